"""Host replica of the counter-based Dropout2d channel mask used by the HIP kernels
(fluidnet_amd/csrc/fnet_common.h: drop_scale).  keep(n, c) depends only on (seed, site, n, c), so the forward
and backward kernels agree without storing a mask, and tests can inject the identical mask into the oracle."""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & _M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M
    x ^= x >> np.uint64(16)
    return x


def drop_scale_table(seed: int, site: int, B: int, C: int, p: float) -> np.ndarray:
    """[B, C] float32 multipliers: 0 for dropped channels, 1/(1-p) for kept ones (Dropout2d semantics)."""
    if p <= 0.0:
        return np.ones((B, C), np.float32)
    n = np.arange(B, dtype=np.uint64)[:, None]
    c = np.arange(C, dtype=np.uint64)[None, :]
    h0 = _mix32(np.array((site * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF, dtype=np.uint64))
    h = _mix32(np.uint64(seed & 0xFFFFFFFF) ^ h0)
    h = _mix32(h ^ ((n * np.uint64(0x85EBCA77) + np.uint64(0x165667B1)) & _M))
    h = _mix32(h ^ ((c * np.uint64(0xC2B2AE3D) + np.uint64(0x27D4EB2F)) & _M))
    u = (h >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    p32 = np.float32(p)
    return np.where(u < p32, np.float32(0.0), np.float32(1.0) / (np.float32(1.0) - p32)).astype(np.float32)
