"""ctypes front end of the native rollout C ABI (include/fluidnet_hip.h, fnet_rollout_*): the entry points a C++ host such
as the reference's viewer (engine/src/Simulation.cpp:236-367) would call, used here by the tests and by Python callers
that want the device-resident frame loop without torch tensors (numpy in, numpy out)."""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import runtime as rt


class FnetUNetConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("in_channels", "out_channels", "base_channels", "depth", "act", "padding_mode", "upsample",
                                             "use_residual", "bottleneck_blocks", "output_activation", "norm", "downsample",
                                             "group_norm_groups")]


def native_config(cfg) -> FnetUNetConfig:
    """UNetConfig (reference field names) -> the C struct; rejects what the native family does not cover."""
    norms, downs = {"instance": 0, "batch": 1, "group": 2}, {"stride": 0, "avgpool": 1, "maxpool": 2}
    if cfg.norm not in norms or cfg.downsample not in downs or cfg.upsample not in ("nearest", "bilinear"):
        raise NotImplementedError("native rollout covers norm in {instance, batch, group}, downsample in {stride, avgpool, maxpool}, "
                                  "upsample in {nearest, bilinear}")
    return FnetUNetConfig(cfg.in_channels, cfg.out_channels, cfg.base_channels, cfg.depth, rt.ACT[cfg.act], rt.PAD[cfg.padding_mode],
                          1 if cfg.upsample == "bilinear" else 0, 1 if cfg.use_residual else 0, cfg.bottleneck_blocks,
                          rt.OUTACT[cfg.output_activation], norms[cfg.norm], downs[cfg.downsample], cfg.group_norm_groups)


def checkpoint_floats(state_dict) -> np.ndarray:
    """The array fnet_rollout_create takes: every floating-point tensor of a (reference-format) state dict, in its order, flattened
    and concatenated -- BatchNorm2d's int64 num_batches_tracked is skipped."""
    parts = [np.asarray(v.detach().cpu().float().numpy() if hasattr(v, "detach") else v, dtype=np.float32).reshape(-1)
             for k, v in state_dict.items() if not k.endswith("num_batches_tracked")]
    return np.concatenate(parts)


def _f32(a: Optional[np.ndarray], shape) -> Optional[np.ndarray]:
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


class NativeRollout:
    def __init__(self, cfg, weights: np.ndarray, B: int, H: int, W: int) -> None:
        """weights: the trainable tensors in state-dict order, flattened and concatenated (fp32)."""
        self._L = rt.lib()
        self.c = native_config(cfg)
        n = ctypes.c_int64(0)
        rt.check(self._L.fnet_unet_param_count(ctypes.byref(self.c), ctypes.byref(n)), "fnet_unet_param_count")
        w = np.ascontiguousarray(weights, dtype=np.float32).reshape(-1)
        if w.size != n.value:
            raise ValueError(f"this configuration has {n.value} parameters, got {w.size}")
        self.B, self.H, self.W = B, H, W
        self._h = ctypes.c_void_p()
        rt.check(self._L.fnet_rollout_create(ctypes.byref(self.c), w.ctypes.data, w.size, B, H, W, ctypes.byref(self._h)), "fnet_rollout_create")

    def set_masks(self, emitter: Optional[np.ndarray], collider: Optional[np.ndarray]) -> None:
        e, c = _f32(emitter, (self.B, self.H, self.W)), _f32(collider, (self.B, self.H, self.W))
        rt.check(self._L.fnet_rollout_set_masks(self._h, e.ctypes.data if e is not None else None, c.ctypes.data if c is not None else None),
                 "fnet_rollout_set_masks")

    def set_state(self, state: Optional[np.ndarray] = None, prev_density: Optional[np.ndarray] = None) -> None:
        s, p = _f32(state, (self.B, 3, self.H, self.W)), _f32(prev_density, (self.B, self.H, self.W))
        rt.check(self._L.fnet_rollout_set_state(self._h, s.ctypes.data if s is not None else None, p.ctypes.data if p is not None else None),
                 "fnet_rollout_set_state")

    def add_impulse(self, impulse_xy: np.ndarray) -> None:
        a = _f32(impulse_xy, (self.B, 2, self.H, self.W))
        rt.check(self._L.fnet_rollout_add_impulse(self._h, a.ctypes.data, None), "fnet_rollout_add_impulse")
        self.read_state()            # the upload is asynchronous: keep `a` alive until it has been consumed

    def step(self, frames: int = 1) -> None:
        rt.check(self._L.fnet_rollout_step(self._h, frames, None), "fnet_rollout_step")

    def read_state(self) -> np.ndarray:
        out = np.empty((self.B, 3, self.H, self.W), dtype=np.float32)
        rt.check(self._L.fnet_rollout_read_state(self._h, out.ctypes.data, None), "fnet_rollout_read_state")
        return out

    def close(self) -> None:
        if self._h:
            self._L.fnet_rollout_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
