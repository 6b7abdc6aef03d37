"""Synthetic NPZ-shaped batches: density U[0,1], velocities clip(N(0,0.3),-1,1), one disc emitter and
one rectangular collider per scene (a few % coverage, non-overlapping), constant over the K window."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def synth_rollout_batch(B: int, K: int, R: int, seed: int = 1234) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> x0 [B,6,R,R], y_seq [B,K,3,R,R], masks [B,K,2,R,R] (float32), the tuple layout of FluidNPZSequenceDataset."""
    rs = np.random.RandomState(seed)

    def fields(n):
        d = rs.uniform(0, 1, (n, 1, R, R))
        v = np.clip(rs.normal(0, 0.3, (n, 2, R, R)), -1, 1)
        return np.concatenate([d, v], 1).astype(np.float32)

    yy, xx = np.mgrid[0:R, 0:R]
    em = np.zeros((B, R, R), np.float32)
    co = np.zeros((B, R, R), np.float32)
    for b in range(B):
        r = rs.randint(max(2, R // 12), max(3, R // 7))
        cy, cx = rs.randint(r, R - r, 2)
        em[b] = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
        h, w = rs.randint(max(1, R // 16), max(2, R // 5)), rs.randint(max(1, R // 16), max(2, R // 3))
        y0, x0 = rs.randint(0, R - h), rs.randint(0, R - w)
        rect = np.zeros((R, R), np.float32)
        rect[y0:y0 + h, x0:x0 + w] = 1
        co[b] = rect * (1 - em[b])
    st = fields(B)
    dprev = rs.uniform(0, 1, (B, 1, R, R)).astype(np.float32)
    x0 = np.concatenate([st, dprev, em[:, None], co[:, None]], 1)
    y = np.stack([fields(B) for _ in range(K)], 1)
    masks = np.broadcast_to(np.stack([em, co], 1)[:, None], (B, K, 2, R, R)).copy()
    return x0, y, masks


def simple_infer_masks(res: int, scale: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """Static circle emitter + rectangular collider of ml/scripts/simple_infer.py:179-205 (scale=1: the reference's pixel
    geometry; scale=res/128 keeps the picture at other resolutions)."""
    yy, xx = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
    radius = int(round(10 * scale))
    cx, cy = res // 2, res // 2 + int(round(50 * scale))
    emitter = (np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2) <= radius).astype(np.float32)
    cw, chh = int(round(40 * scale)), int(round(8 * scale))
    ccx, ccy = res // 2, res // 2 - int(round(40 * scale))
    collider = np.zeros((res, res), np.float32)
    x0, x1 = max(0, ccx - cw // 2), min(res, ccx + cw // 2)
    y0, y1 = max(0, ccy - chh // 2), min(res, ccy + chh // 2)
    collider[y0:y1, x0:x1] = 1.0
    return emitter, collider
