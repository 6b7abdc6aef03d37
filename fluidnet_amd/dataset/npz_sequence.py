"""FluidNPZSequenceDataset with the reference's API and sample layouts (ml/dataset/npz_sequence.py).

Files: <npz_dir>/<split>/seq_*.npz with float32 arrays density, velx, velz (and optionally emitter, collider) of shape
(T,H,W) (vdb-tools/vdb_core/batch_processing.py:354-361).  Index = every (sequence, t) with
t in range(1 + offset, T - rollout_steps, stride) (:68-75); set_epoch rotates the stride offset (:282-284).
rollout_steps == 1 -> (x[6,H,W], y[3,H,W]) with x = [d_t, vx_t, vz_t, d_{t-1}, emitter_t, collider_t], y = state_{t+1};
rollout_steps  > 1 -> (x_0[6], y_seq[K,3], masks): NOTE the two mask conventions of the reference are kept exactly --
the on-disk path returns K+1 mask frames t..t+K (:185,:200,:210), the preloaded path K frames t+1..t+K (:406,:421).
Host-side numpy only (no kernel).  Pinned to the reference by the g6 fixtures: tuples, index lists, epoch offsets, normalisation and
flipped samples produced by the REAL ml/dataset module for the files of tests/fixture_inputs.py (tests/golden/make_golden.py imports it
with a pydantic_settings stub; tests/test_golden_next_rows.py compares bit for bit), plus the oracle's restatement in tests/test_dataset.py.
Differences from the reference: `norm_scales` / `stats_path` can be passed explicitly instead of being read from the
project config; preloading never prints.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from torch.utils.data import Dataset

from .augmentations import apply_augmentation, apply_rollout_augmentation
from .normalization import load_normalization_scales

_KEYS = ("density", "velx", "velz")


def _validate_sequence_data(path, d, vx, vz, emitter, collider) -> None:
    if d.ndim != 3 or vx.ndim != 3 or vz.ndim != 3:
        raise ValueError(f"Expected (T,H,W) arrays in {path}")
    if not (d.shape == vx.shape == vz.shape):
        raise ValueError(f"Shape mismatch in {path}: d={d.shape}, vx={vx.shape}, vz={vz.shape}")
    for name, m in (("emitter", emitter), ("collider", collider)):
        if m is not None and (m.ndim != 3 or m.shape != d.shape):
            raise ValueError(f"{name} shape mismatch in {path}: {name}={m.shape}, density={d.shape}")
    if d.shape[0] < 3:
        raise ValueError(f"Animation is less than 3 frames, not enough to form samples. {path}: d={d.shape}")


def _build_indices_for_offset(frame_counts: List[int], rollout_steps: int, stride: int, offset: int = 0) -> List[Tuple[int, int]]:
    return [(si, t) for si, T in enumerate(frame_counts) for t in range(1 + offset, T - rollout_steps, stride)]


def _read_sequence(path: Path) -> Dict[str, Optional[np.ndarray]]:
    with np.load(path) as data:      # allow_pickle stays False
        seq = {k: data[k].astype(np.float32, copy=True) for k in _KEYS}
        seq["emitter"] = data["emitter"].astype(np.float32, copy=True) if "emitter" in data else None
        seq["collider"] = data["collider"].astype(np.float32, copy=True) if "collider" in data else None
    _validate_sequence_data(path, seq["density"], seq["velx"], seq["velz"], seq["emitter"], seq["collider"])
    return seq


def _single_sample(seq, t: int, scales: Optional[Dict[str, float]]):
    d, vx, vz = seq["density"], seq["velx"], seq["velz"]
    em = seq["emitter"] if seq["emitter"] is not None else np.zeros_like(d)
    co = seq["collider"] if seq["collider"] is not None else np.zeros_like(d)
    sd, sx, sz = (scales["S_density"], scales["S_velx"], scales["S_velz"]) if scales else (None, None, None)
    n = (lambda a, s: a / s) if scales else (lambda a, s: a)
    x = np.stack([n(d[t], sd), n(vx[t], sx), n(vz[t], sz), n(d[t - 1], sd), em[t], co[t]], axis=0)
    y = np.stack([n(d[t + 1], sd), n(vx[t + 1], sx), n(vz[t + 1], sz)], axis=0)
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32))


def _rollout_sample(seq, t: int, K: int, scales: Optional[Dict[str, float]], with_initial_mask: bool):
    d, vx, vz = seq["density"], seq["velx"], seq["velz"]
    em = seq["emitter"] if seq["emitter"] is not None else np.zeros_like(d)
    co = seq["collider"] if seq["collider"] is not None else np.zeros_like(d)
    sd, sx, sz = (scales["S_density"], scales["S_velx"], scales["S_velz"]) if scales else (None, None, None)
    n = (lambda a, s: a / s) if scales else (lambda a, s: a)
    x0 = np.stack([n(d[t], sd), n(vx[t], sx), n(vz[t], sz), n(d[t - 1], sd), em[t], co[t]], axis=0)
    y = np.stack([np.stack([n(d[t + k], sd), n(vx[t + k], sx), n(vz[t + k], sz)], axis=0) for k in range(1, K + 1)], axis=0)
    first = 0 if with_initial_mask else 1
    mk = np.stack([np.stack([em[t + k], co[t + k]], axis=0) for k in range(first, K + 1)], axis=0)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    return f(x0), f(y), f(mk)


def _load_sample(path: Path, t: int, normalize: bool, norm_scales: Optional[Dict[str, float]]):
    """Module-level on-disk loader (npz_sequence.py:115-156)."""
    return _single_sample(_read_sequence(path), t, norm_scales if normalize else None)


def _load_rollout_sample(path: Path, t_start: int, rollout_steps: int, normalize: bool, norm_scales: Optional[Dict[str, float]]):
    """Module-level on-disk rollout loader (npz_sequence.py:159-212): masks carry K+1 frames t..t+K."""
    return _rollout_sample(_read_sequence(path), t_start, rollout_steps, norm_scales if normalize else None, True)


class FluidNPZSequenceDataset(Dataset):
    def __init__(self, npz_dir: Union[str, Path], split: str, normalize: bool = False, is_training: bool = False,
                 augmentation_config: Optional[dict] = None, preload: bool = False, rollout_steps: int = 1, stride: int = 1,
                 *, norm_scales: Optional[Dict[str, float]] = None, stats_path: Optional[Union[str, Path]] = None) -> None:
        self.npz_dir = npz_dir
        self.normalize = normalize
        self.rollout_steps = rollout_steps
        self.is_training = is_training
        self.augmentation_config = augmentation_config or {}
        self.enable_augmentation = self.augmentation_config.get("enable_augmentation", False)
        self.flip_probability = self.augmentation_config.get("flip_probability", 0.5)

        split_dir = Path(npz_dir) / split
        if not split_dir.exists():
            raise FileNotFoundError(f"Split directory not found: {split_dir}")
        self.seq_paths: List[Path] = sorted(f for f in split_dir.iterdir() if f.name.startswith("seq_") and f.name.endswith(".npz"))
        if not self.seq_paths:
            raise FileNotFoundError(f"No seq_*.npz files found in {split_dir}")
        self.num_real_sequences = len(self.seq_paths)

        self._norm_scales: Optional[Dict[str, float]] = None
        if self.normalize:
            if norm_scales is not None:
                self._norm_scales = dict(norm_scales)
            else:
                if stats_path is None:       # reference default: <project root>/data/_field_stats.yaml (config.yaml)
                    stats_path = Path(npz_dir).parent / "_field_stats.yaml"
                self._norm_scales = load_normalization_scales(Path(stats_path))

        frame_counts: List[int] = []
        for p in self.seq_paths:
            with np.load(p) as data:
                d, vx, vz = data["density"], data["velx"], data["velz"]
                _validate_sequence_data(p, d, vx, vz, data["emitter"] if "emitter" in data else None,
                                        data["collider"] if "collider" in data else None)
                frame_counts.append(d.shape[0])
        stride = max(1, int(stride))
        self._indices_by_offset = [_build_indices_for_offset(frame_counts, rollout_steps, stride, off) for off in range(stride)]
        self._index = self._indices_by_offset[0]
        if not self._index:
            raise RuntimeError("No valid samples found (need T>=3 per sequence)")

        self.preload = preload
        self._preloaded_sequences: Optional[Dict[int, Dict[str, Optional[np.ndarray]]]] = None
        if self.preload:
            self._preloaded_sequences = {si: _read_sequence(p) for si, p in enumerate(self.seq_paths)}

    def set_epoch(self, epoch: int) -> None:
        self._index = self._indices_by_offset[epoch % len(self._indices_by_offset)]

    def __len__(self) -> int:
        return len(self._index)

    def __getitem__(self, idx: int):
        si, t = self._index[idx]
        scales = self._norm_scales if self.normalize else None
        mem = self._preloaded_sequences[si] if (self.preload and self._preloaded_sequences is not None) else None
        if self.rollout_steps > 1:
            if mem is not None:
                x, y_seq, masks = _rollout_sample(mem, t, self.rollout_steps, scales, with_initial_mask=False)
            else:
                x, y_seq, masks = _rollout_sample(_read_sequence(self.seq_paths[si]), t, self.rollout_steps, scales, with_initial_mask=True)
            if self.is_training and self.enable_augmentation:
                x, y_seq, masks = apply_rollout_augmentation(x, y_seq, masks, self.flip_probability)
            return x, y_seq, masks
        x, y = _single_sample(mem if mem is not None else _read_sequence(self.seq_paths[si]), t, scales)
        if self.is_training and self.enable_augmentation:
            x, y = apply_augmentation(x, y, self.flip_probability)
        return x, y
