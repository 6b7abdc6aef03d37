"""HBM-resident dataset (no counterpart in the reference, whose DataLoader assembles every
sample on the host: ml/dataset/npz_sequence.py:349-433).

An MI355X has 288 GB of HBM3E; a FluidNet training set (hundreds of 128^2 sequences, 5 fp32 planes per frame) is a few GB.
HbmSequenceStore uploads every seq_*.npz of a split once, normalised, into ONE device tensor [5][T_total][H][W];
HbmBatchLoader then builds each (x0, y_seq, masks) batch with one GATHER_BATCH kernel launch directly into the training
plan's input buffers -- flip augmentation (ml/dataset/augmentations.py:20-35) included -- so that an optimisation step has no
host->device copy and no DataLoader worker on its path.  Sample indexing, the epoch-dependent stride offset, tuple layouts
and mask conventions are FluidNPZSequenceDataset's (the store is built FROM one, so the two always agree on the index).

Data parallel: rank r of `world` takes rows [r*B, (r+1)*B) of every global batch of the same seeded permutation
(SURVEY.md section 8e), so ranks never see the same sample in an epoch and no sample exchange is needed.
"""
from __future__ import annotations

import contextlib
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch

from ..engine import runtime as rt
from .npz_sequence import FluidNPZSequenceDataset, _read_sequence

_FIELDS = ("density", "velx", "velz", "emitter", "collider")


class HbmSequenceStore:
    """All sequences of a FluidNPZSequenceDataset as one device tensor frames[5][T_total][H][W] (fp32, normalised)."""

    def __init__(self, dataset: FluidNPZSequenceDataset, device: str = "cuda") -> None:
        if not str(device).startswith("cuda"):
            raise RuntimeError("HbmSequenceStore keeps the dataset in GPU memory; there is no CPU fallback")
        self.dataset = dataset
        scales = dataset._norm_scales if dataset.normalize else None
        div = [scales["S_density"], scales["S_velx"], scales["S_velz"], None, None] if scales else [None] * 5
        seqs = []
        self.seq_offset: List[int] = []
        total = 0
        for si, path in enumerate(dataset.seq_paths):
            mem = dataset._preloaded_sequences[si] if dataset._preloaded_sequences is not None else _read_sequence(path)
            d = mem["density"]
            planes = []
            for f, name in enumerate(_FIELDS):
                a = mem[name] if mem[name] is not None else np.zeros_like(d)
                planes.append((a / np.float32(div[f])).astype(np.float32) if div[f] is not None else a.astype(np.float32, copy=False))
            seqs.append(np.stack(planes, 0))                 # [5][T][H][W]
            self.seq_offset.append(total)
            total += d.shape[0]
        self.H, self.W = seqs[0].shape[-2:]
        if any(s.shape[-2:] != (self.H, self.W) for s in seqs):
            raise ValueError("all sequences of a split must share one resolution")
        self.t_total = total
        self.frames = torch.from_numpy(np.concatenate(seqs, axis=1)).to(device)     # the one upload
        self.nbytes = self.frames.numel() * 4

    def frame_index(self, sample_indices) -> np.ndarray:
        """Global frame index of sample i's time t, for the dataset's CURRENT epoch offset (set_epoch)."""
        idx = self.dataset._index
        return np.asarray([self.seq_offset[idx[i][0]] + idx[i][1] for i in sample_indices], dtype=np.int32)


class HbmBatchLoader:
    """Iterates the batches of one epoch; every batch is one GATHER_BATCH launch into caller-owned device buffers.

    mask_frames: K (the reference's preloaded-dataset convention, masks of t+1..t+K) or K+1 (its on-disk convention, t..t+K);
    flip augmentation follows the dataset's augmentation_config and is_training flags."""

    def __init__(self, store: HbmSequenceStore, batch_size: int, *, shuffle: bool = True, drop_last: bool = True, seed: int = 0,
                 rank: int = 0, world: int = 1, mask_frames: Optional[int] = None) -> None:
        self.store, self.B, self.shuffle, self.drop_last = store, batch_size, shuffle, drop_last
        self.seed, self.rank, self.world = seed, rank, world
        self.sharded = True          # Trainer: rows are already rank-disjoint, do not split the batch again
        ds = store.dataset
        self.K = max(1, ds.rollout_steps)
        self.Km = mask_frames if mask_frames is not None else self.K
        if self.Km not in (self.K, self.K + 1):
            raise ValueError("mask_frames must be K or K+1")
        # first mask frame relative to t: the on-disk convention (K+1 frames) starts at t, the preloaded one (K frames) at t+1; a
        # single-step dataset (rollout_steps == 1) has no mask tensor at all: its masks are channels 4,5 of x, i.e. frame t
        # (npz_sequence.py:144-153, trainer.py:187-188), so a K = 1 plan fed from here must see frame t as well
        self.mask_first = 0 if (ds.rollout_steps <= 1 or self.Km == self.K + 1) else 1
        self.flip_p = ds.flip_probability if (ds.is_training and ds.enable_augmentation) else 0.0
        self.epoch = 0
        dev = store.frames.device
        self._idx = torch.zeros(batch_size, dtype=torch.int32, device=dev)
        self._flip = torch.zeros(batch_size, dtype=torch.int32, device=dev)
        # Host staging for the per-batch indices and flip flags: two PINNED slots used in turn, each guarded by an event recorded behind
        # its upload.  (An asynchronous copy from a pageable temporary may still be reading it after the temporary is gone; a blocking copy
        # would stall the host behind the previous step's kernels.)
        self._host = torch.zeros(2, 2, batch_size, dtype=torch.int32).pin_memory() if dev.type == "cuda" else torch.zeros(2, 2, batch_size, dtype=torch.int32)
        self._host_ev = [None, None]
        self._slot = 0
        self._table = self._table_host = self._table_ev = None      # upload_epoch(): the whole epoch's indices / flags on the device

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
        self.store.dataset.set_epoch(epoch)

    def __len__(self) -> int:
        n = len(self.store.dataset) // (self.B * self.world)
        if not self.drop_last and len(self.store.dataset) % (self.B * self.world):
            n += 1
        return n

    def batches(self) -> List[Tuple[np.ndarray, np.ndarray]]:
        """[(sample indices, flip flags)] of this rank for the current epoch (host side, deterministic in (seed, epoch))."""
        n = len(self.store.dataset)
        rs = np.random.RandomState((self.seed * 1000003 + self.epoch) & 0x7FFFFFFF)
        order = rs.permutation(n) if self.shuffle else np.arange(n)
        gb = self.B * self.world
        out = []
        for b in range(len(self)):
            rows = order[b * gb + self.rank * self.B: b * gb + (self.rank + 1) * self.B]
            if len(rows) < self.B:                      # ragged last batch (drop_last=False): wrap around
                rows = np.concatenate([rows, order[: self.B - len(rows)]])
            flips = (rs.random_sample(gb) < self.flip_p)[self.rank * self.B: (self.rank + 1) * self.B]
            out.append((rows, flips.astype(np.int32)))
        return out

    def _gather_op(self, B: int, idx_ptr: int, flip_ptr: int, x0: torch.Tensor, y: torch.Tensor, masks: torch.Tensor) -> np.ndarray:
        st = self.store
        for t_, shp in ((x0, (B, 6, st.H, st.W)), (y, (B, self.K, 3, st.H, st.W)), (masks, (B, self.Km, 2, st.H, st.W))):
            if tuple(t_.shape) != shp or t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != st.frames.device:
                raise ValueError(f"output buffer must be contiguous fp32 {shp} on {st.frames.device}, got {tuple(t_.shape)} {t_.dtype}")
        ops = np.zeros(1, dtype=rt.OP_DTYPE)
        o = ops[0]
        o["op"] = rt.OPC["GATHER_BATCH"]
        o["i"][:7] = [B, st.H, st.W, self.K, self.Km, self.mask_first, st.t_total]
        o["p"][:6] = [st.frames.data_ptr(), idx_ptr, flip_ptr, x0.data_ptr(), y.data_ptr(), masks.data_ptr()]
        return ops

    def upload_epoch(self) -> int:
        """The current epoch's frame indices and flip flags as ONE device table [n_batches][2][B] (one pinned upload per epoch instead
        of two small copies per step: each such copy is a blit kernel of its own in front of the step's first launch, ~0.1 ms of idle
        queue per step in the kernel trace).  Returns the number of batches; gather_uploaded(b, ...) then forms batch b."""
        bs = self.batches()
        dev = self.store.frames.device
        host = torch.empty(max(len(bs), 1), 2, self.B, dtype=torch.int32)
        if dev.type == "cuda":
            host = host.pin_memory()
        for b, (rows, flips) in enumerate(bs):
            host[b, 0] = torch.from_numpy(self.store.frame_index(rows))
            host[b, 1] = torch.from_numpy(np.ascontiguousarray(flips, dtype=np.int32))
        if self._table_ev is not None:
            self._table_ev.synchronize()               # the previous epoch's upload has left its pinned buffer
        self._table = host.to(dev, non_blocking=True)
        self._table_host = host                        # keep the pinned source alive until the copy has run
        if dev.type == "cuda":
            self._table_ev = torch.cuda.Event()
            self._table_ev.record()
        self._table_epoch = self.epoch
        return len(bs)

    def gather_uploaded(self, b: int, x0: torch.Tensor, y: torch.Tensor, masks: torch.Tensor, stream=None) -> None:
        """Batch b of the epoch table made by upload_epoch(): one GATHER_BATCH launch, no host -> device copy."""
        if self._table is None or self._table_epoch != self.epoch:
            raise RuntimeError("upload_epoch() must be called after set_epoch() and before gather_uploaded()")
        if stream is not None and self._table_ev is not None:
            torch.cuda.ExternalStream(stream).wait_event(self._table_ev)       # the table was uploaded on torch's current stream
        rt.run_ops(self._gather_op(self.B, self._table[b, 0].data_ptr(), self._table[b, 1].data_ptr(), x0, y, masks), stream)

    def gather(self, rows: np.ndarray, flips: np.ndarray, x0: torch.Tensor, y: torch.Tensor, masks: torch.Tensor, stream=None) -> None:
        """One launch: fill x0 [B,6,H,W], y [B,K,3,H,W], masks [B,Km,2,H,W] (fp32, contiguous, on the store's device)."""
        st = self.store
        B = len(rows)
        ops = self._gather_op(B, self._idx.data_ptr(), self._flip.data_ptr(), x0, y, masks)
        # the index / flag uploads and the kernel go to ONE stream (the reused index buffers would otherwise race)
        ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream)) if stream is not None else contextlib.nullcontext()
        slot = self._slot
        self._slot ^= 1
        if self._host_ev[slot] is not None:
            self._host_ev[slot].synchronize()          # the upload that last used this slot (two gathers ago) has been consumed
        self._host[slot, 0, :B].copy_(torch.from_numpy(st.frame_index(rows)))
        self._host[slot, 1, :B].copy_(torch.from_numpy(np.ascontiguousarray(flips, dtype=np.int32)))
        with ctx:
            self._idx[:B].copy_(self._host[slot, 0, :B], non_blocking=True)
            self._flip[:B].copy_(self._host[slot, 1, :B], non_blocking=True)
            if self._idx.is_cuda:
                self._host_ev[slot] = torch.cuda.Event()
                self._host_ev[slot].record()
            rt.run_ops(ops, stream)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """DataLoader-style iteration (fresh tensors per batch) for code that wants tuples; the zero-copy way is gather() into a
        RolloutTrainPlan's x0 / y / masks buffers (Trainer does that when given an HbmBatchLoader)."""
        st = self.store
        dev = st.frames.device
        for rows, flips in self.batches():
            x0 = torch.empty(len(rows), 6, st.H, st.W, device=dev)
            y = torch.empty(len(rows), self.K, 3, st.H, st.W, device=dev)
            mk = torch.empty(len(rows), self.Km, 2, st.H, st.W, device=dev)
            self.gather(rows, flips, x0, y, mk)
            if st.dataset.rollout_steps > 1:
                yield x0, y, mk
            else:
                yield x0, y[:, 0]
