"""Per-shape UNet forward/backward engine behind the nn.Module API (autograd path).

One UNetEngine serves one (B, H, W, training) shape.  It owns a pool of "application slots": each slot has its own
tape (saved activations), so that K chained applications of a rollout can all be alive until loss.backward() walks
back through them, exactly like the K autograd graphs the reference keeps (ml/training/trainer.py:355-385).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import runtime as rt
from .plan import Buf, OpRec, PlanBuilder, Ref, allocate, lower


class _Slot:
    def __init__(self, idx: int):
        self.idx = idx
        self.busy = False
        self.fwd: np.ndarray = None
        self.bwd: np.ndarray = None
        self.x: Buf = None
        self.pred: Buf = None
        self.dpred: Buf = None
        self.dx: Buf = None


class UNetEngine:
    def __init__(self, cfg, B: int, H: int, W: int, *, params_flat: torch.Tensor, grads_flat: torch.Tensor, dtype: str,
                 conv_impl: str, training: bool, seed: int = 0, norm_state: Optional[torch.Tensor] = None):
        self.cfg, self.B, self.H, self.W = cfg, B, H, W
        self.training = training
        self.device = params_flat.device
        self.pb = PlanBuilder(cfg, B, H, W, dtype=dtype, conv_impl=conv_impl, training=training, seed=seed)
        assert params_flat.numel() == self.pb.n_params_padded
        self.pb.params.tensor = params_flat
        self.pb.grads.tensor = grads_flat
        if norm_state is not None:             # BatchNorm2d running statistics owned by the caller (UNet's buffers)
            assert norm_state.numel() == self.pb.nstate.numel and norm_state.dtype == torch.float32
            self.pb.nstate.tensor = norm_state
        self.slots: List[_Slot] = []
        self._prep: Optional[np.ndarray] = None
        self._prepped_version = -1
        self._zero_dwk: Optional[np.ndarray] = None
        self._unprep_acc: Optional[np.ndarray] = None
        self.calls = 0

    # ------------------------------------------------------------------ slots
    def _new_slot(self) -> _Slot:
        pb = self.pb
        s = _Slot(len(self.slots))
        B, H, W = self.B, self.H, self.W
        HW = H * W
        cin = self.cfg.in_channels
        s.x = pb.new_buf(f"slot{s.idx}/x", (B, cin, H, W), "f32")
        s.dpred = pb.new_buf(f"slot{s.idx}/dpred", (B, self.cfg.out_channels, H, W), "f32")
        s.dx = pb.new_buf(f"slot{s.idx}/dx", (B, cin, H, W), "f32")
        fwd: List[OpRec] = []
        tape = pb.emit_forward(fwd, s.idx, ("nchw", Ref(s.x), cin))
        s.pred = tape.pred
        bwd: List[OpRec] = []
        pb.emit_backward(bwd, tape, s.dpred, s.dx, cin)
        if self._prep is None:
            prep: List[OpRec] = []
            pb.emit_prep_weights(prep)
            zero: List[OpRec] = []
            pb.emit_zero(zero, pb.dwk_slab)
            unprep: List[OpRec] = []
            pb.emit_unprep(unprep, accumulate=True)        # .grad accumulates across backward calls, as autograd's does
            allocate(pb, self.device)
            self._prep, self._zero_dwk, self._unprep_acc = lower(pb, prep), lower(pb, zero), lower(pb, unprep)
        else:
            allocate(pb, self.device)
        s.fwd, s.bwd = lower(pb, fwd), lower(pb, bwd)
        self.slots.append(s)
        return s

    def acquire(self) -> _Slot:
        for s in self.slots:
            if not s.busy:
                s.busy = True
                return s
        s = self._new_slot()
        s.busy = True
        return s

    def release_all(self) -> None:
        for s in self.slots:
            s.busy = False

    # ------------------------------------------------------------------ execution
    def ensure_weights(self) -> None:
        v = self.pb.params.tensor._version
        if self._prep is None:
            self._new_slot()
        if v != self._prepped_version:
            rt.run_ops(self._prep)
            self._prepped_version = v

    def forward(self, x: torch.Tensor, keep: bool) -> Tuple[torch.Tensor, Optional[_Slot]]:
        """x: [B,6,H,W] fp32 on the engine's device.  keep=True reserves the slot's tape for a later backward()."""
        self.ensure_weights()
        s = self.acquire()
        s.x.tensor.copy_(x)
        if self.training and self.cfg.dropout > 0.0:
            seed = (self.pb.seed * 0x9E3779B1 + self.calls * 0x85EBCA6B + 12345) & 0x7FFFFFFF
            rt.patch_seed(s.fwd, seed)
            rt.patch_seed(s.bwd, seed)
        rt.run_ops(s.fwd)
        self.calls += 1
        out = s.pred.tensor.clone()
        if not keep:
            s.busy = False
            return out, None
        return out, s

    def backward(self, s: _Slot, dpred: torch.Tensor) -> torch.Tensor:
        """Accumulates parameter gradients into grads_flat (+=) and returns d(input) [B,6,H,W]."""
        s.dpred.tensor.copy_(dpred)
        rt.run_ops(self._zero_dwk)
        rt.run_ops(s.bwd)
        rt.run_ops(self._unprep_acc)
        s.busy = False
        return s.dx.tensor.clone()
