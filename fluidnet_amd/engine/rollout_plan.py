"""Autoregressive inference rollout (ml/scripts/simple_infer.py:226-260, ml/training/test_evaluation.py:192-292) as a
replayable native plan: one frame = pack -> UNet forward (eval) -> state hand-off on device, captured in a hipGraph."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import runtime as rt
from .plan import OpRec, PlanBuilder, Ref, allocate, lower


class RolloutInferPlan:
    def __init__(self, cfg, B: int, H: int, W: int, *, params_flat: torch.Tensor, dtype: str = "bf16", conv_impl: str = "auto",
                 norm_state: Optional[torch.Tensor] = None):
        self.B, self.H, self.W = B, H, W
        pb = self.pb = PlanBuilder(cfg, B, H, W, dtype=dtype, conv_impl=conv_impl, training=False)
        HW = H * W
        self.cur = pb.new_buf("state_current", (B, 3, H, W), "f32")
        self.prev = pb.new_buf("state_prev_density", (B, 1, H, W), "f32")
        self.emitter = pb.new_buf("emitter", (B, H, W), "f32")
        self.collider = pb.new_buf("collider", (B, H, W), "f32")
        self.prep: List[OpRec] = []
        pb.emit_prep_weights(self.prep)
        frame: List[OpRec] = []
        planes = [(Ref(self.cur, c * HW), 3 * HW) for c in range(3)] + [(Ref(self.prev), HW), (Ref(self.emitter), HW), (Ref(self.collider), HW)]
        tape = pb.emit_forward(frame, 0, planes)
        self.pred = tape.pred
        # state_prev <- old state_current[0:1]; state_current <- output   (simple_infer.py:259-260)
        pb.op(frame, "COPY_PLANES", [B, HW, 3 * HW, HW], p=[Ref(self.cur), Ref(self.prev)], tag="prev<-cur")
        pb.op(frame, "COPY_PLANES", [B, 3 * HW, 3 * HW, 3 * HW], p=[Ref(self.pred), Ref(self.cur)], tag="cur<-pred")
        self.frame = frame
        pb.params.tensor = params_flat
        pb.grads.tensor = params_flat           # unused by inference ops; bound so that lowering never sees a null buffer
        if norm_state is not None:              # BatchNorm2d running statistics of the checkpoint (eval mode reads them)
            assert norm_state.numel() == pb.nstate.numel
            pb.nstate.tensor = norm_state
        self.bytes = allocate(pb, params_flat.device)
        self._prep = lower(pb, self.prep)
        self._frame = lower(pb, frame)
        self._graph: Optional[rt.Graph] = None
        self._version = -1
        self._params = params_flat

    def reset(self, emitter: torch.Tensor, collider: torch.Tensor, state: Optional[torch.Tensor] = None,
              prev_density: Optional[torch.Tensor] = None) -> None:
        self.emitter.tensor.copy_(emitter.reshape(self.B, self.H, self.W))
        self.collider.tensor.copy_(collider.reshape(self.B, self.H, self.W))
        if state is None:
            self.cur.tensor.zero_()
        else:
            self.cur.tensor.copy_(state)
        if prev_density is None:
            self.prev.tensor.zero_()
        else:
            self.prev.tensor.copy_(prev_density.reshape(self.B, 1, self.H, self.W))

    def _weights(self) -> None:
        if self._params._version != self._version:
            rt.run_ops(self._prep)
            self._version = self._params._version

    def step(self, frames: int = 1, use_graph: bool = True) -> torch.Tensor:
        """Advance `frames` frames; returns the state after the last one (a view of the engine's buffer)."""
        self._weights()
        if use_graph:
            if self._graph is None:
                self._graph = rt.Graph(self._frame)
            self._graph.launch(frames)
        else:
            for _ in range(frames):
                rt.run_ops(self._frame)
        return self.cur.tensor

    def rollout(self, frames: int) -> torch.Tensor:
        """[frames, B, 3, H, W]: every frame's output (one small D2D copy per frame between graph replays)."""
        out = torch.empty((frames, self.B, 3, self.H, self.W), dtype=torch.float32, device=self._params.device)
        for f in range(frames):
            out[f].copy_(self.step(1))
        return out
