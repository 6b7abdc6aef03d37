"""clip_grad_norm_ + Adam on the flat parameter / gradient buffers of fluidnet_amd.UNet, as two HIP launches
(GRAD_NORM, ADAM in csrc/fnet_optim.hip) with the clip coefficient computed on the device -- replaces
torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step of ml/training/trainer.py:190-195 (88 norms + ~10 foreach kernels and
one host sync in the reference).  Same arithmetic: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad."""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from ..engine import runtime as rt
from ..engine.runtime import OPC


class FusedClipAdam:
    def __init__(self, model, lr: float, max_norm: Optional[float] = 1.0, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        self.model = model
        self.max_norm = max_norm
        self.betas, self.eps = betas, eps
        self.param_groups = [{"lr": float(lr)}]        # torch-optimizer-shaped, for schedulers / logging
        self.step_count = 0
        self._m: Optional[torch.Tensor] = None
        self._v: Optional[torch.Tensor] = None
        self._normsq: Optional[torch.Tensor] = None

    def _state(self):
        p = self.model.flat_params
        if self._m is None or self._m.device != p.device or self._m.numel() != p.numel():
            self._m = torch.zeros_like(p)
            self._v = torch.zeros_like(p)
            self._normsq = torch.zeros(2, dtype=torch.float64, device=p.device)
        return p, self.model.flat_grads, self._m, self._v, self._normsq

    def zero_grad(self, set_to_none: bool = True) -> None:
        self.model.flat_grads.zero_()
        for q in self.model.parameters():
            q.grad = None

    @torch.no_grad()
    def step(self) -> None:
        p, g, m, v, nsq = self._state()
        if not p.is_cuda:
            raise RuntimeError("FusedClipAdam runs on MI355X only; there is no CPU fallback")
        self.model._attach_grads()
        self.step_count += 1
        b1, b2 = self.betas
        n = p.numel()
        ops = np.zeros(2, dtype=rt.OP_DTYPE)
        nsq.zero_()
        ops[0]["op"] = OPC["GRAD_NORM"]
        ops[0]["i"][:2] = [n & 0x7FFFFFFF, n >> 31]
        ops[0]["p"][:2] = [g.data_ptr(), nsq.data_ptr()]
        ops[1]["op"] = OPC["ADAM"]
        ops[1]["i"][:2] = [n & 0x7FFFFFFF, n >> 31]
        ops[1]["f"][:8] = [self.param_groups[0]["lr"], b1, b2, self.eps, 1.0 - b1 ** self.step_count,
                           math.sqrt(1.0 - b2 ** self.step_count), self.max_norm or 0.0, 1.0]
        ops[1]["p"][:5] = [p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), nsq.data_ptr() if self.max_norm else 0]
        rt.run_ops(ops if self.max_norm else ops[1:])
        p.add_(0)          # bump flat_params' autograd version so that engines refresh their bf16 weight copies

    def state_dict(self) -> Dict:
        p, g, m, v, _ = self._state()
        return {"step": self.step_count, "lr": self.param_groups[0]["lr"], "exp_avg": m.detach().cpu(), "exp_avg_sq": v.detach().cpu(),
                "betas": self.betas, "eps": self.eps, "max_norm": self.max_norm}

    def load_state_dict(self, sd: Dict) -> None:
        p, g, m, v, _ = self._state()
        self.step_count = int(sd["step"])
        self.param_groups[0]["lr"] = float(sd["lr"])
        m.copy_(sd["exp_avg"])
        v.copy_(sd["exp_avg_sq"])
