"""Batch-sharded data parallelism for the native K-step training plan (no counterpart in the reference, which is
single-device: ml/scripts/train.py:113-130).

One process per GPU; rank r owns B samples end to end (InstanceNorm and Dropout2d are per-sample, all K steps of a sample
stay on one GPU).  Two exchange steps per optimisation step:
  1. all-reduce (SUM) of the K fluid-cell counts -- divergence_loss / gradient_loss normalise by the BATCH-wide count
     (ml/training/physics_loss.py:119-124,153-161), so every shard must use the global denominator for the sum of shard
     gradients to be the exact global-batch gradient (LOSS_FINALIZE / LOSS_GRAD take `gcnt` and `world`);
  2. all-reduce (AVG) of the flat fp32 gradient buffer (24.1 MB for UNet_medium).  It is issued once, after backward:
     gradients of a K-step rollout accumulate over all K applications and are only final when the last one (k = 0) has
     finished, so there is nothing to overlap earlier; 24 MB over xGMI is ~0.3 ms against a ~45 ms step.
Clip-norm + Adam then run identically on every rank on the averaged gradient.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


def dp_train_step(plan, world: int, all_reduce: Optional[Callable] = None, run: Optional[Callable] = None, avg: bool = True) -> None:
    """One optimisation step of `plan` (RolloutTrainPlan built with world=world).

    all_reduce(tensor, op) with op in {"sum", "avg"} reduces in place across ranks (torch.distributed in production);
    run(section_name) executes one plan section (default: plan.run on the current HIP stream)."""
    run = run or (lambda name: plan.run((name,)))
    if world == 1:
        for name in ("fwd", "fin", "bwd", "opt"):
            run(name)
        return
    assert all_reduce is not None
    run("fwd")
    # sums[k][5] = this shard's fluid-cell count of rollout step k (fp64)
    plan.gcnt.tensor.copy_(torch.stack([s.tensor[5] for s in plan.sums]))
    all_reduce(plan.gcnt.tensor, "sum")
    run("fin")
    run("bwd")
    all_reduce(plan.pb.grads.tensor, "avg" if avg else "sum")
    run("opt")


def torch_all_reduce(dist):
    """Adapter from torch.distributed (backend nccl = RCCL on ROCm, or gloo in the CPU tests) to dp_train_step's callback."""
    def f(t: torch.Tensor, op: str) -> None:
        if dist.get_backend() == "gloo":                      # CPU tests / rehearsal: no AVG, and device tensors go through the host
            h = t.detach().cpu() if t.is_cuda else t
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            if op == "avg":
                h.div_(dist.get_world_size())
            if t.is_cuda:
                t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.AVG if op == "avg" else dist.ReduceOp.SUM)
    return f
