"""Batch-sharded data parallelism for the native K-step training plan (no counterpart in the reference, which is
single-device: ml/scripts/train.py:113-130).

One process per GPU; rank r owns B samples end to end (InstanceNorm and Dropout2d are per-sample, all K steps of a sample
stay on one GPU).  Two exchange steps per optimisation step:
  1. all-reduce (SUM) of the K fluid-cell counts -- divergence_loss / gradient_loss normalise by the BATCH-wide count
     (ml/training/physics_loss.py:119-124,153-161), so every shard must use the global denominator for the sum of shard
     gradients to be the exact global-batch gradient (LOSS_FINALIZE / LOSS_GRAD take `gcnt` and `world`).  The counts depend on
     the masks alone, which are INPUTS: they are counted and exchanged when the batch is loaded, before the forward pass, and
     the (asynchronous) all-reduce is only waited for when the losses are finalised;
  2. all-reduce (AVG) of the flat fp32 gradient buffer (24.1 MB for UNet_medium) in BUCKETS, decoder first: gradients of a
     K-step rollout accumulate over all K applications, so a layer's gradient is final once the last backward pass (k = 0)
     has visited it; the plan marks those points (RolloutTrainPlan.grad_buckets: head + ups.*, mid.*, the deepest encoder
     level, the rest) and each bucket's all-reduce is started there, on the stream that produced its last piece, while the
     remaining layers of the pass are still being computed.  RCCL runs it on its own stream; only clip + Adam wait for it.
Clip-norm + Adam then run identically on every rank on the averaged gradient.
"""
from __future__ import annotations

import time
from typing import Callable, List, Optional

import torch

from ..engine import runtime as rt


def fluid_counts(plan) -> torch.Tensor:
    """fp64 [K]: this shard's fluid-cell count of every rollout step, from the mask planes the plan was loaded with -- the
    same number LOSS_REDUCE leaves in sums[k][5] (cells with emitter == 0 and collider == 0, physics_loss.py:113-117)."""
    m = plan.masks.tensor
    off = plan.mask_offset
    sel = m[:, off:off + plan.K]
    return ((sel[:, :, 0] == 0) & (sel[:, :, 1] == 0)).sum(dim=(0, 2, 3)).to(torch.float64)


class ExchangeProbe:
    """Measures what the exchange costs the step (bench.py's `allreduce_exposed_ms`): HIP events on the CURRENT stream right before and
    right after the stream is made to wait for the pending collectives.  The first event completes when everything the step queued before
    the wait has run, the second when the collectives have landed too, so their distance is the time the compute stream stood still for
    the exchange -- 0 when the all-reduce was hidden behind the backward pass.  Host-side collectives (gloo rehearsal) block the host
    instead; their wall time is accumulated as `host_ms`.  Events are read once, after the timed region (`exposed_ms`)."""

    def __init__(self, use_events: bool):
        self.use_events = use_events
        self.pairs: List = []
        self.host_ms = 0.0
        self.steps = 0

    def before_wait(self):
        if not self.use_events:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def after_wait(self, e0) -> None:
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.pairs.append((e0, e1))

    def exposed_ms(self) -> float:
        """Total stall (ms) over the probed steps; call after a device synchronisation."""
        return float(sum(a.elapsed_time(b) for a, b in self.pairs)) + self.host_ms


def dp_train_step(plan, world: int, all_reduce: Optional[Callable] = None, run: Optional[Callable] = None, avg: bool = True,
                  run_slice: Optional[Callable] = None, probe: Optional[ExchangeProbe] = None) -> None:
    """One optimisation step of `plan` (RolloutTrainPlan built with world=world) whose x0 / y / masks are already loaded.

    all_reduce(tensor, op, stream=None) with op in {"sum", "avg"} reduces in place across ranks and returns None or a handle
    with .wait() (torch.distributed in production: torch_all_reduce); `stream` = the hipStream_t whose queued work the
    reduction must follow (None: the current stream).
    run(section_name) executes one plan section (default: plan.run on the current HIP stream); run_slice(name, a, b) executes
    ops [a, b) of a section (default: the lowered plan; the CPU tests pass the twin's interpreter for both)."""
    run = run or (lambda name: plan.run((name,)))
    if world == 1:
        for name in ("fwd", "fin", "bwd", "opt"):
            run(name)
        return
    assert all_reduce is not None
    if run_slice is None:
        run_slice = lambda name, a, b: rt.run_ops(plan._lowered[name][a:b])      # noqa: E731
    pending: List = []
    plan.gcnt.tensor.copy_(fluid_counts(plan))
    pending.append(all_reduce(plan.gcnt.tensor, "sum"))         # overlaps the forward pass
    run("fwd")
    _wait(pending, probe)
    run("fin")
    g = plan.pb.grads.tensor
    a = 0
    for end, (lo, hi), lane in plan.grad_buckets:
        run_slice("bwd", a, end)
        a = end
        if hi > lo:
            pending.append(all_reduce(g[lo:hi], "avg" if avg else "sum", stream=rt.lane_stream(lane) if (lane and g.is_cuda) else None))
    _wait(pending, probe)
    run("opt")
    if probe is not None:
        probe.steps += 1


def _wait(pending: List, probe: Optional[ExchangeProbe] = None) -> None:
    e0 = probe.before_wait() if probe is not None else None
    for h in pending:
        if h is not None:
            h.wait()                     # torch.distributed: the current stream waits for the collective's stream
    if probe is not None:
        probe.after_wait(e0)
    pending.clear()


def torch_all_reduce(dist, probe: Optional[ExchangeProbe] = None):
    """Adapter from torch.distributed (backend nccl = RCCL on ROCm, or gloo in the CPU tests) to dp_train_step's callback."""
    def f(t: torch.Tensor, op: str, stream: Optional[int] = None):
        if dist.get_backend() == "gloo":                      # CPU tests / rehearsal: no AVG, and device tensors go through the host
            t0 = time.perf_counter()
            if t.is_cuda:
                # `.cpu()` orders itself behind torch's CURRENT stream only.  A bucket whose last piece was written on a side lane
                # (UNPREP_WGRAD on the weight-gradient stream) must be complete before the host reads it, and nothing may still
                # be writing the range when the reduced values are copied back: drain that lane first.
                if stream is not None:
                    torch.cuda.ExternalStream(stream).synchronize()
                h = t.detach().cpu()
            else:
                h = t
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            if op == "avg":
                h.div_(dist.get_world_size())
            if t.is_cuda:
                t.copy_(h)
            if probe is not None:
                probe.host_ms += (time.perf_counter() - t0) * 1e3
            return None
        rop = dist.ReduceOp.AVG if op == "avg" else dist.ReduceOp.SUM
        if stream is None:
            return dist.all_reduce(t, op=rop, async_op=True)
        # RCCL's stream waits for what is queued on the CURRENT stream at call time: make that the lane that produced the bucket
        with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
            return dist.all_reduce(t, op=rop, async_op=True)
    return f
