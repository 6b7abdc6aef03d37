"""Lock-step parity harness (test infrastructure): every op of a plan is executed once by the CPU twin
(tests/sim_ops.py) and once by the HIP library through the C ABI (fnet_plan_run on a one-op plan) from IDENTICAL
inputs; outputs are compared and the GPU buffers are then re-synchronised from the twin, so a failure names the
exact kernel and cannot be masked or amplified by earlier ones."""
from __future__ import annotations

import numpy as np
import torch

from fluidnet_amd.engine import runtime as rt
from fluidnet_amd.engine.plan import lower
from sim_ops import Sim

OUT_SLOTS = {
    "PACK_INPUT": [6], "CONV3X3": [3], "WGRAD": [2], "IN_STATS": [1], "APPLY": [5], "UPSAMPLE2X": [1], "SUBSAMPLE_BIAS": [2],
    "PAD_RING": [0], "HEAD_FWD": [3, 4], "LOSS_REDUCE": [5], "LOSS_FINALIZE": [2, 3], "LOSS_GRAD": [9], "HEAD_BWD": [5, 6, 7],
    "INACT_BWD_REDUCE": [6, 7, 8, 9], "INACT_BWD_APPLY": [6], "AFFINE_GRAD": [1, 2], "AFFINE_GRAD_MULTI": [1], "FOLD_SUM": [2], "POOL2X": [1], "POOL2X_BWD": [2], "DEPTH2SPACE": [2], "SPACE2DEPTH": [1], "NORM_COMBINE": [0, 1, 2], "NORM_COMBINE_BWD": [2], "BIAS_GRAD": [1],
    "ZERO_INSERT": [1], "UPSAMPLE2X_BWD": [1], "UNPACK_GRAD": [1], "ZERO": [0], "PREP_WEIGHTS": [1, 2], "GRAD_NORM": [1],
    "ADAM": [0, 2, 3], "COPY_PLANES": [1], "UNPREP_WGRAD": [1], "STENCIL": [2, 3], "NOP": [], "EVENT_RECORD": [], "EVENT_WAIT": [],
}


def tol_for(buf, plan_dtype):
    if buf.kind == "f64":
        return 1e-5            # the twin sums in fp32, the kernel in fp64
    if buf.kind == "T" and plan_dtype == rt.BF16:
        return 1.2e-2          # one bf16 ulp (2^-8) of the largest element, with margin
    return 3e-4


def lockstep(pb, recs, device="cuda", verbose=False):
    """Returns a list of failure strings (empty = every op matched).  Buffers must be bound to CPU tensors."""
    cpu = {id(b): b.tensor for b in pb.bufs}
    gpu = {id(b): b.tensor.to(device) for b in pb.bufs if b.parent is None}
    for b in pb.bufs:                      # slices of other buffers stay slices on the device side
        if b.parent is not None:
            par, off = b.parent
            gpu[id(b)] = gpu[id(par)][off:off + b.numel].view(b.shape)
    sim = Sim(pb.dtype)
    failures = []
    for k, r in enumerate(recs):
        name = rt.OPNAME[r.code]
        # GPU run of this single op
        for b in pb.bufs:
            b.tensor = gpu[id(b)]
        arr = lower(pb, [r])
        try:
            rt.run_ops(arr)
            torch.cuda.synchronize()
        except Exception as e:  # argument check failure etc.
            failures.append(f"op#{k} {name} [{r.tag}] raised {e}")
            for b in pb.bufs:
                b.tensor = cpu[id(b)]
            sim.run([r])
            for j in OUT_SLOTS[name]:
                if j < len(r.p) and r.p[j] is not None:
                    gpu[id(r.p[j].buf)].copy_(cpu[id(r.p[j].buf)])
            continue
        # CPU twin
        for b in pb.bufs:
            b.tensor = cpu[id(b)]
        sim.run([r])
        for j in OUT_SLOTS[name]:
            if j >= len(r.p) or r.p[j] is None:
                continue
            b = r.p[j].buf
            want = cpu[id(b)].double()
            got = gpu[id(b)].cpu().double()
            scale = float(want.abs().max()) + 1e-30
            bad = ~torch.isfinite(got)
            err = float((got - want).abs().max()) / scale if not bad.any() else float("inf")
            if verbose:
                print(f"op#{k:4d} {name:18s} {r.tag:28s} slot{j} err={err:.3e} scale={scale:.3e}")
            if not (err <= tol_for(b, pb.dtype)):
                failures.append(f"op#{k} {name} [{r.tag}] slot{j} buf={b.name} shape={b.shape} rel-max-err={err:.3e} (scale {scale:.3e})")
            gpu[id(b)].copy_(cpu[id(b)])      # re-synchronise
    for b in pb.bufs:
        b.tensor = cpu[id(b)]
    return failures
