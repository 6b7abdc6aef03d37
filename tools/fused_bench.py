#!/usr/bin/env python3
"""Micro-benchmark of the fused IN_STATS + APPLY launch (k_stats_apply) against the three separate launches, per layer class of UNet_medium
K4 B64 128^2 (GPU box): python tools/fused_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import torch

from fluidnet_amd.engine import runtime as rt
from fluidnet_amd.engine.plan import lower
from test_gpu_fused_small import _mk

for (H, C) in ((128, 32), (128, 64), (64, 64), (64, 128), (32, 128), (32, 256), (16, 256)):
    pb, ops, st, dst = _mk(64, H, H, C, C, 0, False, "gelu", 1, 0.1)
    arr = lower(pb, ops)
    out = []
    for name, chunks in (("fused", [arr]), ("separate", [arr[0:1], arr[1:2]])):
        for _ in range(3):
            for c in chunks:
                rt.run_ops(c)
        t = rt.Timer()
        t.start()
        for _ in range(50):
            for c in chunks:
                rt.run_ops(c)
        t.stop()
        out.append(f"{name} {t.elapsed_ms() / 50 * 1e3:6.1f} us")
    mb = 64 * (H + 2) * (H + 2) * C * 2 / 1e6
    print(f"{H}x{H} C={C} ({mb:5.1f} MB per tensor): " + " | ".join(out), flush=True)

from test_gpu_fused_small import _mk_bwd

print("adjoint pair (INACT_BWD_REDUCE + INACT_BWD_APPLY, single source):")
for (H, C) in ((128, 64), (64, 64), (64, 128), (32, 128), (32, 256), (16, 256)):
    pb, ops, S, draw = _mk_bwd(64, H, H, C, C, 0, 1, "gelu", 1, 0.1)
    arr = lower(pb, ops)
    out = []
    for name, chunks in (("fused", [arr]), ("separate", [arr[0:1], arr[1:2]])):
        for _ in range(3):
            for c in chunks:
                rt.run_ops(c)
        t = rt.Timer()
        t.start()
        for _ in range(50):
            for c in chunks:
                rt.run_ops(c)
        t.stop()
        out.append(f"{name} {t.elapsed_ms() / 50 * 1e3:6.1f} us")
    print(f"{H}x{H} C={C}: " + " | ".join(out), flush=True)
