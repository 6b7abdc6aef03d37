#!/usr/bin/env python3
"""Experiment: how much do MFMA-bound and HBM/VALU-bound kernels of the training step overlap when two half-batch steps run on two
streams at once?  Two independent 32-sample K4 plans on two streams against one 64-sample plan (same total work per round).
Upper bound for a micro-batched, two-lane training step.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import TrainStepper  # noqa: E402


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    one = TrainStepper(64, 128, 4, "bf16", 1, 0)
    for _ in range(3):
        one.step()
    ms_one = timed(one.step, n)
    del one
    torch.cuda.empty_cache()
    halves = [TrainStepper(32, 128, 4, "bf16", 1, r) for r in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]

    def seq():
        for h in halves:
            h.step()

    def conc():
        for h, s in zip(halves, streams):
            with torch.cuda.stream(s):
                h.step()

    for _ in range(3):
        seq()
    ms_seq = timed(seq, n)
    for _ in range(3):
        conc()
    ms_conc = timed(conc, n)
    print(json.dumps({"one_plan_b64_ms": round(ms_one, 3), "two_b32_sequential_ms": round(ms_seq, 3), "two_b32_concurrent_ms": round(ms_conc, 3),
                      "samples_per_s": {"b64": round(64e3 / ms_one, 1), "2xb32_seq": round(64e3 / ms_seq, 1), "2xb32_conc": round(64e3 / ms_conc, 1)}}))


if __name__ == "__main__":
    main()
