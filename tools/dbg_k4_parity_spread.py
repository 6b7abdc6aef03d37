"""How the trained-regime numbers of tests/test_gpu_k4_parity.py spread over repetitions of the (atomics-noisy, hence not bit-reproducible)
200-step warm-up: whole-gradient cosine / rel-L2 of bf16 vs fp32, gradient norm, absolute error, and the batch-to-batch spread of the fp32
gradient (two held-out batches) as the scale that does not vanish near a minimum.  Usage: python tools/dbg_k4_parity_spread.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

import test_gpu_k4_parity as T
from fluidnet_amd.data.synthetic import synth_rollout_batch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
batch = synth_rollout_batch(T.B, 4, T.R, seed=4321)
h1 = synth_rollout_batch(T.B, 4, T.R, seed=8765)
h2 = synth_rollout_batch(T.B, 4, T.R, seed=9876)
for rep in range(reps):
    trained = T._warm_up(batch)
    for K in (1, 2, 4):
        a32 = T._run(T._model("fp32", trained), "fp32", K, h1)
        b32 = T._run(T._model("fp32", trained), "fp32", K, h2)
        a16 = T._run(T._model("bf16", trained), "bf16", K, h1)
        r = T._compare(a32, a16, K)
        err = float((a16["grads"] - a32["grads"]).norm())
        spread = float((b32["grads"] - a32["grads"]).norm())
        again = T._run(T._model("fp32", trained), "fp32", K, h1)
        rr = float((again["grads"] - a32["grads"]).norm())
        print(f"rep {rep} K{K}: cos {r['grad_cos']:.4f} rel {r['grad_rel']:.3f} gnorm32 {r['gnorm32']:.4f} err {err:.5f} batch-spread {spread:.5f} "
              f"err/spread {err / spread:.3f} fp32 rerun {rr:.2e} loss {r['loss32']:.5f} pred_rel {r['pred_rel'][-1]:.4f}", flush=True)
