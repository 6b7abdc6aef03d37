#!/usr/bin/env python3
"""Debug: accuracy of the device GELU (fp32 APPLY kernel with identity statistics) against float64 erf on a dense grid."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from fluidnet_amd.engine import runtime as rt  # noqa: E402
from fluidnet_amd.engine.plan import PlanBuilder, Ref, lower  # noqa: E402
from helpers import tiny_cfg  # noqa: E402

H = W = 254
C = 8
pb = PlanBuilder(tiny_cfg(base_channels=16), 1, 16, 16, dtype="fp32", conv_impl="ref")
pb.bufs.clear()
pb.B = 1
raw = pb.new_buf("raw", (1, H + 2, W + 2, C), "T")
dst = pb.new_buf("dst", (1, H + 2, W + 2, C), "T")
st = pb.new_buf("st", (1, C, 2), "f32")
ga = pb.new_buf("ga", (C,), "f32")
be = pb.new_buf("be", (C,), "f32")
n = (H + 2) * (W + 2) * C
u = torch.linspace(-7, 7, n, dtype=torch.float64)
raw.tensor = u.float().view(raw.shape).cuda()
dst.tensor = torch.zeros(dst.shape, device="cuda")
s0 = torch.zeros(1, C, 2); s0[..., 1] = 1.0
st.tensor = s0.cuda(); ga.tensor = torch.ones(C, device="cuda"); be.tensor = torch.zeros(C, device="cuda")
ops = []
pb.op(ops, "APPLY", [1, H, W, C, C, 0, 8, 0, C, 0, rt.ACT["gelu"], rt.PAD["zeros"], 0, 0], f=[0.0],
      p=[Ref(raw), Ref(st), Ref(ga), Ref(be), None, Ref(dst)], tag="gelu")
rt.run_ops(lower(pb, ops))
torch.cuda.synchronize()
x = raw.tensor[:, 1:-1, 1:-1].double().cpu().flatten()
got = dst.tensor[:, 1:-1, 1:-1].double().cpu().flatten()
ref = 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))
err = (got - ref).abs()
i = int(err.argmax())
print("GELU fp32 device: max abs err %.3e at u=%.4f; rms %.3e; max rel err (|u|>0.1) %.3e" % (float(err.max()), float(x[i]), float((err ** 2).mean().sqrt()),
      float((err / ref.abs().clamp_min(1e-30))[x.abs() > 0.1].max())))
