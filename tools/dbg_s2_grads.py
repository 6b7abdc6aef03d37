#!/usr/bin/env python3
"""Debug: per-tensor gradient agreement of the two forms of the stride-2 convolutions (FNET_S2=1 / 0) on UNet_medium, K1, 128^2, bf16."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluidnet_oracle as orc  # noqa: E402
from fluidnet_amd.engine.train_plan import RolloutTrainPlan  # noqa: E402
from fluidnet_amd.models.unet import UNet, UNetConfig  # noqa: E402
from fluidnet_amd.training.physics_loss import PhysicsAwareLoss  # noqa: E402
from helpers import unpack  # noqa: E402

B, R = 2, 128
cfg = orc.medium_config(dropout=0.0)
crit = PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")
x0, y, mk = orc.synth_batch(B, 1, R, seed=202)
res = {}
for s2 in ("1", "0"):
    os.environ["FNET_S2"] = s2
    m = UNet(UNetConfig(**cfg.__dict__), precision="bf16")
    m.load_state_dict(orc.make_params(cfg, seed=7))
    m.to("cuda")
    p = RolloutTrainPlan(m.cfg, crit, B, R, R, 1, dtype="bf16", clip_norm=None)
    p.bind(torch.device("cuda"), m.flat_params, m.flat_grads)
    p.x0.tensor.copy_(torch.from_numpy(x0)); p.y.tensor.copy_(torch.from_numpy(y)); p.masks.tensor.copy_(torch.from_numpy(mk))
    p.run(("fwd", "fin", "bwd"))
    torch.cuda.synchronize()
    res[s2] = (p.pb, m.flat_grads.detach().double().cpu().clone(), p.loss_dict()["total"], p.tapes[0].pred.tensor.cpu().clone())
(pb1, g1, l1, p1), (pb0, g0, l0, p0) = res["1"], res["0"]
print("loss", l1, l0, "pred rel", float((p1 - p0).norm() / p0.norm()))
for k in pb1.shapes:
    a, b = unpack(pb1, g1, k).flatten(), unpack(pb0, g0, k).flatten()
    cos = float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))
    print(f"{k:32s} cos {cos:8.4f}  norm {float(a.norm()):10.4e} / {float(b.norm()):10.4e}")
