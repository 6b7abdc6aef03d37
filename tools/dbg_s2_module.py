#!/usr/bin/env python3
"""Debug: per-tensor gradients through the nn.Module (autograd) path, stride-2 forms FNET_S2=1 / 0 against the oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluidnet_oracle as orc  # noqa: E402
from fluidnet_amd.models.unet import UNet, UNetConfig  # noqa: E402
from fluidnet_amd.training.physics_loss import PhysicsAwareLoss  # noqa: E402

B, R = 2, 128
cfg = orc.medium_config(dropout=0.0)
crit = PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")
params = orc.make_params(cfg, seed=7)
x0, y, _ = orc.synth_batch(B, 1, R, seed=202)
p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
xo = torch.from_numpy(x0).requires_grad_(True)
oout = orc.unet_forward(p, cfg, xo, training=True)
lc = orc.LossConfig(mse_weight=1.0, divergence_weight=0.01, gradient_weight=0.001, emitter_weight=0.01, grid_spacing=2.0 / R, padding_mode="replicate", stencil_mode="forward")
oloss, _ = orc.physics_loss(oout, torch.from_numpy(y[:, 0]), xo, lc)
oloss.backward()
res = {}
for s2 in ("1", "0"):
    os.environ["FNET_S2"] = s2
    m = UNet(UNetConfig(**cfg.__dict__), precision="bf16", conv_impl="mfma")
    m.load_state_dict(params)
    m.to("cuda")
    x = torch.from_numpy(x0).cuda().requires_grad_(True)
    out = m(x)
    loss, d = crit(out, torch.from_numpy(y[:, 0]).cuda(), x)
    loss.backward()
    res[s2] = {k: v.grad.detach().double().cpu().flatten() for k, v in m.named_parameters()}
for k in res["1"]:
    o = p[k].grad.double().flatten()
    a, b = res["1"][k], res["0"][k]
    c = lambda u, v: float(torch.dot(u, v) / (u.norm() * v.norm() + 1e-300))  # noqa: E731
    print(f"{k:32s} s2-vs-oracle {c(a, o):8.4f}  full-vs-oracle {c(b, o):8.4f}  norms {float(a.norm()):10.3e} {float(b.norm()):10.3e} {float(o.norm()):10.3e}")
