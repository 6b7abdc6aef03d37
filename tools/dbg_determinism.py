#!/usr/bin/env python3
"""Debug: run-to-run agreement of the gradient of one K-step plan, with and without the weight-gradient lane (FNET_WGRAD_LANE) and the other
plan-level options.  usage: dbg_determinism.py [B R K]"""
import os
import sys

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

B, R, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 64, 2)
cfg = orc.medium_config(dropout=0.1)
crit = PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")
x0, y, mk = orc.synth_batch(B, K, R, seed=91)
for env in ({"FNET_WGRAD_LANE": "0", "dtype": "fp32", "impl": "ref"}, {"FNET_WGRAD_LANE": "0", "dtype": "bf16", "impl": "ref"},
            {"FNET_WGRAD_LANE": "0", "dtype": "bf16", "impl": "auto"}):
    for k in ("FNET_WGRAD_LANE", "FNET_KEEP_G", "FNET_S2"):
        os.environ.pop(k, None)
    os.environ.update({k: v for k, v in env.items() if k.startswith("FNET")})
    m = UNet(UNetConfig(**cfg.__dict__), precision=env["dtype"], conv_impl=env["impl"])
    m.load_state_dict(orc.make_params(cfg, seed=7))
    m.to("cuda")
    p = RolloutTrainPlan(m.cfg, crit, B, R, R, K, dtype=env["dtype"], conv_impl=env["impl"], seed=5, clip_norm=1.0)
    p.bind(torch.device("cuda"), m.flat_params, m.flat_grads)
    gs = []
    for rep in range(4):
        p.x0.tensor.copy_(torch.from_numpy(x0)); p.y.tensor.copy_(torch.from_numpy(y)); p.masks.tensor.copy_(torch.from_numpy(mk))
        p.run(("fwd", "fin", "bwd"))
        torch.cuda.synchronize()
        gs.append(m.flat_grads.detach().double().clone())
    rel = [float((g - gs[0]).norm() / gs[0].norm()) for g in gs[1:]] + [float((gs[3] - gs[2]).norm() / gs[2].norm())]
    worst = []
    for k in p.pb.shapes:
        a, b = unpack(p.pb, gs[1], k), unpack(p.pb, gs[0], k)
        d = float((a - b).norm() / (b.norm() + 1e-30))
        if float(b.norm()) > 1e-3 * float(gs[0].norm()):
            worst.append((d, k))
    worst.sort(reverse=True)
    print(env, "rel-L2 of runs 2-4 against run 1, then run 4 against run 3:", ["%.2e" % v for v in rel], "worst tensors:", [(f"{d:.1e}", k) for d, k in worst[:4]], flush=True)
    del p, m
    torch.cuda.empty_cache()
