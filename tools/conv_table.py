#!/usr/bin/env python3
"""Table of the 3x3 convolution launches of one K1 training step (forward + data gradient) of a model config at a batch / resolution:
shape, executed GFLOP (the flattened GEMM over padded pixels), tile variant the launcher picks.  Host only."""
import sys
from collections import defaultdict

sys.path.insert(0, "oracle")
sys.path.insert(0, ".")
import fluidnet_oracle as orc  # noqa: E402
from fluidnet_amd.engine.plan import OPC  # noqa: E402
from fluidnet_amd.engine.train_plan import RolloutTrainPlan  # noqa: E402
from fluidnet_amd.models.unet import UNetConfig  # noqa: E402
from fluidnet_amd.training.physics_loss import PhysicsAwareLoss  # noqa: E402

B, R = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
cfg = UNetConfig(**orc.medium_config().__dict__)
inv = {v: k for k, v in OPC.items()}
crit = PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")
p = RolloutTrainPlan(cfg, crit, B, R, R, 1, dtype="bf16")
rows, tot = [], 0.0
for sec, recs in p.sections.items():
    for r in recs:
        if inv[r.code] == "CONV3X3":
            Npix, Wp, Cin, Cout = r.i[0], r.i[1], r.i[2], r.i[5]
            fl = 2.0 * Npix * 9 * Cin * Cout
            big = False
            if Cout > 32:
                tnb = 128 if Cout > 64 else 64
                big = ((Npix + 511) // 512) * ((Cout + tnb - 1) // tnb) >= 768
            TN = (128 if Cout > 64 else 64) if big else (64 if Cout > 32 else 32)
            rows.append((sec, r.tag, Wp - 2, Cin, Cout, fl / 1e9, f"<{TN},{8 if big else 4}>"))
            tot += fl
for r in rows:
    print("%-4s %-28s W=%-4d %4d->%-4d %7.2f GF %s" % r)
print("total GF", tot / 1e9)
d = defaultdict(float)
for r in rows:
    d[(r[2], r[3], r[4], r[-1])] += r[5]
for k, v in sorted(d.items(), key=lambda kv: -kv[1]):
    print(k, "%.1f GF  %.1f%%" % (v, 100 * v / tot * 1e9))
