"""The RCCL leg of the data-parallel step on the one GPU this box has: a 1-rank `nccl` process group (RCCL accepts a single-rank communicator;
two ranks on one device it refuses) driving dp_train_step's real exchange code -- asynchronous all-reduces, the gradient buckets enqueued
behind the weight-gradient lane through torch.cuda.ExternalStream, the waits before clip + Adam -- on a plan built for world = 2.  With one rank
SUM and AVG are the identity, so the step must equal the same plan stepped with an identity callback that does nothing asynchronous.
Runs in a child process under a hard time limit: a communicator that cannot initialise must fail this test, not hang the suite.
(What it cannot show is a second GPU; the 2-rank arithmetic is tests/test_dp_gloo.py.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import fluidnet_oracle as orc
    from fluidnet_amd.engine.train_plan import RolloutTrainPlan
    from fluidnet_amd.models.unet import UNet, UNetConfig
    from fluidnet_amd.training.dp import dp_train_step, torch_all_reduce
    from fluidnet_amd.training.physics_loss import PhysicsAwareLoss
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    B, R, K = 4, 64, 2
    cfg = orc.medium_config(dropout=0.1)
    crit = PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")
    x0, y, mk = orc.synth_batch(B, K, R, seed=91)
    out = []
    for real in (True, False, False):
        m = UNet(UNetConfig(**cfg.__dict__), precision="bf16")
        m.load_state_dict(orc.make_params(cfg, seed=7)); m.to("cuda")
        p = RolloutTrainPlan(m.cfg, crit, B, R, R, K, dtype="bf16", world=2, seed=5, clip_norm=1.0)
        p.bind(torch.device("cuda"), m.flat_params, m.flat_grads)
        assert len(p.grad_buckets) >= 3 and any(lane for _, _, lane in p.grad_buckets), p.grad_buckets
        ar = torch_all_reduce(dist) if real else (lambda t, op, stream=None: None)
        p.x0.tensor.copy_(torch.from_numpy(x0)); p.y.tensor.copy_(torch.from_numpy(y)); p.masks.tensor.copy_(torch.from_numpy(mk))
        p.set_adam_scalars(1e-3, 1)
        dp_train_step(p, 2, ar)
        torch.cuda.synchronize()
        out.append((m.flat_params.detach().clone(), m.flat_grads.detach().clone(), p.loss_dict()["total"], float(torch.sqrt(p.normsq.tensor[0]))))
        del p, m
    (p1, g1, l1, n1), (p0, g0, l0, n0), (p2, g2, l2, n2) = out
    rel = lambda a, b: float((a - b).norm() / b.norm())
    floor = rel(g2, g0)                   # two identical runs: the weight-gradient reduction and the norm sums end in float atomics
    gd = rel(g1, g0)
    print("RCCL-1RANK loss", l1, l0, "grad rel-L2 diff", gd, "(two identical runs:", floor, ") norms", n1, n0, "param max abs diff", float((p1 - p0).abs().max()))
    nfloor = abs(n2 - n0) / n0            # the clip norm of two identical runs moves with that noise too (1.4e-4 seen)
    assert l1 == l0 and abs(n1 - n0) < max(1e-3, 10 * nfloor) * n0 and gd < max(1e-4, 10 * floor), (l1, l0, n1, n0, gd, floor, nfloor)
    # one Adam step: every element moves by at most lr whatever its gradient, so rounding-level gradient noise is bounded by 2 lr
    assert float((p1 - p0).abs().max()) <= 2e-3 * 1.01
    dist.destroy_process_group()
    print("RCCL-1RANK OK")
''')


def test_bucketed_exchange_through_rccl_single_rank():
    import socket
    with socket.socket() as sk:                 # a free rendezvous port on this host
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], env=env, capture_output=True, text=True, timeout=240)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "RCCL-1RANK OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
