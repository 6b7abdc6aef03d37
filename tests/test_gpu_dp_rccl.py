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


GLOO_CHILD = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import fluidnet_oracle as orc
    from fluidnet_amd.engine.train_plan import RolloutTrainPlan
    from fluidnet_amd.models.unet import UNet, UNetConfig
    from fluidnet_amd.training.dp import ExchangeProbe, dp_train_step, torch_all_reduce
    from fluidnet_amd.training.physics_loss import PhysicsAwareLoss
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    torch.cuda.set_device(0)
    B, R, K = 4, 64, 2
    cfg = orc.medium_config(dropout=0.0)
    crit = PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")
    x0, y, mk = orc.synth_batch(B, K, R, seed=91)           # BOTH ranks hold the same shard: the average is each rank's own gradient
    out = []
    for real in (True, False, False):
        m = UNet(UNetConfig(**cfg.__dict__), precision="bf16")
        m.load_state_dict(orc.make_params(cfg, seed=7)); m.to("cuda")
        p = RolloutTrainPlan(m.cfg, crit, B, R, R, K, dtype="bf16", world=2, seed=5, clip_norm=1.0)
        p.bind(torch.device("cuda"), m.flat_params, m.flat_grads)
        assert any(lane for _, _, lane in p.grad_buckets), p.grad_buckets      # buckets finished on the weight-gradient lane's stream
        probe = ExchangeProbe(False)
        if real:
            ar = torch_all_reduce(dist, probe)
        else:                                                # what a perfect 2-rank exchange of identical shards leaves: counts doubled
            def ar(t, op, stream=None):
                if op == "sum":
                    t.mul_(2)
        p.x0.tensor.copy_(torch.from_numpy(x0)); p.y.tensor.copy_(torch.from_numpy(y)); p.masks.tensor.copy_(torch.from_numpy(mk))
        p.set_adam_scalars(1e-3, 1)
        dp_train_step(p, 2, ar, probe=probe)
        torch.cuda.synchronize()
        assert not real or (probe.steps == 1 and probe.host_ms > 0)
        out.append((m.flat_params.detach().clone(), m.flat_grads.detach().clone(), p.loss_dict()["total"]))
        del p, m
    (p1, g1, l1), (p0, g0, l0), (p2, g2, l2) = out
    rel = lambda a, b: float((a - b).norm() / b.norm())
    floor, gd = rel(g2, g0), rel(g1, g0)
    print(f"GLOO-2RANK rank {rank} loss {l1} {l0} grad rel-L2 diff {gd} (two identical runs: {floor})", flush=True)
    # the sum of two atomics-noisy copies averaged: within a few times the run-to-run floor; a bucket read before its lane had
    # finished would miss whole tensors (rel-L2 of order 0.1 - 1)
    assert abs(l1 - l0) <= 1e-6 * abs(l0) and gd < max(1e-4, 10 * floor), (l1, l0, gd, floor)
    assert float((p1 - p0).abs().max()) <= 2e-3 * 1.01
    # and both ranks end with identical parameters (they averaged the same two gradients)
    h = p1.cpu(); ref = h.clone(); dist.broadcast(ref, 0)
    assert torch.equal(h, ref)
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"ok": True}), flush=True)
''')


def test_gloo_adapter_on_device_tensors_waits_for_the_lane(tmp_path):
    """torch_all_reduce's gloo branch on CUDA tensors (bench.py's rehearsal mode) must drain the side lane that wrote a bucket
    before the host reads it.  Two gloo ranks share this GPU with identical shards; the step must equal the same plan stepped with an
    exact stand-in exchange."""
    from fluidnet_amd.training import launch
    f = tmp_path / "gloo_child.py"
    f.write_text("ROOT = %r\n" % ROOT + GLOO_CHILD)
    code, line, why = launch.spawn_ranks([sys.executable, str(f)], 2, timeout_s=240)
    assert code == 0 and line is not None, why


def test_bench_gpus_2_runs_by_itself_over_gloo():
    """`python bench.py --gpus 2` with NO launcher on the command line spawns its ranks, and the line says what the
    collective layer saw.  Over gloo here (two ranks share the one GPU of this box: a flow test, not a scaling figure)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FNET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "8",
                        "--no-profile", "--spawn-timeout", "300"], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    o = json.loads(lines[0])
    ex = o["exchange"]
    assert o["n_gpus"] == 2 and o["steps"] == 4 and o["config"]["global_batch"] == 16 and o["config"]["parallelism"] == "dp2"
    assert ex["backend"] == "gloo" and ex["world"] == 2 and len(ex["ranks"]) == 2 and {q["rank"] for q in ex["ranks"]} == {0, 1}
    assert len({q["pid"] for q in ex["ranks"]}) == 2 and ex["allreduce_ms_per_step"] > 0 and ex["allreduce_exposed_ms"] > 0
    assert sum(b["elements"] for b in ex["buckets"]) * 4 == ex["gradient_bytes"]
    assert ex["rank_samples_per_s_min"] <= ex["rank_samples_per_s_max"] and o["value"] <= 2 * ex["rank_samples_per_s_max"] * 1.001
