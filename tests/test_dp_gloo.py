"""Data-parallel path on CPU: two gloo ranks, each running its shard of the native training plan through the CPU twin
(tests/sim_ops.py), must reproduce the single-process global-batch step (losses, gradient, parameters after clip + Adam).
Shards are built with DIFFERENT fluid-cell counts so that the global-denominator exchange (fluidnet_amd/training/dp.py,
step 1) is actually exercised: mean-of-shard-losses would differ."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fluidnet_oracle as orc
from fluidnet_amd.engine.train_plan import RolloutTrainPlan
from fluidnet_amd.training.dp import dp_train_step, torch_all_reduce
from helpers import loss_c2, pack_params, tiny_cfg
from sim_ops import Sim

B, R, K = 4, 16, 2


def _batch():
    x0, y, mk = orc.synth_batch(B, K, R, seed=321)
    mk[2:, :, 1] = 0.0            # second shard: no colliders -> more fluid cells than the first shard
    mk[3, :, 0] = 0.0
    x0[:, 4:6] = mk[:, 0]
    return x0, y, mk


def _plan(b, world):
    cfg, lc = tiny_cfg(), loss_c2(R)
    params = orc.make_params(cfg, seed=9)
    plan = RolloutTrainPlan(cfg, lc, b, R, R, K, dtype="fp32", conv_impl="ref", world=world, clip_norm=0.05, seed=0)
    plan.bind("cpu", pack_params(plan.pb, params), torch.zeros(plan.pb.n_params_padded))
    plan.set_adam_scalars(3e-3, 1)
    return plan


def _load(plan, x0, y, mk):
    plan.x0.tensor.copy_(torch.from_numpy(x0))
    plan.y.tensor.copy_(torch.from_numpy(y))
    plan.masks.tensor.copy_(torch.from_numpy(mk))


def _worker(rank, world, port, outdir):
    torch.set_num_threads(2)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x0, y, mk = _batch()
    sl = slice(rank * (B // world), (rank + 1) * (B // world))
    plan = _plan(B // world, world)
    _load(plan, x0[sl], y[sl], mk[sl])
    sim = Sim(plan.pb.dtype)
    dp_train_step(plan, world, torch_all_reduce(dist), run=lambda name: sim.run(plan.sections[name]))
    # logged losses: average of the shard values (each shard already used the global denominator / world rescale)
    acc = plan.accum.tensor.clone()
    dist.all_reduce(acc)
    acc /= world
    if rank == 0:
        np.savez(os.path.join(outdir, "dp.npz"), grads=plan.pb.grads.tensor.numpy(), params=plan.pb.params.tensor.numpy(), acc=acc.numpy())
    dist.destroy_process_group()


def test_dp2_gloo_equals_global_batch():
    x0, y, mk = _batch()
    ref = _plan(B, 1)
    _load(ref, x0, y, mk)
    sim = Sim(ref.pb.dtype)
    dp_train_step(ref, 1, None, run=lambda name: sim.run(ref.sections[name]))
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, 29500 + os.getpid() % 500, d), nprocs=2, join=True)
        got = np.load(os.path.join(d, "dp.npz"))
    g_ref = ref.pb.grads.tensor.numpy()
    assert np.linalg.norm(got["grads"] - g_ref) < 2e-5 * np.linalg.norm(g_ref)
    # MSE / emitter terms are plain means (shard-additive); divergence / gradient use the global fluid count
    a_ref = ref.accum.tensor.numpy()
    assert np.allclose(got["acc"][:5], a_ref[:5], rtol=2e-5, atol=1e-7), (got["acc"][:5], a_ref[:5])
    # Adam divides by sqrt(v) + 1e-8: elements whose clipped gradient sits at the 1e-8 noise level take arbitrary steps of up
    # to lr in any implementation (see tests/test_plan_sim.py), so only well-conditioned elements are matched tightly
    gn = float(np.sqrt((g_ref.astype(np.float64) ** 2).sum()))
    coef = min(1.0, 0.05 / (gn + 1e-6))
    d = np.abs(got["params"] - ref.pb.params.tensor.numpy())
    good = np.abs(g_ref * coef) > 1e-5
    assert d.max() <= 3e-3 * 1.01 and good.sum() > 1000 and d[good].max() < 3e-5, (d.max(), d[good].max())
    # sanity: the shards really have different fluid counts (otherwise the test would not exercise the exchange)
    fluid = ((mk[:, :, 0] == 0) & (mk[:, :, 1] == 0)).sum(axis=(1, 2, 3))
    assert fluid[:2].sum() != fluid[2:].sum()
