"""BASELINE.json configs at FULL size -- UNet_medium, K = 4, 64 samples, 128x128, bf16 MFMA plan, through the C ABI -- checked
through size-independent properties (the oracle cannot run 64 x 4 applications at 128^2 in test time, so it checks a slice):

  * the forward pass is bit-reproducible run to run;
  * a sample's predictions do not depend on what else is in the batch or where it sits in it: reversing the batch
    reverses the predictions bit for bit; samples 0..1 agree with a 2-sample plan (different statistics chunking) and,
    application by application, with the fp32 CPU oracle (bf16 tolerance of tests/test_gpu_api.py, stated below);
  * data parallelism: two 32-sample shards driven through dp_train_step's exchange (fluid counts, then gradients) give the
    64-sample gradient and loss;
  * a few optimisation steps on a fixed batch lower the loss and keep everything finite.
"""
import sys

import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.data.synthetic import synth_rollout_batch
from fluidnet_amd.engine.train_plan import RolloutTrainPlan
from fluidnet_amd.models.unet import UNet, UNetConfig
from fluidnet_amd.training.dp import dp_train_step
from fluidnet_amd.training.physics_loss import PhysicsAwareLoss

pytestmark = pytest.mark.gpu
B, K, R = 64, 4, 128


def _crit():
    return PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")


def _model(dropout):
    cfg = orc.medium_config(dropout=dropout)
    m = UNet(UNetConfig(**cfg.__dict__), precision="bf16")
    m.load_state_dict(orc.make_params(cfg, seed=7))
    return m.to("cuda"), cfg


def _plan(model, nb, batch, lo=0, world=1, grads=None, seed=99):
    p = RolloutTrainPlan(model.cfg, _crit(), nb, R, R, K, dtype="bf16", world=world, weight_decay=1.10, clip_norm=1.0, seed=seed)
    p.bind(torch.device("cuda"), model.flat_params, model.flat_grads if grads is None else grads)
    x0, y, mk = batch
    p.x0.tensor.copy_(torch.from_numpy(x0[lo:lo + nb]))
    p.y.tensor.copy_(torch.from_numpy(y[lo:lo + nb]))
    p.masks.tensor.copy_(torch.from_numpy(mk[lo:lo + nb]))
    p.set_adam_scalars(3e-4, 1)
    return p


@pytest.fixture(scope="module")
def batch():
    return synth_rollout_batch(B, K, R, seed=4321)


def test_forward_bit_reproducible_and_batch_independent(batch):
    model, cfg = _model(0.0)
    big = _plan(model, B, batch)
    big.run(("fwd", "fin"))
    preds1 = [t.pred.tensor.clone() for t in big.tapes]
    loss1 = big.loss_dict()
    big.run(("fwd", "fin"))
    for k in range(K):
        assert torch.equal(preds1[k], big.tapes[k].pred.tensor), f"rollout step {k} differs between two runs"
    for key, v in big.loss_dict().items():        # the loss sums are fp64 atomics: reproducible to rounding, not bitwise
        assert abs(v - loss1[key]) <= 1e-6 * max(1.0, abs(v)), key
    assert all(np.isfinite(v) for v in loss1.values())

    # permutation equivariance, bitwise: with the batch reversed, sample i sits at position 63-i, its pixels fall into other
    # tiles of the flattened pixel axis and other blocks of the elementwise kernels, but every per-pixel and per-(n,c) sum keeps
    # its order, so each sample's K predictions must be identical bit for bit
    rev = _plan(model, B, tuple(np.ascontiguousarray(a[::-1]) for a in batch))
    rev.run(("fwd",))
    for k in range(K):
        assert torch.equal(rev.tapes[k].pred.tensor.flip(0), preds1[k]), f"rollout step {k} depends on the batch position"

    # batch-size independence (first application): the 2-sample plan chunks the InstanceNorm sums differently (fp32 summation
    # order, then bf16 rounding flips), and the random-init UNet_medium amplifies a one-off bf16 rounding of its inputs to
    # ~3e-2 rel-L2 at its output (the "conditioning floor" of tests/test_gpu_api.py); a batch-dependent bug would be O(1)
    small = _plan(model, 2, batch)
    small.run(("fwd",))
    rel = float((preds1[0][:2] - small.tapes[0].pred.tensor).norm() / small.tapes[0].pred.tensor.norm())
    print(f"[full size] 64-sample plan vs 2-sample plan, first application: rel-L2 {rel:.3e}")
    assert rel < 4e-2, rel

    # oracle slice: every one of the K applications of samples 0..1 against the fp32 CPU oracle applied to the SAME input (the
    # GPU's own previous prediction -- chaining the oracle on its own outputs would compound the conditioning, see above);
    # bf16 bar as in tests/test_gpu_api.py: rel-L2 <= 8e-2
    x0, y, mk = batch
    p = orc.make_params(cfg, seed=7)
    cur, prev = torch.from_numpy(x0[:2, :3]), torch.from_numpy(x0[:2, 3:4])
    with torch.no_grad():
        for k in range(K):
            inp = torch.cat([cur, prev, torch.from_numpy(mk[:2, k, 0:1]), torch.from_numpy(mk[:2, k, 1:2])], 1)
            pred = orc.unet_forward(p, cfg, inp, training=True)
            got = preds1[k][:2].cpu()
            rel = float((got - pred).norm() / pred.norm())
            print(f"[full size] application {k}: 64-sample plan vs fp32 oracle on the same input rel-L2 {rel:.3e}")
            assert rel < 8e-2, (k, rel)
            prev, cur = got[:, 0:1], got


@pytest.mark.parametrize("k_steps", [1, 4])
def test_two_shards_equal_the_global_batch(batch, k_steps, monkeypatch):
    monkeypatch.setattr(sys.modules[__name__], "K", k_steps)
    batch = (batch[0], batch[1][:, :k_steps], batch[2][:, :k_steps])
    model, _ = _model(0.0)
    glob = _plan(model, B, batch)
    glob.run(("fwd", "fin", "bwd"))
    g_ref = model.flat_grads.clone()
    loss_ref = glob.loss_dict()

    half = B // 2
    grads = [torch.zeros_like(model.flat_grads) for _ in range(2)]
    shards = [_plan(model, half, batch, lo=r * half, world=2, grads=grads[r]) for r in range(2)]
    # dp_train_step's protocol with the two ranks interleaved on one GPU; "opt" is skipped (it would step the shared weights)
    for s in shards:
        s.run(("fwd",))
    cnt = sum(torch.stack([q.tensor[5] for q in s.sums]) for s in shards)
    for s in shards:
        s.gcnt.tensor.copy_(cnt)
        s.run(("fin", "bwd"))
    g = (grads[0] + grads[1]) / 2
    rel = float((g - g_ref).norm() / g_ref.norm())
    cos = float(torch.dot(g, g_ref) / (g.norm() * g_ref.norm()))
    print(f"[full size] K={k_steps}: 2 x 32-sample shards vs 64-sample batch: gradient rel-L2 {rel:.3e} cos {cos:.5f} "
          f"norms {float(g.norm()):.3f} / {float(g_ref.norm()):.3f}")
    # each shard's loss dict uses the global fluid count, so their mean is the global-batch loss (forward only: tight)
    for key in ("mse", "divergence", "gradient", "emitter", "total"):
        mean = 0.5 * (shards[0].loss_dict()[key] + shards[1].loss_dict()[key])
        assert abs(mean - loss_ref[key]) < 2e-3 * max(1.0, abs(loss_ref[key])), (key, mean, loss_ref[key])
    if k_steps == 1:
        # one application: the shard plans differ from the 64-sample plan only in how the InstanceNorm sums are chunked
        assert rel < 5e-2 and cos > 0.998, (rel, cos)
    else:
        # BPTT through 4 applications of the random-init net is chaotic (gradient norm 2 -> 33, a 2.5e-2 forward perturbation
        # decorrelates the last applications), so element-wise agreement between two differently chunked plans is not a
        # meaningful bar here; the exact K-step equivalence is asserted with identical arithmetic in tests/test_dp_gloo.py
        assert 0.8 < float(g.norm() / g_ref.norm()) < 1.25


def test_optimisation_steps_lower_the_loss(batch):
    model, _ = _model(0.1)
    plan = _plan(model, B, batch)
    losses = []
    for step in range(1, 7):
        plan.set_adam_scalars(1e-3, step)
        plan.set_dropout_seed(1000 + step)
        dp_train_step(plan, 1)
        losses.append(plan.loss_dict()["total"])
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < 0.9 * losses[0], losses
    assert bool(torch.isfinite(model.flat_params).all())
