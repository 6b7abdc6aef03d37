"""BASELINE.json configs 4 and 5 under -m gpu.

c4  "Higher-resolution UNet K2 training at 256x256 and 512x512 (deeper encoder)": UNet_medium with depth 4 at 256^2 and depth 5 at
    512^2, K = 2, 2 samples, through the native training plan (C ABI):
      * fp32 plan (FMA convolutions): the first application's prediction against the fp32 CPU oracle on the same input, and the
        first-step loss terms against the oracle's physics loss (max-abs 2e-3 / 1e-4 rel: 5-6 resolution levels of fp32 rounding);
      * bf16 MFMA plan: both applications against the oracle applied to the GPU's own input (bf16 bar 8e-2 rel-L2, as
        tests/test_gpu_api.py), forward bit-reproducible, batch reversal reverses the predictions bit for bit, backward finite
        and non-zero for every parameter tensor, one clip + Adam step keeps the weights finite.
c5  "Autoregressive inference rollout of 600 frames at 256x256 ... batched 64 scenes per GPU, hipGraph-captured step"
    (ml/scripts/simple_infer.py:226-260): 64 scenes with different masks, bf16, one hipGraph replay per frame:
      * frames 1-4 of scenes 0-1 against the oracle applied to the same state (bf16 bar);
      * 600 frames stay finite and inside the clamp range;
      * scene permutation equivariance, bit for bit, after all 600 frames (no dependence on the position in the batch).
"""
import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.data.synthetic import synth_rollout_batch
from fluidnet_amd.engine.rollout_plan import RolloutInferPlan
from fluidnet_amd.engine.train_plan import RolloutTrainPlan
from fluidnet_amd.models.unet import UNet, UNetConfig
from fluidnet_amd.training.physics_loss import PhysicsAwareLoss
from helpers import unpack

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _crit(R):
    return PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")


def _lc(R):
    return orc.LossConfig(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")


def _plan(model, precision, R, K, batch, reverse=False):
    x0, y, mk = (np.ascontiguousarray(a[::-1]) for a in batch) if reverse else batch
    p = RolloutTrainPlan(model.cfg, _crit(R), x0.shape[0], R, R, K, dtype=precision, conv_impl="ref" if precision == "fp32" else "auto",
                         weight_decay=1.10, clip_norm=1.0)
    p.bind(torch.device(DEV), model.flat_params, model.flat_grads)
    p.x0.tensor.copy_(torch.from_numpy(x0))
    p.y.tensor.copy_(torch.from_numpy(y))
    p.masks.tensor.copy_(torch.from_numpy(mk))
    p.set_adam_scalars(3e-4, 1)
    return p


@pytest.mark.parametrize("R,depth", [(256, 4), (512, 5)])
def test_c4_k2_high_resolution_deeper_encoder(R, depth):
    K, B = 2, 2
    cfg = orc.medium_config(dropout=0.0, depth=depth)
    params = orc.make_params(cfg, seed=7)
    batch = synth_rollout_batch(B, K, R, seed=77)
    x0, y, mk = batch

    # the oracle's first application on sample 0 (CPU fp32) and its loss terms
    with torch.no_grad():
        xin = torch.from_numpy(x0[:1])
        o0 = orc.unet_forward(params, cfg, xin, training=True)
        _, od = orc.physics_loss(o0, torch.from_numpy(y[:1, 0]), xin, _lc(R))

    # fp32 plan on sample 0 alone (the masked means of the loss are batch-wide: same batch as the oracle's)
    m32 = UNet(UNetConfig(**cfg.__dict__), precision="fp32", conv_impl="ref")
    m32.load_state_dict(params)
    m32.to(DEV)
    p32 = _plan(m32, "fp32", R, 1, tuple(a[:1, :1] if a.ndim == 5 else a[:1] for a in batch))
    p32.run(("fwd", "fin"))
    err = float((p32.tapes[0].pred.tensor.cpu() - o0).abs().max())
    d32 = p32.loss_dict()
    print(f"[c4 {R}^2 depth {depth}] fp32 plan vs oracle: pred max-abs {err:.2e}; loss {d32['total']:.6f} vs {float(od['total']):.6f}")
    assert err < 2e-3
    for k in ("mse", "divergence", "gradient", "emitter", "total"):
        assert abs(d32[k] - float(od[k])) < 1e-4 * max(1.0, abs(float(od[k]))), (k, d32[k], float(od[k]))
    del p32, m32
    torch.cuda.empty_cache()

    # bf16 MFMA plan, K = 2, 2 samples
    m = UNet(UNetConfig(**cfg.__dict__), precision="bf16")
    m.load_state_dict(params)
    m.to(DEV)
    p = _plan(m, "bf16", R, K, batch)
    p.run(("fwd", "fin"))
    preds = [t.pred.tensor.clone() for t in p.tapes]
    p.run(("fwd", "fin"))
    for k in range(K):
        assert torch.equal(preds[k], p.tapes[k].pred.tensor), f"application {k} not bit-reproducible"
    cur, prev = torch.from_numpy(x0[:1, :3]), torch.from_numpy(x0[:1, 3:4])
    with torch.no_grad():
        for k in range(K):
            inp = torch.cat([cur, prev, torch.from_numpy(mk[:1, k, 0:1]), torch.from_numpy(mk[:1, k, 1:2])], 1)
            ref = orc.unet_forward(params, cfg, inp, training=True)
            got = preds[k][:1].cpu()
            rel = float((got - ref).norm() / ref.norm())
            print(f"[c4 {R}^2 depth {depth}] bf16 application {k} vs fp32 oracle on the same input: rel-L2 {rel:.3e}")
            assert rel < 8e-2, (k, rel)
            prev, cur = got[:, 0:1], got
    assert all(np.isfinite(v) for v in p.loss_dict().values())
    p.run(("bwd",))
    g = m.flat_grads
    assert bool(torch.isfinite(g).all())
    for k in p.pb.shapes:
        if k.endswith(".bias") and (".c1." in k or ".c2." in k):
            continue
        assert float(unpack(p.pb, g, k).abs().max()) > 0.0, k
    p.run(("opt",))
    assert bool(torch.isfinite(m.flat_params).all())
    del p
    # reversing the batch reverses the predictions bit for bit (same weights: re-load them, the step above moved them)
    m.load_state_dict(params)
    fwd = _plan(m, "bf16", R, K, batch)
    fwd.run(("fwd",))
    a = [t.pred.tensor.clone() for t in fwd.tapes]
    del fwd
    rev = _plan(m, "bf16", R, K, batch, reverse=True)
    rev.run(("fwd",))
    for k in range(K):
        assert torch.equal(rev.tapes[k].pred.tensor.flip(0), a[k]), f"application {k} depends on the batch position"


def test_c5_rollout_600_frames_64_scenes_256():
    R, B, FR = 256, 64, 600
    cfg = orc.medium_config(dropout=0.0)
    params = orc.make_params(cfg, seed=7)
    m = UNet(UNetConfig(**cfg.__dict__), precision="bf16")
    m.load_state_dict(params)
    m.to(DEV).eval()
    x0, _, mk = synth_rollout_batch(B, 1, R, seed=99)          # per-scene emitter discs / collider boxes
    em, co = torch.from_numpy(mk[:, 0, 0]).to(DEV), torch.from_numpy(mk[:, 0, 1]).to(DEV)
    st0 = torch.from_numpy(x0[:, :3]).to(DEV) * 0.1             # a small non-zero start so that scenes differ from frame 1 on
    plan = RolloutInferPlan(m.cfg, B, R, R, params_flat=m.flat_params, dtype="bf16")
    plan.reset(em, co, state=st0, prev_density=st0[:, 0])
    cur, prev = st0[:2].cpu(), st0[:2, 0:1].cpu()
    e2, c2 = em[:2, None].cpu(), co[:2, None].cpu()
    with torch.no_grad():
        for f in range(4):
            got = plan.step(1)[:2].cpu()
            ref = orc.unet_forward(params, cfg, torch.cat([cur, prev, e2, c2], 1), training=False)
            rel = float((got - ref).norm() / (ref.norm() + 1e-12))
            print(f"[c5] frame {f + 1}, scenes 0-1 vs fp32 oracle on the same state: rel-L2 {rel:.3e}")
            assert rel < 8e-2, (f, rel)
            prev, cur = cur[:, 0:1], got
    plan.step(FR - 4)
    end = plan.cur.tensor.clone()
    assert bool(torch.isfinite(end).all())
    assert float(end[:, 0].min()) >= 0.0 and float(end[:, 0].max()) <= 1.0 and float(end[:, 1:].abs().max()) <= 1.0
    assert float(end.std()) > 0.0
    # the same 64 scenes in another order: every scene's 600-frame trajectory must end in the same bits
    perm = torch.from_numpy(np.random.RandomState(5).permutation(B)).to(DEV)
    plan.reset(em[perm], co[perm], state=st0[perm], prev_density=st0[perm, 0])
    plan.step(FR)
    assert torch.equal(plan.cur.tensor, end[perm]), "a scene's rollout depends on its position in the batch"
