"""Graph-wiring check of the plan builder on CPU: the op records of fluidnet_amd.engine are interpreted by
tests/sim_ops.py (a CPU twin of every kernel) and compared with the oracle's autograd.  This proves the
hand-derived backward graph (ring folding, residual/skip gradient routing, BPTT through the rollout) before any
kernel runs; the kernels themselves are checked against the same twins in tests/test_gpu_ops.py."""
import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.engine.train_plan import RolloutTrainPlan
from helpers import drop_masks_for, loss_c2, pack_params, relerr, tiny_cfg, unpack
from sim_ops import Sim

torch.set_num_threads(8)


def run_case(cfg, lc, B, R, K, *, trunc=False, fin=False, seed=5, dropout_seed=0, mask_frames=None, mask_offset=0):
    params = orc.make_params(cfg, seed=seed)
    x0, y, mk = orc.synth_batch(B, K, R, seed=100 + seed)
    for k in range(K):
        mk[:, k] = np.roll(mk[:, k], shift=k, axis=-1)
    plan = RolloutTrainPlan(cfg, lc, B, R, R, K, dtype="fp32", conv_impl="ref", gradient_truncation=trunc,
                            final_step_only=fin, seed=dropout_seed, clip_norm=0.05)
    pb = plan.pb
    plan.bind("cpu", pack_params(pb, params), torch.zeros(pb.n_params_padded))
    plan.x0.tensor.copy_(torch.from_numpy(x0))
    plan.y.tensor.copy_(torch.from_numpy(y))
    plan.masks.tensor.copy_(torch.from_numpy(mk))
    sim = Sim(pb.dtype)
    for s in ("fwd", "fin", "bwd"):
        sim.run(plan.sections[s])
    # oracle
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    dm = None
    if cfg.dropout > 0:
        dm = [drop_masks_for(cfg, B, dropout_seed, k, cfg.dropout) for k in range(K)]
    bn = None
    if cfg.norm == "batch":                     # fresh BatchNorm2d buffers; the K applications update them in turn
        bn = {}
        for k_ in pb.nstate_offsets:
            c = params[k_.rsplit(".", 1)[0] + ".weight"].numel()
            bn[k_] = torch.ones(c) if k_.endswith("running_var") else torch.zeros(c)
    loss, d, pred = orc.rollout_loss(p, cfg, lc, torch.from_numpy(x0), torch.from_numpy(y), torch.from_numpy(mk),
                                     weight_decay=1.10, gradient_truncation=trunc, final_step_only=fin, training=True, drop_masks=dm,
                                     bn_state=bn)
    loss.backward()
    if bn is not None:
        for k_, o in pb.nstate_offsets.items():
            got_rs = pb.nstate.tensor[o:o + bn[k_].numel()]
            assert (got_rs - bn[k_]).abs().max() < 1e-5 * max(1.0, float(bn[k_].abs().max())), k_
    got = plan.loss_dict()
    assert abs(got["total"] - float(loss.detach())) < 2e-6 * max(1, abs(float(loss.detach())))
    for k_, v in d.items():
        assert abs(got[k_] - float(v.detach())) < 5e-6 * max(1, abs(float(v.detach()))), k_
    # fp32 rounding differences are amplified by each further UNet application (random-init InstanceNorm stacks)
    assert (plan.tapes[-1].pred.tensor - pred.detach()).abs().max() < (1e-4 if K == 1 else 2e-3)
    worst = 0.0
    for k_, v in p.items():
        ref = v.grad if v.grad is not None else torch.zeros_like(v)
        e = relerr(unpack(pb, pb.grads.tensor, k_), ref)
        worst = max(worst, e)
        # K >= 3: the fp32 rounding noise of two chained applications is amplified again and flips a few clamp gates
        assert e < (2e-4 if K <= 2 else 6e-3), (k_, e)
    return plan, params, (x0, y, mk), worst


def test_k1_replicate():
    run_case(tiny_cfg(), loss_c2(32), 2, 32, 1)


def test_k2_full_bptt():
    run_case(tiny_cfg(), loss_c2(32), 2, 32, 2)


def test_k3_full_bptt():
    run_case(tiny_cfg(), loss_c2(32), 2, 32, 3)


def test_k3_truncation():
    run_case(tiny_cfg(), loss_c2(32), 2, 32, 3, trunc=True)


def test_k3_final_only():
    run_case(tiny_cfg(), loss_c2(32), 2, 32, 3, fin=True)


@pytest.mark.parametrize("pm", ["zeros", "reflect", "circular"])
def test_padding_modes(pm):
    run_case(tiny_cfg(padding_mode=pm), loss_c2(16, padding_mode=pm, stencil_mode="central"), 2, 16, 2)


def test_variants_nores_nearest_silu_depth1_mid2():
    run_case(tiny_cfg(use_residual=False, upsample="nearest", act="silu", depth=1, bottleneck_blocks=2,
                      output_activation="sigmoid_tanh"), loss_c2(16), 2, 16, 2)


def test_mid0_relu_depth3():
    run_case(tiny_cfg(bottleneck_blocks=0, act="relu", depth=3), loss_c2(32), 1, 32, 1)


@pytest.mark.parametrize("mode,pm", [("avgpool", "replicate"), ("maxpool", "zeros"), ("maxpool", "reflect"), ("avgpool", "circular")])
def test_pooled_downsample(mode, pm):
    """Sequential(pool, conv3x3) `down` of UNet_small / UNet_small-medium / UNet_large (unet.py:93-94)"""
    # K = 1 for max: the argmax is discontinuous, a second application amplifies fp32 noise past the K <= 2 gradient bar
    run_case(tiny_cfg(downsample=mode, padding_mode=pm), loss_c2(16, padding_mode=pm), 2, 16, 2 if mode == "avgpool" else 1)


def test_batch_norm_small_architecture():
    """UNet_small's block types (UNet_small.yaml): BatchNorm2d in training mode incl. running statistics, avgpool, nearest, zeros"""
    run_case(tiny_cfg(norm="batch", downsample="avgpool", upsample="nearest", padding_mode="zeros"),
             loss_c2(16, padding_mode="zeros"), 3, 16, 2)


def test_group_norm_default_config_blocks():
    """UNetConfig's own defaults (unet.py:183-201): GroupNorm, SiLU, nearest, zeros, no residual; 16 channels in 8 groups and
    (base 24, group_norm_groups 5) 48 channels, where the reference's group search settles on 4 groups of 12"""
    run_case(tiny_cfg(norm="group", act="silu", upsample="nearest", padding_mode="zeros", use_residual=False),
             loss_c2(16, padding_mode="zeros"), 2, 16, 2)
    run_case(tiny_cfg(norm="group", base_channels=24, depth=1, group_norm_groups=5), loss_c2(16), 2, 16, 1)


def test_norm_none_conv_bias_blocks():
    """norm="none": ConvBlock convs carry a bias and nothing is normalised (unet.py:55-69)"""
    run_case(tiny_cfg(norm="none", dropout=0.25, act="leaky_relu"), loss_c2(16), 2, 16, 2, dropout_seed=3)
    run_case(tiny_cfg(norm="none", use_residual=False, downsample="maxpool", upsample="nearest"), loss_c2(16), 2, 16, 1)


def test_transposed_conv_upsample():
    """upsample="transpose": nn.ConvTranspose2d(C, C, kernel_size=2, stride=2) with bias in every Up (unet.py:149-150)"""
    run_case(tiny_cfg(upsample="transpose"), loss_c2(16), 2, 16, 2)
    run_case(tiny_cfg(upsample="transpose", padding_mode="reflect", use_residual=False, depth=1, norm="group"),
             loss_c2(16, padding_mode="reflect"), 2, 16, 1)


def test_channel_counts_not_a_multiple_of_8():
    """UNet_small-medium (base 20, UNet_small-medium.yaml): the 20-channel tensors run zero-padded to 32, everything the reference
    sees (losses, predictions, every gradient, real-shaped parameters) is unchanged; and base 12 with GroupNorm / BatchNorm2d, whose
    groups and running statistics are formed over the real channels"""
    run_case(tiny_cfg(base_channels=20, downsample="avgpool", dropout=0.2), loss_c2(16), 2, 16, 2, dropout_seed=7)
    run_case(tiny_cfg(base_channels=12, norm="group", group_norm_groups=3, depth=1), loss_c2(16), 2, 16, 1)
    run_case(tiny_cfg(base_channels=12, norm="batch", depth=1, use_residual=False), loss_c2(16), 2, 16, 2)
    plan, *_ = run_case(tiny_cfg(base_channels=20), loss_c2(16), 1, 16, 1)
    # the zero padding stays zero: the gaps of the flat gradient buffer received nothing
    pb = plan.pb
    used = torch.zeros(pb.n_params_padded, dtype=torch.bool)
    for k, s in pb.shapes.items():
        used[pb.offsets[k]: pb.offsets[k] + int(np.prod(s))] = True
    assert float(pb.grads.tensor[~used].abs().max()) == 0.0


def test_dropout_hash_matches_injected_masks():
    run_case(tiny_cfg(dropout=0.3), loss_c2(16), 3, 16, 2, dropout_seed=1234)


def test_clip_adam_step_matches_oracle():
    cfg, lc = tiny_cfg(), loss_c2(32)
    plan, params, (x0, y, mk), _ = run_case(cfg, lc, 2, 32, 2)
    sim = Sim(plan.pb.dtype)
    p = {k: v.clone() for k, v in params.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    for step in (1, 2):
        plan.set_adam_scalars(3e-3, step)
        if step > 1:
            for s in ("fwd", "fin", "bwd"):
                sim.run(plan.sections[s])
        sim.run(plan.sections["opt"])
        _, _, raw = orc.train_step(p, m, v, step, cfg, lc, torch.from_numpy(x0), torch.from_numpy(y), torch.from_numpy(mk),
                                   lr=3e-3, clip_norm=0.05)
        gn = float(torch.sqrt(sum((t.double() ** 2).sum() for t in raw.values())))
        coef = min(1.0, 0.05 / (gn + 1e-6))
        for k_ in p:
            # Adam divides by sqrt(v) + 1e-8: elements whose (clipped) gradient is at the 1e-8 noise level take
            # arbitrary steps of up to lr in ANY implementation, so only well-conditioned elements are matched tightly.
            good = (raw[k_] * coef).abs() > 1e-5
            d = (unpack(plan.pb, plan.pb.params.tensor, k_) - p[k_]).abs()
            assert d.max() <= 3e-3 * step * 1.01, (step, k_)
            if good.any():
                assert d[good].max() < 3e-5, (step, k_, float(d[good].max()))
        # re-synchronise the noise-level elements so that step 2 starts from identical state on both sides
        from helpers import pack_params
        plan.pb.params.tensor.copy_(pack_params(plan.pb, p))
        plan.m.tensor.copy_(pack_params(plan.pb, m))
        plan.v.tensor.copy_(pack_params(plan.pb, v))


@pytest.mark.parametrize("pm", ["replicate", "zeros", "reflect", "circular"])
def test_stride2_space_to_depth_form_equals_full_resolution_form(pm, monkeypatch):
    """bf16 MFMA plans run Conv2d(C, C, 3, stride=2) (ml/models/unet.py:90-91) as a 2x2-tap convolution over the space-to-depth view of
    its input (CONV3X3 modes 1 / 2, WGRAD mode 1, PREP_WEIGHTS / UNPREP_WGRAD kind 2) instead of a full-resolution convolution +
    SUBSAMPLE_BIAS / ZERO_INSERT.  The twin interprets both forms of the same bf16 plan (MFMA weight layouts and all), here on fp32
    storage so that the two differ by summation order only: same loss, prediction and gradients; the full-resolution form is the one
    the other tests of this file pin to the oracle."""
    import fluidnet_amd.engine.runtime as rt
    cfg = tiny_cfg(base_channels=16, depth=2, padding_mode=pm, dropout=0.1)
    lc = loss_c2(32, padding_mode=pm)
    params = orc.make_params(cfg, seed=11)
    x0, y, mk = orc.synth_batch(2, 2, 32, seed=77)
    out = {}
    for s2 in ("1", "0"):
        monkeypatch.setenv("FNET_S2", s2)
        plan = RolloutTrainPlan(cfg, lc, 2, 32, 32, 2, dtype="bf16", seed=3, clip_norm=None)
        pb = plan.pb
        modes = [r.i[9] for sec in plan.sections.values() for r in sec if rt.OPNAME[r.code] == "CONV3X3" and len(r.i) > 9]
        assert (1 in modes and 2 in modes) == (s2 == "1")
        for b in pb.bufs:                       # fp32 storage for every activation / weight-image buffer
            if b.kind == "T" and not b.external and b.parent is None:
                b.tensor = torch.zeros(b.shape)
        plan.bind("cpu", pack_params(pb, params), torch.zeros(pb.n_params_padded))
        plan.x0.tensor.copy_(torch.from_numpy(x0))
        plan.y.tensor.copy_(torch.from_numpy(y))
        plan.masks.tensor.copy_(torch.from_numpy(mk))
        sim = Sim(rt.F32)
        for s in ("fwd", "fin", "bwd"):
            sim.run(plan.sections[s])
        out[s2] = (plan.loss_dict()["total"], plan.tapes[-1].pred.tensor.clone(), pb.grads.tensor.clone(), pb)
    (l1, p1, g1, pb1), (l0, p0, g0, pb0) = out["1"], out["0"]
    assert abs(l1 - l0) < 2e-6 * abs(l0)
    assert float((p1 - p0).abs().max()) < 2e-3
    for k_ in pb1.shapes:
        e = relerr(unpack(pb1, g1, k_), unpack(pb0, g0, k_))
        assert e < 2e-4, (k_, e)
