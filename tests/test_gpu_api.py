"""End-to-end GPU parity through the reference-shaped Python API (fluidnet_amd.models.UNet + PhysicsAwareLoss with
torch autograd) against the CPU oracle and the golden fixtures generated from the real reference.

Tolerances (stated per BASELINE.json north_star):
  fp32 path : predictions max-abs 1e-4, losses 1e-5 rel, per-tensor gradients rel-L2 2e-4 (+1e-6 abs floor)
  bf16 path : predictions rel-L2 8e-2, losses 1e-2 rel, whole-gradient cosine >= 0.93 and rel-L2 <= 0.45, and never more
              than 2x the "conditioning floor" measured in the same test: the fp32 oracle evaluated on weights and inputs
              rounded ONCE to bf16 (no intermediate rounding) already moves predictions by ~3e-2 rel-L2 and the gradient
              by ~0.19 rel-L2 (cos 0.98) for the random-init UNet_medium, whose 56 % clamped outputs and stacked
              InstanceNorms make it ill-conditioned.  Kernel-level bf16 exactness (1 ulp against a bf16-emulating CPU
              twin, identical inputs) is what tests/test_gpu_lockstep.py asserts.
"""
import os

import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.models.unet import UNet, UNetConfig
from fluidnet_amd.training.physics_loss import PhysicsAwareLoss
from helpers import loss_c2, relerr, tiny_cfg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(cfg, params, precision, conv_impl="auto", train=True):
    m = UNet(UNetConfig(**cfg.__dict__), precision=precision, conv_impl=conv_impl)
    m.load_state_dict(params, strict=cfg.norm != "batch")      # the seeded parameter set has no BatchNorm2d buffers
    m.to(DEV)
    m.train(train)
    return m


def _crit(lc):
    return PhysicsAwareLoss(lc.mse_weight, lc.divergence_weight, lc.gradient_weight, lc.emitter_weight, lc.grid_spacing,
                            lc.enable_divergence, lc.enable_gradient, lc.enable_emitter, lc.padding_mode, lc.stencil_mode)


def test_g1_golden_fp32(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_tiny_step.npz"))
    cfg = tiny_cfg()
    params = orc.make_params(cfg, seed=11)
    m = _model(cfg, params, "fp32")
    assert [k for k, _ in m.named_parameters()] == [str(s) for s in g["param_names"]]
    x0, y, _ = orc.synth_batch(2, 1, 32, seed=101)
    x = torch.from_numpy(x0).to(DEV)
    tgt = torch.from_numpy(y[:, 0]).to(DEV)
    out = m(x)
    loss, d = _crit(loss_c2(32))(out, tgt, x)
    loss.backward()
    assert np.abs(out.detach().cpu().numpy() - g["out"]).max() < 1e-4
    for k in ("mse", "divergence", "gradient", "emitter", "total"):
        assert abs(d[k] - float(g[f"loss_{k}"])) < 1e-5 * max(1.0, abs(float(g[f"loss_{k}"]))), k
    for k, p in m.named_parameters():
        assert relerr(p.grad, g[f"grad/{k}"]) < 2e-4, k


# the alternate block types of the reference's UNet, one combination each (the G5 fixtures of tests/golden/make_golden.py: outputs of
# the imported reference model in training mode)
G5 = {
    "batch_relu_nearest_avgpool": dict(norm="batch", act="relu", upsample="nearest", downsample="avgpool", padding_mode="zeros"),
    "group_silu_transpose_maxpool": dict(norm="group", act="silu", upsample="transpose", downsample="maxpool", padding_mode="reflect", group_norm_groups=3),
    "none_leaky_bilinear_stride_nores": dict(norm="none", act="leaky_relu", use_residual=False, padding_mode="circular"),
    "instance_sigmoid_tanh_mid2": dict(output_activation="sigmoid_tanh", bottleneck_blocks=2, depth=1),
    "instance_zeros_mid0": dict(padding_mode="zeros", bottleneck_blocks=0),
}


@pytest.mark.parametrize("name", list(G5))
def test_g5_golden_variants_forward_fp32(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "g5_variants_fwd.npz"))
    cfg = tiny_cfg(**G5[name])
    m = _model(cfg, orc.make_params(cfg, seed=17), "fp32")
    assert [k for k, _ in m.named_parameters()] == [str(s) for s in g[f"{name}/param_names"]]
    x0, _, _ = orc.synth_batch(2, 1, 32, seed=505)
    with torch.no_grad():
        out = m(torch.from_numpy(x0).to(DEV))
    assert np.abs(out.cpu().numpy() - g[f"{name}/out"]).max() < 1e-4


def test_g2_golden_medium_fp32(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_medium_step.npz"))
    cfg = orc.medium_config(dropout=0.0)
    m = _model(cfg, orc.make_params(cfg, seed=7), "fp32")
    x0, y, _ = orc.synth_batch(2, 1, 128, seed=202)
    x = torch.from_numpy(x0).to(DEV)
    tgt = torch.from_numpy(y[:, 0]).to(DEV)
    out = m(x)
    loss, d = _crit(loss_c2(128))(out, tgt, x)
    loss.backward()
    assert np.abs(out.detach()[:, :, ::8, ::8].cpu().numpy() - g["out_sub8"]).max() < 2e-4
    assert abs(d["total"] - float(g["loss_total"])) < 1e-5
    names = [str(s) for s in g["grad_names"]]
    grads = dict(m.named_parameters())
    for i, k in enumerate(names):
        gl2 = float(grads[k].grad.double().norm())
        assert abs(gl2 - g["grad_l2"][i]) < 2e-3 * g["grad_l2"][i] + 1e-6, k
        head = grads[k].grad.flatten()[:64].cpu().numpy()
        assert np.abs(head - g["grad_head64"][i][: head.size]).max() < 2e-3 * np.abs(g["grad_head64"][i]).max() + 1e-6, k


def _flat(grads):
    return torch.cat([g.detach().reshape(-1).double().cpu() for g in grads])


def _cos_rel(a, b):
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)), float((a - b).norm() / (b.norm() + 1e-300))


def _bf16_floor(cfg, lc, params, x0, y, mk):
    """Oracle (fp32 arithmetic) on weights+inputs rounded once to bf16: the conditioning floor of the problem."""
    p = {k: v.detach().bfloat16().float().requires_grad_(True) for k, v in params.items()}
    loss, _, pred = orc.rollout_loss(p, cfg, lc, torch.from_numpy(x0).bfloat16().float(), torch.from_numpy(y), torch.from_numpy(mk),
                                     training=True)
    loss.backward()
    return pred.detach(), _flat([v.grad for v in p.values()])


def _rollout_api(m, crit, x0, y_seq, masks, wd=1.10):
    """The loop of trainer.py:355-385, written against the drop-in API."""
    K = y_seq.shape[1]
    cur, prev = x0[:, :3], x0[:, 3:4]
    total, wsum, acc = 0.0, 0.0, {}
    for k in range(K):
        inp = torch.cat([cur, prev, masks[:, k, 0:1], masks[:, k, 1:2]], dim=1)
        pred = m(inp)
        l, d = crit(pred, y_seq[:, k], inp)
        w = wd ** k
        total = total + w * l
        wsum += w
        for kk, v in d.items():
            acc[kk] = acc.get(kk, 0.0) + w * v
        prev, cur = pred[:, 0:1], pred
    return total / wsum, {k: v / wsum for k, v in acc.items()}, pred


@pytest.mark.parametrize("precision,base", [("fp32", 8), ("bf16", 16)])
def test_rollout_k2_autograd_api(precision, base):
    cfg = tiny_cfg(base_channels=base)
    params = orc.make_params(cfg, seed=13)
    x0, y, mk = orc.synth_batch(2, 2, 32, seed=303)
    mk[:, 1] = np.roll(mk[:, 1], 1, -1)
    m = _model(cfg, params, precision)
    crit = _crit(loss_c2(32))
    loss, d, pred = _rollout_api(m, crit, *(torch.from_numpy(a).to(DEV) for a in (x0, y, mk)))
    loss.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    oloss, od, opred = orc.rollout_loss(p, cfg, loss_c2(32), torch.from_numpy(x0), torch.from_numpy(y), torch.from_numpy(mk), training=True)
    oloss.backward()
    if precision == "fp32":
        assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-5 * max(1, abs(float(oloss.detach())))
        assert (pred.detach().cpu() - opred.detach()).abs().max() < 2e-3
        worst = max(relerr(v.grad, p[k].grad) for k, v in m.named_parameters())
        print(f"[fp32] K2 rollout: loss {float(loss.detach()):.6f} vs {float(oloss.detach()):.6f}; worst per-tensor grad rel-L2 {worst:.3e}")
        assert worst < 2e-4
    else:
        fpred, fgrad = _bf16_floor(cfg, loss_c2(32), params, x0, y, mk)
        g0 = _flat([v.grad for v in p.values()])
        cos, rel = _cos_rel(_flat([v.grad for _, v in m.named_parameters()]), g0)
        fcos, frel = _cos_rel(fgrad, g0)
        prel = float((pred.detach().cpu() - opred.detach()).norm() / opred.detach().norm())
        print(f"[bf16] K2 rollout: loss {float(loss.detach()):.6f} vs {float(oloss.detach()):.6f}; pred rel-L2 {prel:.3e}; "
              f"grad cos {cos:.4f} rel-L2 {rel:.3e} (floor: cos {fcos:.4f} rel-L2 {frel:.3e})")
        assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-2 * abs(float(oloss.detach()))
        fprel = float((fpred - opred.detach()).norm() / opred.detach().norm())
        # two chained applications of the tiny random-init net are even worse conditioned than UNet_medium: bound by the floor
        assert prel < max(8e-2, 2 * fprel) and cos > min(0.93, fcos - 0.08) and rel < max(0.45, 1.5 * frel)


# the reference's shipped architectures that the HIP engine covers (ml/config/model_architectures/*.yaml)
ARCHS = {
    "medium": (dict(), 128),
    "large": (dict(base_channels=48, depth=4, downsample="avgpool"), 64),        # UNet_large.yaml
    "medium-maxpool": (dict(base_channels=16, downsample="maxpool"), 64),
    "small-medium": (dict(base_channels=20, downsample="avgpool"), 64),          # UNet_small-medium.yaml
    "small": (dict(base_channels=16, depth=2, norm="batch", upsample="nearest", downsample="avgpool", padding_mode="zeros"), 128),  # UNet_small.yaml
}


@pytest.mark.parametrize("arch", list(ARCHS))
def test_medium_bf16_mfma_vs_oracle(arch):
    over, R = ARCHS[arch]
    cfg = orc.medium_config(dropout=0.0, **over)
    params = orc.make_params(cfg, seed=7)
    x0, y, _ = orc.synth_batch(2, 1, R, seed=202)
    m = _model(cfg, params, "bf16", "auto" if cfg.base_channels % 8 else "mfma")
    x = torch.from_numpy(x0).to(DEV).requires_grad_(True)
    tgt = torch.from_numpy(y[:, 0]).to(DEV)
    out = m(x)
    loss, d = _crit(loss_c2(R))(out, tgt, x)
    loss.backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = torch.from_numpy(x0).requires_grad_(True)
    oout = orc.unet_forward(p, cfg, xo, training=True)
    oloss, od = orc.physics_loss(oout, torch.from_numpy(y[:, 0]), xo, loss_c2(R))
    oloss.backward()
    diff = (out.detach().cpu() - oout.detach())
    prel = float(diff.norm() / oout.detach().norm())
    mk = np.zeros((2, 1, 2, R, R), np.float32)
    mk[:, 0] = x0[:, 4:6]
    fpred, fgrad = _bf16_floor(cfg, loss_c2(R), params, x0, y, mk)
    g0 = _flat([v.grad for v in p.values()])
    cos, rel = _cos_rel(_flat([v.grad for _, v in m.named_parameters()]), g0)
    fcos, frel = _cos_rel(fgrad, g0)
    fprel = float((fpred - oout.detach()).norm() / oout.detach().norm())
    dcos, drel = _cos_rel(_flat([x.grad]), _flat([xo.grad]))
    print(f"[bf16 {arch}] pred max-abs {float(diff.abs().max()):.3e} rel-L2 {prel:.3e} (floor {fprel:.3e}); loss {float(loss.detach()):.6f} vs "
          f"{float(oloss.detach()):.6f}; grad cos {cos:.4f} rel-L2 {rel:.3e} (floor: cos {fcos:.4f} rel-L2 {frel:.3e}); dx cos {dcos:.4f} rel-L2 {drel:.3e}")
    assert prel < max(8e-2, 2 * fprel)
    assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-2 * abs(float(oloss.detach()))
    assert cos > 0.93 and rel < max(0.45, 2 * frel)
    # max pooling: bf16 rounding flips the argmax of near-tied windows, on either side; bound by the floor run's own agreement
    assert dcos > min(0.93, fcos - 0.05)


def test_eval_mode_and_no_grad_and_cpu_rejection():
    cfg = tiny_cfg(base_channels=16, dropout=0.5)
    params = orc.make_params(cfg, seed=3)
    m = _model(cfg, params, "bf16", train=False)
    x0, _, _ = orc.synth_batch(3, 1, 32, seed=9)
    with torch.no_grad():
        a = m(torch.from_numpy(x0).to(DEV))
        b = m(torch.from_numpy(x0).to(DEV))
    assert torch.equal(a, b)                       # eval: dropout off, deterministic
    ref = orc.unet_forward(params, cfg, torch.from_numpy(x0), training=False)
    assert float((a.cpu() - ref).norm() / ref.norm()) < 8e-2
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.from_numpy(x0))
    with pytest.raises(NotImplementedError):
        UNet(UNetConfig(in_channels=17, base_channels=16)).to(DEV)(torch.zeros(3, 17, 32, 32, device=DEV))
    with pytest.raises(ValueError, match="depth must be >= 1"):
        UNet(UNetConfig(depth=0))


def test_dropout_training_matches_oracle_with_injected_masks():
    """Dropout2d(p) in training mode: the engine's hash mask, replayed into the oracle, gives the same step."""
    from helpers import drop_masks_for
    cfg = tiny_cfg(dropout=0.25)
    params = orc.make_params(cfg, seed=21)
    m = _model(cfg, params, "fp32")
    x0, y, _ = orc.synth_batch(4, 1, 32, seed=77)
    x = torch.from_numpy(x0).to(DEV)
    out = m(x)
    eng = next(iter(m._engines.values()))
    seed = (eng.pb.seed * 0x9E3779B1 + 0 * 0x85EBCA6B + 12345) & 0x7FFFFFFF
    dm = drop_masks_for(cfg, 4, seed, 0, cfg.dropout)
    ref = orc.unet_forward(params, cfg, torch.from_numpy(x0), training=True, drop_masks=dm)
    assert (out.detach().cpu() - ref).abs().max() < 1e-4
    frac = np.mean([float((v == 0).float().mean()) for v in dm.values()])
    assert 0.1 < frac < 0.4


def test_batch_norm_module_train_eval_and_state_dict():
    """UNet_small's family through the nn.Module API (fp32 path): training-mode forward/backward with BatchNorm2d batch statistics,
    the running statistics it leaves in the module's buffers (state-dict keys of nn.BatchNorm2d), then eval mode on those."""
    cfg = tiny_cfg(norm="batch", downsample="avgpool", upsample="nearest", padding_mode="zeros", base_channels=16)
    params = orc.make_params(cfg, seed=21)
    m = _model(cfg, params, "fp32")
    sd_keys = set(m.state_dict().keys())
    assert {"downs.0.block.n1.running_mean", "downs.0.block.n1.running_var", "downs.0.block.n1.num_batches_tracked"} <= sd_keys
    x0, y, _ = orc.synth_batch(3, 1, 32, seed=77)
    x = torch.from_numpy(x0).to(DEV)
    bn = {}
    for k, v in m.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            bn[k] = v.detach().cpu().clone()
    crit = _crit(loss_c2(32, padding_mode="zeros"))
    for step in range(2):                                  # two training-mode forwards: statistics accumulate
        out = m(x)
        loss, _ = crit(out, torch.from_numpy(y[:, 0]).to(DEV), x)
        m.zero_grad()
        loss.backward()
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        oout = orc.unet_forward(p, cfg, torch.from_numpy(x0), training=True, bn_state=bn)
        oloss, _ = orc.physics_loss(oout, torch.from_numpy(y[:, 0]), torch.from_numpy(x0), loss_c2(32, padding_mode="zeros"))
        oloss.backward()
        assert (out.detach().cpu() - oout.detach()).abs().max() < 2e-4
        worst = max(relerr(v.grad, p[k].grad) for k, v in m.named_parameters())
        assert worst < 2e-4, worst
    sd = m.state_dict()
    for k, v in bn.items():
        assert (sd[k].cpu() - v).abs().max() < 1e-5 * max(1.0, float(v.abs().max())), k
    assert int(sd["mid.0.b.n2.num_batches_tracked"]) == 2
    m.eval()
    with torch.no_grad():
        e = m(x)
    eref = orc.unet_forward(params, cfg, torch.from_numpy(x0), training=False, bn_state=bn)
    assert (e.cpu() - eref).abs().max() < 2e-4
    # eval-mode backward: BatchNorm2d is an affine map of the running statistics (nothing is centred)
    xg = torch.from_numpy(x0).to(DEV).requires_grad_(True)
    m.zero_grad()
    m(xg).square().sum().backward()
    pe = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = torch.from_numpy(x0).requires_grad_(True)
    orc.unet_forward(pe, cfg, xo, training=False, bn_state=bn).square().sum().backward()
    assert relerr(xg.grad, xo.grad) < 2e-4
    assert max(relerr(v.grad, pe[k].grad) for k, v in m.named_parameters()) < 2e-4
    # a reference-format checkpoint round trip keeps the statistics (buffers stay views of the flat norm state)
    m2 = _model(cfg, orc.make_params(cfg, seed=5), "fp32", train=False)
    m2.load_state_dict(sd)
    with torch.no_grad():
        assert torch.equal(m2(x), e)


def test_default_constructor_config():
    """`UNet()` with the reference's own defaults (unet.py:183-201: 7 input channels, GroupNorm, SiLU, nearest, zeros, no residual,
    depth 2): forward, input gradient and parameter gradients against the oracle, fp32 path."""
    m = UNet(precision="fp32")
    cfg = orc.OracleUNetConfig(**m.cfg.__dict__)
    assert cfg.in_channels == 7 and cfg.norm == "group"
    params = orc.make_params(cfg, seed=31)
    m.load_state_dict(params)
    m.to(DEV).train()
    rs = np.random.RandomState(2)
    x0 = rs.rand(2, 7, 32, 32).astype(np.float32)
    x = torch.from_numpy(x0).to(DEV).requires_grad_(True)
    out = m(x)
    w = torch.from_numpy(rs.randn(*out.shape).astype(np.float32))
    (out * w.to(DEV)).sum().backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = torch.from_numpy(x0).requires_grad_(True)
    oout = orc.unet_forward(p, cfg, xo, training=True)
    (oout * w).sum().backward()
    assert (out.detach().cpu() - oout.detach()).abs().max() < 1e-4
    assert relerr(x.grad, xo.grad) < 2e-4
    assert max(relerr(v.grad, p[k].grad) for k, v in m.named_parameters()) < 2e-4
