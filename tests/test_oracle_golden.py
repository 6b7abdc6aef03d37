"""The CPU oracle (oracle/fluidnet_oracle.py) against fixtures generated from the real reference
(tests/golden/make_golden.py).  CPU only; this is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

import fluidnet_oracle as orc

torch.set_num_threads(8)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def tiny_cfg(**over):
    kw = dict(in_channels=6, out_channels=3, base_channels=8, depth=2, norm="instance", act="gelu", dropout=0.0,
              upsample="bilinear", downsample="stride", padding_mode="replicate", use_residual=True,
              bottleneck_blocks=1, output_activation="linear_clamp")
    kw.update(over)
    return orc.OracleUNetConfig(**kw)


def loss_c2(R, **over):
    kw = dict(mse_weight=1.0, divergence_weight=0.01, gradient_weight=0.001, emitter_weight=0.01,
              grid_spacing=2.0 / R, padding_mode="replicate", stencil_mode="forward")
    kw.update(over)
    return orc.LossConfig(**kw)


def relerr(a, b, rtol_for_floor=2e-4, atol=1e-6):
    """Relative L2 error with an absolute floor: returns ||a-b|| / (||b|| + (atol/rtol)*sqrt(n)), so that
    `relerr < rtol` is equivalent to ||a-b|| < rtol*||b|| + atol*sqrt(n).  (Some gradients are analytically
    zero, e.g. a bias that feeds an InstanceNorm, and are pure rounding noise in both implementations.)"""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + (atol / rtol_for_floor) * (b.numel() ** 0.5)))


def test_param_inventory_matches_reference(golden_dir):
    g = _load(golden_dir, "g2_medium_step.npz")
    shapes = orc.param_shapes(orc.medium_config())
    assert list(shapes.keys()) == [str(s) for s in g["param_names"]]
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["param_count"]) == 6026755
    assert len(shapes) == 88


def test_g1_tiny_step(golden_dir):
    g = _load(golden_dir, "g1_tiny_step.npz")
    cfg = tiny_cfg()
    p = {k: v.requires_grad_(True) for k, v in orc.make_params(cfg, seed=11).items()}
    assert abs(sum(float(v.detach().double().abs().sum()) for v in p.values()) - float(g["param_checksum"])) < 1e-9
    x0, y, _ = orc.synth_batch(2, 1, 32, seed=101)
    x, tgt = torch.from_numpy(x0), torch.from_numpy(y[:, 0])
    out = orc.unet_forward(p, cfg, x, training=True)
    loss, d = orc.physics_loss(out, tgt, x, loss_c2(32))
    loss.backward()
    assert np.abs(out.detach().numpy() - g["out"]).max() < 2e-5
    assert abs(float(loss.detach()) - float(g["loss_total"])) < 1e-6
    for k in ("mse", "divergence", "gradient", "emitter", "total"):
        assert abs(float(d[k].detach()) - float(g[f"loss_{k}"])) < 1e-5 * max(1.0, abs(float(g[f"loss_{k}"]))), k
    for k, v in p.items():
        assert relerr(v.grad, g[f"grad/{k}"]) < 2e-4, k


def test_g2_medium_step(golden_dir):
    g = _load(golden_dir, "g2_medium_step.npz")
    cfg = orc.medium_config(dropout=0.0)
    p = {k: v.requires_grad_(True) for k, v in orc.make_params(cfg, seed=7).items()}
    x0, y, _ = orc.synth_batch(2, 1, 128, seed=202)
    x, tgt = torch.from_numpy(x0), torch.from_numpy(y[:, 0])
    out = orc.unet_forward(p, cfg, x, training=True)
    loss, d = orc.physics_loss(out, tgt, x, loss_c2(128))
    loss.backward()
    assert np.abs(out.detach()[:, :, ::8, ::8].numpy() - g["out_sub8"]).max() < 5e-5
    assert abs(float(loss.detach()) - float(g["loss_total"])) < 1e-6
    names = [str(s) for s in g["grad_names"]]
    for i, k in enumerate(names):
        gl2 = float(p[k].grad.double().norm())
        assert abs(gl2 - g["grad_l2"][i]) < 1e-3 * g["grad_l2"][i] + 1e-6, k
        head = p[k].grad.flatten()[:64].numpy()
        assert np.abs(head - g["grad_head64"][i][: head.size]).max() < 1e-3 * np.abs(g["grad_head64"][i]).max() + 1e-6, k


@pytest.mark.parametrize("tag,trunc,fin", [("full", False, False), ("trunc", True, False), ("final", False, True)])
def test_g3_rollout_k4(golden_dir, tag, trunc, fin):
    g = _load(golden_dir, "g3_tiny_rollout_k4.npz")
    cfg = tiny_cfg()
    p = {k: v.requires_grad_(True) for k, v in orc.make_params(cfg, seed=13).items()}
    x0, y, _ = orc.synth_batch(2, 4, 32, seed=303)
    mk = torch.from_numpy(g["masks"])
    loss, d, pred = orc.rollout_loss(p, cfg, loss_c2(32), torch.from_numpy(x0), torch.from_numpy(y), mk,
                                     weight_decay=1.10, gradient_truncation=trunc, final_step_only=fin, training=True)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[f"{tag}/loss_total"])) < 2e-6
    assert np.abs(pred.detach().numpy() - g[f"{tag}/pred"]).max() < 5e-5
    for k in ("mse", "divergence", "gradient", "emitter", "total"):
        assert abs(float(d[k].detach()) - float(g[f"{tag}/loss_{k}"])) < 2e-5 * max(1.0, abs(float(g[f"{tag}/loss_{k}"]))), k
    for k, v in p.items():
        ref = g[f"{tag}/grad/{k}"]
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert relerr(got, ref) < 5e-4, k


def test_g4_loss_matrix(golden_dir):
    g = _load(golden_dir, "g4_loss_matrix.npz")
    out, tgt, inp = (torch.from_numpy(g[k]) for k in ("out", "tgt", "inp"))
    R = out.shape[2]
    e, c = g["emit_partial"], g["coll_partial"]
    kinds = {"partial": (e, c), "none": (np.zeros_like(e), np.zeros_like(c)), "all_emitter": (np.ones_like(e), np.zeros_like(c))}
    for pm in ("zeros", "replicate", "reflect", "circular"):
        for st in ("forward", "central"):
            gx, gy = orc.compute_spatial_gradients(out[:, 0], 0.5, 0.25, pm, st)
            div = orc.compute_divergence(out[:, 1], out[:, 2], 0.5, 0.25, pm, st)
            assert np.abs(gx.numpy() - g[f"{pm}/{st}/gx"]).max() == 0
            assert np.abs(gy.numpy() - g[f"{pm}/{st}/gy"]).max() == 0
            assert np.abs(div.numpy() - g[f"{pm}/{st}/div"]).max() == 0
            for mk, (em, co) in kinds.items():
                i2 = inp.clone()
                i2[:, 4], i2[:, 5] = torch.from_numpy(em), torch.from_numpy(co)
                o = out.clone().requires_grad_(True)
                lc = orc.LossConfig(1.0, 0.3, 0.2, 0.4, 2.0 / R, True, True, True, pm, st)
                loss, d = orc.physics_loss(o, tgt, i2, lc)
                loss.backward()
                pre = f"{pm}/{st}/{mk}"
                assert abs(float(loss.detach()) - float(g[f"{pre}/total"])) <= 1e-6 * max(1, abs(float(loss))), pre
                for k in ("mse", "divergence", "gradient", "emitter"):
                    assert abs(float(d[k].detach()) - float(g[f"{pre}/{k}"])) <= 1e-6 * max(1, abs(float(d[k]))), (pre, k)
                assert np.abs(o.grad.numpy() - g[f"{pre}/dout"]).max() < 1e-7, pre
    v = orc.gradient_loss(out[:, 0], tgt[:, 0], padding_mode="replicate", mode="central")
    assert abs(float(v) - float(g["nomask_gradient_loss"])) < 1e-6
    v = orc.emitter_spawn_loss(out[:, 0:1], inp[:, 0:1], torch.from_numpy(e[:, None]))
    assert abs(float(v) - float(g["emitter_loss"])) < 1e-7


G5 = {
    "batch_relu_nearest_avgpool": dict(norm="batch", act="relu", upsample="nearest", downsample="avgpool", padding_mode="zeros"),
    "group_silu_transpose_maxpool": dict(norm="group", act="silu", upsample="transpose", downsample="maxpool", padding_mode="reflect", group_norm_groups=3),
    "none_leaky_bilinear_stride_nores": dict(norm="none", act="leaky_relu", use_residual=False, padding_mode="circular"),
    "instance_sigmoid_tanh_mid2": dict(output_activation="sigmoid_tanh", bottleneck_blocks=2, depth=1),
    "instance_zeros_mid0": dict(padding_mode="zeros", bottleneck_blocks=0),
}


@pytest.mark.parametrize("name", list(G5))
def test_g5_variants_forward(golden_dir, name):
    g = _load(golden_dir, "g5_variants_fwd.npz")
    cfg = tiny_cfg(**G5[name])
    p = orc.make_params(cfg, seed=17)
    assert list(p.keys()) == [str(s) for s in g[f"{name}/param_names"]]
    x0, _, _ = orc.synth_batch(2, 1, 32, seed=505)
    out = orc.unet_forward(p, cfg, torch.from_numpy(x0), training=True)
    assert np.abs(out.numpy() - g[f"{name}/out"]).max() < 5e-5


def test_g7_adam_clip(golden_dir):
    g = _load(golden_dir, "g7_adam_clip.npz")
    cfg = tiny_cfg()
    p = orc.make_params(cfg, seed=19)
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v = {k: torch.zeros_like(t) for k, t in p.items()}
    for step in range(1, 4):
        x0, y, mk = orc.synth_batch(2, 2, 32, seed=700 + step)
        loss, d, raw = orc.train_step(p, m, v, step, cfg, loss_c2(32), torch.from_numpy(x0), torch.from_numpy(y),
                                      torch.from_numpy(mk), lr=3e-3, clip_norm=0.05)
        assert abs(loss - float(g[f"step{step}/loss"])) < 1e-5 * max(1, abs(loss)), step
        norm = float(torch.sqrt(sum((t.double() ** 2).sum() for t in raw.values())))
        assert abs(norm - float(g[f"step{step}/grad_norm"])) < 1e-3 * norm
        if step in (1, 3):
            for k in p:
                # A bias that feeds an InstanceNorm has an analytically zero gradient; Adam turns the
                # rounding noise into +-lr steps, so those tensors are only bounded, not matched.
                noise = float(raw[k].norm()) < 1e-6
                tol = 3e-3 * step * 1.01 if noise else (2e-5 if step == 1 else 1.5e-4)   # Adam amplifies fp32 noise of tiny grads
                assert np.abs(p[k].numpy() - g[f"step{step}/param/{k}"]).max() < tol, (step, k)


def test_g8_rollout16(golden_dir):
    g = _load(golden_dir, "g8_rollout16.npz")
    cfg = tiny_cfg()
    p = orc.make_params(cfg, seed=23)
    em, co = orc.simple_infer_masks(32, scale=32 / 128.0)
    assert np.array_equal(em, g["emitter"]) and np.array_equal(co, g["collider"])
    fr = orc.rollout_infer(p, cfg, torch.from_numpy(em)[None], torch.from_numpy(co)[None], 16)
    assert np.abs(fr[:, 0].numpy() - g["frames"]).max() < 1e-4
