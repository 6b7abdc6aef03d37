"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference/ml).

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Inputs and weights come from seeded numpy formulas in oracle/fluidnet_oracle.py (make_params,
synth_batch), so the fixtures store expected OUTPUTS plus small inputs; nothing of the reference's
source is stored.  Importable reference pieces: models.unet (UNet, UNetConfig) and
training.physics_loss (everything).  training.trainer needs mlflow/torchmetrics (absent), so the
K-step rollout loss is pinned by composing the reference UNet + reference PhysicsAwareLoss in the
loop shape of trainer.py:355-385 below.
Round 2 additions: dataset.* imports config.config -> pydantic_settings (absent): a stub module
(BaseSettings = pydantic.BaseModel, SettingsConfigDict = dict) in sys.modules lets the REAL dataset module import -> g6;
scripts.variant_manager needs only yaml -> g9; training.metrics imports torchmetrics (absent) for SSIM only: a stub
torchmetrics.functional.image module lets its seven other functions import -> g10 (SSIM itself stays unpinned).
Their seeded inputs come from tests/fixture_inputs.py (our code), shared with the tests.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference/ml")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import fluidnet_oracle as orc  # noqa: E402
from models.unet import UNet, UNetConfig  # noqa: E402  (reference)
from training import physics_loss as ref_pl  # noqa: E402  (reference)

torch.set_num_threads(8)
torch.manual_seed(0)


def ref_model(cfg: orc.OracleUNetConfig, params, train: bool):
    m = UNet(UNetConfig(**cfg.__dict__))
    missing = m.load_state_dict(params, strict=(cfg.norm != "batch"))
    m.train(train)
    return m


def tiny_cfg(**over):
    kw = dict(in_channels=6, out_channels=3, base_channels=8, depth=2, norm="instance", act="gelu", dropout=0.0,
              upsample="bilinear", downsample="stride", padding_mode="replicate", use_residual=True,
              bottleneck_blocks=1, output_activation="linear_clamp")
    kw.update(over)
    return orc.OracleUNetConfig(**kw)


def loss_cfg_c2(R, padding_mode="replicate", stencil="forward"):
    return dict(mse_weight=1.0, divergence_weight=0.01, gradient_weight=0.001, emitter_weight=0.01,
                grid_spacing=2.0 / R, enable_divergence=True, enable_gradient=True, enable_emitter=True,
                padding_mode=padding_mode, stencil_mode=stencil)


def grads_of(model):
    return {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
            for k, p in model.named_parameters()}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ---------------------------------------------------------------- G1: tiny net, one step, all grads
def g1():
    cfg = tiny_cfg()
    R, B = 32, 2
    params = orc.make_params(cfg, seed=11)
    x0, y, _ = orc.synth_batch(B, 1, R, seed=101)
    x = torch.from_numpy(x0)
    tgt = torch.from_numpy(y[:, 0])
    m = ref_model(cfg, params, train=True)
    crit = ref_pl.PhysicsAwareLoss(**loss_cfg_c2(R))
    out = m(x)
    loss, d = crit(out, tgt, x)
    loss.backward()
    g = grads_of(m)
    arrs = {"out": out, "loss_total": loss.detach(), **{f"loss_{k}": np.float32(v) for k, v in d.items()}}
    arrs.update({f"grad/{k}": v for k, v in g.items()})
    arrs["param_names"] = np.array(list(m.state_dict().keys()))
    arrs["param_checksum"] = np.float64(sum(float(v.double().abs().sum()) for v in params.values()))
    save("g1_tiny_step.npz", **arrs)


# ---------------------------------------------------------------- G2: medium @128^2, B=2
def g2():
    cfg = orc.medium_config(dropout=0.0)
    R, B = 128, 2
    params = orc.make_params(cfg, seed=7)
    x0, y, _ = orc.synth_batch(B, 1, R, seed=202)
    x = torch.from_numpy(x0)
    tgt = torch.from_numpy(y[:, 0])
    m = ref_model(cfg, params, train=True)
    crit = ref_pl.PhysicsAwareLoss(**loss_cfg_c2(R))
    out = m(x)
    loss, d = crit(out, tgt, x)
    loss.backward()
    g = grads_of(m)
    names = list(g.keys())
    arrs = {
        "out_sub8": out[:, :, ::8, ::8],
        "out_stats": torch.stack([out.mean((0, 2, 3)), out.std((0, 2, 3)), out.amin((0, 2, 3)), out.amax((0, 2, 3))]),
        "loss_total": loss.detach(),
        **{f"loss_{k}": np.float32(v) for k, v in d.items()},
        "grad_names": np.array(names),
        "grad_l2": np.array([float(g[k].double().norm()) for k in names]),
        "grad_head64": np.stack([np.pad(g[k].flatten()[:64].numpy(), (0, max(0, 64 - g[k].numel()))) for k in names]),
        "param_count": np.int64(sum(p.numel() for p in m.parameters())),
        "param_names": np.array(list(m.state_dict().keys())),
    }
    save("g2_medium_step.npz", **arrs)


# ---------------------------------------------------------------- G3: K=4 rollout loss on the tiny net
def ref_rollout(m, crit, x0, y_seq, masks, wd, truncation, final_only):
    """Loop shape of trainer.py:321-385, composed from reference UNet + reference PhysicsAwareLoss."""
    K = y_seq.shape[1]
    cur, prev = x0[:, :3], x0[:, 3:4]
    total = torch.tensor(0.0)
    acc, wsum = {}, 0.0
    for k in range(K):
        inp = torch.cat([cur, prev, masks[:, k, 0:1], masks[:, k, 1:2]], dim=1)
        pred = m(inp)
        if final_only:
            if k == K - 1:
                l, d = crit(pred, y_seq[:, k], inp)
                return l, d, pred
        else:
            l, d = crit(pred, y_seq[:, k], inp)
            w = wd ** k
            total = total + w * l
            wsum += w
            for kk, v in d.items():
                acc[kk] = acc.get(kk, 0.0) + w * v
        if truncation:
            prev, cur = pred[:, 0:1].detach(), pred.detach()
        else:
            prev, cur = pred[:, 0:1], pred
    return total / wsum, {k_: v / wsum for k_, v in acc.items()}, pred


def g3():
    cfg = tiny_cfg()
    R, B, K = 32, 2, 4
    params = orc.make_params(cfg, seed=13)
    x0, y, mk = orc.synth_batch(B, K, R, seed=303)
    # make the masks vary over k so that the mask indexing is pinned too
    for k in range(K):
        mk[:, k] = np.roll(mk[:, k], shift=k, axis=-1)
    arrs = {"masks": mk}
    for tag, trunc, fin in (("full", False, False), ("trunc", True, False), ("final", False, True)):
        m = ref_model(cfg, params, train=True)
        crit = ref_pl.PhysicsAwareLoss(**loss_cfg_c2(R))
        loss, d, pred = ref_rollout(m, crit, torch.from_numpy(x0), torch.from_numpy(y), torch.from_numpy(mk), 1.10, trunc, fin)
        loss.backward()
        g = grads_of(m)
        arrs[f"{tag}/loss_total"] = loss.detach()
        arrs[f"{tag}/pred"] = pred
        for k_, v in d.items():
            arrs[f"{tag}/loss_{k_}"] = np.float32(v)
        for k_, v in g.items():
            arrs[f"{tag}/grad/{k_}"] = v
    save("g3_tiny_rollout_k4.npz", **arrs)


# ---------------------------------------------------------------- G4: loss-only matrix
def g4():
    R, B = 24, 2
    rs = np.random.RandomState(404)
    out = rs.normal(0, 0.6, (B, 3, R, R + 4)).astype(np.float32)       # non-square on purpose
    tgt = rs.normal(0, 0.6, (B, 3, R, R + 4)).astype(np.float32)
    inp = rs.normal(0, 0.6, (B, 6, R, R + 4)).astype(np.float32)
    mask_kinds = {}
    e = (rs.uniform(size=(B, R, R + 4)) > 0.8).astype(np.float32)
    c = (rs.uniform(size=(B, R, R + 4)) > 0.85).astype(np.float32)
    mask_kinds["partial"] = (e, c)
    mask_kinds["none"] = (np.zeros_like(e), np.zeros_like(c))
    mask_kinds["all_emitter"] = (np.ones_like(e), np.zeros_like(c))
    arrs = {"out": out, "tgt": tgt, "inp": inp, "emit_partial": e, "coll_partial": c}
    for pm in ("zeros", "replicate", "reflect", "circular"):
        for st in ("forward", "central"):
            gx, gy = ref_pl.compute_spatial_gradients(torch.from_numpy(out[:, 0]), 0.5, 0.25, pm, st)
            div = ref_pl.compute_divergence(torch.from_numpy(out[:, 1]), torch.from_numpy(out[:, 2]), 0.5, 0.25, pm, st)
            arrs[f"{pm}/{st}/gx"], arrs[f"{pm}/{st}/gy"], arrs[f"{pm}/{st}/div"] = gx, gy, div
            for mk, (em, co) in mask_kinds.items():
                i2 = inp.copy()
                i2[:, 4], i2[:, 5] = em, co
                o = torch.from_numpy(out).requires_grad_(True)
                crit = ref_pl.PhysicsAwareLoss(mse_weight=1.0, divergence_weight=0.3, gradient_weight=0.2,
                                               emitter_weight=0.4, grid_spacing=2.0 / R, padding_mode=pm, stencil_mode=st)
                loss, d = crit(o, torch.from_numpy(tgt), torch.from_numpy(i2))
                loss.backward()
                pre = f"{pm}/{st}/{mk}"
                arrs[f"{pre}/total"] = loss.detach()
                for k_, v in d.items():
                    arrs[f"{pre}/{k_}"] = np.float32(v)
                arrs[f"{pre}/dout"] = o.grad
    # gradient_loss without masks (l1 branch, physics_loss.py:162-164) and emitter loss alone
    arrs["nomask_gradient_loss"] = ref_pl.gradient_loss(torch.from_numpy(out[:, 0]), torch.from_numpy(tgt[:, 0]),
                                                        padding_mode="replicate", mode="central")
    arrs["emitter_loss"] = ref_pl.emitter_spawn_loss(torch.from_numpy(out[:, 0:1]), torch.from_numpy(inp[:, 0:1]),
                                                     torch.from_numpy(e[:, None]))
    save("g4_loss_matrix.npz", **arrs)


# ---------------------------------------------------------------- G5: alternate block types, forward only
G5_VARIANTS = {
    "batch_relu_nearest_avgpool": dict(norm="batch", act="relu", upsample="nearest", downsample="avgpool", padding_mode="zeros"),
    "group_silu_transpose_maxpool": dict(norm="group", act="silu", upsample="transpose", downsample="maxpool", padding_mode="reflect", group_norm_groups=3),
    "none_leaky_bilinear_stride_nores": dict(norm="none", act="leaky_relu", use_residual=False, padding_mode="circular"),
    "instance_sigmoid_tanh_mid2": dict(output_activation="sigmoid_tanh", bottleneck_blocks=2, depth=1),
    "instance_zeros_mid0": dict(padding_mode="zeros", bottleneck_blocks=0),
}


def g5():
    R, B = 32, 2
    x0, _, _ = orc.synth_batch(B, 1, R, seed=505)
    arrs = {}
    for name, over in G5_VARIANTS.items():
        cfg = tiny_cfg(**over)
        params = orc.make_params(cfg, seed=17)
        m = ref_model(cfg, params, train=True)       # train mode: batch norm uses batch statistics
        assert list(params.keys()) == [k for k, _ in m.named_parameters()], name
        arrs[f"{name}/out"] = m(torch.from_numpy(x0))
        arrs[f"{name}/param_names"] = np.array(list(params.keys()))
    save("g5_variants_fwd.npz", **arrs)


# ---------------------------------------------------------------- G7: clip + Adam, 3 steps
def g7():
    cfg = tiny_cfg()
    R, B, K = 32, 2, 2
    params = orc.make_params(cfg, seed=19)
    m = ref_model(cfg, params, train=True)
    crit = ref_pl.PhysicsAwareLoss(**loss_cfg_c2(R))
    opt = torch.optim.Adam(m.parameters(), lr=3e-3)                       # trainer.py:52
    arrs = {}
    for step in range(1, 4):
        x0, y, mk = orc.synth_batch(B, K, R, seed=700 + step)
        opt.zero_grad()
        loss, d, _ = ref_rollout(m, crit, torch.from_numpy(x0), torch.from_numpy(y), torch.from_numpy(mk), 1.10, False, False)
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.05)       # trainer.py:192-193 (small norm -> clip is active)
        opt.step()
        arrs[f"step{step}/loss"] = loss.detach()
        arrs[f"step{step}/grad_norm"] = norm
        if step in (1, 3):
            for k_, p in m.named_parameters():
                arrs[f"step{step}/param/{k_}"] = p.detach().clone()
    save("g7_adam_clip.npz", **arrs)


# ---------------------------------------------------------------- G8: 16-frame autoregressive rollout (eval)
def g8():
    cfg = tiny_cfg()
    R, frames = 32, 16
    params = orc.make_params(cfg, seed=23)
    m = ref_model(cfg, params, train=False)
    em, co = orc.simple_infer_masks(R, scale=R / 128.0)
    prev = np.zeros((3, R, R), np.float32)
    cur = np.zeros((3, R, R), np.float32)
    outs = []
    with torch.no_grad():
        for _ in range(frames):                                              # simple_infer.py:226-260
            inp = np.concatenate([cur, prev[0:1], em[None], co[None]], 0)[None]
            out = m(torch.from_numpy(inp)).numpy()[0]
            prev = cur.copy()
            cur = out
            outs.append(out)
    save("g8_rollout16.npz", frames=np.stack(outs), emitter=em, collider=co)


# ---------------------------------------------------------------- G6: dataset tuples + augmentation from the real dataset module
def _stub_pydantic_settings():
    import types

    import pydantic
    if "pydantic_settings" not in sys.modules:
        st = types.ModuleType("pydantic_settings")
        st.BaseSettings = pydantic.BaseModel
        st.SettingsConfigDict = dict
        sys.modules["pydantic_settings"] = st


def g6():
    import tempfile
    from pathlib import Path

    import fixture_inputs as fi
    _stub_pydantic_settings()
    from dataset import npz_sequence as ref_ds           # reference
    from dataset import augmentations as ref_aug         # reference
    arrs = {}
    with tempfile.TemporaryDirectory() as td:
        root = fi.write_synth_npz(Path(td))
        for K in (1, 4):
            for preload in (False, True):
                for stride in (1, 3):
                    ds = ref_ds.FluidNPZSequenceDataset(root, "train", rollout_steps=K, preload=preload, stride=stride)
                    pre = f"K{K}/pre{int(preload)}/s{stride}"
                    arrs[f"{pre}/len"] = np.int64(len(ds))
                    arrs[f"{pre}/index0"] = np.array(ds._index, np.int64)
                    ds.set_epoch(1)
                    arrs[f"{pre}/index1"] = np.array(ds._index, np.int64)
                    ds.set_epoch(0)
                    for idx in (0, len(ds) // 2, len(ds) - 1):
                        for j, t in enumerate(ds[idx]):
                            arrs[f"{pre}/item{idx}/{j}"] = t
        ds = ref_ds.FluidNPZSequenceDataset(root, "val", rollout_steps=2, preload=True)      # the file without masks
        for j, t in enumerate(ds[1]):
            arrs[f"val_nomask/K2/item1/{j}"] = t
        ds = ref_ds.FluidNPZSequenceDataset(root, "val", rollout_steps=1, preload=False)
        for j, t in enumerate(ds[2]):
            arrs[f"val_nomask/K1/item2/{j}"] = t
        # normalisation through the module-level loaders (the class would read <project>/data/_field_stats.yaml, absent)
        p0 = sorted((root / "train").iterdir())[0]
        for j, t in enumerate(ref_ds._load_sample(p0, 3, True, fi.G6_SCALES)):
            arrs[f"norm/single/{j}"] = t
        for j, t in enumerate(ref_ds._load_rollout_sample(p0, 2, 4, True, fi.G6_SCALES)):
            arrs[f"norm/rollout/{j}"] = t
        # preloaded + normalised: the in-memory path with the scales injected after construction
        ds = ref_ds.FluidNPZSequenceDataset(root, "train", rollout_steps=4, preload=True)
        ds.normalize, ds._norm_scales = True, dict(fi.G6_SCALES)
        for j, t in enumerate(ds[1]):
            arrs[f"norm/preloaded_rollout/{j}"] = t
        # flip augmentation forced on / off (augmentations.py:14-35)
        ds1 = ref_ds.FluidNPZSequenceDataset(root, "train", rollout_steps=1, preload=True)
        ds4 = ref_ds.FluidNPZSequenceDataset(root, "train", rollout_steps=4, preload=False)
        x, y = ds1[2]
        for tag, p in (("on", 1.0), ("off", 0.0)):
            fx, fy = ref_aug.apply_augmentation(x.clone(), y.clone(), p)
            arrs[f"flip_{tag}/single/0"], arrs[f"flip_{tag}/single/1"] = fx, fy
            x0, ys, mk = ds4[2]
            for j, t in enumerate(ref_aug.apply_rollout_augmentation(x0.clone(), ys.clone(), mk.clone(), p)):
                arrs[f"flip_{tag}/rollout/{j}"] = t
    save("g6_dataset.npz", **arrs)


# ---------------------------------------------------------------- G9: VariantManager (curriculum configuration plumbing)
def g9():
    import json
    import tempfile
    from pathlib import Path

    import fixture_inputs as fi
    from scripts.variant_manager import VariantManager   # reference

    def dump_tree(cfg_root, ck_root):
        m = VariantManager(Path(cfg_root), Path(ck_root))
        names = sorted(m._variants_cache)
        out = {"variants": names, "configs": {}, "chains": {}, "parents": {}, "relative_dirs": {}, "warmstart": {}}
        for n in names:
            cfg = m.build_config_with_inheritance(n)
            out["configs"][n] = json.loads(json.dumps(cfg, default=str))
            out["chains"][n] = m.resolve_dependency_chain(n)
            out["parents"][n] = m.get_parent_variant(n)
            out["relative_dirs"][n] = str(m.get_variant_relative_dir(n))
            try:
                w = m.get_warmstart_checkpoint(n)
            except FileNotFoundError as e:      # no parent checkpoint on disk: the path it looked for is in the message
                w = str(e).split("Parent checkpoint not found: ")[1].split("\n")[0]
            out["warmstart"][n] = None if w is None else str(Path(w).relative_to(ck_root))
        out["topo_all"] = m.topological_sort(names)
        out["topo_reversed"] = m.topological_sort(list(reversed(names)))
        return out

    res = {}
    with tempfile.TemporaryDirectory() as td:
        res["synthetic"] = dump_tree(fi.write_synth_config_tree(Path(td)), Path(td) / "ck")
        res["synthetic_topo_subset"] = VariantManager(Path(td) / "config", Path(td) / "ck").topological_sort(["A-K4", "B-solo"])
    # the reference's own configuration tree (the test that reads it only runs where /root/reference exists)
    res["reference_tree"] = dump_tree("/root/reference/ml/config", "/tmp/ck_root")
    path = os.path.join(HERE, "g9_variants.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(f"wrote g9_variants.json: {os.path.getsize(path) / 1024:.1f} KiB, {len(res['reference_tree']['variants'])} reference variants")


# ---------------------------------------------------------------- G10: the seven non-SSIM validation metrics
def g10():
    import types

    import fixture_inputs as fi
    if "torchmetrics" not in sys.modules:
        tm, tf_, ti = types.ModuleType("torchmetrics"), types.ModuleType("torchmetrics.functional"), types.ModuleType("torchmetrics.functional.image")

        def _absent(*a, **k):
            raise RuntimeError("torchmetrics is not installed: SSIM stays unpinned")
        ti.structural_similarity_index_measure = _absent
        tm.functional, tf_.image = tf_, ti
        sys.modules.update({"torchmetrics": tm, "torchmetrics.functional": tf_, "torchmetrics.functional.image": ti})
    from training import metrics as ref_m                # reference
    pred, tgt, em, co = (torch.from_numpy(a) for a in fi.metrics_inputs())
    arrs = {}
    for k, v in ref_m.compute_per_channel_mse(pred, tgt).items():
        arrs[k] = np.float64(v)
    for pm in ("zeros", "replicate", "reflect", "circular"):
        for st in ("forward", "central"):
            arrs[f"gradient_l1/{pm}/{st}"] = np.float64(ref_m.compute_gradient_l1(pred[:, 0], tgt[:, 0], 0.0625, 0.0625, pm, st))
            arrs[f"divergence_norm/{pm}/{st}"] = np.float64(ref_m.compute_divergence_norm(pred[:, 1], pred[:, 2], 0.0625, 0.0625, pm, st))
    arrs["kinetic_energy"] = np.float64(ref_m.compute_kinetic_energy(pred[:, 1], pred[:, 2]))
    arrs["collider_violation"] = np.float64(ref_m.compute_collider_violation(pred[:, 0], co))
    arrs["collider_violation_none"] = np.float64(ref_m.compute_collider_violation(pred[2:, 0], co[2:]))
    arrs["emitter_density_accuracy"] = np.float64(ref_m.compute_emitter_density_accuracy(pred[:, 0], em, 0.8))
    arrs["emitter_density_accuracy_none"] = np.float64(ref_m.compute_emitter_density_accuracy(pred[:, 0], torch.zeros_like(em), 0.8))
    tr = ref_m.MetricsTracker()
    tr.update({"a": 1.0, "b": 2.0})
    tr.update({"a": 3.0})
    for k, v in tr.compute_averages().items():
        arrs[f"tracker/{k}"] = np.float64(v)
    save("g10_metrics.npz", **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10"]
    for w in which:
        globals()[w]()
