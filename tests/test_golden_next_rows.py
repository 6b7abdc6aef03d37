"""The "next" rows of SURVEY.md section 8 against fixtures generated from the REAL reference modules (tests/golden/make_golden.py,
round 2): g6 = ml/dataset/npz_sequence.py + augmentations.py (imported with a pydantic_settings stub), g9 =
ml/scripts/variant_manager.py, g10 = the seven non-SSIM functions of ml/training/metrics.py (imported with a torchmetrics stub).
Inputs are regenerated from tests/fixture_inputs.py on both sides.  CPU only; the GPU kernels (GATHER_BATCH, METRICS) meet the same
fixtures in tests/test_hbm_loader.py and tests/test_gpu_metrics.py."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

import fixture_inputs as fi
import fluidnet_oracle as orc
from fluidnet_amd.dataset import npz_sequence as ds_mod
from fluidnet_amd.dataset.augmentations import apply_augmentation, apply_rollout_augmentation
from fluidnet_amd.training.variants import VariantManager


@pytest.fixture(scope="module")
def g6(golden_dir):
    return np.load(os.path.join(golden_dir, "g6_dataset.npz"))


@pytest.fixture()
def npz_root(tmp_path):
    return fi.write_synth_npz(tmp_path)


def _same(t, ref):
    t = t.numpy() if torch.is_tensor(t) else np.asarray(t)
    return t.shape == ref.shape and t.dtype == ref.dtype and np.array_equal(t, ref)


@pytest.mark.parametrize("K", [1, 4])
@pytest.mark.parametrize("preload", [False, True])
@pytest.mark.parametrize("stride", [1, 3])
def test_g6_dataset_tuples_index_and_epoch_offsets(g6, npz_root, K, preload, stride):
    ds = ds_mod.FluidNPZSequenceDataset(npz_root, "train", rollout_steps=K, preload=preload, stride=stride)
    pre = f"K{K}/pre{int(preload)}/s{stride}"
    assert len(ds) == int(g6[f"{pre}/len"])
    assert np.array_equal(np.array(ds._index, np.int64), g6[f"{pre}/index0"])
    ds.set_epoch(1)
    assert np.array_equal(np.array(ds._index, np.int64), g6[f"{pre}/index1"])
    ds.set_epoch(0)
    for idx in (0, len(ds) // 2, len(ds) - 1):
        item = ds[idx]
        assert len(item) == (2 if K == 1 else 3)
        for j, t in enumerate(item):        # bit for bit, shapes included: the on-disk path carries K+1 mask frames, the preloaded one K
            assert _same(t, g6[f"{pre}/item{idx}/{j}"]), (pre, idx, j)


def test_g6_files_without_masks_normalisation_and_flip(g6, npz_root):
    for j, t in enumerate(ds_mod.FluidNPZSequenceDataset(npz_root, "val", rollout_steps=2, preload=True)[1]):
        assert _same(t, g6[f"val_nomask/K2/item1/{j}"])
    for j, t in enumerate(ds_mod.FluidNPZSequenceDataset(npz_root, "val", rollout_steps=1, preload=False)[2]):
        assert _same(t, g6[f"val_nomask/K1/item2/{j}"])
    p0 = sorted((npz_root / "train").iterdir())[0]
    for j, t in enumerate(ds_mod._load_sample(p0, 3, True, fi.G6_SCALES)):
        assert _same(t, g6[f"norm/single/{j}"])
    for j, t in enumerate(ds_mod._load_rollout_sample(p0, 2, 4, True, fi.G6_SCALES)):
        assert _same(t, g6[f"norm/rollout/{j}"])
    ds = ds_mod.FluidNPZSequenceDataset(npz_root, "train", normalize=True, rollout_steps=4, preload=True, norm_scales=fi.G6_SCALES)
    for j, t in enumerate(ds[1]):
        assert _same(t, g6[f"norm/preloaded_rollout/{j}"])
    ds1 = ds_mod.FluidNPZSequenceDataset(npz_root, "train", rollout_steps=1, preload=True)
    ds4 = ds_mod.FluidNPZSequenceDataset(npz_root, "train", rollout_steps=4, preload=False)
    x, y = ds1[2]
    for tag, p in (("on", 1.0), ("off", 0.0)):
        fx, fy = apply_augmentation(x.clone(), y.clone(), p)
        assert _same(fx, g6[f"flip_{tag}/single/0"]) and _same(fy, g6[f"flip_{tag}/single/1"])
        for j, t in enumerate(apply_rollout_augmentation(*(a.clone() for a in ds4[2]), p)):
            assert _same(t, g6[f"flip_{tag}/rollout/{j}"])
    # the oracle's independent restatement meets the same fixture
    seq = {k: v for k, v in np.load(p0).items()}
    for j, t in enumerate(orc.make_sample(seq, 2, 4, preload=False, scales=fi.G6_SCALES)):
        assert np.array_equal(np.asarray(t), g6[f"norm/rollout/{j}"])


def _check_tree(m: VariantManager, ck_root: Path, want: dict):
    assert sorted(m._variants_cache) == want["variants"]
    for n in want["variants"]:
        cfg = json.loads(json.dumps(m.build_config_with_inheritance(n), default=str))
        assert cfg == want["configs"][n], n
        assert m.resolve_dependency_chain(n) == want["chains"][n]
        assert m.get_parent_variant(n) == want["parents"][n]
        assert str(m.get_variant_relative_dir(n)) == want["relative_dirs"][n]
        if want["warmstart"][n] is None:
            assert m.get_warmstart_checkpoint(n) is None
        else:
            path = ck_root / want["warmstart"][n]
            if not path.exists():                               # (siblings share a parent: an earlier iteration may have made it)
                with pytest.raises(FileNotFoundError) as e:     # no checkpoint on disk: same path, same message as the reference
                    m.get_warmstart_checkpoint(n)
                assert str(e.value).startswith(f"Parent checkpoint not found: {path}\n")
                path.parent.mkdir(parents=True, exist_ok=True)
                path.touch()
            assert m.get_warmstart_checkpoint(n) == path
    # the reference's order among independent roots depends on set iteration (string hashing); what it guarantees -- and what is
    # compared -- is the node set and every parent before its children
    for req, ref in ((want["variants"], want["topo_all"]), (list(reversed(want["variants"])), want["topo_reversed"])):
        got = m.topological_sort(req)
        assert sorted(got) == sorted(ref)
        for v in got:
            par = want["parents"][v]
            assert par is None or got.index(par) < got.index(v)


def test_g9_variant_manager_synthetic_tree(golden_dir, tmp_path):
    want = json.load(open(os.path.join(golden_dir, "g9_variants.json")))
    m = VariantManager(fi.write_synth_config_tree(tmp_path), tmp_path / "ck")
    _check_tree(m, tmp_path / "ck", want["synthetic"])
    assert sorted(m.topological_sort(["A-K4", "B-solo"])) == sorted(want["synthetic_topo_subset"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/ml/config"), reason="the reference's config tree only exists in the build container")
def test_g9_variant_manager_reference_tree(golden_dir, tmp_path):
    want = json.load(open(os.path.join(golden_dir, "g9_variants.json")))["reference_tree"]
    assert len(want["variants"]) >= 40
    _check_tree(VariantManager(Path("/root/reference/ml/config"), tmp_path / "ck"), tmp_path / "ck", want)


def test_g10_oracle_metrics(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_metrics.npz"))
    pred, tgt, em, co = (torch.from_numpy(a) for a in fi.metrics_inputs())
    for k, v in orc.per_channel_mse(pred, tgt).items():
        assert abs(v - float(g[k])) < 1e-7
    for pm in ("zeros", "replicate", "reflect", "circular"):
        for st in ("forward", "central"):
            assert abs(orc.gradient_l1(pred[:, 0], tgt[:, 0], 0.0625, 0.0625, pm, st) - float(g[f"gradient_l1/{pm}/{st}"])) < 1e-5
            assert abs(orc.divergence_norm(pred[:, 1], pred[:, 2], 0.0625, 0.0625, pm, st) - float(g[f"divergence_norm/{pm}/{st}"])) < 1e-5
    assert abs(orc.kinetic_energy(pred[:, 1], pred[:, 2]) - float(g["kinetic_energy"])) < 1e-3
    assert abs(orc.collider_violation(pred[:, 0], co) - float(g["collider_violation"])) < 1e-7
    assert orc.collider_violation(pred[2:, 0], co[2:]) == float(g["collider_violation_none"]) == 0.0
    assert abs(orc.emitter_density_accuracy(pred[:, 0], em, 0.8) - float(g["emitter_density_accuracy"])) < 1e-7
    assert orc.emitter_density_accuracy(pred[:, 0], torch.zeros_like(em), 0.8) == float(g["emitter_density_accuracy_none"]) == 0.0
