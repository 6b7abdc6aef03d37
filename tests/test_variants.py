"""VariantManager / curriculum planning (fluidnet_amd/training/variants.py, curriculum.py) on a synthetic config tree and,
when the reference checkout is present (this container, not the GPU box), on the reference's own ml/config tree."""
import os
from pathlib import Path

import pytest
import yaml

from fluidnet_amd.training.curriculum import dict_to_training_config, plan_curriculum, variant_training_config
from fluidnet_amd.training.variants import VariantManager

REF_CFG = Path("/root/reference/ml/config")


@pytest.fixture()
def tree(tmp_path):
    root = tmp_path / "config"
    (root / "model_architectures").mkdir(parents=True)
    (root / "variants" / "a" / "b").mkdir(parents=True)
    (root / "base_config.yaml").write_text(yaml.safe_dump({"batch_size": 8, "epochs": 3, "physics_loss": {"mse_weight": 1.0, "divergence_weight": 0.5},
                                                           "auto_inference_enabled": False}))
    (root / "model_architectures" / "Small.yaml").write_text(yaml.safe_dump({"base_channels": 8, "depth": 2, "dropout": 0.0}))
    v = root / "variants" / "a" / "b"
    (v / "K1.yaml").write_text(yaml.safe_dump({"variant_name": "K1", "model_architecture": "Small", "rollout_step": 1, "learning_rate": 1e-3}))
    (v / "K2.yaml").write_text(yaml.safe_dump({"variant_name": "K2", "parent_variant": "K1", "rollout_step": 2,
                                               "physics_loss": {"divergence_weight": 0.1}}))
    (v / "K3.yaml").write_text(yaml.safe_dump({"variant_name": "K3", "parent_variant": "K2", "rollout_step": 3, "epochs": 1}))
    (root / "variants" / "a" / "note.yaml").write_text("just: a note without variant_name\n")
    return root, tmp_path / "ck"


def test_merge_chain_and_metadata(tree):
    root, ck = tree
    m = VariantManager(root, ck)
    assert sorted(m._variants_cache) == ["K1", "K2", "K3"]
    assert m.resolve_dependency_chain("K3") == ["K1", "K2", "K3"]
    c = m.build_config_with_inheritance("K3")
    assert c["rollout_step"] == 3 and c["epochs"] == 1 and c["learning_rate"] == 1e-3 and c["base_channels"] == 8
    assert c["physics_loss"] == {"mse_weight": 1.0, "divergence_weight": 0.1}          # deep merge keeps sibling keys
    meta = c["_variant_metadata"]
    assert meta["full_model_name"] == "Small_K3" and meta["parent_variant"] == "K2" and meta["relative_dir"] == Path("a/b")
    assert m.topological_sort(["K3"]) == ["K1", "K2", "K3"] and m.topological_sort(["K2", "K1"]) == ["K1", "K2"]
    with pytest.raises(FileNotFoundError):
        m.get_variant_yaml_path("nope")
    with pytest.raises(FileNotFoundError, match="Train K1 before"):
        m.get_warmstart_checkpoint("K2")
    assert m.get_warmstart_checkpoint("K1") is None


def test_errors(tree):
    root, ck = tree
    v = root / "variants" / "a" / "b"
    (v / "dup.yaml").write_text(yaml.safe_dump({"variant_name": "K1", "model_architecture": "Small"}))
    with pytest.raises(ValueError, match="Duplicate"):
        VariantManager(root, ck)
    (v / "dup.yaml").unlink()
    (v / "X.yaml").write_text(yaml.safe_dump({"variant_name": "X", "parent_variant": "Y"}))
    (v / "Y.yaml").write_text(yaml.safe_dump({"variant_name": "Y", "parent_variant": "X"}))
    m = VariantManager(root, ck)
    with pytest.raises(ValueError, match="Circular"):
        m.resolve_dependency_chain("X")
    (v / "Z.yaml").write_text(yaml.safe_dump({"variant_name": "Z", "parent_variant": "K1", "model_architecture": "Other"}))
    with pytest.raises(ValueError, match="Architecture mismatch"):
        VariantManager(root, ck).build_config_with_inheritance("Z")


def test_curriculum_plan_and_config(tree):
    root, ck = tree
    m = VariantManager(root, ck)
    assert plan_curriculum(m, ["K3"]) == ["K1", "K2", "K3"]
    best = ck / "a" / "b" / "Small_K1" / "best_model.pth"
    best.parent.mkdir(parents=True)
    best.write_bytes(b"x")
    assert m.check_checkpoint_exists("K1") and not m.check_checkpoint_exists("K2")
    assert plan_curriculum(m, ["K3"]) == ["K2", "K3"]                   # trained dependency skipped
    assert plan_curriculum(m, ["K3", "K1"]) == ["K1", "K2", "K3"]       # explicitly requested: retrained
    assert plan_curriculum(m, ["K3"], force_retrain=True) == ["K1", "K2", "K3"]
    assert plan_curriculum(m, ["K3"], no_deps=True) == ["K3"]
    cfg = variant_training_config("K2", m)
    assert cfg.rollout_step == 2 and cfg.batch_size == 8 and cfg.physics_loss.divergence_weight == 0.1
    assert cfg.variant.warmstart_checkpoint == best and cfg.variant.full_model_name == "Small_K2"
    assert cfg.checkpoint_dir_variant == ck / "a" / "b" / "Small_K2"
    assert dict_to_training_config({"batch_size": 2, "unknown_key": 1, "variant_name": "v"}).batch_size == 2


@pytest.mark.skipif(not REF_CFG.exists(), reason="reference checkout not present (GPU box)")
def test_reference_config_tree_loads():
    m = VariantManager(REF_CFG, Path("/nonexistent/checkpoints"))
    assert len(m._variants_cache) >= 10
    name = "Full-K4-Unet_medium-div_grad"
    chain = m.resolve_dependency_chain(name)
    assert chain[-1] == name and chain[0].startswith("Full-K1") and len(chain) == 4
    d = m.build_config_with_inheritance(name)
    assert d["rollout_step"] == 4 and d["base_channels"] == 32 and d["depth"] == 3 and d["downsample"] == "stride"
    cfg = dict_to_training_config(dict(d))
    assert cfg.rollout_step == 4 and cfg.norm == "instance"
    assert m.topological_sort([name]) == chain
