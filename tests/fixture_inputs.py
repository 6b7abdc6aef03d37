"""Seeded INPUTS of the g6 / g9 / g10 fixtures, shared by the generator (tests/golden/make_golden.py, which feeds them to the imported
reference) and by the tests (which feed them to fluidnet_amd).  Only numpy / yaml; nothing here comes from the reference."""
from pathlib import Path

import numpy as np
import yaml

G6_SCALES = {"S_density": 1.7, "S_velx": 0.45, "S_velz": 0.8}


def write_synth_npz(root: Path) -> Path:
    """<root>/npz/{train,val}/seq_NNNN.npz: two train sequences (T = 12 and 9) and one val sequence (T = 7), 16 x 20 grids
    (non-square on purpose), masks that move from frame to frame so that the time index of every mask plane is pinned."""
    rs = np.random.RandomState(606)
    H, W = 16, 20
    for split, lens in (("train", (12, 9)), ("val", (7,))):
        d = Path(root) / "npz" / split
        d.mkdir(parents=True, exist_ok=True)
        for i, T in enumerate(lens):
            em = np.zeros((T, H, W), np.float32)
            co = np.zeros((T, H, W), np.float32)
            for t in range(T):
                em[t, 2:5, (1 + t) % W:(4 + t) % W or W] = 1.0
                co[t, 9:12, (W - 5 - t) % W:(W - 2 - t) % W or W] = 1.0
            arrs = dict(density=rs.uniform(0, 1, (T, H, W)).astype(np.float32), velx=rs.normal(0, 0.3, (T, H, W)).astype(np.float32),
                        velz=rs.normal(0, 0.3, (T, H, W)).astype(np.float32), emitter=em, collider=co)
            if split == "val":                      # a file without masks: the loaders substitute zeros
                del arrs["emitter"], arrs["collider"]
            np.savez_compressed(d / f"seq_{i:04d}.npz", **arrs)
    return Path(root) / "npz"


def write_synth_config_tree(root: Path) -> Path:
    """A configuration tree of the reference's shape (base_config.yaml, model_architectures/*.yaml, variants/**/*.yaml) with a
    3-stage curriculum chain, a second architecture, nested overrides and a variant with no parent."""
    c = Path(root) / "config"
    (c / "model_architectures").mkdir(parents=True, exist_ok=True)
    (c / "variants" / "a" / "chain").mkdir(parents=True, exist_ok=True)
    (c / "variants" / "b").mkdir(parents=True, exist_ok=True)
    dump = lambda p, o: Path(p).write_text(yaml.safe_dump(o, sort_keys=False))  # noqa: E731
    dump(c / "base_config.yaml", {"batch_size": 8, "epochs": 3, "learning_rate": 1e-3, "rollout_step": 1, "normalize": False,
                                  "physics_loss": {"mse_weight": 1.0, "divergence_weight": 0.001, "enable_divergence": True,
                                                   "enable_gradient": False, "grid_spacing": 0.0625},
                                  "augmentation": {"enable_augmentation": True, "flip_probability": 0.5}})
    dump(c / "model_architectures" / "ArchA.yaml", {"base_channels": 8, "depth": 2, "norm": "instance", "act": "gelu", "dropout": 0.1,
                                                    "physics_loss": {"divergence_weight": 0.01}})
    dump(c / "model_architectures" / "ArchB.yaml", {"base_channels": 16, "depth": 1, "norm": "group", "act": "silu"})
    dump(c / "variants" / "a" / "chain" / "K1.yaml", {"variant_name": "A-K1", "model_architecture": "ArchA", "rollout_step": 1,
                                                      "physics_loss": {"enable_gradient": True, "gradient_weight": 0.002}})
    dump(c / "variants" / "a" / "chain" / "K2.yaml", {"variant_name": "A-K2", "parent_variant": "A-K1", "rollout_step": 2,
                                                      "learning_rate": 5e-4})
    dump(c / "variants" / "a" / "chain" / "K4.yaml", {"variant_name": "A-K4", "parent_variant": "A-K2", "rollout_step": 4, "epochs": 5,
                                                      "physics_loss": {"divergence_weight": 0.05}, "augmentation": {"flip_probability": 0.25}})
    dump(c / "variants" / "b" / "solo.yaml", {"variant_name": "B-solo", "model_architecture": "ArchB", "batch_size": 4})
    return c


def metrics_inputs():
    """pred / target [3,3,20,24], emitter / collider masks [3,20,24] (one scene without any collider cell)."""
    rs = np.random.RandomState(1010)
    B, H, W = 3, 20, 24
    pred = rs.normal(0.2, 0.5, (B, 3, H, W)).astype(np.float32)
    tgt = rs.normal(0.2, 0.5, (B, 3, H, W)).astype(np.float32)
    em = (rs.uniform(size=(B, H, W)) > 0.85).astype(np.float32)
    co = (rs.uniform(size=(B, H, W)) > 0.9).astype(np.float32)
    co[2] = 0.0
    return pred, tgt, em, co
