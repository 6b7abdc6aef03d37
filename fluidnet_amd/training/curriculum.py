"""Curriculum driver with the reference's entry points (ml/scripts/train.py:22-383): train one variant from its merged YAML
configuration -- dataset, model, optional warm start from the parent's best_model.pth, Trainer -- and a list of variants
in dependency order (K1 -> K2 -> K3 -> K4), skipping dependencies whose checkpoint already exists.

Differences from the reference, all deliberate: checkpoints are read with weights_only=True (nothing in the file executes);
MLflow logging, the post-training test/rollout evaluation plots and auto-inference are out of scope;
`use_hbm_loader=True` feeds the Trainer from the HBM-resident store instead of a DataLoader.

    python -m fluidnet_amd.training.curriculum --config-root <ml/config> --npz-dir <data/npz/128> --checkpoints <dir> VARIANT...
"""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Dict, List, Optional

import torch
from torch.utils.data import DataLoader

from ..config.training_config import AugmentationConfig, PhysicsLossConfig, TrainingConfig, VariantMetadata
from ..dataset import FluidNPZSequenceDataset
from ..models.unet import UNet, UNetConfig
from .trainer import Trainer
from .variants import VariantManager


def dict_to_training_config(config_dict: dict) -> TrainingConfig:
    """ml/scripts/train.py:22-35; keys TrainingConfig does not know (auto_inference_enabled, ...) are dropped."""
    d = {k: v for k, v in config_dict.items() if k in TrainingConfig.model_fields}
    if isinstance(d.get("augmentation"), dict):
        d["augmentation"] = AugmentationConfig(**d["augmentation"])
    if isinstance(d.get("physics_loss"), dict):
        d["physics_loss"] = PhysicsLossConfig(**d["physics_loss"])
    return TrainingConfig(**d)


def variant_training_config(variant_name: str, manager: VariantManager, *, checkpoint_dir: Optional[Path] = None) -> TrainingConfig:
    merged = manager.build_config_with_inheritance(variant_name)
    meta = merged["_variant_metadata"]
    config = dict_to_training_config(merged)
    config.checkpoint_dir = Path(checkpoint_dir) if checkpoint_dir is not None else manager.checkpoints_dir
    config.variant = VariantMetadata(
        variant_name=variant_name, model_architecture_name=meta["model_architecture_name"], full_model_name=meta["full_model_name"],
        parent_variant=meta["parent_variant"], warmstart_checkpoint=manager.get_warmstart_checkpoint(variant_name),
        variant_yaml_path=manager.get_variant_yaml_path(variant_name), relative_dir=meta["relative_dir"])
    return config


def train_single_variant(variant_name: str, manager: VariantManager, npz_dir: Path, *, device: str = "cuda", precision: Optional[str] = None,
                         use_hbm_loader: bool = False, seed: Optional[int] = None, engine: str = "plan", evaluate: bool = False,
                         stats_path: Optional[Path] = None) -> Trainer:
    """ml/scripts/train.py:38-262 without the MLflow tail.  Returns the finished Trainer; with `evaluate`, the reference's post-training
    evaluation (train.py:247-256: best checkpoint -> run_test_evaluation + run_rollout_evaluation on the test split) is run and left in
    `trainer.evaluation`."""
    config = variant_training_config(variant_name, manager)
    config.device = device
    torch.manual_seed(config.split_seed if seed is None else seed)
    K = config.rollout_step
    aug = {"enable_augmentation": config.augmentation.enable_augmentation, "flip_probability": config.augmentation.flip_probability}
    train_stride = max(1, round(config.rollout_stride * max(K, 1)))
    train_ds = FluidNPZSequenceDataset(npz_dir, "train", normalize=config.normalize, is_training=True, augmentation_config=aug,
                                       preload=config.preload_dataset, rollout_steps=max(K, 1), stride=train_stride)
    val_ds = FluidNPZSequenceDataset(npz_dir, "val", normalize=config.normalize, is_training=False, augmentation_config=None,
                                     preload=config.preload_dataset, rollout_steps=max(K, 1) if config.validation_use_rollout_k else 1)
    if use_hbm_loader:
        from ..dataset.hbm_resident import HbmBatchLoader, HbmSequenceStore
        train_loader = HbmBatchLoader(HbmSequenceStore(train_ds, device), config.batch_size, shuffle=True, drop_last=True, seed=config.split_seed)
    else:
        train_loader = DataLoader(train_ds, batch_size=config.batch_size, shuffle=True, num_workers=0, drop_last=False)
    val_loader = DataLoader(val_ds, batch_size=config.batch_size, shuffle=False, num_workers=0)
    model = UNet(UNetConfig(in_channels=config.in_channels, out_channels=config.out_channels, base_channels=config.base_channels,
                            depth=config.depth, norm=config.norm, act=config.act, group_norm_groups=config.group_norm_groups,
                            dropout=config.dropout, upsample=config.upsample, downsample=config.downsample, padding_mode=config.padding_mode,
                            use_residual=config.use_residual, bottleneck_blocks=config.bottleneck_blocks,
                            output_activation=config.output_activation),
                 precision=precision or ("bf16" if config.amp_enabled else "fp32")).to(device)
    if config.variant.warmstart_checkpoint:
        print(f"Warm-start from parent: {config.variant.parent_variant}\nCheckpoint: {config.variant.warmstart_checkpoint}")
        ck = torch.load(config.variant.warmstart_checkpoint, map_location="cpu", weights_only=True)
        model.load_state_dict(ck["model_state_dict"])
    trainer = Trainer(model, train_loader, val_loader, config, device, engine=engine)
    trainer.train()
    print(f"Training complete: {config.variant.full_model_name}\nBest model: {trainer.config.checkpoint_dir / 'best_model.pth'}")
    trainer.evaluation = None
    if evaluate and (Path(npz_dir) / "test").exists():
        from .test_evaluation import run_rollout_evaluation, run_test_evaluation
        best = trainer.config.checkpoint_dir / "best_model.pth"
        if best.exists():
            model.load_state_dict(torch.load(best, map_location="cpu", weights_only=True)["model_state_dict"])
        trainer.evaluation = {"test": run_test_evaluation(model, config, device, npz_dir=npz_dir, stats_path=stats_path),
                              "rollout": run_rollout_evaluation(model, config, device, npz_dir=npz_dir, stats_path=stats_path)}
    return trainer


def plan_curriculum(manager: VariantManager, variants: List[str], *, no_deps: bool = False, force_retrain: bool = False) -> List[str]:
    """The variants to train, in order (ml/scripts/train.py:326-352): dependencies first; a dependency whose checkpoint exists is
    skipped unless force_retrain; explicitly requested variants are always (re)trained."""
    order = list(variants) if no_deps else manager.topological_sort(variants)
    if force_retrain:
        return order
    return [v for v in order if v in variants or not manager.check_checkpoint_exists(v)]


def train_curriculum(manager: VariantManager, variants: List[str], npz_dir: Path, **kw) -> Dict[str, Trainer]:
    no_deps, force = kw.pop("no_deps", False), kw.pop("force_retrain", False)
    out: Dict[str, Trainer] = {}
    for v in plan_curriculum(manager, variants, no_deps=no_deps, force_retrain=force):
        print(f"\n{'=' * 80}\nTraining variant: {v}\n{'=' * 80}\n")
        out[v] = train_single_variant(v, manager, npz_dir, **kw)
    return out


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description="Train FluidNet variants in dependency order on MI355X")
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--config-root", type=Path, required=True, help="directory with base_config.yaml, model_architectures/, variants/")
    ap.add_argument("--npz-dir", type=Path, required=True, help="directory with train/ and val/ seq_*.npz")
    ap.add_argument("--checkpoints", type=Path, required=True)
    ap.add_argument("--evaluate", action="store_true", help="run the test-split and 30-step rollout evaluations after each variant")
    ap.add_argument("--stats-path", type=Path, default=None, help="_field_stats.yaml (needed by --evaluate when the config normalises)")
    ap.add_argument("--no-deps", action="store_true")
    ap.add_argument("--force-retrain", action="store_true")
    ap.add_argument("--hbm-loader", action="store_true", help="keep the training split in HBM and gather batches on device")
    a = ap.parse_args(argv)
    m = VariantManager(a.config_root, a.checkpoints)
    train_curriculum(m, a.variants, a.npz_dir, no_deps=a.no_deps, force_retrain=a.force_retrain, use_hbm_loader=a.hbm_loader,
                     evaluate=a.evaluate, stats_path=a.stats_path)


if __name__ == "__main__":
    main()
