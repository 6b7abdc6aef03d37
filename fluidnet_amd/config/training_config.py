"""The training-side configuration records of the hot path, under the reference's names: `PhysicsLossConfig`,
`AugmentationConfig`, `VariantMetadata`, `TrainingConfig` (field names, defaults and validation errors follow
ml/config/training_config.py:10-155 because YAML variants and callers address them by name).

The records are generated from the schema tables below (one row per field, grouped by the subsystem that reads it) with
`pydantic.create_model`, so the set of knobs the HIP plan honours is a table one can diff against the reference rather than
four hand-kept classes.  Values derived from the reference's project config (grid resolution 128 and 6 input channels:
config.yaml:1-24) are constants here; the MLflow names and variant bookkeeping are carried as inert data."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, Literal, Optional, Tuple

from pydantic import BaseModel, create_model, field_validator

# the closed value sets of the block options (ml/models/unet.py:9-14) and of the loss / schedule selectors: a YAML typo fails at load time
NormType = Literal["none", "batch", "instance", "group"]
ActType = Literal["relu", "leaky_relu", "gelu", "silu"]
UpsampleType = Literal["nearest", "bilinear", "transpose"]
DownsampleType = Literal["stride", "avgpool", "maxpool"]
PaddingType = Literal["zeros", "reflect", "replicate", "circular"]
OutputActivationType = Literal["sigmoid_tanh", "linear_clamp"]
StencilMode = Literal["central", "forward"]

GRID_RESOLUTION = 128      # config.yaml simulation.grid_resolution
INPUT_CHANNELS = 6         # config.yaml simulation.input_channels

_REQUIRED = ...


def _record(name: str, *groups: Dict[str, Tuple[Any, Any]], base=BaseModel, doc: str = ""):
    fields: Dict[str, Tuple[Any, Any]] = {}
    for g in groups:
        fields.update(g)
    model = create_model(name, __base__=base, __module__=__name__, **fields)
    model.__doc__ = doc
    return model


# ---------------------------------------------------------------------------------------------------------------- loss
PhysicsLossConfig = _record("PhysicsLossConfig", {
    # weights of the four terms of training/physics_loss.py, and which of them are evaluated at all
    "mse_weight": (float, 1.0), "divergence_weight": (float, 0.001),
    "gradient_weight": (float, 0.002), "emitter_weight": (float, 0.05),
    "enable_divergence": (bool, True), "enable_gradient": (bool, False), "enable_emitter": (bool, False),
    # finite-difference stencil ("forward" | "central") and cell size of the [-1, 1] domain
    "stencil_mode": (StencilMode, "forward"), "grid_spacing": (float, 2.0 / GRID_RESOLUTION),
}, doc="ml/config/training_config.py:10-20")

AugmentationConfig = _record("AugmentationConfig", {
    "enable_augmentation": (bool, True), "flip_probability": (float, 0.5), "flip_axis": (str, "x"),
}, doc="ml/config/training_config.py:23-27")

VariantMetadata = _record("VariantMetadata", {
    "variant_name": (str, _REQUIRED), "model_architecture_name": (str, _REQUIRED), "full_model_name": (str, _REQUIRED),
    "parent_variant": (Optional[str], None), "warmstart_checkpoint": (Optional[Path], None),
    "variant_yaml_path": (Path, _REQUIRED), "relative_dir": (Path, _REQUIRED),
}, doc="ml/config/training_config.py:30-37")

# ------------------------------------------------------------------------------------------------------------ training
_RUN = {
    "batch_size": (int, 4), "learning_rate": (float, 0.0007), "epochs": (int, 200), "device": (Optional[str], "cuda"),
    # reference: fp16 autocast + GradScaler; here: selects the bf16 MFMA plan (no loss scaling needed)
    "amp_enabled": (bool, True),
    "num_workers": (int, 4),
}
_DATA = {
    "npz_dir": (Path, Path("data/npz")), "normalize": (bool, True),
    "split_ratios": (Tuple[float, float, float], (0.70, 0.15, 0.15)), "split_seed": (int, 42),
    "augmentation": (AugmentationConfig, AugmentationConfig()), "preload_dataset": (bool, True),
}
_MODEL = {      # UNetConfig fields (models/unet.py), flattened into the training record as the reference does
    "in_channels": (int, INPUT_CHANNELS), "out_channels": (int, 3), "base_channels": (int, 32), "depth": (int, 3),
    "norm": (NormType, "instance"), "act": (ActType, "gelu"), "group_norm_groups": (int, 8), "dropout": (float, 0.0),
    "upsample": (UpsampleType, "bilinear"), "downsample": (DownsampleType, "stride"), "padding_mode": (PaddingType, "replicate"),
    "use_residual": (bool, True), "bottleneck_blocks": (int, 1), "output_activation": (OutputActivationType, "linear_clamp"),
}
_BOOKKEEPING = {
    "mlflow_tracking_uri": (str, "./mlruns"), "mlflow_experiment_name": (str, "fluid_baseline_v1"),
    "checkpoint_dir": (Path, Path("data/pytorch-models")),
    "save_every_n_epochs": (int, 5), "keep_last_n_checkpoints": (int, 20),
    "variant": (Optional[VariantMetadata], None),
}
_OPTIMISER = {
    "gradient_clip_norm": (float, 1.0), "gradient_clip_enabled": (bool, True),
    "use_lr_scheduler": (bool, True), "lr_scheduler_type": (str, "plateau"),
    "lr_scheduler_patience": (int, 4), "lr_scheduler_factor": (float, 0.5), "lr_scheduler_min_lr": (float, 1e-6),
    "lr_scheduler_step_size": (int, 30), "lr_scheduler_t_max": (int, 100),
    "use_early_stopping": (bool, True), "early_stop_patience": (int, 10), "early_stop_min_delta": (float, 0),
    "physics_loss": (PhysicsLossConfig, PhysicsLossConfig()),
}
_ROLLOUT = {    # K-step rollout training (training/trainer.py:210-330)
    "rollout_step": (int, 0), "rollout_stride": (float, 1.0), "rollout_weight_decay": (float, 1.10),
    "rollout_gradient_truncation": (bool, False), "validation_use_rollout_k": (bool, True),
    "rollout_final_step_only": (bool, False),
}


class _TrainingBase(BaseModel):
    @property
    def checkpoint_dir_variant(self) -> Path:
        """Where this run's checkpoints go (ml/config/training_config.py:131-135)."""
        v = self.variant
        leaf = Path("Unet") if not v else Path(v.relative_dir) / v.full_model_name
        return self.checkpoint_dir / leaf

    @field_validator("rollout_step", check_fields=False)
    @classmethod
    def _nonnegative_k(cls, k: int) -> int:
        if k < 0:
            raise ValueError(f"rollout_step must be >= 0, got {k}")
        return k

    @field_validator("rollout_stride", check_fields=False)
    @classmethod
    def _unit_interval_stride(cls, s: float) -> float:
        if not 0.0 < s <= 1.0:
            raise ValueError(f"rollout_stride must be in (0.0, 1.0], got {s}")
        return s


TrainingConfig = _record("TrainingConfig", _RUN, _DATA, _MODEL, _BOOKKEEPING, _OPTIMISER, _ROLLOUT, base=_TrainingBase,
                         doc="ml/config/training_config.py:40-155")
