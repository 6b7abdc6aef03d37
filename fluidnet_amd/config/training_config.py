"""TrainingConfig / PhysicsLossConfig / AugmentationConfig with the reference's field names and defaults
(ml/config/training_config.py:10-155), as plain pydantic models.  Project-config derived defaults (grid resolution 128,
6 input channels: config.yaml:1-24) are constants here; MLflow / variant bookkeeping fields are kept as inert data."""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

from pydantic import BaseModel, field_validator

GRID_RESOLUTION = 128      # config.yaml simulation.grid_resolution
INPUT_CHANNELS = 6         # config.yaml simulation.input_channels


class PhysicsLossConfig(BaseModel):
    mse_weight: float = 1.0
    divergence_weight: float = 0.001
    gradient_weight: float = 0.002
    emitter_weight: float = 0.05
    enable_divergence: bool = True
    enable_gradient: bool = False
    enable_emitter: bool = False
    stencil_mode: str = "forward"
    grid_spacing: float = 2.0 / GRID_RESOLUTION


class AugmentationConfig(BaseModel):
    enable_augmentation: bool = True
    flip_probability: float = 0.5
    flip_axis: str = "x"


class VariantMetadata(BaseModel):
    """ml/config/training_config.py:30-37"""
    variant_name: str
    model_architecture_name: str
    full_model_name: str
    parent_variant: Optional[str] = None
    warmstart_checkpoint: Optional[Path] = None
    variant_yaml_path: Path
    relative_dir: Path


class TrainingConfig(BaseModel):
    batch_size: int = 4
    learning_rate: float = 0.0007
    epochs: int = 200
    device: Optional[str] = "cuda"
    amp_enabled: bool = True           # reference: fp16 autocast + GradScaler; here: selects the bf16 MFMA plan (no loss scaling)
    num_workers: int = 4

    npz_dir: Path = Path("data/npz")
    normalize: bool = True
    split_ratios: Tuple[float, float, float] = (0.70, 0.15, 0.15)
    split_seed: int = 42
    augmentation: AugmentationConfig = AugmentationConfig()
    preload_dataset: bool = True

    in_channels: int = INPUT_CHANNELS
    out_channels: int = 3
    base_channels: int = 32
    depth: int = 3
    norm: str = "instance"
    act: str = "gelu"
    group_norm_groups: int = 8
    dropout: float = 0.0
    upsample: str = "bilinear"
    downsample: str = "stride"
    padding_mode: str = "replicate"
    use_residual: bool = True
    bottleneck_blocks: int = 1
    output_activation: str = "linear_clamp"

    mlflow_tracking_uri: str = "./mlruns"
    mlflow_experiment_name: str = "fluid_baseline_v1"

    checkpoint_dir: Path = Path("data/pytorch-models")
    save_every_n_epochs: int = 5
    keep_last_n_checkpoints: int = 20

    gradient_clip_norm: float = 1.0
    gradient_clip_enabled: bool = True

    use_lr_scheduler: bool = True
    lr_scheduler_type: str = "plateau"
    lr_scheduler_patience: int = 4
    lr_scheduler_factor: float = 0.5
    lr_scheduler_min_lr: float = 1e-6
    lr_scheduler_step_size: int = 30
    lr_scheduler_t_max: int = 100

    use_early_stopping: bool = True
    early_stop_patience: int = 10
    early_stop_min_delta: float = 0

    physics_loss: PhysicsLossConfig = PhysicsLossConfig()
    variant: Optional[VariantMetadata] = None

    rollout_step: int = 0
    rollout_stride: float = 1.0
    rollout_weight_decay: float = 1.10
    rollout_gradient_truncation: bool = False
    validation_use_rollout_k: bool = True
    rollout_final_step_only: bool = False

    @property
    def checkpoint_dir_variant(self) -> Path:
        """ml/config/training_config.py:131-135"""
        if self.variant:
            return self.checkpoint_dir / self.variant.relative_dir / self.variant.full_model_name
        return self.checkpoint_dir / "Unet"

    @field_validator("rollout_step")
    @classmethod
    def _v_rollout_step(cls, v: int) -> int:
        if v < 0:
            raise ValueError(f"rollout_step must be >= 0, got {v}")
        return v

    @field_validator("rollout_stride")
    @classmethod
    def _v_rollout_stride(cls, v: float) -> float:
        if v <= 0.0 or v > 1.0:
            raise ValueError(f"rollout_stride must be in (0.0, 1.0], got {v}")
        return v
