"""Post-training evaluation with the reference's function names and result dictionaries (ml/training/test_evaluation.py):

* `run_test_evaluation` (`:78-143`)  -- single-step metrics of the model and of the persistence baseline over the test split;
* `run_rollout_evaluation` (`:364-476`) -- the 30-step autoregressive degradation curves from the reference's three starting points
  per test sequence, ground-truth emitter / collider masks fed frame by frame (`_run_single_rollout`, `:192-292`).

MI355X shape of the second one: the reference runs its rollouts one after the other at batch 1; here every (sequence, start) pair of
the test split is one scene of ONE batched RolloutInferPlan (a hipGraph replay per frame, state resident in HBM), and the per-step
metrics are then taken per scene -- the reference's numbers are per-rollout means, and `divergence_norm` (a root of a mean) does not
commute with batching.  MLflow logging and the matplotlib figure stay out of scope: the returned
dictionaries are what the reference logs.  Metrics come from the METRICS kernel (fluidnet_amd/training/metrics.py; SSIM parity
unpinned, see there)."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from ..dataset.normalization import load_normalization_scales
from ..dataset.npz_sequence import FluidNPZSequenceDataset
from ..engine.rollout_plan import RolloutInferPlan
from .metrics import MetricsTracker, validation_metrics

ROLLOUT_STEPS = 30
STARTING_POINTS: List[Union[int, str]] = [20, "middle", -31]
ROLLOUT_KEYS = ("mse_density", "mse_velx", "mse_vely", "ssim_density", "divergence_norm", "divergence_norm_gt", "gradient_l1")


def _get_starting_frames(total_frames: int, steps: int = ROLLOUT_STEPS, points: Sequence[Union[int, str]] = STARTING_POINTS) -> List[int]:
    """test_evaluation.py:177-189: frame 20, the middle frame and the 31st from the end, kept when the rollout fits"""
    starts = []
    for sp in points:
        if sp == "middle":
            t = total_frames // 2
        elif isinstance(sp, int) and sp < 0:
            t = total_frames + sp
        else:
            t = int(sp)
        if t >= 1 and t + steps < total_frames:
            starts.append(t)
    return starts


def _batch_metrics(outputs: torch.Tensor, targets: torch.Tensor, inputs: torch.Tensor, config) -> Dict[str, float]:
    """test_evaluation.py:38-75 (the nine validation metrics of one batch)"""
    pl = config.physics_loss
    return validation_metrics(outputs, targets, inputs[:, 4], inputs[:, 5], dx=pl.grid_spacing, padding_mode=config.padding_mode,
                              mode=pl.stencil_mode)


def run_test_evaluation(model, config, device: str = "cuda", npz_dir: Optional[Path] = None, test_loader=None,
                        stats_path: Optional[Path] = None) -> Dict[str, Dict[str, float]]:
    """{"model": averages, "persistence": averages} over the test split (rollout_steps = 1 tuples).  `npz_dir` is the directory that
    holds `test/seq_*.npz` (the reference derives it from its project config: config.npz_dir / grid resolution)."""
    if test_loader is None:
        from torch.utils.data import DataLoader
        ds = FluidNPZSequenceDataset(npz_dir=Path(npz_dir if npz_dir is not None else config.npz_dir), split="test", normalize=config.normalize,
                                     is_training=False, augmentation_config=None, preload=config.preload_dataset, rollout_steps=1,
                                     stats_path=stats_path)
        test_loader = DataLoader(ds, batch_size=config.batch_size, shuffle=False, num_workers=0)
    model_tracker, persistence_tracker = MetricsTracker(), MetricsTracker()
    was_training = model.training
    model.eval()
    with torch.no_grad():
        for inputs, targets in test_loader:
            inputs = inputs.to(device, non_blocking=True).float().contiguous()
            targets = targets.to(device, non_blocking=True).float().contiguous()
            model_tracker.update(_batch_metrics(model(inputs), targets, inputs, config))
            # persistence baseline: the next frame is predicted to equal the current one
            persistence_tracker.update(_batch_metrics(inputs[:, 0:3].contiguous(), targets, inputs, config))
    model.train(was_training)
    return {"model": model_tracker.compute_averages(), "persistence": persistence_tracker.compute_averages()}


def _scene_arrays(data: Dict[str, np.ndarray], norm_scales: Optional[Dict[str, float]]):
    def norm(a, key):
        return a / norm_scales[key] if norm_scales else a
    return (norm(data["density"], "S_density").astype(np.float32), norm(data["velx"], "S_velx").astype(np.float32),
            norm(data["velz"], "S_velz").astype(np.float32), data["emitter"].astype(np.float32), data["collider"].astype(np.float32))


def rollout_step_metrics(model, scenes: Sequence[Dict[str, np.ndarray]], starts: Sequence[int], config, *, steps: int = ROLLOUT_STEPS,
                         norm_scales: Optional[Dict[str, float]] = None, max_batch: int = 64) -> List[List[Dict[str, float]]]:
    """`_run_single_rollout` (test_evaluation.py:192-292) for many (scene, start) pairs at once: result[i][k] = the reference's
    step_metrics[k] of rollout i.  All rollouts of a chunk advance together in one batched plan; masks of frame t_start + k and the
    ground truth of frame t_start + k + 1 are staged per step."""
    assert len(scenes) == len(starts)
    pl = config.physics_loss
    out: List[List[Dict[str, float]]] = [[] for _ in scenes]
    H, W = scenes[0]["density"].shape[-2:]
    dev = model.flat_params.device
    for c0 in range(0, len(scenes), max_batch):
        idx = list(range(c0, min(len(scenes), c0 + max_batch)))
        B = len(idx)
        arrs = [_scene_arrays(scenes[i], norm_scales) for i in idx]
        plan = RolloutInferPlan(model.cfg, B, H, W, params_flat=model.flat_params, dtype=model.precision, conv_impl=model.conv_impl,
                                norm_state=model.flat_norm_state)

        def frames(field: int, offset: int) -> torch.Tensor:
            return torch.from_numpy(np.stack([a[field][starts[i] + offset] for a, i in zip(arrs, idx)])).to(dev)

        state = torch.stack([frames(0, 0), frames(1, 0), frames(2, 0)], 1)
        plan.reset(frames(3, 0), frames(4, 0), state=state, prev_density=frames(0, -1))
        for k in range(steps):
            if k:
                plan.emitter.tensor.copy_(frames(3, k))
                plan.collider.tensor.copy_(frames(4, k))
            pred = plan.step(1)
            gt = torch.stack([frames(0, k + 1), frames(1, k + 1), frames(2, k + 1)], 1)
            for j, i in enumerate(idx):
                p1, g1 = pred[j:j + 1], gt[j:j + 1]
                m = validation_metrics(p1, g1, None, None, dx=pl.grid_spacing, padding_mode=config.padding_mode, mode=pl.stencil_mode)
                mg = validation_metrics(g1, g1, None, None, dx=pl.grid_spacing, padding_mode=config.padding_mode, mode=pl.stencil_mode, ssim=False)
                out[i].append({"mse_density": m["mse_density"], "mse_velx": m["mse_velx"], "mse_vely": m["mse_vely"],
                               "ssim_density": m.get("ssim_density", 0.0), "divergence_norm": m["divergence_norm"],
                               "divergence_norm_gt": mg["divergence_norm"], "gradient_l1": m["gradient_l1"]})
    return out


def run_rollout_evaluation(model, config, device: str = "cuda", npz_dir: Optional[Path] = None, stats_path: Optional[Path] = None,
                           steps: int = ROLLOUT_STEPS) -> Dict[str, List[Dict[str, float]]]:
    """{"avg_per_step": [per-step averages over every (test sequence, starting point) rollout]}; {} without a test split."""
    test_dir = Path(npz_dir if npz_dir is not None else config.npz_dir) / "test"
    if not test_dir.exists():
        print(f"Test split directory not found: {test_dir}, skipping rollout evaluation.")
        return {}
    paths = sorted(p for p in test_dir.iterdir() if p.name.startswith("seq_") and p.name.endswith(".npz"))
    norm_scales = load_normalization_scales(Path(stats_path)) if (config.normalize and stats_path is not None) else None
    if config.normalize and stats_path is None:
        raise ValueError("config.normalize is set: pass stats_path (the reference reads it from its project config)")
    scenes, starts = [], []
    for p in paths:
        with np.load(p) as npz:
            d = npz["density"].astype(np.float32)
            data = {"density": d, "velx": npz["velx"].astype(np.float32), "velz": npz["velz"].astype(np.float32),
                    "emitter": (npz["emitter"] if "emitter" in npz.files else np.zeros_like(d)).astype(np.float32),
                    "collider": (npz["collider"] if "collider" in npz.files else np.zeros_like(d)).astype(np.float32)}
        for t in _get_starting_frames(d.shape[0], steps):
            scenes.append(data)
            starts.append(t)
    was_training = model.training
    model.eval()
    per = rollout_step_metrics(model, scenes, starts, config, steps=steps, norm_scales=norm_scales) if scenes else []
    model.train(was_training)
    avg_per_step = []
    for k in range(steps):
        if not per:
            avg_per_step.append({key: 0.0 for key in ROLLOUT_KEYS})
        else:
            avg_per_step.append({key: sum(r[k][key] for r in per) / len(per) for key in ROLLOUT_KEYS})
    return {"avg_per_step": avg_per_step}


__all__ = ["run_test_evaluation", "run_rollout_evaluation", "rollout_step_metrics", "ROLLOUT_STEPS", "STARTING_POINTS"]
