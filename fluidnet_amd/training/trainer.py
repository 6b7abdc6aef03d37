"""Trainer with the reference's constructor and method surface (ml/training/trainer.py:34-42,142,210,311,606,717,765) over
the MI355X engine.

  Trainer(model, train_loader, val_loader, config: TrainingConfig, device)
  .train()  .train_epoch() -> dict  .validate() -> dict  ._compute_rollout_loss(x_0, y_seq, masks) -> (loss, dict, pred)
  .save_checkpoint(epoch, final=False)  .load_checkpoint(path) -> epoch

Two execution engines for the optimisation step of train_epoch (trainer.py:142-208):
  engine="plan" (default): the whole step -- K forwards, physics losses, BPTT backward, clip, Adam -- is ONE native plan
      (fluidnet_amd.engine.train_plan.RolloutTrainPlan) replayed per batch with no host sync; batches whose shape has no
      plan yet build one (at most `max_plans` shapes, e.g. the ragged last batch), others fall back to the autograd engine.
  engine="autograd": the literal loop of the reference (_compute_rollout_loss + loss.backward()) through the
      torch.autograd wrappers of fluidnet_amd.UNet / PhysicsAwareLoss, followed by FusedClipAdam.
Both give the loss dict keys of the reference (mse, divergence, gradient, emitter, total).
Not carried over: MLflow logging, matplotlib plots, GradScaler (bf16 needs none), the torchmetrics-based validation
metrics -- validate() returns the averaged loss terms only.
With torch.distributed initialised (one process per GPU) the plan engine runs the data-parallel step of
fluidnet_amd/training/dp.py.
"""
from __future__ import annotations

import time
from pathlib import Path
from typing import Any, Dict, Optional, Tuple

import torch

from ..engine.train_plan import RolloutTrainPlan
from .dp import dp_train_step, torch_all_reduce
from .early_stopping import EarlyStopping
from .optim import FusedClipAdam
from .physics_loss import PhysicsAwareLoss
from .metrics import MetricsTracker, validation_metrics


class _Scheduler:
    """plateau / cosine / step learning-rate schedules with torch's semantics, acting on FusedClipAdam.param_groups."""

    def __init__(self, opt: FusedClipAdam, cfg) -> None:
        import math
        self.opt, self.cfg, self.kind = opt, cfg, cfg.lr_scheduler_type
        if self.kind not in ("plateau", "cosine", "step"):
            raise ValueError(f"Unknown scheduler type: {self.kind}")
        self.base_lr = opt.param_groups[0]["lr"]
        self.epoch = 0
        self.best, self.bad = float("inf"), 0
        self._math = math

    def step(self, metric: Optional[float] = None) -> None:
        c, g = self.cfg, self.opt.param_groups[0]
        self.epoch += 1
        if self.kind == "plateau":          # ReduceLROnPlateau(mode="min", threshold=1e-4 rel)
            if metric is not None and metric < self.best * (1 - 1e-4):
                self.best, self.bad = metric, 0
            else:
                self.bad += 1
            if self.bad > c.lr_scheduler_patience:
                g["lr"] = max(g["lr"] * c.lr_scheduler_factor, c.lr_scheduler_min_lr)
                self.bad = 0
        elif self.kind == "cosine":         # CosineAnnealingLR(T_max, eta_min), closed form
            g["lr"] = c.lr_scheduler_min_lr + (self.base_lr - c.lr_scheduler_min_lr) * \
                (1 + self._math.cos(self._math.pi * self.epoch / c.lr_scheduler_t_max)) / 2
        else:                               # StepLR(step_size, gamma)
            g["lr"] = self.base_lr * c.lr_scheduler_factor ** (self.epoch // c.lr_scheduler_step_size)

    def state_dict(self) -> dict:
        return {"epoch": self.epoch, "best": self.best, "bad": self.bad, "base_lr": self.base_lr}

    def load_state_dict(self, s: dict) -> None:
        self.epoch, self.best, self.bad, self.base_lr = s["epoch"], s["best"], s["bad"], s["base_lr"]


def step_seed(step: int, rank: int = 0) -> int:
    """Dropout2d seed of optimisation step `step` on data-parallel rank `rank`: ranks must not share channel masks for their
    local sample n (the hash mixes seed, layer site, n and c: fnet_common.h drop_scale)."""
    x = (step * 7919 + 13) ^ ((rank * 0x9E3779B1 + 0x7F4A7C15) & 0x7FFFFFFF if rank else 0)
    return x & 0x7FFFFFFF


class Trainer:
    def __init__(self, model, train_loader, val_loader, config, device: str, *, engine: str = "plan", max_plans: int = 2) -> None:
        if engine not in ("plan", "autograd"):
            raise ValueError("engine must be 'plan' or 'autograd'")
        self.model, self.train_loader, self.val_loader = model, train_loader, val_loader
        self.config = config.model_copy()
        self.device = device
        self.engine, self.max_plans = engine, max_plans
        self.gradient_clip_norm = config.gradient_clip_norm
        self.gradient_clip_enabled = config.gradient_clip_enabled
        pl = config.physics_loss
        self.criterion = PhysicsAwareLoss(
            mse_weight=pl.mse_weight, divergence_weight=pl.divergence_weight, gradient_weight=pl.gradient_weight,
            emitter_weight=pl.emitter_weight, grid_spacing=pl.grid_spacing, enable_divergence=pl.enable_divergence,
            enable_gradient=pl.enable_gradient, enable_emitter=pl.enable_emitter, padding_mode=config.padding_mode,
            stencil_mode=pl.stencil_mode)
        self.optimizer = FusedClipAdam(model, config.learning_rate, config.gradient_clip_norm if config.gradient_clip_enabled else None)
        self.scaler = None                                   # bf16 storage / fp32 accumulate needs no loss scaling
        self.scheduler = _Scheduler(self.optimizer, config) if config.use_lr_scheduler else None
        if getattr(config, "variant", None):                 # curriculum runs: <checkpoints>/<relative dir>/<arch>_<variant> (trainer.py:71-74)
            ckdir = Path(config.checkpoint_dir_variant)
        else:
            ckdir = Path(config.checkpoint_dir) / str(getattr(model, "model_name", "Unet"))
        self.early_stopping = EarlyStopping(config.early_stop_patience, config.early_stop_min_delta, ckdir) if config.use_early_stopping else None
        self.config.checkpoint_dir = ckdir
        ckdir.mkdir(parents=True, exist_ok=True)
        self.history: Dict[str, list] = {"train_total": [], "train_mse": [], "val_total": [], "val_mse": [], "learning_rate": []}
        self._plans: Dict[Tuple, RolloutTrainPlan] = {}
        self._dist = None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                self._dist = dist
        except Exception:
            self._dist = None
        self._rank = self._dist.get_rank() if self._dist is not None else 0
        self._shard_checked = False
        self._world = self._dist.get_world_size() if self._dist is not None else 1
        if self._dist is not None and engine != "plan":
            # the autograd engine has no gradient exchange: ranks would silently train different models
            raise NotImplementedError("data-parallel training (torch.distributed initialised with world_size > 1) needs engine='plan'")

    # ------------------------------------------------------------------ the K-step rollout loss, literal loop
    def _compute_rollout_loss(self, x_0: torch.Tensor, y_seq: torch.Tensor, masks: torch.Tensor):
        """ml/training/trainer.py:311-385."""
        K = y_seq.shape[1]
        cfg = self.config
        state_current, state_prev = x_0[:, :3], x_0[:, 3:4]
        if cfg.rollout_final_step_only:
            for k in range(K):
                model_input = torch.cat([state_current, state_prev, masks[:, k, 0:1], masks[:, k, 1:2]], dim=1)
                pred = self.model(model_input)
                if k == K - 1:
                    loss, d = self.criterion(pred, y_seq[:, k], model_input)
                    return loss, d, pred
                if cfg.rollout_gradient_truncation:
                    state_prev, state_current = pred[:, 0:1].detach(), pred.detach()
                else:
                    state_prev, state_current = pred[:, 0:1], pred
            raise RuntimeError("Should have returned on final step")
        total = torch.zeros((), device=x_0.device)
        acc: Dict[str, float] = {}
        wsum = 0.0
        pred = None
        for k in range(K):
            model_input = torch.cat([state_current, state_prev, masks[:, k, 0:1], masks[:, k, 1:2]], dim=1)
            pred = self.model(model_input)
            loss_k, d_k = self.criterion(pred, y_seq[:, k], model_input)
            w = cfg.rollout_weight_decay ** k
            total = total + w * loss_k
            wsum += w
            for key, val in d_k.items():
                acc[key] = acc.get(key, 0.0) + w * val
            if cfg.rollout_gradient_truncation:
                state_prev, state_current = pred[:, 0:1].detach(), pred.detach()
            else:
                state_prev, state_current = pred[:, 0:1], pred
        return total / wsum, {k: v / wsum for k, v in acc.items()}, pred

    # ------------------------------------------------------------------ native step
    def _plan_for(self, B: int, H: int, W: int, K: int, mask_frames: int) -> Optional[RolloutTrainPlan]:
        key = (B, H, W, K, mask_frames)
        p = self._plans.get(key)
        if p is None and len(self._plans) < self.max_plans:
            cfg = self.config
            world = self._dist.get_world_size() if self._dist is not None else 1
            p = RolloutTrainPlan(self.model.cfg, self.criterion, B, H, W, K, dtype=self.model.precision, conv_impl=self.model.conv_impl,
                                 weight_decay=cfg.rollout_weight_decay, gradient_truncation=cfg.rollout_gradient_truncation,
                                 final_step_only=cfg.rollout_final_step_only, mask_frames=mask_frames, world=world,
                                 clip_norm=self.gradient_clip_norm if self.gradient_clip_enabled else None,
                                 betas=self.optimizer.betas, eps=self.optimizer.eps, seed=self.model.seed)
            # one Adam state for every shape: the plans share the optimizer's moment buffers
            _, _, m, v, _ = self.optimizer._state()
            p.bind(self.model.flat_params.device, self.model.flat_params, self.model.flat_grads, adam_m=m, adam_v=v,
                   norm_state=self.model.flat_norm_state)
            self._plans[key] = p
        return p

    def _native_step(self, inputs: torch.Tensor, targets: torch.Tensor, masks: Optional[torch.Tensor]) -> Optional[Dict[str, float]]:
        B, _, H, W = inputs.shape
        if masks is None:                       # single-step batch (x, y): K = 1, masks are channels 4,5 of x (trainer.py:187-188)
            targets = targets[:, None]
            masks = inputs[:, None, 4:6]
        K, Km = targets.shape[1], masks.shape[1]
        plan = self._plan_for(B, H, W, K, Km)
        if plan is None:
            return None
        plan.x0.tensor.copy_(inputs)
        plan.y.tensor.copy_(targets)
        plan.masks.tensor.copy_(masks)
        return self._run_plan(plan)

    def _run_plan(self, plan: RolloutTrainPlan) -> Dict[str, float]:
        """One optimisation step of a plan whose x0 / y / masks buffers are already filled."""
        self.optimizer.step_count += 1
        plan.set_adam_scalars(self.optimizer.param_groups[0]["lr"], self.optimizer.step_count)
        plan.set_dropout_seed(step_seed(self.optimizer.step_count, self._rank))
        world = self._world
        dp_train_step(plan, world, torch_all_reduce(self._dist) if world > 1 else None)
        self.model.count_batches(plan.K)        # BatchNorm2d bookkeeping: the plan applied the model K times in training mode
        self.model.flat_params.add_(0)          # bump the autograd version: autograd-path engines refresh their bf16 weights
        return plan.loss_dict()

    # ------------------------------------------------------------------ epoch loops
    def _to_device(self, batch):
        return tuple(t.to(self.device, non_blocking=True) for t in batch)

    def _shard(self, batch):
        """Data parallel with a plain (unsharded) loader: every rank receives the same global batch and keeps rows
        [rank*b, (rank+1)*b), b = batch/world -- the convention of HbmBatchLoader and SURVEY.md section 8e.  A loader that already
        shards (DistributedSampler, HbmBatchLoader(rank=, world=)) declares it with `loader.sharded = True`."""
        if self._dist is None or getattr(self.train_loader, "sharded", False) or \
                type(getattr(self.train_loader, "sampler", None)).__name__ == "DistributedSampler":
            return batch
        n = batch[0].shape[0]
        if n % self._world:
            raise RuntimeError(f"data-parallel training: a global batch of {n} samples cannot be split over {self._world} ranks "
                               "(use drop_last=True and a batch size divisible by the world size)")
        if not self._shard_checked:
            # nothing makes two ranks' plain DataLoaders draw the same permutation (shuffle=True uses each process's own RNG): check
            # once that every rank holds the same global batch before slicing it, instead of training on duplicated / missing rows
            self._shard_checked = True
            mine = [float(t.double().sum()) for t in batch] + [float(n)]
            alls = [None] * self._world
            self._dist.all_gather_object(alls, mine)
            if any(a != alls[0] for a in alls):
                raise RuntimeError("data-parallel training with a plain DataLoader: the ranks drew different global batches (seed the "
                                   "loader's generator identically on every rank, or use DistributedSampler / HbmBatchLoader(rank=, world=))")
        b = n // self._world
        return tuple(t[self._rank * b:(self._rank + 1) * b] for t in batch)

    def _train_epoch_hbm(self, loader) -> Dict[str, float]:
        """HbmBatchLoader + plan engine: every batch is gathered by one kernel straight into the plan's input buffers (no host
        batch, no copy), then the native K-step plan runs."""
        st = loader.store
        plan = self._plan_for(loader.B, st.H, st.W, loader.K, loader.Km)
        if plan is None:
            raise RuntimeError("no plan slot left for the HBM loader's batch shape (raise max_plans)")
        acc: Dict[str, float] = {}
        n = 0
        for b in range(loader.upload_epoch()):               # one index upload per epoch, one gather launch per batch
            loader.gather_uploaded(b, plan.x0.tensor, plan.y.tensor, plan.masks.tensor)
            d = self._run_plan(plan)
            n += 1
            for k, v in d.items():
                acc[k] = acc.get(k, 0.0) + v
        return {k: v / max(n, 1) for k, v in acc.items()}

    def train_epoch(self) -> Dict[str, float]:
        self.model.train()
        if self.engine == "plan" and hasattr(self.train_loader, "gather") and hasattr(self.train_loader, "store"):
            return self._train_epoch_hbm(self.train_loader)
        acc: Dict[str, float] = {}
        n = 0
        for batch in self.train_loader:
            batch = self._to_device(self._shard(batch))
            rollout = len(batch) == 3
            inputs, targets = batch[0], batch[1]
            masks = batch[2] if rollout else None
            d = self._native_step(inputs, targets, masks) if self.engine == "plan" else None
            if d is None and self._dist is not None:
                # the autograd fallback has no gradient all-reduce: refusing beats ranks that silently diverge
                raise RuntimeError(f"no plan slot left for batch shape {tuple(inputs.shape)} under data parallelism (raise max_plans, "
                                   "or use drop_last=True so that every batch has one shape)")
            if d is None:
                self.optimizer.zero_grad()
                if rollout:
                    loss, d, _ = self._compute_rollout_loss(inputs, targets, masks)
                else:
                    outputs = self.model(inputs)
                    loss, d = self.criterion(outputs, targets, inputs)
                loss.backward()
                self.optimizer.step()
            n += 1
            for k, v in d.items():
                acc[k] = acc.get(k, 0.0) + v
        return {k: v / max(n, 1) for k, v in acc.items()}

    def validate(self) -> Dict[str, float]:
        """ml/training/trainer.py:206-310: averaged loss terms + the validation metrics of the last prediction (one METRICS
        kernel pass and one host copy per batch instead of the reference's ten .item() syncs)."""
        self.model.eval()
        acc: Dict[str, float] = {}
        tracker = MetricsTracker()
        n = 0
        pl = self.config.physics_loss
        with torch.no_grad():
            for batch in self.val_loader:
                batch = self._to_device(batch)
                if len(batch) == 3:
                    _, d, outputs = self._compute_rollout_loss(*batch)
                    target, emitter, collider = batch[1][:, -1], batch[2][:, -1, 0], batch[2][:, -1, 1]
                else:
                    outputs = self.model(batch[0])
                    _, d = self.criterion(outputs, batch[1], batch[0])
                    target, emitter, collider = batch[1], batch[0][:, 4], batch[0][:, 5]
                n += 1
                for k, v in d.items():
                    acc[k] = acc.get(k, 0.0) + v
                tracker.update(validation_metrics(outputs, target, emitter, collider, dx=pl.grid_spacing, padding_mode=self.config.padding_mode,
                                                  mode=pl.stencil_mode, expected_injection=0.8))
        return {**{k: v / max(n, 1) for k, v in acc.items()}, **tracker.compute_averages()}

    def train(self) -> None:
        for epoch in range(self.config.epochs):
            t0 = time.time()
            ds = getattr(self.train_loader, "dataset", None)
            if hasattr(self.train_loader, "set_epoch"):          # HbmBatchLoader: reshuffles and moves the dataset's stride offset
                self.train_loader.set_epoch(epoch)
            elif hasattr(ds, "set_epoch"):
                ds.set_epoch(epoch)
            tr = self.train_epoch()
            va = self.validate()
            lr = self.optimizer.param_groups[0]["lr"]
            self.history["train_total"].append(tr["total"])
            self.history["train_mse"].append(tr.get("mse", 0.0))
            self.history["val_total"].append(va["total"])
            self.history["val_mse"].append(va.get("mse", 0.0))
            self.history["learning_rate"].append(lr)
            print(f"Epoch {epoch + 1}/{self.config.epochs} | Train Loss: {tr['total']:.6f} | Val Loss: {va['total']:.6f} | "
                  f"Time: {time.time() - t0:.2f}s")
            if self.scheduler is not None:
                self.scheduler.step(va["total"])
            if self.early_stopping is not None:
                # data parallel: every rank validates the same weights on the same split, so the decisions agree; rank 0 alone writes
                if self.early_stopping(va["total"], epoch):
                    if self._rank == 0:
                        self.early_stopping.save_checkpoint(self.model, self.optimizer, None, self.config, epoch)
                    break
                if self.early_stopping.counter == 0 and self._rank == 0:
                    self.early_stopping.save_checkpoint(self.model, self.optimizer, None, self.config, epoch)
            if (epoch + 1) % self.config.save_every_n_epochs == 0:
                self.save_checkpoint(epoch)
        if self._dist is not None:
            self._dist.barrier()                 # rank 0's best_model.pth is complete before any rank reads it back
        if self.early_stopping is not None:
            self.early_stopping.load_best_checkpoint(self.model, self.optimizer, None, self.device)
        else:
            self.save_checkpoint(self.config.epochs - 1, final=True)

    # ------------------------------------------------------------------ checkpoints (trainer.py:717-784)
    def save_checkpoint(self, epoch: int, final: bool = False) -> None:
        if self._rank != 0:                      # replicas hold identical state: one writer per checkpoint directory
            return
        ck: Dict[str, Any] = {"epoch": epoch, "model_state_dict": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                              "optimizer_state_dict": self.optimizer.state_dict(), "config": self.config.model_dump(mode="json")}
        if self.scheduler is not None:
            ck["scheduler_state_dict"] = self.scheduler.state_dict()
        if self.early_stopping is not None:
            ck["early_stopping_state"] = self.early_stopping.state_dict()
        path = self.config.checkpoint_dir / ("final_model.pth" if final else f"checkpoint_epoch_{epoch + 1}.pth")
        torch.save(ck, path)
        if not final:
            cks = sorted(self.config.checkpoint_dir.glob("checkpoint_epoch_*.pth"), key=lambda x: x.stat().st_mtime)
            while len(cks) > self.config.keep_last_n_checkpoints:
                cks.pop(0).unlink()

    def load_checkpoint(self, checkpoint_path) -> int:
        ck = torch.load(Path(checkpoint_path), map_location=self.device, weights_only=True)
        self.model.load_state_dict(ck["model_state_dict"])
        self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        if self.scheduler is not None and "scheduler_state_dict" in ck:
            self.scheduler.load_state_dict(ck["scheduler_state_dict"])
        if self.early_stopping is not None and "early_stopping_state" in ck:
            self.early_stopping.load_state_dict(ck["early_stopping_state"])
        return int(ck.get("epoch", 0))
