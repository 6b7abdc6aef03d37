"""EarlyStopping with the reference's behaviour and best_model.pth layout (ml/training/early_stopping.py:8-72):
checkpoint = {epoch, model_state_dict, optimizer_state_dict, config, val_loss}."""
from __future__ import annotations

from pathlib import Path
from typing import Any

import torch


class EarlyStopping:
    def __init__(self, patience: int, min_delta: float, checkpoint_dir: Path) -> None:
        self.patience, self.min_delta = patience, min_delta
        self.checkpoint_dir = Path(checkpoint_dir)
        self.counter = 0
        self.best_loss = float("inf")
        self.best_epoch = -1

    def __call__(self, val_loss: float, epoch: int) -> bool:
        if val_loss < self.best_loss - self.min_delta:
            self.best_loss, self.best_epoch, self.counter = val_loss, epoch, 0
            return False
        self.counter += 1
        return self.counter >= self.patience

    def save_checkpoint(self, model, optimizer: Any, scaler, config: Any, epoch: int) -> None:
        ck = {"epoch": epoch, "model_state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
              "optimizer_state_dict": optimizer.state_dict(), "config": config.model_dump(mode="json"), "val_loss": self.best_loss}
        self.checkpoint_dir.mkdir(parents=True, exist_ok=True)
        torch.save(ck, self.checkpoint_dir / "best_model.pth")

    def load_best_checkpoint(self, model, optimizer: Any, scaler, device: str) -> None:
        path = self.checkpoint_dir / "best_model.pth"
        if not path.exists():
            return
        ck = torch.load(path, map_location=device, weights_only=True)     # this file is written by save_checkpoint above
        model.load_state_dict(ck["model_state_dict"])
        optimizer.load_state_dict(ck["optimizer_state_dict"])

    def state_dict(self) -> dict:
        return {"counter": self.counter, "best_loss": self.best_loss, "best_epoch": self.best_epoch}

    def load_state_dict(self, state: dict) -> None:
        self.counter, self.best_loss, self.best_epoch = state["counter"], state["best_loss"], state["best_epoch"]
