"""InferenceBackend / PyTorchBackend with the reference's interface (ml/inference/backend.py:15-100):
load_model(path, device); infer(np.float32[B,6,H,W]) -> np.float32[B,3,H,W]; .device.

Deviation (documented in SURVEY section 8 row a20): the reference rebuilds the UNet WITHOUT `padding_mode`, silently
zero-padding checkpoints trained with replicate padding (backend.py:58-74); here every architecture key of the saved
config, padding_mode included, is honoured.  Checkpoints are read with weights_only=True (nothing in the file executes).
The ONNX backend (backend.py:103-147) is out of scope (onnxruntime is absent offline).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from pathlib import Path
from typing import Optional, Union

import numpy as np
import torch

from ..models.unet import UNet, UNetConfig


class InferenceBackend(ABC):
    @abstractmethod
    def load_model(self, model_path: Union[str, Path], device: str) -> None: ...

    @abstractmethod
    def infer(self, input_data: np.ndarray) -> np.ndarray: ...

    @property
    @abstractmethod
    def device(self) -> str: ...


class HipBackend(InferenceBackend):
    def __init__(self, precision: str = "bf16") -> None:
        self._device = "cuda"
        self._model: Optional[UNet] = None
        self._precision = precision

    def load_model(self, model_path: Union[str, Path], device: str = "cuda") -> None:
        if not str(device).startswith("cuda"):
            raise RuntimeError("fluidnet_amd inference runs on MI355X only; there is no CPU fallback")
        self._device = device
        ck = torch.load(model_path, map_location="cpu", weights_only=True)
        if "config" not in ck:
            raise KeyError(f"Checkpoint missing 'config' key at {model_path}")
        c = ck["config"]
        missing = [k for k in ("in_channels", "out_channels", "base_channels", "depth") if k not in c]
        if missing:
            raise KeyError(f"Checkpoint config missing: {missing}")
        cfg = UNetConfig(in_channels=c["in_channels"], out_channels=c["out_channels"], base_channels=c["base_channels"], depth=c["depth"],
                         norm=c.get("norm", "group"), act=c.get("act", "silu"), group_norm_groups=c.get("group_norm_groups", 8),
                         dropout=c.get("dropout", 0.0), upsample=c.get("upsample", "nearest"), downsample=c.get("downsample", "stride"),
                         padding_mode=c.get("padding_mode", "zeros"), use_residual=c.get("use_residual", True),
                         bottleneck_blocks=c.get("bottleneck_blocks", 1), output_activation=c.get("output_activation", "linear_clamp"))
        self._model = UNet(cfg, precision=self._precision)
        self._model.load_state_dict(ck["model_state_dict"])
        self._model.eval()
        self._model.to(device)

    def infer(self, input_data: np.ndarray) -> np.ndarray:
        if self._model is None:
            raise RuntimeError("Model not loaded. Call load_model first.")
        x = torch.from_numpy(np.ascontiguousarray(input_data, dtype=np.float32)).to(self._device)
        with torch.no_grad():
            out = self._model(x)
        return out.cpu().numpy()

    @property
    def device(self) -> str:
        return self._device


PyTorchBackend = HipBackend      # drop-in name of the reference's default backend
