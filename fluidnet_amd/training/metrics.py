"""Validation metrics with the reference's API (ml/training/metrics.py:7-118) on the METRICS kernel of libfluidnet_hip.

`validation_metrics(...)` computes every metric that Trainer.validate logs (ml/training/trainer.py:256-302) in ONE pass over
(outputs, target, masks) and one device->host copy; the reference's eight functions are kept with their names, arguments
and return types (Python floats) and are thin selections of that pass.  CPU tensors raise: there is no CPU fallback.

SSIM: the reference calls torchmetrics.functional.image.structural_similarity_index_measure(data_range=1.0) with library
defaults; torchmetrics is not available offline, so the kernel follows the library's published algorithm (11x11 Gaussian
window, sigma 1.5, k1 0.01, k2 0.03, reflect-pad then crop 5 pixels per side) and its parity is UNPINNED.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from ..engine import runtime as rt
from .physics_loss import StencilMode, _need_gpu, _plane

_SLOTS = 16


def _sums(outputs: torch.Tensor, target: torch.Tensor, emitter: Optional[torch.Tensor], collider: Optional[torch.Tensor], dx: float,
          padding_mode: str, mode: str, ssim: bool) -> np.ndarray:
    _need_gpu(outputs, target, emitter, collider)
    if outputs.ndim != 4 or outputs.shape[1] != 3 or target.shape != outputs.shape:
        raise ValueError(f"expected outputs/target of shape (B,3,H,W), got {tuple(outputs.shape)} / {tuple(target.shape)}")
    B, _, H, W = outputs.shape
    pred = outputs.detach().contiguous().float()
    tgt, ts = _plane(target.detach(), 3)
    zeros = None
    if emitter is None or collider is None:
        zeros = torch.zeros(B, H, W, device=pred.device)
    em, se = _plane((emitter if emitter is not None else zeros).detach().reshape(B, H, W), 1)
    co, sc = _plane((collider if collider is not None else zeros).detach().reshape(B, H, W), 1)
    sums = torch.zeros(_SLOTS, dtype=torch.float64, device=pred.device)
    ops = np.zeros(1, dtype=rt.OP_DTYPE)
    o = ops[0]
    o["op"] = rt.OPC["METRICS"]
    o["i"][:9] = [B, H, W, ts, se, sc, rt.PAD[padding_mode], rt.STENCIL[mode], 1 if ssim else 0]
    o["f"][0] = dx
    o["p"][:5] = [pred.data_ptr(), tgt.data_ptr(), em.data_ptr(), co.data_ptr(), sums.data_ptr()]
    rt.run_ops(ops)
    return sums.cpu().numpy()          # the one host sync


def validation_metrics(outputs: torch.Tensor, target: torch.Tensor, emitter_mask: torch.Tensor, collider_mask: torch.Tensor, *,
                       dx: float = 1.0, padding_mode: str = "zeros", mode: StencilMode = "central",
                       expected_injection: float = 0.8, eps: float = 1e-8, ssim: bool = True) -> Dict[str, float]:
    """All of Trainer.validate's batch metrics (same keys as ml/training/trainer.py:270-300) from one kernel pass."""
    B, _, H, W = outputs.shape
    ssim = ssim and H > 10 and W > 10
    m = _sums(outputs, target, emitter_mask, collider_mask, dx, padding_mode, mode, ssim)
    n = float(B * H * W)
    out = {"mse_density": m[0] / n, "mse_velx": m[1] / n, "mse_vely": m[2] / n,
           "divergence_norm": float(np.sqrt(m[5] / n)),
           "kinetic_energy": 0.5 * m[6],
           "collider_violation": 0.0 if m[7] < eps else float(np.float32(m[8]) / (np.float32(m[7]) + np.float32(eps))),
           "emitter_density_accuracy": 0.0 if m[9] < eps else abs(float(np.float32(m[10]) / (np.float32(m[9]) + np.float32(eps))) - expected_injection),
           "gradient_l1": m[3] / n + m[4] / n}
    if ssim:
        out["ssim_density"] = m[11] / float(B * (H - 10) * (W - 10))
    return {k: float(v) for k, v in out.items()}


def _as_b3(density: torch.Tensor, velx: Optional[torch.Tensor] = None, vely: Optional[torch.Tensor] = None) -> torch.Tensor:
    d = density.reshape(density.shape[0], *density.shape[-2:])
    z = torch.zeros_like(d)
    return torch.stack([d, z if velx is None else velx.reshape(d.shape), z if vely is None else vely.reshape(d.shape)], 1)


def compute_per_channel_mse(pred: torch.Tensor, target: torch.Tensor) -> Dict[str, float]:
    """metrics.py:7-12"""
    m = validation_metrics(pred, target, None, None, ssim=False)
    return {k: m[k] for k in ("mse_density", "mse_velx", "mse_vely")}


def compute_ssim_density(pred: torch.Tensor, target: torch.Tensor) -> float:
    """metrics.py:15-21 (parity unpinned, see the module docstring)"""
    if pred.shape[-1] <= 10 or pred.shape[-2] <= 10:
        raise ValueError("SSIM needs images larger than the 11x11 window")
    return validation_metrics(pred[:, :3], target[:, :3], None, None, ssim=True)["ssim_density"]


def compute_gradient_l1(density_pred: torch.Tensor, density_target: torch.Tensor, dx: float = 1.0, dy: float = 1.0,
                        padding_mode: str = "zeros", mode: StencilMode = "central") -> float:
    """metrics.py:24-38"""
    if dx != dy:
        raise NotImplementedError("the METRICS kernel takes one grid spacing (the reference always passes dx == dy)")
    return validation_metrics(_as_b3(density_pred), _as_b3(density_target), None, None, dx=dx, padding_mode=padding_mode, mode=mode,
                              ssim=False)["gradient_l1"]


def compute_divergence_norm(velx: torch.Tensor, vely: torch.Tensor, dx: float = 1.0, dy: float = 1.0, padding_mode: str = "zeros",
                            mode: StencilMode = "central") -> float:
    """metrics.py:41-51"""
    if dx != dy:
        raise NotImplementedError("the METRICS kernel takes one grid spacing (the reference always passes dx == dy)")
    p = _as_b3(torch.zeros_like(velx), velx, vely)
    return validation_metrics(p, p, None, None, dx=dx, padding_mode=padding_mode, mode=mode, ssim=False)["divergence_norm"]


def compute_kinetic_energy(velx: torch.Tensor, vely: torch.Tensor) -> float:
    """metrics.py:54-61"""
    p = _as_b3(torch.zeros_like(velx), velx, vely)
    return validation_metrics(p, p, None, None, ssim=False)["kinetic_energy"]


def compute_collider_violation(density: torch.Tensor, collider_mask: torch.Tensor, eps: float = 1e-8) -> float:
    """metrics.py:64-78"""
    p = _as_b3(density)
    return validation_metrics(p, p, None, collider_mask, eps=eps, ssim=False)["collider_violation"]


def compute_emitter_density_accuracy(density: torch.Tensor, emitter_mask: torch.Tensor, expected_injection: float = 0.8,
                                     eps: float = 1e-8) -> float:
    """metrics.py:81-97"""
    p = _as_b3(density)
    return validation_metrics(p, p, emitter_mask, None, expected_injection=expected_injection, eps=eps, ssim=False)["emitter_density_accuracy"]


class MetricsTracker:
    """metrics.py:100-118"""

    def __init__(self) -> None:
        self.metrics: Dict[str, float] = {}
        self.counts: Dict[str, int] = {}

    def reset(self) -> None:
        self.metrics = {}
        self.counts = {}

    def update(self, metrics_dict: Dict[str, float]) -> None:
        for key, value in metrics_dict.items():
            if key not in self.metrics:
                self.metrics[key] = 0.0
                self.counts[key] = 0
            self.metrics[key] += value
            self.counts[key] += 1

    def compute_averages(self) -> Dict[str, float]:
        return {k: self.metrics[k] / self.counts[k] for k in self.metrics}
