"""Physics loss with the reference's API (ml/training/physics_loss.py) on the HIP kernels of libfluidnet_hip.

Same free functions and the same PhysicsAwareLoss(...).forward(outputs, targets, inputs) -> (loss tensor, dict of
floats) contract.  All arithmetic happens in the LOSS_REDUCE / LOSS_FINALIZE / LOSS_GRAD / STENCIL kernels; CUDA-less
tensors raise (no CPU fallback).  The dict costs ONE device->host copy instead of the reference's five .item() syncs.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Literal, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..engine import runtime as rt

StencilMode = Literal["central", "forward"]


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fluidnet_amd physics loss runs on MI355X only (got a CPU tensor); there is no CPU fallback")


def _planes3(t: torch.Tensor) -> Tuple[torch.Tensor, bool]:
    sq = t.ndim == 3
    if t.ndim == 4:
        if t.shape[1] != 1:
            raise ValueError("expected (B,H,W) or (B,1,H,W)")
        t = t[:, 0]
    return t.contiguous().float(), sq


def compute_spatial_gradients(field: torch.Tensor, dx: float = 1.0, dy: float = 1.0, padding_mode: str = "zeros",
                              mode: StencilMode = "central") -> Tuple[torch.Tensor, torch.Tensor]:
    """ml/training/physics_loss.py:19-50 (forward pass only; the training path differentiates inside LOSS_GRAD)."""
    _need_gpu(field)
    shape = field.shape
    f, _ = _planes3(field)
    B, H, W = f.shape
    gx, gy = torch.empty_like(f), torch.empty_like(f)
    rc = rt.lib().fnet_spatial_gradients(f.data_ptr(), gx.data_ptr(), gy.data_ptr(), B, H, W, float(dx), float(dy),
                                         rt.PAD[padding_mode], rt.STENCIL[mode], ctypes.c_void_p(rt.current_stream_ptr()))
    rt.check(rc, "fnet_spatial_gradients")
    return gx.view(shape), gy.view(shape)


def compute_divergence(velx: torch.Tensor, vely: torch.Tensor, dx: float = 1.0, dy: float = 1.0, padding_mode: str = "zeros",
                       mode: StencilMode = "central") -> torch.Tensor:
    """ml/training/physics_loss.py:53-91."""
    _need_gpu(velx, vely)
    shape = velx.shape
    a, _ = _planes3(velx)
    b, _ = _planes3(vely)
    B, H, W = a.shape
    out = torch.empty_like(a)
    rc = rt.lib().fnet_divergence(a.data_ptr(), b.data_ptr(), out.data_ptr(), B, H, W, float(dx), float(dy), rt.PAD[padding_mode],
                                  rt.STENCIL[mode], ctypes.c_void_p(rt.current_stream_ptr()))
    rt.check(rc, "fnet_divergence")
    return out.view(shape)


class _LossCfg:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _bstride_ok(t: torch.Tensor, planes: int) -> bool:
    """t[n] must be `planes` contiguous H*W planes; the batch stride is free."""
    H, W = t.shape[-2:]
    if t.stride(-1) != 1 or t.stride(-2) != W:
        return False
    if t.ndim == 4 and t.shape[1] > 1 and t.stride(1) != H * W:
        return False
    return True


def _plane(t: torch.Tensor, planes: int) -> Tuple[torch.Tensor, int]:
    if not _bstride_ok(t, planes) or t.dtype != torch.float32:
        t = t.contiguous().float()
    return t, (t.stride(0) if t.shape[0] > 1 else planes * t.shape[-1] * t.shape[-2])


def _loss_ops(cfg: _LossCfg, pred, tgt, in0, em, co, sums, terms, dpred=None, gscale=None, wk: float = 1.0) -> np.ndarray:
    B, _, H, W = pred.shape
    flags = (1 if cfg.div else 0) | (2 if cfg.grad else 0) | (4 if cfg.emit else 0)
    tgt, ts = _plane(tgt, 3)
    in0, s0 = _plane(in0, 1)
    em, se = _plane(em, 1)
    co, sc = _plane(co, 1)
    keep = (tgt, in0, em, co)
    i10 = [B, H, W, ts, s0, se, sc, rt.PAD[cfg.padding_mode], rt.STENCIL[cfg.stencil_mode], flags]
    n = 3 if dpred is None else 1
    ops = np.zeros(n, dtype=rt.OP_DTYPE)
    if dpred is None:
        ops[0]["op"] = rt.OPC["ZERO"]
        ops[0]["i"][:2] = [64, 0]
        ops[0]["p"][0] = sums.data_ptr()
        o = ops[1]
        o["op"] = rt.OPC["LOSS_REDUCE"]
        o["i"][:10] = i10
        o["f"][0] = cfg.grid_spacing
        o["p"][:6] = [pred.data_ptr(), tgt.data_ptr(), in0.data_ptr(), em.data_ptr(), co.data_ptr(), sums.data_ptr()]
        o = ops[2]
        o["op"] = rt.OPC["LOSS_FINALIZE"]
        o["i"][:5] = [B, H, W, flags, 1]
        o["f"][:5] = [cfg.mse_weight, cfg.divergence_weight, cfg.gradient_weight, cfg.emitter_weight, wk]
        o["p"][:4] = [sums.data_ptr(), 0, terms.data_ptr(), 0]
    else:
        o = ops[0]
        o["op"] = rt.OPC["LOSS_GRAD"]
        o["i"][:12] = i10 + [1, 0]
        o["f"][:6] = [cfg.grid_spacing, cfg.mse_weight, cfg.divergence_weight, cfg.gradient_weight, cfg.emitter_weight, wk]
        o["p"][:10] = [pred.data_ptr(), tgt.data_ptr(), in0.data_ptr(), em.data_ptr(), co.data_ptr(), sums.data_ptr(), 0, 0,
                       gscale.data_ptr(), dpred.data_ptr()]
    return ops, keep


class _PhysicsLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outputs, targets, inputs, cfg):
        pred = outputs.contiguous().float()
        sums = torch.empty(8, dtype=torch.float64, device=pred.device)
        terms = torch.empty(8, dtype=torch.float32, device=pred.device)
        ops, keep = _loss_ops(cfg, pred, targets, inputs[:, 0], inputs[:, 4], inputs[:, 5], sums, terms)
        rt.run_ops(ops)
        ctx.cfg = cfg
        ctx.save_for_backward(pred, targets, inputs, sums)
        ctx.mark_non_differentiable(terms)
        return terms[4].clone(), terms

    @staticmethod
    def backward(ctx, gtotal, _gterms):
        pred, targets, inputs, sums = ctx.saved_tensors
        dpred = torch.empty_like(pred)
        gscale = gtotal.reshape(1).contiguous().float()
        ops, keep = _loss_ops(ctx.cfg, pred, targets, inputs[:, 0], inputs[:, 4], inputs[:, 5], sums, None, dpred=dpred, gscale=gscale)
        rt.run_ops(ops)
        return dpred, None, None, None


def _check_loss_shapes(outputs, targets, inputs):
    if outputs.ndim != 4 or outputs.shape[1] != 3 or targets.shape != outputs.shape:
        raise ValueError(f"outputs/targets must be [B,3,H,W] (got {tuple(outputs.shape)}, {tuple(targets.shape)})")
    if inputs.ndim != 4 or inputs.shape[1] < 6 or inputs.shape[0] != outputs.shape[0] or inputs.shape[2:] != outputs.shape[2:]:
        raise ValueError(f"inputs must be [B,6,H,W] (got {tuple(inputs.shape)})")


class PhysicsAwareLoss(nn.Module):
    """ml/training/physics_loss.py:180-278."""

    def __init__(self, mse_weight: float = 1.0, divergence_weight: float = 0.1, gradient_weight: float = 0.1,
                 emitter_weight: float = 0.1, grid_spacing: float = 1.0, enable_divergence: bool = True,
                 enable_gradient: bool = True, enable_emitter: bool = True, padding_mode: str = "zeros",
                 stencil_mode: StencilMode = "forward") -> None:
        super().__init__()
        self.mse_weight = mse_weight
        self.divergence_weight = divergence_weight
        self.gradient_weight = gradient_weight
        self.emitter_weight = emitter_weight
        self.grid_spacing = grid_spacing
        self.padding_mode = padding_mode
        self.stencil_mode: StencilMode = stencil_mode
        self.enable_divergence = enable_divergence
        self.enable_gradient = enable_gradient
        self.enable_emitter = enable_emitter

    def _cfg(self) -> _LossCfg:
        return _LossCfg(mse_weight=self.mse_weight, divergence_weight=self.divergence_weight, gradient_weight=self.gradient_weight,
                        emitter_weight=self.emitter_weight, grid_spacing=self.grid_spacing, padding_mode=self.padding_mode,
                        stencil_mode=self.stencil_mode,
                        div=self.enable_divergence and self.divergence_weight > 0,
                        grad=self.enable_gradient and self.gradient_weight > 0,
                        emit=self.enable_emitter and self.emitter_weight > 0)

    def forward_device(self, outputs, targets, inputs) -> Tuple[torch.Tensor, torch.Tensor]:
        """(total, terms[8]) both on the device, no host sync: terms = mse, divergence, gradient, emitter, total, fluid count."""
        _need_gpu(outputs, targets, inputs)
        _check_loss_shapes(outputs, targets, inputs)
        return _PhysicsLossFn.apply(outputs, targets, inputs, self._cfg())

    def forward(self, outputs: torch.Tensor, targets: torch.Tensor, inputs: torch.Tensor) -> Tuple[torch.Tensor, Dict[str, float]]:
        total, terms = self.forward_device(outputs, targets, inputs)
        t = terms.detach().cpu().numpy()          # the one host sync (the reference has five .item() calls)
        cfg = self._cfg()
        d: Dict[str, float] = {"mse": float(t[0])}
        if cfg.div:
            d["divergence"] = float(t[1])
        if cfg.grad:
            d["gradient"] = float(t[2])
        if cfg.emit:
            d["emitter"] = float(t[3])
        d["total"] = float(t[4])
        return total, d


def _standalone(kind: str, a, b, emitter_mask, collider_mask, dx, padding_mode, mode) -> torch.Tensor:
    """divergence_loss / gradient_loss as single-term PhysicsAwareLoss evaluations (same kernels).  The value returned is the
    module's `total` with that one term at weight 1, so it is differentiable with respect to `a` (velx / density_pred) and, for
    the divergence, `b` (vely) through LOSS_GRAD; the reference's gradient with respect to density_target is not provided."""
    _need_gpu(a, b, emitter_mask, collider_mask)
    a3, _ = _planes3(a)
    b3, _ = _planes3(b)
    z = torch.zeros_like(a3)
    em = z if emitter_mask is None else _planes3(emitter_mask)[0]
    co = z if collider_mask is None else _planes3(collider_mask)[0]
    if kind == "div":
        outputs = torch.stack([z, a3, b3], 1)
        targets = outputs.detach()
        crit = PhysicsAwareLoss(0.0, 1.0, 0.0, 0.0, dx, True, False, False, padding_mode, mode)
    else:
        outputs = torch.stack([a3, z, z], 1)
        targets = torch.stack([b3.detach(), z, z], 1)
        crit = PhysicsAwareLoss(0.0, 0.0, 1.0, 0.0, dx, False, True, False, padding_mode, mode)
    inputs = torch.stack([z, z, z, z, em, co], 1)
    total, _ = crit.forward_device(outputs, targets, inputs)
    return total


def divergence_loss(velx, vely, emitter_mask, collider_mask=None, dx: float = 1.0, dy: float = 1.0, eps: float = 1e-8,
                    padding_mode: str = "zeros", stencil_mode: StencilMode = "central") -> torch.Tensor:
    """ml/training/physics_loss.py:94-126.  dx != dy: the stencil is linear, so d(vx)/dx + d(vy)/dy is evaluated as the unit-spacing
    divergence of (vx/dx, vy/dy).  eps is the kernels' compile-time 1e-8 (the only value the reference ever passes): anything
    else is refused, not approximated."""
    if eps != 1e-8:
        raise NotImplementedError("fluidnet_amd: divergence_loss evaluates the masked mean with eps = 1e-8 (the reference's only value)")
    if dx != dy:
        return _standalone("div", velx / dx, vely / dy, emitter_mask, collider_mask, 1.0, padding_mode, stencil_mode)
    return _standalone("div", velx, vely, emitter_mask, collider_mask, dx, padding_mode, stencil_mode)


def gradient_loss(density_pred, density_target, emitter_mask=None, collider_mask=None, dx: float = 1.0, dy: float = 1.0,
                  eps: float = 1e-8, padding_mode: str = "zeros", mode: StencilMode = "central") -> torch.Tensor:
    """ml/training/physics_loss.py:129-166.  With emitter_mask=None the reference takes plain L1 means and IGNORES collider_mask
    (:162-164); that branch is the masked form with an all-fluid mask (up to the 1e-8 in the denominator), so the collider is
    dropped here too.  dx != dy or eps != 1e-8 are refused (the x and y sums share one kernel constant; the reference never
    passes either)."""
    if dx != dy or eps != 1e-8:
        raise NotImplementedError("fluidnet_amd: gradient_loss supports dx == dy and eps == 1e-8 (the reference's only values)")
    if emitter_mask is None:
        collider_mask = None
    return _standalone("grad", density_pred, density_target, emitter_mask, collider_mask, dx, padding_mode, mode)


def emitter_spawn_loss(density_pred, density_current, emitter_mask, threshold: float = 0.01) -> torch.Tensor:
    """ml/training/physics_loss.py:169-177; differentiable with respect to density_pred.  threshold is the kernels' compile-time
    0.01 (the reference's default and only value): anything else is refused."""
    if threshold != 0.01:
        raise NotImplementedError("fluidnet_amd: emitter_spawn_loss supports threshold == 0.01 (the reference's only value)")
    _need_gpu(density_pred, density_current, emitter_mask)
    d, _ = _planes3(density_pred)
    c, _ = _planes3(density_current)
    e, _ = _planes3(emitter_mask)
    z = torch.zeros_like(d)
    outputs = torch.stack([d, z, z], 1)
    inputs = torch.stack([c.detach(), z, z, z, e, z], 1)
    crit = PhysicsAwareLoss(0.0, 0.0, 0.0, 1.0, 1.0, False, False, True)
    total, _ = crit.forward_device(outputs, outputs.detach(), inputs)
    return total
