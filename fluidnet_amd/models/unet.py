"""UNet with the reference's API (ml/models/unet.py:183-303) whose compute runs in libfluidnet_hip.

Same UNetConfig fields, same sub-module tree and therefore the same 88 state-dict keys/shapes for UNet_medium
(stem, downs[i].{block.{c1,n1,c2,n2}, res.b.{...}, down}, mid[j].b.{...}, ups[i].{block, res.b}, head), so reference
checkpoints (`model_state_dict`, OIHW fp32) load unchanged.  The torch sub-modules only own the parameters; forward
never calls them: it runs the HIP engine (fluidnet_amd.engine) and there is NO CPU fallback -- a CPU tensor raises.

Parameters are views into one flat fp32 buffer (and .grad views into one flat gradient buffer), which is what the
engine, the fused optimiser and the RCCL gradient all-reduce operate on.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Literal, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..engine.plan import check_supported, norm_state_layout, param_layout
from ..engine.unet_engine import UNetEngine

NormType = Literal["none", "batch", "instance", "group"]
ActType = Literal["relu", "leaky_relu", "gelu", "silu"]
UpsampleType = Literal["nearest", "bilinear", "transpose"]
DownsampleType = Literal["stride", "avgpool", "maxpool"]
PaddingType = Literal["zeros", "reflect", "replicate", "circular"]
OutputActivationType = Literal["sigmoid_tanh", "linear_clamp"]


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 7
    out_channels: int = 3

    base_channels: int = 32
    depth: int = 2

    norm: NormType = "group"
    act: ActType = "silu"
    group_norm_groups: int = 8
    dropout: float = 0.0

    upsample: UpsampleType = "nearest"
    downsample: DownsampleType = "stride"
    padding_mode: PaddingType = "zeros"
    use_residual: bool = False
    bottleneck_blocks: int = 1
    output_activation: OutputActivationType = "linear_clamp"


def _norm(norm: str, ch: int, groups: int) -> nn.Module:
    if norm == "none":
        return nn.Identity()
    if norm == "batch":
        return nn.BatchNorm2d(ch)
    if norm == "instance":
        return nn.InstanceNorm2d(ch, affine=True)
    g = min(groups, ch)
    while g > 1 and ch % g:
        g -= 1
    return nn.GroupNorm(g, ch)


class _ParamHolder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("sub-modules of fluidnet_amd.UNet only hold parameters; call the UNet itself")


class ConvBlock(_ParamHolder):
    def __init__(self, in_ch, out_ch, cfg: UNetConfig):
        super().__init__()
        bias = cfg.norm == "none"
        self.c1 = nn.Conv2d(in_ch, out_ch, 3, padding=1, padding_mode=cfg.padding_mode, bias=bias)
        self.n1 = _norm(cfg.norm, out_ch, cfg.group_norm_groups)
        self.c2 = nn.Conv2d(out_ch, out_ch, 3, padding=1, padding_mode=cfg.padding_mode, bias=bias)
        self.n2 = _norm(cfg.norm, out_ch, cfg.group_norm_groups)


class ResBlock(_ParamHolder):
    def __init__(self, ch, cfg):
        super().__init__()
        self.b = ConvBlock(ch, ch, cfg)


class Down(_ParamHolder):
    def __init__(self, in_ch, out_ch, cfg):
        super().__init__()
        self.block = ConvBlock(in_ch, out_ch, cfg)
        self.res = ResBlock(out_ch, cfg) if cfg.use_residual else nn.Identity()
        conv = nn.Conv2d(out_ch, out_ch, 3, stride=2 if cfg.downsample == "stride" else 1, padding=1, padding_mode=cfg.padding_mode)
        self.down = conv if cfg.downsample == "stride" else nn.Sequential(nn.Identity(), conv)


class Up(_ParamHolder):
    def __init__(self, in_ch, skip_ch, out_ch, cfg):
        super().__init__()
        if cfg.upsample == "transpose":
            self.up = nn.ConvTranspose2d(in_ch, in_ch, kernel_size=2, stride=2)
        else:
            self.up = nn.Identity()
        self.block = ConvBlock(in_ch + skip_ch, out_ch, cfg)
        self.res = ResBlock(out_ch, cfg) if cfg.use_residual else nn.Identity()


class _SlotGuard:
    """Returns the engine slot (its tape) to the pool when the autograd graph that holds it is freed."""

    def __init__(self, slot):
        self.slot = slot

    def release(self):
        if self.slot is not None:
            self.slot.busy = False
            self.slot = None

    def __del__(self):
        self.release()


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, model):
        eng = model._engine(x, training=model.training)
        out, slot = eng.forward(x, keep=True)
        ctx.eng, ctx.guard, ctx.model = eng, _SlotGuard(slot), model
        return out

    @staticmethod
    def backward(ctx, dout):
        slot = ctx.guard.slot
        if slot is None:
            raise RuntimeError("fluidnet_amd.UNet: this forward's tape was already consumed (retain_graph is not supported)")
        ctx.model._attach_grads()
        dx = ctx.eng.backward(slot, dout.contiguous())
        ctx.guard.slot = None
        return dx, None, None


class UNet(nn.Module):
    """U-Net, MI355X engine.  `precision`: "bf16" (MFMA path, bf16 activations / fp32 accumulate) or "fp32" (exact path)."""

    model_name: str = "Unet"

    def __init__(self, cfg: Optional[UNetConfig] = None, *, precision: str = "bf16", conv_impl: str = "auto", seed: int = 0) -> None:
        super().__init__()
        self.cfg = cfg or UNetConfig()
        if self.cfg.depth < 1:
            raise ValueError("depth must be >= 1")
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.precision, self.conv_impl, self.seed = precision, conv_impl, seed
        c = self.cfg
        self.stem = nn.Conv2d(c.in_channels, c.base_channels, 3, padding=1, padding_mode=c.padding_mode)
        downs, skip_chs = [], []
        ch = c.base_channels
        for _ in range(c.depth):
            downs.append(Down(ch, ch * 2, c))
            skip_chs.append(ch * 2)
            ch *= 2
        self.downs = nn.ModuleList(downs)
        mids = [ResBlock(ch, c) for _ in range(max(0, c.bottleneck_blocks))]
        self.mid = nn.Sequential(*mids) if mids else nn.Identity()
        ups = []
        for i in reversed(range(c.depth)):
            ups.append(Up(ch, skip_chs[i], skip_chs[i] // 2, c))
            ch = skip_chs[i] // 2
        self.ups = nn.ModuleList(ups)
        self.head = nn.Conv2d(ch, c.out_channels, 1)
        self._engines: Dict[Tuple, UNetEngine] = {}
        self._flat: Optional[torch.Tensor] = None
        self._gflat: Optional[torch.Tensor] = None
        self._anchor: Optional[torch.Tensor] = None
        self._offsets: Optional[Dict[str, int]] = None
        # BatchNorm2d running statistics, flat like the parameters (the engine updates them in place); the modules' buffers are views
        self._nflat: Optional[torch.Tensor] = None
        self._noffsets, self._nstate_len = norm_state_layout(self.cfg)

    # ------------------------------------------------------------------ flat parameter storage
    def _layout(self):
        shapes, offsets, total = param_layout(self.cfg)
        names = [n for n, _ in self.named_parameters()]
        if names != list(shapes.keys()):
            raise NotImplementedError("fluidnet_amd HIP engine: unsupported parameter set for this UNetConfig "
                                      f"(norm={self.cfg.norm}, upsample={self.cfg.upsample}, downsample={self.cfg.downsample})")
        return shapes, offsets, total

    def _flatten(self) -> None:
        """(Re)pack all parameters into one flat buffer on their current device; parameters become views of it."""
        shapes, offsets, total = self._layout()
        dev = self.stem.weight.device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for n, p in self.named_parameters():
                o, cnt = offsets[n], p.numel()
                flat[o:o + cnt].copy_(p.detach().reshape(-1).float())
                p.data = flat[o:o + cnt].view(p.shape)
                p.grad = None
        self._flat, self._gflat, self._offsets = flat, gflat, offsets
        nflat = torch.zeros(max(self._nstate_len, 8), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for name, o in self._noffsets.items():
                mod_name, leaf = name.rsplit(".", 1)
                mod = self.get_submodule(mod_name)
                buf = getattr(mod, leaf)
                nflat[o:o + buf.numel()].copy_(buf.detach().float())
                setattr(mod, leaf, nflat[o:o + buf.numel()])
        self._nflat = nflat
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        self._engines.clear()

    def _is_flat(self) -> bool:
        if self._flat is None or self._flat.device != self.stem.weight.device:
            return False
        base = self._flat.data_ptr()
        for n, p in self.named_parameters():
            if p.data_ptr() != base + 4 * self._offsets[n] or p.dtype != torch.float32:
                return False
        for name, o in self._noffsets.items():
            mod_name, leaf = name.rsplit(".", 1)
            if getattr(self.get_submodule(mod_name), leaf).data_ptr() != self._nflat.data_ptr() + 4 * o:
                return False
        return True

    def _attach_grads(self) -> None:
        """Make every p.grad a view of the flat gradient buffer (zeroing segments whose grad was None)."""
        for n, p in self.named_parameters():
            o, cnt = self._offsets[n], p.numel()
            view = self._gflat[o:o + cnt].view(p.shape)
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view

    @property
    def flat_params(self) -> torch.Tensor:
        if not self._is_flat():
            self._flatten()
        return self._flat

    @property
    def flat_grads(self) -> torch.Tensor:
        if not self._is_flat():
            self._flatten()
        return self._gflat

    def _engine(self, x: torch.Tensor, training: bool) -> UNetEngine:
        if not self._is_flat():
            self._flatten()
        B, C, H, W = x.shape
        key = (B, H, W, bool(training))
        eng = self._engines.get(key)
        if eng is None:
            eng = UNetEngine(self.cfg, B, H, W, params_flat=self._flat, grads_flat=self._gflat, dtype=self.precision,
                             conv_impl=self.conv_impl, training=training, seed=self.seed, norm_state=self._nflat)
            self._engines[key] = eng
        return eng

    @property
    def flat_norm_state(self) -> torch.Tensor:
        """BatchNorm2d running_mean / running_var of every norm layer in one buffer (plan.norm_state_layout); 8 zeros otherwise."""
        if not self._is_flat():
            self._flatten()
        return self._nflat

    def count_batches(self, n: int = 1) -> None:
        """BatchNorm2d.num_batches_tracked += n (bookkeeping only: momentum is the fixed 0.1 of unet.py:21)."""
        if self.cfg.norm == "batch":
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.num_batches_tracked += n

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("fluidnet_amd.UNet runs on MI355X only (input is on %s); there is no CPU fallback" % x.device)
        if x.ndim != 4 or x.shape[1] != self.cfg.in_channels:
            raise ValueError(f"expected input [B,{self.cfg.in_channels},H,W], got {tuple(x.shape)}")
        check_supported(self.cfg, x.shape[2], x.shape[3])
        if self.stem.weight.device != x.device:
            raise RuntimeError(f"model is on {self.stem.weight.device}, input on {x.device}")
        if not self._is_flat():
            self._flatten()
        x = x.contiguous().float()
        if self.training:
            self.count_batches(1)
        if not torch.is_grad_enabled():
            out, _ = self._engine(x, training=self.training).forward(x, keep=False)
            return out
        return _UNetFunction.apply(x, self._anchor, self)


__all__ = ["UNet", "UNetConfig", "DownsampleType", "OutputActivationType", "PaddingType"]
