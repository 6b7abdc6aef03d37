"""CPU oracle for the FluidNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a from-scratch CPU restatement (plain PyTorch fp32, functional style)
of the arithmetic the reference performs on its hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline.  Nothing under
``fluidnet_amd/`` imports it: the product path is HIP-only and fails loudly when
``libfluidnet_hip.so`` is missing.

Parity pinning: the reference holds no golden vectors for UNet / rollout / Adam
(SURVEY.md section 8c); this restatement is pinned by fixtures under
``tests/golden/`` that ``tests/golden/make_golden.py`` generated in the build
container by importing the real reference modules (``models.unet``,
``training.physics_loss``) from /root/reference/ml, plus the 16 property tests
of ``ml/tests/test_physics_loss.py`` restated in ``tests/test_oracle_*``.

Each function cites the reference file:line it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# configuration (field names = ml/models/unet.py:183-201)
# --------------------------------------------------------------------------
@dataclass(frozen=True)
class OracleUNetConfig:
    in_channels: int = 7
    out_channels: int = 3
    base_channels: int = 32
    depth: int = 2
    norm: str = "group"
    act: str = "silu"
    group_norm_groups: int = 8
    dropout: float = 0.0
    upsample: str = "nearest"
    downsample: str = "stride"
    padding_mode: str = "zeros"
    use_residual: bool = False
    bottleneck_blocks: int = 1
    output_activation: str = "linear_clamp"


def medium_config(**over) -> OracleUNetConfig:
    """ml/config/model_architectures/UNet_medium.yaml:1-14 + training_config.py:57-70."""
    kw = dict(in_channels=6, out_channels=3, base_channels=32, depth=3, norm="instance",
              act="gelu", group_norm_groups=8, dropout=0.1, upsample="bilinear",
              downsample="stride", padding_mode="replicate", use_residual=True,
              bottleneck_blocks=1, output_activation="linear_clamp")
    kw.update(over)
    return OracleUNetConfig(**kw)


def _group_count(groups: int, ch: int) -> int:
    # ml/models/unet.py:24-28
    g = min(groups, ch)
    while g > 1 and ch % g != 0:
        g -= 1
    return g


# --------------------------------------------------------------------------
# parameter inventory (state-dict keys/shapes of ml/models/unet.py:204-279)
# --------------------------------------------------------------------------
def _convblock_shapes(prefix: str, cin: int, cout: int, cfg: OracleUNetConfig, out: "OrderedDict[str, tuple]"):
    has_bias = cfg.norm == "none"                      # unet.py:55,59
    for idx, (ci, co) in enumerate(((cin, cout), (cout, cout)), start=1):
        out[f"{prefix}.c{idx}.weight"] = (co, ci, 3, 3)
        if has_bias:
            out[f"{prefix}.c{idx}.bias"] = (co,)
        if cfg.norm in ("batch", "instance", "group"):
            out[f"{prefix}.n{idx}.weight"] = (co,)
            out[f"{prefix}.n{idx}.bias"] = (co,)


def param_shapes(cfg: OracleUNetConfig) -> "OrderedDict[str, tuple]":
    """Trainable tensors in state-dict order (buffers of BatchNorm excluded)."""
    if cfg.depth < 1:
        raise ValueError("depth must be >= 1")           # unet.py:215-216
    out: "OrderedDict[str, tuple]" = OrderedDict()
    out["stem.weight"] = (cfg.base_channels, cfg.in_channels, 3, 3)
    out["stem.bias"] = (cfg.base_channels,)
    ch = cfg.base_channels
    skip = []
    for L in range(cfg.depth):
        oc = ch * 2
        _convblock_shapes(f"downs.{L}.block", ch, oc, cfg, out)
        if cfg.use_residual:
            _convblock_shapes(f"downs.{L}.res.b", oc, oc, cfg, out)
        dkey = f"downs.{L}.down" if cfg.downsample == "stride" else f"downs.{L}.down.1"
        out[f"{dkey}.weight"] = (oc, oc, 3, 3)
        out[f"{dkey}.bias"] = (oc,)
        skip.append(oc)
        ch = oc
    for j in range(max(0, cfg.bottleneck_blocks)):
        _convblock_shapes(f"mid.{j}.b", ch, ch, cfg, out)
    for i, L in enumerate(reversed(range(cfg.depth))):
        sc = skip[L]
        oc = sc // 2
        if cfg.upsample == "transpose":
            out[f"ups.{i}.up.weight"] = (ch, ch, 2, 2)
            out[f"ups.{i}.up.bias"] = (ch,)
        _convblock_shapes(f"ups.{i}.block", ch + sc, oc, cfg, out)
        if cfg.use_residual:
            _convblock_shapes(f"ups.{i}.res.b", oc, oc, cfg, out)
        ch = oc
    out["head.weight"] = (cfg.out_channels, ch, 1, 1)
    out["head.bias"] = (cfg.out_channels,)
    return out


def make_params(cfg: OracleUNetConfig, seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, Tensor]":
    """Seeded numpy-formula weights (reproducible on both sides without torch RNG).

    conv weights ~ N(0, 2/fan_in) * 0.8, norm gamma = 1 + 0.1 N, betas/biases = 0.1 N.
    """
    rs = np.random.RandomState(seed)
    out: "OrderedDict[str, Tensor]" = OrderedDict()
    for name, shp in param_shapes(cfg).items():
        if len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            a = rs.standard_normal(shp) * (0.8 * math.sqrt(2.0 / fan_in))
        elif name.endswith(".weight"):
            a = 1.0 + 0.1 * rs.standard_normal(shp)
        else:
            a = 0.1 * rs.standard_normal(shp)
        out[name] = torch.from_numpy(a.astype(np.float32)).to(dtype)
    return out


# --------------------------------------------------------------------------
# UNet forward (ml/models/unet.py)
# --------------------------------------------------------------------------
_PAD = {"zeros": "constant", "reflect": "reflect", "replicate": "replicate", "circular": "circular"}


def _conv3x3(x: Tensor, w: Tensor, b: Optional[Tensor], padding_mode: str, stride: int = 1) -> Tensor:
    # nn.Conv2d(padding=1, padding_mode=...) == explicit pad + valid conv  (unet.py:55,59,92,218-220)
    if padding_mode == "zeros":
        return F.conv2d(x, w, b, stride=stride, padding=1)
    xp = F.pad(x, (1, 1, 1, 1), mode=_PAD[padding_mode])
    return F.conv2d(xp, w, b, stride=stride, padding=0)


def _apply_norm(x: Tensor, p: Dict[str, Tensor], key: str, cfg: OracleUNetConfig,
                bn_state: Optional[Dict[str, Tensor]], training: bool) -> Tensor:
    # ml/models/unet.py:17-28
    if cfg.norm == "none":
        return x
    g, b = p[f"{key}.weight"], p[f"{key}.bias"]
    if cfg.norm == "instance":
        # InstanceNorm2d(affine=True): per-(n,c) biased variance, eps 1e-5, no running stats
        return F.instance_norm(x, None, None, g, b, use_input_stats=True, eps=1e-5)
    if cfg.norm == "group":
        return F.group_norm(x, _group_count(cfg.group_norm_groups, x.shape[1]), g, b, eps=1e-5)
    if cfg.norm == "batch":
        rm = rv = None
        if bn_state is not None:
            rm = bn_state.get(f"{key}.running_mean")
            rv = bn_state.get(f"{key}.running_var")
        use_batch = training or rm is None
        return F.batch_norm(x, rm, rv, g, b, training=use_batch, momentum=0.1, eps=1e-5)
    raise ValueError(cfg.norm)


def _apply_act(x: Tensor, act: str) -> Tensor:
    # ml/models/unet.py:31-39
    if act == "relu":
        return F.relu(x)
    if act == "leaky_relu":
        return F.leaky_relu(x, 0.1)
    if act == "gelu":
        return F.gelu(x)            # exact erf form
    if act == "silu":
        return F.silu(x)
    raise ValueError(act)


def _conv_block(x: Tensor, p: Dict[str, Tensor], key: str, cfg: OracleUNetConfig, bn_state, training: bool,
                drop_masks: Optional[Dict[str, Tensor]]) -> Tensor:
    # ml/models/unet.py:65-69
    x = _conv3x3(x, p[f"{key}.c1.weight"], p.get(f"{key}.c1.bias"), cfg.padding_mode)
    x = _apply_act(_apply_norm(x, p, f"{key}.n1", cfg, bn_state, training), cfg.act)
    if cfg.dropout > 0.0 and training:
        if drop_masks is not None:
            # injected channel mask [B,C] of {0,1}: Dropout2d zeroes whole channels and rescales by 1/(1-p)
            m = drop_masks[key].to(x.dtype)
            x = x * (m / (1.0 - cfg.dropout))[:, :, None, None]
        else:
            x = F.dropout2d(x, cfg.dropout, training=True)
    x = _conv3x3(x, p[f"{key}.c2.weight"], p.get(f"{key}.c2.bias"), cfg.padding_mode)
    x = _apply_act(_apply_norm(x, p, f"{key}.n2", cfg, bn_state, training), cfg.act)
    return x


def dropout_keys(cfg: OracleUNetConfig) -> List[Tuple[str, int]]:
    """(ConvBlock key, channel count) for every Dropout2d site, in forward order."""
    keys: List[Tuple[str, int]] = []
    ch = cfg.base_channels
    skip = []
    for L in range(cfg.depth):
        oc = ch * 2
        keys.append((f"downs.{L}.block", oc))
        if cfg.use_residual:
            keys.append((f"downs.{L}.res.b", oc))
        skip.append(oc)
        ch = oc
    for j in range(max(0, cfg.bottleneck_blocks)):
        keys.append((f"mid.{j}.b", ch))
    for i, L in enumerate(reversed(range(cfg.depth))):
        oc = skip[L] // 2
        keys.append((f"ups.{i}.block", oc))
        if cfg.use_residual:
            keys.append((f"ups.{i}.res.b", oc))
        ch = oc
    return keys


def unet_forward(p: Dict[str, Tensor], cfg: OracleUNetConfig, x: Tensor, *, training: bool = False,
                 bn_state: Optional[Dict[str, Tensor]] = None,
                 drop_masks: Optional[Dict[str, Tensor]] = None,
                 return_preact: bool = False):
    """ml/models/unet.py:281-303."""
    pm = cfg.padding_mode
    x = _conv3x3(x, p["stem.weight"], p["stem.bias"], pm)                       # :282
    skips: List[Tensor] = []
    for L in range(cfg.depth):                                                  # :285-287 / Down.forward :122-127
        x = _conv_block(x, p, f"downs.{L}.block", cfg, bn_state, training, drop_masks)
        if cfg.use_residual:
            x = x + _conv_block(x, p, f"downs.{L}.res.b", cfg, bn_state, training, drop_masks)   # :86-87
        skips.append(x)
        if cfg.downsample == "stride":                                          # :90-94
            x = _conv3x3(x, p[f"downs.{L}.down.weight"], p[f"downs.{L}.down.bias"], pm, stride=2)
        else:
            x = F.avg_pool2d(x, 2) if cfg.downsample == "avgpool" else F.max_pool2d(x, 2)
            x = _conv3x3(x, p[f"downs.{L}.down.1.weight"], p[f"downs.{L}.down.1.bias"], pm)
    for j in range(max(0, cfg.bottleneck_blocks)):                              # :289
        x = x + _conv_block(x, p, f"mid.{j}.b", cfg, bn_state, training, drop_masks)
    for i, L in enumerate(reversed(range(cfg.depth))):                          # :291-292 / Up.forward :167-180
        skip = skips[L]
        if cfg.upsample == "transpose":
            x = F.conv_transpose2d(x, p[f"ups.{i}.up.weight"], p[f"ups.{i}.up.bias"], stride=2)
        elif cfg.upsample == "bilinear":
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)
        else:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        if x.shape[-2:] != skip.shape[-2:]:                                     # :171-175
            dh = skip.shape[-2] - x.shape[-2]
            dw = skip.shape[-1] - x.shape[-1]
            skip = skip[..., dh // 2: dh // 2 + x.shape[-2], dw // 2: dw // 2 + x.shape[-1]]
        x = torch.cat([x, skip], dim=1)                                          # :177 (x first)
        x = _conv_block(x, p, f"ups.{i}.block", cfg, bn_state, training, drop_masks)
        if cfg.use_residual:
            x = x + _conv_block(x, p, f"ups.{i}.res.b", cfg, bn_state, training, drop_masks)
    z = F.conv2d(x, p["head.weight"], p["head.bias"])                            # :294
    if cfg.output_activation == "sigmoid_tanh":                                  # :296-301
        out = torch.cat([torch.sigmoid(z[:, 0:1]), torch.tanh(z[:, 1:3])], dim=1)
    else:
        out = torch.cat([torch.clamp(z[:, 0:1], 0.0, 1.0), torch.clamp(z[:, 1:3], -1.0, 1.0)], dim=1)
    if return_preact:
        return out, z
    return out


# --------------------------------------------------------------------------
# physics loss (ml/training/physics_loss.py)
# --------------------------------------------------------------------------
def compute_spatial_gradients(fld: Tensor, dx: float = 1.0, dy: float = 1.0, padding_mode: str = "zeros",
                              mode: str = "central") -> Tuple[Tensor, Tensor]:
    """physics_loss.py:19-50 (accepts (B,H,W) or (B,C,H,W))."""
    squeeze = fld.ndim == 3
    if squeeze:
        fld = fld.unsqueeze(1)
    pmode = _PAD[padding_mode]
    kw = {"value": 0.0} if pmode == "constant" else {}
    if mode == "forward":
        q = F.pad(fld, (0, 1, 0, 1), mode=pmode, **kw)
        base = q[:, :, :-1, :-1]
        gx = (q[:, :, :-1, 1:] - base) / dx
        gy = (q[:, :, 1:, :-1] - base) / dy
    else:
        q = F.pad(fld, (1, 1, 1, 1), mode=pmode, **kw)
        gx = (q[:, :, 1:-1, 2:] - q[:, :, 1:-1, :-2]) / (2.0 * dx)
        gy = (q[:, :, 2:, 1:-1] - q[:, :, :-2, 1:-1]) / (2.0 * dy)
    if squeeze:
        gx, gy = gx.squeeze(1), gy.squeeze(1)
    return gx, gy


def compute_divergence(velx: Tensor, vely: Tensor, dx: float = 1.0, dy: float = 1.0, padding_mode: str = "zeros",
                       mode: str = "central") -> Tensor:
    """physics_loss.py:53-91."""
    if mode == "forward":
        squeeze = velx.ndim == 3
        if squeeze:
            velx, vely = velx.unsqueeze(1), vely.unsqueeze(1)
        pmode = _PAD[padding_mode]
        kw = {"value": 0.0} if pmode == "constant" else {}
        qx = F.pad(velx, (0, 1, 0, 0), mode=pmode, **kw)
        qy = F.pad(vely, (0, 0, 0, 1), mode=pmode, **kw)
        div = (qx[:, :, :, 1:] - qx[:, :, :, :-1]) / dx + (qy[:, :, 1:, :] - qy[:, :, :-1, :]) / dy
        return div.squeeze(1) if squeeze else div
    # NB: the reference does not forward `mode` here, so this is always the central stencil (:89-90)
    gxx, _ = compute_spatial_gradients(velx, dx, dy, padding_mode)
    _, gyy = compute_spatial_gradients(vely, dx, dy, padding_mode)
    return gxx + gyy


def _fluid_mask(emitter: Tensor, collider: Optional[Tensor]) -> Tensor:
    m = (emitter == 0.0).float()
    if collider is not None:
        m = m * (collider == 0.0).float()
    return m


def divergence_loss(velx, vely, emitter_mask, collider_mask=None, dx=1.0, dy=1.0, eps=1e-8,
                    padding_mode="zeros", stencil_mode="central") -> Tensor:
    """physics_loss.py:94-126."""
    div = compute_divergence(velx, vely, dx, dy, padding_mode, mode=stencil_mode)
    fm = _fluid_mask(emitter_mask, collider_mask)
    n = fm.sum()
    if n < eps:
        return torch.tensor(0.0, dtype=velx.dtype)
    return (div.abs() * fm).sum() / (n + eps)


def gradient_loss(density_pred, density_target, emitter_mask=None, collider_mask=None, dx=1.0, dy=1.0,
                  eps=1e-8, padding_mode="zeros", mode="central") -> Tensor:
    """physics_loss.py:129-166."""
    px, py = compute_spatial_gradients(density_pred, dx, dy, padding_mode, mode=mode)
    tx, ty = compute_spatial_gradients(density_target, dx, dy, padding_mode, mode=mode)
    if emitter_mask is None:
        return F.l1_loss(px, tx) + F.l1_loss(py, ty)
    fm = _fluid_mask(emitter_mask, collider_mask)
    n = fm.sum()
    if n < eps:
        return torch.tensor(0.0, dtype=density_pred.dtype)
    return ((px - tx).abs() * fm).sum() / (n + eps) + ((py - ty).abs() * fm).sum() / (n + eps)


def emitter_spawn_loss(density_pred, density_current, emitter_mask, threshold: float = 0.01) -> Tensor:
    """physics_loss.py:169-177: mean over ALL cells of density where spawning is forbidden."""
    allowed = (emitter_mask > 0.01) | (density_current > threshold)
    return (density_pred * (~allowed).float()).mean()


@dataclass
class LossConfig:
    """Constructor arguments of PhysicsAwareLoss (physics_loss.py:181-192)."""
    mse_weight: float = 1.0
    divergence_weight: float = 0.1
    gradient_weight: float = 0.1
    emitter_weight: float = 0.1
    grid_spacing: float = 1.0
    enable_divergence: bool = True
    enable_gradient: bool = True
    enable_emitter: bool = True
    padding_mode: str = "zeros"
    stencil_mode: str = "forward"


def physics_loss(outputs: Tensor, targets: Tensor, inputs: Tensor, lc: LossConfig) -> Tuple[Tensor, Dict[str, Tensor]]:
    """PhysicsAwareLoss.forward, physics_loss.py:210-278.  Dict values stay tensors (no .item())."""
    d_pred, vx_pred, vy_pred = outputs[:, 0], outputs[:, 1], outputs[:, 2]
    d_tgt = targets[:, 0]
    emitter, collider = inputs[:, 4], inputs[:, 5]
    terms: Dict[str, Tensor] = {}
    mse = F.mse_loss(outputs, targets)
    terms["mse"] = mse
    total = lc.mse_weight * mse
    if lc.enable_divergence and lc.divergence_weight > 0:
        t = divergence_loss(vx_pred, vy_pred, emitter, collider, dx=lc.grid_spacing, dy=lc.grid_spacing,
                            padding_mode=lc.padding_mode, stencil_mode=lc.stencil_mode)
        terms["divergence"] = t
        total = total + lc.divergence_weight * t
    if lc.enable_gradient and lc.gradient_weight > 0:
        t = gradient_loss(d_pred, d_tgt, emitter, collider, dx=lc.grid_spacing, dy=lc.grid_spacing,
                          padding_mode=lc.padding_mode, mode=lc.stencil_mode)
        terms["gradient"] = t
        total = total + lc.gradient_weight * t
    if lc.enable_emitter and lc.emitter_weight > 0:
        t = emitter_spawn_loss(outputs[:, 0:1], inputs[:, 0:1], inputs[:, 4:5])
        terms["emitter"] = t
        total = total + lc.emitter_weight * t
    terms["total"] = total
    return total, terms


# --------------------------------------------------------------------------
# K-step rollout loss (ml/training/trainer.py:311-385)
# --------------------------------------------------------------------------
def rollout_loss(p: Dict[str, Tensor], cfg: OracleUNetConfig, lc: LossConfig, x0: Tensor, y_seq: Tensor,
                 masks: Tensor, *, weight_decay: float = 1.10, gradient_truncation: bool = False,
                 final_step_only: bool = False, training: bool = True,
                 drop_masks: Optional[List[Dict[str, Tensor]]] = None,
                 bn_state=None) -> Tuple[Tensor, Dict[str, Tensor], Tensor]:
    K = y_seq.shape[1]
    cur = x0[:, :3]                  # :323
    prev = x0[:, 3:4]                # :324
    total = torch.zeros((), dtype=x0.dtype)
    acc: Dict[str, Tensor] = {}
    wsum = 0.0
    pred = None
    for k in range(K):
        inp = torch.cat([cur, prev, masks[:, k, 0:1], masks[:, k, 1:2]], dim=1)        # :356-358
        dm = drop_masks[k] if drop_masks is not None else None
        pred = unet_forward(p, cfg, inp, training=training, drop_masks=dm, bn_state=bn_state)
        if final_step_only:                                                            # :326-346
            if k == K - 1:
                loss, d = physics_loss(pred, y_seq[:, k], inp, lc)
                return loss, d, pred
        else:
            lk, dk = physics_loss(pred, y_seq[:, k], inp, lc)                         # :362-363
            w = weight_decay ** k                                                       # :365
            total = total + w * lk
            wsum += w
            for key, v in dk.items():
                acc[key] = acc.get(key, 0.0) + w * v.detach()
        if gradient_truncation:                                                        # :374-379 / :339-344
            prev, cur = pred[:, 0:1].detach(), pred.detach()
        else:
            prev, cur = pred[:, 0:1], pred          # NB: prev is the NEW density, as in the reference
    total = total / wsum                                                               # :383
    return total, {k_: v / wsum for k_, v in acc.items()}, pred


# --------------------------------------------------------------------------
# optimisation step (ml/training/trainer.py:164-195; torch clip_grad_norm_ + Adam defaults)
# --------------------------------------------------------------------------
def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tuple[Dict[str, Tensor], Tensor]:
    """torch.nn.utils.clip_grad_norm_(error_if_nonfinite=False): coef = min(1, max/(norm+1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, total


def adam_step(p: Dict[str, Tensor], g: Dict[str, Tensor], m: Dict[str, Tensor], v: Dict[str, Tensor], step: int,
              lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam (amsgrad=False, weight_decay=0), in place. `step` is 1-based."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for k in p:
        m[k].mul_(beta1).add_(g[k], alpha=1.0 - beta1)
        v[k].mul_(beta2).addcmul_(g[k], g[k], value=1.0 - beta2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p[k].addcdiv_(m[k], denom, value=-lr / bc1)


def train_step(p, m, v, step, cfg, lc, x0, y_seq, masks, *, lr, clip_norm: Optional[float] = 1.0,
               weight_decay: float = 1.10, gradient_truncation=False, final_step_only=False,
               drop_masks=None) -> Tuple[float, Dict[str, float], Dict[str, Tensor]]:
    """One optimisation step of Trainer.train_epoch (trainer.py:164-195), K>=1 rollout form.

    For K == 1 with y_seq[:, 0] and masks from x0 this equals the single-step branch (:187-188).
    Updates p, m, v in place; returns (loss, loss dict, raw gradients before clipping).
    """
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in p.items()}
    loss, d, _ = rollout_loss(leaves, cfg, lc, x0, y_seq, masks, weight_decay=weight_decay,
                              gradient_truncation=gradient_truncation, final_step_only=final_step_only,
                              training=True, drop_masks=drop_masks)
    loss.backward()
    grads = {k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in leaves.items()}
    raw = {k: g.clone() for k, g in grads.items()}
    if clip_norm is not None:
        grads, _ = clip_grad_norm(grads, clip_norm)
    with torch.no_grad():
        adam_step(p, grads, m, v, step, lr)
    return float(loss.detach()), {k: float(t) for k, t in d.items()}, raw


# --------------------------------------------------------------------------
# autoregressive inference (ml/scripts/simple_infer.py:176-260, ml/training/test_evaluation.py:192-292)
# --------------------------------------------------------------------------
def simple_infer_masks(res: int, scale: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """Static circle emitter + rectangle collider of simple_infer.py:179-205.

    scale=1 reproduces the reference's hard-coded pixel geometry (radius 10, centre +50 rows, 40x8 collider
    at -40 rows); scale=res/128 keeps the same picture at other resolutions.
    """
    yy, xx = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
    radius = int(round(10 * scale))
    cx, cy = res // 2, res // 2 + int(round(50 * scale))
    emitter = (np.sqrt((xx - cx) ** 2 + (yy - cy) ** 2) <= radius).astype(np.float32)
    cw, chh = int(round(40 * scale)), int(round(8 * scale))
    ccx, ccy = res // 2, res // 2 - int(round(40 * scale))
    collider = np.zeros((res, res), np.float32)
    x0, x1 = max(0, ccx - cw // 2), min(res, ccx + cw // 2)
    y0, y1 = max(0, ccy - chh // 2), min(res, ccy + chh // 2)
    collider[y0:y1, x0:x1] = 1.0
    return emitter, collider


def rollout_infer(p, cfg, emitter: Tensor, collider: Tensor, frames: int, *, state0: Optional[Tensor] = None,
                  prev0: Optional[Tensor] = None) -> Tensor:
    """Autoregressive loop: input = [state_current(3), state_prev[0:1], emitter, collider];
    state_prev <- state_current (old), state_current <- output (simple_infer.py:226-260;
    test_evaluation.py:289-290).  emitter/collider: [B,H,W] (static) or [frames,B,H,W].
    Returns [frames,B,3,H,W]."""
    B, H, W = emitter.shape[-3:]
    cur = torch.zeros(B, 3, H, W) if state0 is None else state0.clone()
    prev = torch.zeros(B, 1, H, W) if prev0 is None else prev0.clone()
    outs = []
    with torch.no_grad():
        for f in range(frames):
            e = emitter if emitter.ndim == 3 else emitter[f]
            c = collider if collider.ndim == 3 else collider[f]
            inp = torch.cat([cur, prev, e[:, None], c[:, None]], dim=1)
            out = unet_forward(p, cfg, inp, training=False)
            prev = cur[:, 0:1]
            cur = out
            outs.append(out)
    return torch.stack(outs, 0)


# --------------------------------------------------------------------------
# validation metrics (ml/training/metrics.py:7-97)
# PARITY UNPINNED by execution: the reference module imports torchmetrics at load, which is not installed offline, so no
# fixture could be generated from it.  Everything except SSIM is a mean/sum over the stencil functions above (which ARE
# pinned by the physics-loss fixtures); SSIM restates torchmetrics' structural_similarity_index_measure defaults
# (gaussian 11x11, sigma 1.5, k1 0.01, k2 0.03, reflect pad -> depthwise conv -> crop pad -> mean).
# --------------------------------------------------------------------------
def per_channel_mse(pred: Tensor, target: Tensor) -> Dict[str, float]:                       # metrics.py:7-12
    return {"mse_density": float(((pred[:, 0] - target[:, 0]) ** 2).mean()), "mse_velx": float(((pred[:, 1] - target[:, 1]) ** 2).mean()),
            "mse_vely": float(((pred[:, 2] - target[:, 2]) ** 2).mean())}


def ssim_density(pred: Tensor, target: Tensor, data_range: float = 1.0) -> float:             # metrics.py:15-21
    p, t = pred[:, 0:1].double(), target[:, 0:1].double()
    sigma, ks = 1.5, int(3.5 * 1.5 + 0.5) * 2 + 1
    pad = (ks - 1) // 2
    dist = torch.arange((1 - ks) / 2, (1 + ks) / 2, 1, dtype=torch.float64)
    g = torch.exp(-((dist / sigma) ** 2) / 2)
    g = (g / g.sum())[None]
    kernel = (g.t() @ g)[None, None]
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    p = F.pad(p, (pad, pad, pad, pad), mode="reflect")
    t = F.pad(t, (pad, pad, pad, pad), mode="reflect")
    out = F.conv2d(torch.cat((p, t, p * p, t * t, p * t)), kernel)
    mu_p, mu_t, e_pp, e_tt, e_pt = out.split(p.shape[0])
    s_pp = torch.clamp(e_pp - mu_p ** 2, min=0.0)
    s_tt = torch.clamp(e_tt - mu_t ** 2, min=0.0)
    s_pt = e_pt - mu_p * mu_t
    full = ((2 * mu_p * mu_t + c1) * (2 * s_pt + c2)) / ((mu_p ** 2 + mu_t ** 2 + c1) * (s_pp + s_tt + c2))
    idx = full[..., pad:-pad, pad:-pad]
    return float(idx.reshape(idx.shape[0], -1).mean(-1).mean())


def gradient_l1(density_pred: Tensor, density_target: Tensor, dx=1.0, dy=1.0, padding_mode="zeros", mode="central") -> float:   # :24-38
    gpx, gpy = compute_spatial_gradients(density_pred, dx, dy, padding_mode, mode)
    gtx, gty = compute_spatial_gradients(density_target, dx, dy, padding_mode, mode)
    return float((gpx - gtx).abs().mean()) + float((gpy - gty).abs().mean())


def divergence_norm(velx: Tensor, vely: Tensor, dx=1.0, dy=1.0, padding_mode="zeros", mode="central") -> float:                 # :41-51
    return float(torch.sqrt((compute_divergence(velx, vely, dx, dy, padding_mode, mode) ** 2).mean()))


def kinetic_energy(velx: Tensor, vely: Tensor) -> float:                                       # :54-61
    return 0.5 * float((velx ** 2 + vely ** 2).sum())


def collider_violation(density: Tensor, collider_mask: Tensor, eps: float = 1e-8) -> float:    # :64-78
    n = collider_mask.sum()
    if n < eps:
        return 0.0
    return float((density * collider_mask).sum() / (n + eps))


def emitter_density_accuracy(density: Tensor, emitter_mask: Tensor, expected_injection: float = 0.8, eps: float = 1e-8) -> float:  # :81-97
    n = emitter_mask.sum()
    if n < eps:
        return 0.0
    return float(((density * emitter_mask).sum() / (n + eps) - expected_injection).abs())


def validation_metrics(outputs, target, emitter, collider, dx, padding_mode, mode) -> Dict[str, float]:
    """The batch_metrics dict of Trainer.validate (ml/training/trainer.py:270-300)."""
    d = per_channel_mse(outputs, target)
    d["divergence_norm"] = divergence_norm(outputs[:, 1], outputs[:, 2], dx, dx, padding_mode, mode)
    d["kinetic_energy"] = kinetic_energy(outputs[:, 1], outputs[:, 2])
    d["collider_violation"] = collider_violation(outputs[:, 0], collider)
    d["emitter_density_accuracy"] = emitter_density_accuracy(outputs[:, 0], emitter, 0.8)
    d["ssim_density"] = ssim_density(outputs, target)
    d["gradient_l1"] = gradient_l1(outputs[:, 0], target[:, 0], dx, dx, padding_mode, mode)
    return d


# --------------------------------------------------------------------------
# dataset sample layout (ml/dataset/npz_sequence.py:68-75,115-212,349-433; augmentations.py:4-35)
# --------------------------------------------------------------------------
def build_indices(frame_counts: List[int], rollout_steps: int, stride: int, offset: int = 0) -> List[Tuple[int, int]]:
    return [(si, t) for si, T in enumerate(frame_counts) for t in range(1 + offset, T - rollout_steps, stride)]


def make_sample(seq: Dict[str, np.ndarray], t: int, K: int, *, preload: bool, scales: Optional[Dict[str, float]] = None):
    """K==1 -> (x[6],y[3]); K>1 -> (x0[6], y_seq[K,3], masks).  On-disk path returns masks for t..t+K
    (K+1 entries, :185,:200,:210), the preloaded path for t+1..t+K (K entries, :406,:421)."""
    s = scales or {"S_density": 1.0, "S_velx": 1.0, "S_velz": 1.0}
    d = seq["density"].astype(np.float32) / np.float32(s["S_density"]) if scales else seq["density"].astype(np.float32)
    vx = seq["velx"].astype(np.float32) / np.float32(s["S_velx"]) if scales else seq["velx"].astype(np.float32)
    vz = seq["velz"].astype(np.float32) / np.float32(s["S_velz"]) if scales else seq["velz"].astype(np.float32)
    em = seq.get("emitter")
    co = seq.get("collider")
    em = np.zeros_like(d) if em is None else em.astype(np.float32)
    co = np.zeros_like(d) if co is None else co.astype(np.float32)
    x = np.stack([d[t], vx[t], vz[t], d[t - 1], em[t], co[t]], 0)
    if K == 1:
        return x, np.stack([d[t + 1], vx[t + 1], vz[t + 1]], 0)
    y = np.stack([np.stack([d[t + k], vx[t + k], vz[t + k]], 0) for k in range(1, K + 1)], 0)
    first = 1 if preload else 0
    mk = np.stack([np.stack([em[t + k], co[t + k]], 0) for k in range(first, K + 1)], 0)
    return x, y, mk


def flip_sample(*arrs):
    """Horizontal flip + negate vx (augmentations.py:4-11, 26-33)."""
    out = [np.ascontiguousarray(a[..., ::-1]).copy() for a in arrs]
    out[0][1] = -out[0][1]
    if out[1].ndim == 3:
        out[1][1] = -out[1][1]
    else:
        out[1][:, 1] = -out[1][:, 1]
    return tuple(out)


# --------------------------------------------------------------------------
# synthetic data of BASELINE.md section 3 (density U[0,1], vel clip(N(0,.3)), disc/rect masks)
# --------------------------------------------------------------------------
def synth_batch(B: int, K: int, R: int, seed: int = 1234) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    rs = np.random.RandomState(seed)

    def fields(n):
        d = rs.uniform(0, 1, (n, 1, R, R))
        v = np.clip(rs.normal(0, 0.3, (n, 2, R, R)), -1, 1)
        return np.concatenate([d, v], 1).astype(np.float32)

    yy, xx = np.mgrid[0:R, 0:R]
    em = np.zeros((B, R, R), np.float32)
    co = np.zeros((B, R, R), np.float32)
    for b in range(B):
        r = rs.randint(R // 12, R // 7)
        cy, cx = rs.randint(r, R - r, 2)
        em[b] = ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
        h, w = rs.randint(R // 16, R // 5), rs.randint(R // 16, R // 3)
        y0, x0 = rs.randint(0, R - h), rs.randint(0, R - w)
        rect = np.zeros((R, R), np.float32)
        rect[y0:y0 + h, x0:x0 + w] = 1
        co[b] = rect * (1 - em[b])
    st = fields(B)
    dprev = rs.uniform(0, 1, (B, 1, R, R)).astype(np.float32)
    x0 = np.concatenate([st, dprev, em[:, None], co[:, None]], 1)
    y = np.stack([fields(B) for _ in range(K)], 1)
    masks = np.broadcast_to(np.stack([em, co], 1)[:, None], (B, K, 2, R, R)).copy()
    return x0, y, masks
