"""Shared test helpers (test infrastructure)."""
import numpy as np
import torch

import fluidnet_oracle as orc


def tiny_cfg(**over):
    kw = dict(in_channels=6, out_channels=3, base_channels=8, depth=2, norm="instance", act="gelu", dropout=0.0,
              upsample="bilinear", downsample="stride", padding_mode="replicate", use_residual=True,
              bottleneck_blocks=1, output_activation="linear_clamp")
    kw.update(over)
    return orc.OracleUNetConfig(**kw)


def loss_c2(R, **over):
    kw = dict(mse_weight=1.0, divergence_weight=0.01, gradient_weight=0.001, emitter_weight=0.01,
              grid_spacing=2.0 / R, padding_mode="replicate", stencil_mode="forward")
    kw.update(over)
    return orc.LossConfig(**kw)


def relerr(a, b, rtol_for_floor=2e-4, atol=1e-6):
    """||a-b|| / (||b|| + (atol/rtol)*sqrt(n)): `relerr < rtol`  <=>  ||a-b|| < rtol*||b|| + atol*sqrt(n)."""
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).double().cpu()
    return float((a - b).norm() / (b.norm() + (atol / rtol_for_floor) * (b.numel() ** 0.5)))


def pack_params(pb, params, device="cpu"):
    flat = torch.zeros(pb.n_params_padded, dtype=torch.float32, device=device)
    for k, v in params.items():
        o = pb.offsets[k]
        flat[o:o + v.numel()] = v.reshape(-1).to(device)
    return flat


def unpack(pb, flat, name):
    o = pb.offsets[name]
    shp = pb.shapes[name]
    return flat[o:o + int(np.prod(shp))].view(shp)


def drop_masks_for(pb_cfg, B, seed, app, p):
    """Injected Dropout2d keep-masks for the oracle that reproduce the HIP engine's hash (site = app*64 + block index)."""
    from fluidnet_amd.engine.dropout import drop_scale_table
    out = {}
    for idx, (key, ch) in enumerate(orc.dropout_keys(pb_cfg)):
        tbl = drop_scale_table(seed & 0x7FFFFFFF, app * 64 + idx, B, ch, p)
        out[key] = torch.from_numpy((tbl > 0).astype(np.float32))
    return out
