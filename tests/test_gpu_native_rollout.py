"""Native rollout C ABI (fnet_rollout_*, no torch on its path) against the Python-built RolloutInferPlan on the same weights:
both hosts emit the same op records, so the states must agree bit for bit; plus the viewer's impulse contract and argument
checking.  The Python plan itself is checked against the oracle in tests/test_gpu_api.py / test_gpu_lockstep.py; the native library is
also compared with the oracle directly (the G8 fixture's geometry and masks, every frame from the library's own previous state)."""
import os

import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.engine.native_rollout import NativeRollout, checkpoint_floats
from fluidnet_amd.engine.rollout_plan import RolloutInferPlan
from fluidnet_amd.engine.runtime import FluidNetHipError
from fluidnet_amd.models.unet import UNet, UNetConfig
from helpers import tiny_cfg

pytestmark = pytest.mark.gpu


def _weights(cfg, seed):
    params = orc.make_params(cfg, seed=seed)
    return params, np.concatenate([v.reshape(-1).numpy() for v in params.values()]).astype(np.float32)


@pytest.mark.parametrize("name,cfg,B,R", [
    ("tiny", tiny_cfg(), 2, 32),
    ("tiny-noresid-nearest-zeros", tiny_cfg(use_residual=False, upsample="nearest", padding_mode="zeros", act="silu"), 1, 16),
    ("medium", orc.medium_config(dropout=0.0), 3, 64),
    # the other shipped architectures' block types (ml/config/model_architectures/): UNet_small, UNet_large, and UNetConfig's defaults
    ("small-batchnorm-avgpool", orc.medium_config(dropout=0.0, base_channels=16, depth=2, norm="batch", upsample="nearest",
                                                  downsample="avgpool", padding_mode="zeros"), 2, 64),
    ("large-avgpool", orc.medium_config(dropout=0.0, base_channels=48, depth=4, downsample="avgpool"), 1, 64),
    ("small-medium-base20", orc.medium_config(dropout=0.0, base_channels=20, downsample="avgpool"), 2, 32),      # UNet_small-medium.yaml
    ("groupnorm-maxpool", tiny_cfg(norm="group", act="silu", downsample="maxpool", base_channels=16, group_norm_groups=4), 2, 32),
    # GroupNorm over channel counts that run zero-padded (20 -> 24, 12 -> 16): the groups are formed over the REAL channels
    ("groupnorm-base20", orc.medium_config(dropout=0.0, base_channels=20, norm="group", group_norm_groups=8, downsample="avgpool"), 2, 32),
    ("groupnorm-base12", tiny_cfg(norm="group", base_channels=12, group_norm_groups=3), 2, 32),
])
def test_native_rollout_equals_python_plan(name, cfg, B, R):
    params, flat = _weights(cfg, seed=3)
    model = UNet(UNetConfig(**cfg.__dict__), precision="bf16")
    model.load_state_dict(params, strict=cfg.norm != "batch")
    if cfg.norm == "batch":                    # a trained checkpoint's running statistics: anything but (0, 1)
        g = torch.Generator().manual_seed(5)
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                m.running_var.copy_(1.0 + 0.3 * torch.rand(m.running_var.shape, generator=g))
    model = model.cuda().eval()
    flat = checkpoint_floats(model.state_dict())
    em, co = orc.simple_infer_masks(R, scale=R / 128.0)
    e = np.repeat(em[None], B, 0).astype(np.float32)
    c = np.repeat(co[None], B, 0).astype(np.float32)
    plan = RolloutInferPlan(model.cfg, B, R, R, params_flat=model.flat_params, norm_state=model.flat_norm_state)
    plan.reset(torch.from_numpy(e).cuda(), torch.from_numpy(c).cuda())
    nat = NativeRollout(model.cfg, flat, B, R, R)
    nat.set_masks(e, c)
    nat.set_state(None, None)
    for frames in (1, 4):
        want = plan.step(frames).cpu().numpy()
        nat.step(frames)
        got = nat.read_state()
        assert np.isfinite(got).all()
        assert np.array_equal(got, want), (name, frames, float(np.abs(got - want).max()))
    # impulse: velx += ix, vely += iy on the current state (Simulation.cpp:263-271)
    rs = np.random.RandomState(0)
    imp = (0.1 * rs.randn(B, 2, R, R)).astype(np.float32)
    before = nat.read_state()
    nat.add_impulse(imp)
    after = nat.read_state()
    assert np.array_equal(after[:, 0], before[:, 0]) and np.allclose(after[:, 1:], before[:, 1:] + imp, atol=1e-7)
    plan.cur.tensor[:, 1:] += torch.from_numpy(imp).cuda()
    nat.step(1)
    assert np.array_equal(nat.read_state(), plan.step(1).cpu().numpy())
    nat.close()


def test_native_rollout_against_oracle_on_g8_geometry(golden_dir):
    """ml/scripts/simple_infer.py:226-260 through the C ABI alone: masks of the G8 fixture, zero initial state, 16 frames; frame f against
    the oracle's UNet applied to the library's own state of frames f-1 / f-2 (bf16 bar of DESIGN.md section 5: rel-L2 <= 8e-2)."""
    g = np.load(os.path.join(golden_dir, "g8_rollout16.npz"))
    cfg = tiny_cfg(base_channels=16)             # the MFMA kernels need >= 16 channels (the fixture's own net has 8)
    params, _ = _weights(cfg, seed=23)
    model = UNet(UNetConfig(**cfg.__dict__), precision="bf16")
    model.load_state_dict(params)
    flat = checkpoint_floats(model.state_dict())
    em, co = g["emitter"].astype(np.float32), g["collider"].astype(np.float32)
    nat = NativeRollout(model.cfg, flat, 1, 32, 32)
    nat.set_masks(em[None], co[None])
    nat.set_state(None, None)
    e, c = torch.from_numpy(em)[None, None], torch.from_numpy(co)[None, None]
    cur, prev = torch.zeros(1, 3, 32, 32), torch.zeros(1, 1, 32, 32)
    worst = 0.0
    with torch.no_grad():
        for f in range(16):
            nat.step(1)
            got = torch.from_numpy(nat.read_state())
            ref = orc.unet_forward(params, cfg, torch.cat([cur, prev, e, c], 1), training=False)
            worst = max(worst, float((got - ref).norm() / (ref.norm() + 1e-12)))
            prev, cur = cur[:, 0:1], got
    nat.close()
    print(f"[native rollout vs oracle] 16 frames, worst rel-L2 {worst:.3e}")
    assert np.isfinite(worst) and worst < 8e-2


def test_native_rollout_rejects_bad_arguments():
    cfg = tiny_cfg()
    _, flat = _weights(cfg, seed=1)
    mcfg = UNetConfig(**cfg.__dict__)
    with pytest.raises(ValueError):
        NativeRollout(mcfg, flat[:-1], 1, 32, 32)
    with pytest.raises(FluidNetHipError):
        NativeRollout(mcfg, flat, 1, 30, 32)            # not divisible by 2^depth
    with pytest.raises(NotImplementedError):
        NativeRollout(UNetConfig(**tiny_cfg(norm="none").__dict__), flat, 1, 32, 32)
    nat = NativeRollout(mcfg, flat, 1, 32, 32)
    with pytest.raises(ValueError):
        nat.set_masks(np.zeros((2, 32, 32), np.float32), None)
    nat.close()
