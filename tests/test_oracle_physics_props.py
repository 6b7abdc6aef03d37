"""The reference's 16 property tests (ml/tests/test_physics_loss.py:13-208), restated against the CPU oracle.
The same properties are checked against the HIP path in tests/test_gpu_physics_loss.py."""
import pytest
import torch

import fluidnet_oracle as orc


def _ramp(n, axis):
    a = torch.arange(n).float()
    g = a[None, :].expand(n, n) if axis == "x" else a[:, None].expand(n, n)
    return g[None].expand(2, -1, -1)


def test_constant_field_zero_gradient():
    gx, gy = orc.compute_spatial_gradients(torch.ones(2, 32, 32), padding_mode="replicate")
    assert gx.abs().max() < 1e-6 and gy.abs().max() < 1e-6


def test_linear_field_constant_gradient():
    gx, gy = orc.compute_spatial_gradients(_ramp(32, "x"), 1.0, 1.0, padding_mode="replicate")
    assert abs(float(gx.mean()) - 1.0) < 0.1 and abs(float(gy.mean())) < 0.1


def test_gradient_shape():
    f = torch.randn(4, 64, 64)
    gx, gy = orc.compute_spatial_gradients(f)
    assert gx.shape == f.shape == gy.shape


def test_rotation_field_divergence_free():
    div = orc.compute_divergence(-_ramp(32, "y"), _ramp(32, "x"), 1.0, 1.0, padding_mode="replicate")
    assert div.abs().mean() < 0.1


def test_divergence_loss_all_emitter_is_zero():
    l = orc.divergence_loss(torch.randn(2, 32, 32), torch.randn(2, 32, 32), torch.ones(2, 32, 32), padding_mode="replicate")
    assert abs(float(l)) < 1e-6


def test_divergence_loss_expansion_positive():
    l = orc.divergence_loss(_ramp(32, "x"), _ramp(32, "y"), torch.zeros(2, 32, 32), padding_mode="replicate")
    assert float(l) > 0


def test_gradient_loss_identical_zero():
    d = torch.randn(2, 64, 64)
    assert abs(float(orc.gradient_loss(d, d.clone(), padding_mode="replicate"))) < 1e-6


def test_gradient_loss_different_positive():
    assert float(orc.gradient_loss(torch.randn(2, 64, 64), torch.randn(2, 64, 64), padding_mode="replicate")) > 0


def test_gradient_loss_fully_masked_zero():
    one = torch.ones(2, 32, 32)
    l = orc.gradient_loss(torch.randn(2, 32, 32), torch.randn(2, 32, 32), one, one, padding_mode="replicate")
    assert abs(float(l)) < 1e-6


def test_gradient_loss_partial_mask_finite():
    p = torch.zeros(2, 32, 32)
    t = torch.zeros(2, 32, 32)
    p[:, :10] = torch.linspace(0, 10, 10)[None, :, None]
    p[:, 10:22] = 1.0
    t[:, 10:22] = 1.0
    p[:, 22:] = torch.linspace(10, 0, 10)[None, :, None]
    e = torch.zeros(2, 32, 32)
    e[:, :10] = 1
    c = torch.zeros(2, 32, 32)
    c[:, 22:] = 1
    l = orc.gradient_loss(p, t, e, c, padding_mode="replicate")
    assert float(l) >= 0 and torch.isfinite(l)


def _rand_case():
    o, t, i = torch.randn(2, 3, 64, 64), torch.randn(2, 3, 64, 64), torch.randn(2, 6, 64, 64)
    i[:, 4] = (i[:, 4] > 0).float()
    i[:, 5] = (i[:, 5] > 0).float()
    return o, t, i


def test_physics_loss_returns_scalar_and_keys():
    loss, d = orc.physics_loss(*_rand_case(), orc.LossConfig())
    assert loss.ndim == 0 and {"total", "mse", "divergence", "gradient"} <= set(d)


def test_disable_flags():
    _, d = orc.physics_loss(*_rand_case(), orc.LossConfig(enable_divergence=False, enable_gradient=False))
    assert "divergence" not in d and "gradient" not in d and "mse" in d


def test_components_finite():
    loss, d = orc.physics_loss(*_rand_case(), orc.LossConfig())
    assert torch.isfinite(loss) and all(torch.isfinite(v) for v in d.values())


@pytest.mark.parametrize("pm", ["zeros", "replicate", "reflect"])
def test_padding_modes(pm):
    f = torch.randn(2, 32, 32)
    gx, gy = orc.compute_spatial_gradients(f, padding_mode=pm)
    assert gx.shape == f.shape and torch.isfinite(gx).all() and torch.isfinite(gy).all()
