"""Oracle restatement of ml/training/metrics.py: known-answer properties (the reference module cannot be imported here --
torchmetrics is absent -- so these pin the restatement by construction, not against the reference)."""
import numpy as np
import torch

import fluidnet_oracle as orc


def _fields(B=2, R=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(B, 3, R, R, generator=g)
    t = torch.rand(B, 3, R, R, generator=g)
    return p, t


def test_mse_and_energy_known_answers():
    p, t = _fields()
    assert orc.per_channel_mse(p, p) == {"mse_density": 0.0, "mse_velx": 0.0, "mse_vely": 0.0}
    q = p.clone(); q[:, 1] += 0.5
    d = orc.per_channel_mse(q, p)
    assert abs(d["mse_velx"] - 0.25) < 1e-6 and d["mse_density"] == 0.0
    ones = torch.ones(2, 8, 8)
    assert orc.kinetic_energy(ones, 2 * ones) == 0.5 * 5 * 128


def test_ssim_identity_range_and_crop():
    p, t = _fields()
    assert abs(orc.ssim_density(p, p) - 1.0) < 1e-12
    s = orc.ssim_density(p, t)
    assert -1.0 <= s < 0.2                      # independent noise: near zero
    # constant images: sigma = 0, ssim = (2 mu_p mu_t + c1) / (mu_p^2 + mu_t^2 + c1)
    a, b = torch.full((1, 3, 24, 24), 0.2), torch.full((1, 3, 24, 24), 0.6)
    want = (2 * 0.2 * 0.6 + 1e-4) / (0.04 + 0.36 + 1e-4)
    assert abs(orc.ssim_density(a, b) - want) < 1e-6


def test_divergence_norm_and_gradient_l1_of_linear_fields():
    R = 16
    yy, xx = torch.meshgrid(torch.arange(R, dtype=torch.float32), torch.arange(R, dtype=torch.float32), indexing="ij")
    vx, vy = (2.0 * xx)[None], (3.0 * yy)[None]
    # replicate padding, central stencil: interior divergence 2 + 3 = 5, edges see half of one term
    dn = orc.divergence_norm(vx, vy, 1.0, 1.0, "replicate", "central")
    div = orc.compute_divergence(vx, vy, 1.0, 1.0, "replicate", "central")
    assert float(div[0, 5, 5]) == 5.0 and abs(dn - float(torch.sqrt((div ** 2).mean()))) < 1e-6
    assert orc.gradient_l1(vx, vx, 1.0, 1.0, "zeros", "forward") == 0.0
    assert abs(orc.gradient_l1(vx + xx[None], vx, 1.0, 1.0, "replicate", "forward") - (R - 1) / R) < 1e-6


def test_mask_metrics():
    d = torch.full((2, 8, 8), 0.5)
    m = torch.zeros(2, 8, 8); m[:, 2:4, 2:4] = 1
    assert abs(orc.collider_violation(d, m) - 0.5) < 1e-6
    assert orc.collider_violation(d, torch.zeros_like(m)) == 0.0
    assert abs(orc.emitter_density_accuracy(d, m, 0.8) - 0.3) < 1e-6
    assert orc.emitter_density_accuracy(d, torch.zeros_like(m)) == 0.0


def test_validation_dict_keys_match_the_reference_trainer():
    p, t = _fields()
    m = torch.zeros(2, 32, 32); m[:, 4:8, 4:8] = 1
    d = orc.validation_metrics(p, t, m, torch.zeros_like(m), 2.0 / 32, "replicate", "forward")
    assert set(d) == {"mse_density", "mse_velx", "mse_vely", "divergence_norm", "kinetic_energy", "collider_violation",
                      "emitter_density_accuracy", "ssim_density", "gradient_l1"}
    assert all(np.isfinite(v) for v in d.values())
