"""Validation metrics (METRICS kernel, through the C ABI) against the oracle restatement of ml/training/metrics.py.
fp32 sums in fp64 accumulators on both sides: 1e-5 relative.  SSIM parity against the reference is unpinned (torchmetrics is
absent); here it is checked against the oracle's independent formulation (reflect pad + conv2d + crop, fp64)."""
import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.training import metrics as M

pytestmark = pytest.mark.gpu


def _batch(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(B, 3, H, W, generator=g)
    p[:, 1:] = p[:, 1:] * 2 - 1
    t = (p + 0.1 * torch.randn(B, 3, H, W, generator=g)).clamp(-1, 1)
    em = torch.zeros(B, H, W); em[:, H // 2: H // 2 + 3, 2:6] = 1
    co = torch.zeros(B, H, W); co[:, 1:4, W - 6: W - 2] = 1
    return p, t, em, co


@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 20, 44), (64, 128, 128)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("pm,mode", [("replicate", "forward"), ("zeros", "central"), ("reflect", "central"), ("circular", "forward")])
def test_validation_metrics_match_oracle(shape, pm, mode):
    B, H, W = shape
    p, t, em, co = _batch(B, H, W, seed=B + H)
    want = orc.validation_metrics(p, t, em, co, 2.0 / W, pm, mode)
    got = M.validation_metrics(p.cuda(), t.cuda(), em.cuda(), co.cuda(), dx=2.0 / W, padding_mode=pm, mode=mode)
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (k, got[k], want[k])


def test_reference_named_functions_and_strided_views():
    p, t, em, co = _batch(4, 24, 40, seed=5)
    pc, tc = p.cuda(), t.cuda()
    x = torch.cat([pc, pc[:, :1], em.cuda()[:, None], co.cuda()[:, None]], 1)           # masks as channel views of a (B,6,H,W) input
    d = M.compute_per_channel_mse(pc, tc)
    w = orc.per_channel_mse(p, t)
    assert all(abs(d[k] - w[k]) < 1e-6 for k in w)
    assert abs(M.compute_ssim_density(pc, tc) - orc.ssim_density(p, t)) < 1e-5
    assert abs(M.compute_gradient_l1(pc[:, 0], tc[:, 0], 0.1, 0.1, "replicate", "forward") - orc.gradient_l1(p[:, 0], t[:, 0], 0.1, 0.1, "replicate", "forward")) < 1e-4
    assert abs(M.compute_divergence_norm(pc[:, 1], pc[:, 2], 0.1, 0.1, "zeros", "central") - orc.divergence_norm(p[:, 1], p[:, 2], 0.1, 0.1, "zeros", "central")) < 1e-4
    assert abs(M.compute_kinetic_energy(pc[:, 1], pc[:, 2]) - orc.kinetic_energy(p[:, 1], p[:, 2])) < 1e-3
    assert abs(M.compute_collider_violation(pc[:, 0], x[:, 5]) - orc.collider_violation(p[:, 0], co)) < 1e-6
    assert abs(M.compute_emitter_density_accuracy(pc[:, 0], x[:, 4]) - orc.emitter_density_accuracy(p[:, 0], em)) < 1e-6
    assert M.compute_collider_violation(pc[:, 0], torch.zeros_like(pc[:, 0])) == 0.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.compute_per_channel_mse(p, t)
    tr = M.MetricsTracker()
    tr.update({"a": 1.0}); tr.update({"a": 3.0, "b": 2.0})
    assert tr.compute_averages() == {"a": 2.0, "b": 2.0}


def test_g10_reference_metrics_fixture(golden_dir):
    """The HIP METRICS kernel (through the reference-named functions) against the values the REAL ml/training/metrics.py produced for
    tests/fixture_inputs.py::metrics_inputs (tests/golden/make_golden.py g10); SSIM is not in the fixture (torchmetrics absent)."""
    import os

    import fixture_inputs as fi
    g = np.load(os.path.join(golden_dir, "g10_metrics.npz"))
    pred, tgt, em, co = (torch.from_numpy(a).cuda() for a in fi.metrics_inputs())
    for k, v in M.compute_per_channel_mse(pred, tgt).items():
        assert abs(v - float(g[k])) < 1e-6 * max(1.0, abs(float(g[k]))), k
    for pm in ("zeros", "replicate", "reflect", "circular"):
        for st in ("forward", "central"):
            ref = float(g[f"gradient_l1/{pm}/{st}"])
            assert abs(M.compute_gradient_l1(pred[:, 0], tgt[:, 0], 0.0625, 0.0625, pm, st) - ref) < 1e-5 * max(1.0, abs(ref)), (pm, st)
            ref = float(g[f"divergence_norm/{pm}/{st}"])
            assert abs(M.compute_divergence_norm(pred[:, 1], pred[:, 2], 0.0625, 0.0625, pm, st) - ref) < 1e-5 * max(1.0, abs(ref)), (pm, st)
    ref = float(g["kinetic_energy"])
    assert abs(M.compute_kinetic_energy(pred[:, 1], pred[:, 2]) - ref) < 1e-5 * ref
    assert abs(M.compute_collider_violation(pred[:, 0], co) - float(g["collider_violation"])) < 1e-6
    assert M.compute_collider_violation(pred[2:, 0], co[2:]) == 0.0
    assert abs(M.compute_emitter_density_accuracy(pred[:, 0], em, 0.8) - float(g["emitter_density_accuracy"])) < 1e-6
    assert M.compute_emitter_density_accuracy(pred[:, 0], torch.zeros_like(em), 0.8) == 0.0
