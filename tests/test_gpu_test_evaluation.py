"""run_test_evaluation / run_rollout_evaluation (ml/training/test_evaluation.py:78-143, :192-292, :364-476) on the GPU against the
oracle: the reference's loops restated here with the oracle's UNet and metric functions, rollout by rollout at batch 1 with the
ground-truth masks of every frame, while the code under test advances all rollouts in one batched plan."""
import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.config.training_config import PhysicsLossConfig, TrainingConfig
from fluidnet_amd.models.unet import UNet, UNetConfig
from fluidnet_amd.training.test_evaluation import _get_starting_frames, run_rollout_evaluation, run_test_evaluation
from helpers import tiny_cfg

pytestmark = pytest.mark.gpu
R, T, STEPS = 32, 16, 5


def _write_split(tmp_path, n_seq=2):
    d = tmp_path / "npz" / "test"
    d.mkdir(parents=True)
    seqs = []
    for i in range(n_seq):
        x0, _, mk = orc.synth_batch(T, 1, R, seed=40 + i)
        em = np.stack([np.roll(mk[0, 0, 0], t, axis=-1) for t in range(T)])          # masks that move from frame to frame
        co = np.stack([np.roll(mk[0, 0, 1], -t, axis=-2) for t in range(T)])
        s = dict(density=x0[:, 0], velx=x0[:, 1], velz=x0[:, 2], emitter=em.astype(np.float32), collider=co.astype(np.float32))
        np.savez_compressed(d / f"seq_{i:04d}.npz", **s)
        seqs.append(s)
    return tmp_path / "npz", seqs


def _setup(tmp_path):
    root, seqs = _write_split(tmp_path)
    cfg = tiny_cfg(base_channels=16)
    params = orc.make_params(cfg, seed=9)
    m = UNet(UNetConfig(**cfg.__dict__), precision="fp32")
    m.load_state_dict(params)
    m.to("cuda")
    tc = TrainingConfig(batch_size=4, npz_dir=root, normalize=False, base_channels=16, depth=2, padding_mode="replicate",
                        physics_loss=PhysicsLossConfig(grid_spacing=2.0 / R, stencil_mode="forward"))
    return root, seqs, cfg, params, m, tc


def test_rollout_evaluation_matches_per_rollout_oracle(tmp_path, monkeypatch):
    root, seqs, cfg, params, m, tc = _setup(tmp_path)
    import fluidnet_amd.training.test_evaluation as te
    monkeypatch.setattr(te, "STARTING_POINTS", [3, "middle", -6])
    monkeypatch.setattr(te, "_get_starting_frames", lambda n, steps=STEPS, points=None: _get_starting_frames(n, steps, [3, "middle", -6]))
    got = run_rollout_evaluation(m, tc, "cuda", npz_dir=root, steps=STEPS)["avg_per_step"]
    assert len(got) == STEPS
    # the reference's loop, one rollout at a time
    per_step = [[] for _ in range(STEPS)]
    dx = 2.0 / R
    for s in seqs:
        for t0 in (3, 8, 10):
            cur = torch.from_numpy(np.stack([s["density"][t0], s["velx"][t0], s["velz"][t0]]))[None]
            prev = torch.from_numpy(s["density"][t0 - 1])[None, None]
            for k in range(STEPS):
                inp = torch.cat([cur, prev, torch.from_numpy(s["emitter"][t0 + k])[None, None], torch.from_numpy(s["collider"][t0 + k])[None, None]], 1)
                pred = orc.unet_forward(params, cfg, inp, training=False)
                gt = torch.from_numpy(np.stack([s["density"][t0 + k + 1], s["velx"][t0 + k + 1], s["velz"][t0 + k + 1]]))[None]
                mm = orc.per_channel_mse(pred, gt)
                mm["ssim_density"] = orc.ssim_density(pred, gt)
                mm["divergence_norm"] = orc.divergence_norm(pred[:, 1], pred[:, 2], dx, dx, "replicate", "forward")
                mm["divergence_norm_gt"] = orc.divergence_norm(gt[:, 1], gt[:, 2], dx, dx, "replicate", "forward")
                mm["gradient_l1"] = orc.gradient_l1(pred[:, 0], gt[:, 0], dx, dx, "replicate", "forward")
                per_step[k].append(mm)
                prev, cur = cur[:, 0:1], pred
    for k in range(STEPS):
        for key in got[k]:
            want = float(np.mean([r[key] for r in per_step[k]]))
            # fp32 on both sides; the autoregressive state drifts apart by rounding only
            assert abs(got[k][key] - want) <= 2e-4 * max(1.0, abs(want)) * (k + 1), (k, key, got[k][key], want)
    assert run_rollout_evaluation(m, tc, "cuda", npz_dir=tmp_path / "nowhere") == {}


def test_test_evaluation_model_and_persistence(tmp_path):
    root, seqs, cfg, params, m, tc = _setup(tmp_path)
    res = run_test_evaluation(m, tc, "cuda", npz_dir=root)
    assert set(res) == {"model", "persistence"}
    keys = {"mse_density", "mse_velx", "mse_vely", "divergence_norm", "kinetic_energy", "collider_violation", "emitter_density_accuracy",
            "ssim_density", "gradient_l1"}
    assert set(res["model"]) == keys == set(res["persistence"])
    # oracle: batches of 4 consecutive (t, t+1) pairs in dataset order, averaged over batches like MetricsTracker
    dx = 2.0 / R
    pairs = [(s, t) for s in seqs for t in range(1, T - 1)]
    acc = {"model": [], "persistence": []}
    for b0 in range(0, len(pairs), 4):
        chunk = pairs[b0:b0 + 4]
        inp = torch.from_numpy(np.stack([np.stack([s["density"][t], s["velx"][t], s["velz"][t], s["density"][t - 1], s["emitter"][t], s["collider"][t]])
                                         for s, t in chunk]))
        tgt = torch.from_numpy(np.stack([np.stack([s["density"][t + 1], s["velx"][t + 1], s["velz"][t + 1]]) for s, t in chunk]))
        out = orc.unet_forward(params, cfg, inp, training=False)
        acc["model"].append(orc.validation_metrics(out, tgt, inp[:, 4], inp[:, 5], dx, "replicate", "forward"))
        acc["persistence"].append(orc.validation_metrics(inp[:, :3], tgt, inp[:, 4], inp[:, 5], dx, "replicate", "forward"))
    for who in acc:
        for key in keys:
            want = float(np.mean([a[key] for a in acc[who]]))
            assert abs(res[who][key] - want) <= 2e-4 * max(1.0, abs(want)), (who, key, res[who][key], want)
