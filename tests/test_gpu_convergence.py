"""Convergence-level parity of the benchmarked bf16 plan: does a bf16 K4 run TRACK an fp32 K4 run over hundreds of
optimisation steps?  The point-wise gradient cosine of tests/test_gpu_k4_parity.py cannot say that.

UNet_medium, 128x128, batch 4, dropout 0, on a fixed stream of 8 synthetic batches visited in order:
    150 K1 steps (Adam lr 1e-3, clip 1.0), then 100 K4 steps warm-started from those weights with a fresh optimiser -- the reference's
    curriculum hand-over (ml/scripts/train.py:132-138: the child variant loads the parent's model_state_dict only) around the loop of
    ml/training/trainer.py:142-208 (train_epoch) and :311-385 (_compute_rollout_loss, weighted average, wd 1.10) --
once with RolloutTrainPlan(dtype="fp32", conv_impl="ref") (the plan pinned to the reference by G1/G2/G3/G7) and once with the bf16 MFMA
plan, from the same initial weights and batch order.  Two trainings never agree weight by weight (fp32 atomics noise is amplified by
Adam), so the statement is about the LOSS: per pass over the stream (8 steps) the mean training loss of the two
runs must agree within BAND_REL, in both phases, and the loss of the final weights on held-out batches within HELD_REL.
Stated tolerance: BAND_REL = 5 % per 8-step window (+ 2e-4 absolute), except the first TRANSIENT = 2 windows of a
phase (TRANSIENT_REL = 25 %: the first steps of a fresh Adam are moves of +-lr along the signs of a gradient, and how high the loss spikes
after the K1 -> K4 hand-over differs run to run and is larger in bf16, whose gradient noise floor gives more parameters a random first
move -- first-window differences of 0.7, 2.4, 6.7, 8.7 and 13.0 % in five recorded runs, 2.6 % in the second window of the 8.7 % run,
< 1.5 % from the third on); the mean over a phase's later windows within MEAN_REL = 2 %, the last window within LAST_REL = 2 %;
HELD_REL = 5 %.  The curves of the last run are written to gpurun_out/convergence.json.
"""
import json
import os

import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.data.synthetic import synth_rollout_batch
from fluidnet_amd.engine.train_plan import RolloutTrainPlan
from fluidnet_amd.models.unet import UNet, UNetConfig
from fluidnet_amd.training.dp import dp_train_step
from fluidnet_amd.training.physics_loss import PhysicsAwareLoss

pytestmark = pytest.mark.gpu
B, R, NB = 4, 128, 8
STEPS_K1, STEPS_K4, LR = 150, 100, 1e-3
BAND_REL, BAND_ABS, HELD_REL = 0.05, 2e-4, 0.05
TRANSIENT, TRANSIENT_REL, MEAN_REL, LAST_REL = 2, 0.25, 0.02, 0.02


def _crit():
    return PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")


def _plan(model, precision, K):
    p = RolloutTrainPlan(model.cfg, _crit(), B, R, R, K, dtype=precision, conv_impl="ref" if precision == "fp32" else "auto",
                         weight_decay=1.10, clip_norm=1.0)
    p.bind(torch.device("cuda"), model.flat_params, model.flat_grads)
    return p


def _load(p, batch, K):
    x0, y, mk = batch
    p.x0.tensor.copy_(x0)
    p.y.tensor.copy_(y[:, :K])
    p.masks.tensor.copy_(mk[:, :K])


def _train(precision, stream, held):
    cfg = orc.medium_config(dropout=0.0)
    m = UNet(UNetConfig(**cfg.__dict__), precision=precision, conv_impl="ref" if precision == "fp32" else "auto")
    m.load_state_dict(orc.make_params(cfg, seed=7))
    m.to("cuda")
    curves = {}
    for K, steps in ((1, STEPS_K1), (4, STEPS_K4)):
        p = _plan(m, precision, K)                 # a new plan = a fresh Adam state, like the curriculum's new Trainer
        losses = []
        for s in range(1, steps + 1):
            _load(p, stream[(s - 1) % NB], K)
            p.set_adam_scalars(LR, s)
            dp_train_step(p, 1)
            losses.append(p.loss_dict()["total"])
        curves[K] = np.asarray(losses)
        if K == 4:
            hl = []
            for hb in held:                         # the final weights on batches the run never saw (forward + loss only)
                _load(p, hb, 4)
                p.run(("fwd", "fin"))
                hl.append(p.loss_dict()["total"])
            curves["held"] = float(np.mean(hl))
        del p
    return curves


def _windows(c):
    n = len(c) // NB
    return c[:n * NB].reshape(n, NB).mean(axis=1)


def test_bf16_run_tracks_fp32_run_over_the_k1_k4_curriculum():
    def dev(b):
        return tuple(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in b)
    stream = [dev(synth_rollout_batch(B, 4, R, seed=100 + i)) for i in range(NB)]
    held = [dev(synth_rollout_batch(B, 4, R, seed=900 + i)) for i in range(2)]
    c32 = _train("fp32", stream, held)
    c16 = _train("bf16", stream, held)
    rep = {"config": dict(B=B, R=R, stream_batches=NB, steps_k1=STEPS_K1, steps_k4=STEPS_K4, lr=LR)}
    worst = {}
    for K in (1, 4):
        w32, w16 = _windows(c32[K]), _windows(c16[K])
        rel = np.abs(w16 - w32) / w32
        worst[K] = float(rel.max())
        rep[f"K{K}"] = {"loss_fp32_per_window": w32.tolist(), "loss_bf16_per_window": w16.tolist(), "rel_diff_per_window": rel.tolist(),
                        "first_step": [float(c32[K][0]), float(c16[K][0])], "last_window": [float(w32[-1]), float(w16[-1])]}
        print(f"[convergence] K{K}: fp32 {w32[0]:.5f} -> {w32[-1]:.5f}, bf16 {w16[0]:.5f} -> {w16[-1]:.5f}; worst window rel diff {rel.max():.4f} "
              f"(mean {rel.mean():.4f})")
    rep["held_out"] = {"fp32": c32["held"], "bf16": c16["held"], "rel_diff": abs(c16["held"] - c32["held"]) / c32["held"]}
    print(f"[convergence] held-out K4 loss of the final weights: fp32 {c32['held']:.5f}, bf16 {c16['held']:.5f} (rel diff {rep['held_out']['rel_diff']:.4f})")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "convergence.json"), "w") as f:
        json.dump(rep, f, indent=1)
    # both runs must actually train: K1 loss at least halves, K4 does not blow up after the hand-over and ends below where it started
    for c in (c32, c16):
        assert np.all(np.isfinite(c[1])) and np.all(np.isfinite(c[4]))
        assert _windows(c[1])[-1] < 0.5 * c[1][0]
        assert _windows(c[4])[-1] < _windows(c[4])[0]
    for K in (1, 4):
        w32, w16 = _windows(c32[K]), _windows(c16[K])
        band = np.full_like(w32, BAND_REL)
        band[:TRANSIENT] = TRANSIENT_REL
        assert np.all(np.abs(w16 - w32) <= band * w32 + BAND_ABS), (K, worst[K], (np.abs(w16 - w32) / w32).tolist())
        m32, m16 = w32[TRANSIENT:].mean(), w16[TRANSIENT:].mean()
        assert abs(m16 - m32) <= MEAN_REL * m32, (K, m16, m32)
        assert abs(w16[-1] - w32[-1]) <= LAST_REL * w32[-1], (K, w16[-1], w32[-1])
    assert abs(c16["held"] - c32["held"]) <= HELD_REL * c32["held"], rep["held_out"]
