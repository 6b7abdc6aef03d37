"""K-step rollout training step as one native plan.

Follows Trainer._compute_rollout_loss (ml/training/trainer.py:311-385) + the optimisation step of
Trainer.train_epoch (trainer.py:164-195): K UNet applications chained through their predictions, a physics loss
per step, weights rollout_weight_decay**k normalised by their sum, backward through all K applications, gradient
clipping by global norm, Adam.  K == 1 with the masks taken from x0 is the single-step branch (:187-188).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np

from . import runtime as rt
from .plan import Buf, OpRec, PlanBuilder, Ref, allocate, lower


class RolloutTrainPlan:
    def __init__(self, cfg, lc, B: int, H: int, W: int, K: int, *, dtype: str = "bf16", conv_impl: str = "auto",
                 weight_decay: float = 1.10, gradient_truncation: bool = False, final_step_only: bool = False,
                 mask_frames: Optional[int] = None, mask_offset: int = 0, seed: int = 0, world: int = 1, training: bool = True,
                 clip_norm: Optional[float] = 1.0, betas=(0.9, 0.999), eps: float = 1e-8):
        self.cfg, self.lc, self.B, self.H, self.W, self.K = cfg, lc, B, H, W, K
        self.world = world
        self.mask_offset = mask_offset
        if world > 1 and cfg.norm == "batch":
            raise NotImplementedError("BatchNorm2d under data parallelism would normalise with per-rank batch statistics (the reference "
                                      "has no multi-GPU path to mirror); shard UNet_small runs by experiment, not by batch")
        self.clip_norm = clip_norm
        self.betas, self.eps = betas, eps
        pb = self.pb = PlanBuilder(cfg, B, H, W, dtype=dtype, conv_impl=conv_impl, training=training, seed=seed)
        HW = H * W
        Km = mask_frames if mask_frames is not None else K
        self.x0 = pb.new_buf("x0", (B, 6, H, W), "f32")
        self.y = pb.new_buf("y_seq", (B, K, 3, H, W), "f32")
        self.masks = pb.new_buf("masks", (B, Km, 2, H, W), "f32")
        self.accum = pb.new_buf("loss_accum", (8,), "f32")
        self.normsq = pb.new_buf("grad_normsq", (2,), "f64")
        self.gcnt = pb.new_buf("global_fluid_count", (K,), "f64")
        self.m = pb.new_buf("adam_m", (pb.n_params_padded,), "f32")
        self.v = pb.new_buf("adam_v", (pb.n_params_padded,), "f32")
        if final_step_only:
            w = [0.0] * (K - 1) + [1.0]
        else:
            raw = [weight_decay ** k for k in range(K)]
            w = [x / sum(raw) for x in raw]
        self.weights = w

        # ---------------- section 0: weight re-layout, K forward applications, loss reductions
        s_fwd: List[OpRec] = []
        pb.emit_prep_weights(s_fwd)
        pb.emit_zero(s_fwd, self.accum)
        self.tapes, self.sums, self.terms = [], [], []
        loss_in = []
        for k in range(K):
            mk = k + mask_offset
            em = (Ref(self.masks, (mk * 2 + 0) * HW), Km * 2 * HW)
            co = (Ref(self.masks, (mk * 2 + 1) * HW), Km * 2 * HW)
            if k == 0:
                st = [(Ref(self.x0, c * HW), 6 * HW) for c in range(4)]
            else:
                pr = self.tapes[k - 1].pred
                st = [(Ref(pr, c * HW), 3 * HW) for c in range(3)] + [(Ref(pr, 0), 3 * HW)]   # prev <- NEW density (trainer.py:378)
            sums = pb.new_buf(f"loss_sums{k}", (8,), "f64")
            terms = pb.new_buf(f"loss_terms{k}", (8,), "f32")
            pb.emit_zero(s_fwd, sums)          # before the application: HEAD_FWD and LOSS_REDUCE stay adjacent (one fused launch)
            t = pb.emit_forward(s_fwd, k, st + [em, co])
            self.tapes.append(t)
            self.sums.append(sums)
            self.terms.append(terms)
            tgt = (Ref(self.y, k * 3 * HW), K * 3 * HW)
            loss_in.append((tgt, st[0], em, co))
            i, flags = pb.loss_ops_args(lc, tgt, st[0], em, co)
            pb.op(s_fwd, "LOSS_REDUCE", i, f=[lc.grid_spacing], p=[Ref(t.pred), tgt[0], st[0][0], em[0], co[0], Ref(sums)], tag=f"loss{k}")
        # ---------------- section 1: loss finalisation (after the optional all-reduce of the fluid counts)
        s_fin: List[OpRec] = []
        for k in range(K):
            tgt, in0, em, co = loss_in[k]
            i, flags = pb.loss_ops_args(lc, tgt, in0, em, co)
            pb.op(s_fin, "LOSS_FINALIZE", [B, H, W, flags, world],
                  f=[lc.mse_weight, lc.divergence_weight, lc.gradient_weight, lc.emitter_weight, w[k]],
                  p=[Ref(self.sums[k]), Ref(self.gcnt, k) if world > 1 else None, Ref(self.terms[k]), Ref(self.accum)], tag=f"loss{k}")
        # ---------------- section 2: backward through the K applications
        s_bwd: List[OpRec] = []
        pb.emit_zero_grads(s_bwd)
        dnext = None
        self.dpred = pb.new_buf("dpred", (B, 3, H, W), "f32")
        # Gradient buckets (data parallel): parameter gradients accumulate over all K applications and are final when the LAST
        # backward pass emitted (k = 0 unless truncated) has visited their layer.  That pass reports, decoder first, when a group of
        # layers is complete; the group's 3x3 weight gradients are folded to OIHW right there (on the weight-gradient lane, behind the
        # launches they depend on) and `grad_buckets` records (index into the bwd section, [lo, hi) of the flat gradient buffer), so
        # that training/dp.py can start that range's all-reduce while the rest of the pass still runs.
        self.grad_buckets: List[tuple] = []
        unprepped: List[str] = []

        def bucket_ready(ops, prefixes):
            pb.flush_affine(ops)
            lane = pb.wgrad_lane
            if lane:
                pb.emit_lane_edge(ops, 0, lane)      # gamma / beta / bias gradients of the group were written on lane 0
            unprepped.extend(pb.emit_unprep(ops, prefixes, lane=lane))
            self.grad_buckets.append((len(ops), pb.param_range(prefixes), lane))

        passes, flows = [], False             # the applications that get a backward pass (same rule as the loop below)
        for k in reversed(range(K)):
            if w[k] == 0.0 and not flows:
                continue
            passes.append(k)
            flows = k > 0 and not gradient_truncation
        for k in reversed(range(K)):
            if w[k] == 0.0 and dnext is None:
                continue
            tgt, in0, em, co = loss_in[k]
            def loss_grad(ops, k=k, tgt=tgt, in0=in0, em=em, co=co, dnext=dnext):
                pb.emit_loss_grad(ops, lc, self.tapes[k].pred, tgt, in0, em, co, self.sums[k], dnext, None, self.dpred, w[k],
                                  gcnt=Ref(self.gcnt, k) if world > 1 else None, world=world)
            pass_down = k > 0 and not gradient_truncation
            dx = pb.new_buf(f"dxin{k}", (B, 4, H, W), "f32") if pass_down else None
            last = k == (passes[-1] if passes else 0)
            # (the loss gradient is emitted right in front of HEAD_BWD: the executor runs the pair as one launch)
            pb.emit_backward(s_bwd, self.tapes[k], self.dpred, dx, 4, bucket_cb=bucket_ready if (last and world > 1) else None,
                             pre_head=loss_grad)
            dnext = (Ref(dx), 4 * HW) if pass_down else None
        pb.emit_unprep(s_bwd, skip=unprepped)
        # the rest of the flat buffer (outer encoder levels, stem; everything when there are no earlier buckets) completes here
        covered = [r for _, r, _ in self.grad_buckets]
        lo_rest = 0
        hi_rest = min([r[0] for r in covered], default=pb.n_params_padded)
        self.grad_buckets.append((len(s_bwd), (lo_rest, hi_rest), 0))
        tiles = sorted(r for _, r, _ in self.grad_buckets)
        assert tiles[0][0] == 0 and tiles[-1][1] == pb.n_params_padded and all(a[1] == b[0] for a, b in zip(tiles, tiles[1:])), \
            f"gradient buckets must tile the flat buffer: {tiles}"
        # ---------------- section 3: clip + Adam
        s_opt: List[OpRec] = []
        n = pb.n_params_padded
        pb.emit_zero(s_opt, self.normsq)
        pb.op(s_opt, "GRAD_NORM", [n & 0x7FFFFFFF, n >> 31], p=[Ref(pb.grads), Ref(self.normsq)])
        pb.op(s_opt, "ADAM", [n & 0x7FFFFFFF, n >> 31], f=[0.0, betas[0], betas[1], eps, 1.0, 1.0, clip_norm or 0.0, 1.0],
              p=[Ref(pb.params), Ref(pb.grads), Ref(self.m), Ref(self.v), Ref(self.normsq)])
        self.sections: Dict[str, List[OpRec]] = {"fwd": s_fwd, "fin": s_fin, "bwd": s_bwd, "opt": s_opt}
        self.step_count = 0
        self._lowered: Dict[str, np.ndarray] = {}

    # ------------------------------------------------------------------ host side
    def set_adam_scalars(self, lr: float, step: int) -> None:
        """Write lr and the bias corrections of optimisation step `step` (1-based) into the ADAM op."""
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** step
        bc2s = math.sqrt(1.0 - b2 ** step)
        rec = self.sections["opt"][-1]
        rec.f[0], rec.f[4], rec.f[5] = lr, bc1, bc2s
        if "opt" in self._lowered:
            a = self._lowered["opt"][-1]
            a["f"][0], a["f"][4], a["f"][5] = lr, bc1, bc2s

    def bind(self, device, params_flat, grads_flat, adam_m=None, adam_v=None, norm_state=None) -> int:
        """Allocate all internal buffers on `device` and lower the sections (device must be a GPU for run()).  adam_m / adam_v:
        externally owned Adam moment buffers (so that several plans of different batch shapes share one optimizer state)."""
        self.pb.params.tensor = params_flat
        self.pb.grads.tensor = grads_flat
        if adam_m is not None:
            self.m.tensor, self.v.tensor = adam_m, adam_v
        if norm_state is not None:             # BatchNorm2d running statistics owned by the caller; else a fresh (0, 1) buffer
            assert norm_state.numel() == self.pb.nstate.numel
            self.pb.nstate.tensor = norm_state
        nbytes = allocate(self.pb, device)
        if str(device) != "cpu":
            self._lowered = {k: lower(self.pb, v) for k, v in self.sections.items()}
        return nbytes

    def set_dropout_seed(self, seed: int) -> None:
        """New Dropout2d masks for the next step (the hash also mixes in the application index and the layer)."""
        for n in ("fwd", "bwd"):
            if n in self._lowered:
                rt.patch_seed(self._lowered[n], seed)

    def run(self, names=("fwd", "fin", "bwd", "opt"), stream=None) -> None:
        for n in names:
            rt.run_ops(self._lowered[n], stream)

    def loss_dict(self) -> Dict[str, float]:
        a = self.accum.tensor.detach().cpu().numpy()
        out = {"mse": float(a[0])}
        lc = self.lc
        if lc.enable_divergence and lc.divergence_weight > 0:
            out["divergence"] = float(a[1])
        if lc.enable_gradient and lc.gradient_weight > 0:
            out["gradient"] = float(a[2])
        if lc.enable_emitter and lc.emitter_weight > 0:
            out["emitter"] = float(a[3])
        out["total"] = float(a[4])
        return out
