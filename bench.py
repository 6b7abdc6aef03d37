#!/usr/bin/env python3
"""Headline benchmark: K4 UNet_medium 128^2 rollout-training samples/s (BASELINE.json metric, config[1]).

    python bench.py --gpus N --steps K --warmup W            (N>1: one rank per GPU over RCCL; started by torch.distributed.run, or --
                                                              when no launcher set WORLD_SIZE -- by this script itself, which then only
                                                              spawns N children, relays rank 0's line and never touches the GPU)
    python bench.py --mode rollout                            (600-frame 256^2 x 64-scene inference rollout alone, frames/s)

A step = one pass of the hot path over one synthetic batch resident in HBM: K UNet forwards + physics losses, BPTT
backward, (RCCL gradient all-reduce), clip + Adam.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fluidnet_amd.data.synthetic import simple_infer_masks, synth_rollout_batch  # noqa: E402
from fluidnet_amd.engine import runtime as rt  # noqa: E402
from fluidnet_amd.engine.rollout_plan import RolloutInferPlan  # noqa: E402
from fluidnet_amd.engine.train_plan import RolloutTrainPlan  # noqa: E402
from fluidnet_amd.models.unet import UNet, UNetConfig  # noqa: E402
from fluidnet_amd.training import launch  # noqa: E402
from fluidnet_amd.training.dp import ExchangeProbe, dp_train_step, torch_all_reduce  # noqa: E402
from fluidnet_amd.training.physics_loss import PhysicsAwareLoss  # noqa: E402
from fluidnet_amd.training.trainer import step_seed  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0               # MI355X HBM3E (MI355X_MICROARCH.md: ~8 TB/s)


# the reference's shipped architectures (ml/config/model_architectures/*.yaml); the headline metric is quoted on UNet_medium
ARCHS = {
    "medium": dict(),
    "small": dict(base_channels=16, depth=2, norm="batch", upsample="nearest", downsample="avgpool", padding_mode="zeros", dropout=0.0),
    "small-medium": dict(base_channels=20, depth=3, downsample="avgpool", dropout=0.0),
    "large": dict(base_channels=48, depth=4, downsample="avgpool", dropout=0.0),
}
# the only throughput-like figures the reference publishes: one rollout frame, batch 1, 128^2, through
# ONNX Runtime's CUDA EP on an RTX 3070 -- frames/s = 1 / latency, the small model at the fast end of its "1-2 ms"
PUBLISHED_B1_128_FPS = {"medium": 200.0, "small": 1000.0}


def arch_cfg(arch, dropout=None, depth=None):
    act = os.environ.get("FNET_BENCH_ACT")              # development knob (activation cost studies); the headline is the architecture's own
    kw = dict(in_channels=6, out_channels=3, base_channels=32, depth=3, norm="instance", act="gelu", dropout=0.1, upsample="bilinear",
              downsample="stride", padding_mode="replicate", use_residual=True, bottleneck_blocks=1, output_activation="linear_clamp")
    kw.update(ARCHS[arch])
    if dropout is not None:
        kw["dropout"] = dropout
    if depth is not None:
        kw["depth"] = depth
    if act:
        kw["act"] = act
    return UNetConfig(**kw)


def c2_loss(R):
    return PhysicsAwareLoss(mse_weight=1.0, divergence_weight=0.01, gradient_weight=0.001, emitter_weight=0.01, grid_spacing=2.0 / R,
                            enable_divergence=True, enable_gradient=True, enable_emitter=True, padding_mode="replicate",
                            stencil_mode="forward")


def dist_setup(n):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # FNET_DIST_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks then share devices and the
    # all-reduces go through the host); production is one rank per GPU over nccl (= RCCL on ROCm)
    backend = os.environ.get("FNET_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local >= ndev:
        raise SystemExit(f"LOCAL_RANK {local} but only {ndev} GPUs visible: one rank per GPU")
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def synthetic_store(R, K, world, rank, batch, seed=1234):
    """An HBM-resident synthetic dataset (BASELINE.md section 3 statistics) behind the reference's dataset API: seq_*.npz files written
    to a temporary directory, read by FluidNPZSequenceDataset (preloaded: the reference's default), uploaded once by
    HbmSequenceStore.  Every optimisation step then forms its batch with one GATHER_BATCH launch (flip augmentation included) inside
    the timed region, rank-disjoint rows of one seeded permutation under data parallelism."""
    import tempfile

    from fluidnet_amd.dataset import FluidNPZSequenceDataset
    from fluidnet_amd.dataset.hbm_resident import HbmBatchLoader, HbmSequenceStore
    n_seq, T = 16, max(K + 3, (batch * world * 2) // 16 + K + 2)
    tmp = tempfile.TemporaryDirectory(prefix="fnet_bench_")
    d = os.path.join(tmp.name, "npz", "train")
    os.makedirs(d)
    for i in range(n_seq):
        x0, y, mk = synth_rollout_batch(T, 1, R, seed=seed + 17 * i)
        np.savez(os.path.join(d, f"seq_{i:04d}.npz"), density=x0[:, 0], velx=x0[:, 1], velz=x0[:, 2],
                 emitter=np.repeat(mk[:1, 0, 0], T, 0), collider=np.repeat(mk[:1, 0, 1], T, 0))
    ds = FluidNPZSequenceDataset(os.path.join(tmp.name, "npz"), "train", normalize=False, is_training=True,
                                 augmentation_config={"enable_augmentation": True, "flip_probability": 0.5}, preload=True,
                                 rollout_steps=K, stride=1)
    store = HbmSequenceStore(ds)
    tmp.cleanup()
    return HbmBatchLoader(store, batch, shuffle=True, seed=seed, rank=rank, world=world)


class TrainStepper:
    def __init__(self, B, R, K, precision, world, rank, lr=3e-4, conv_impl="auto", arch="medium", depth=None, loader=True):
        torch.manual_seed(42)
        self.model = UNet(arch_cfg(arch, depth=depth), precision=precision, conv_impl=conv_impl).cuda()
        self.world, self.rank = world, rank
        self.lr = lr
        self.plan = RolloutTrainPlan(self.model.cfg, c2_loss(R), B, R, R, K, dtype=precision, conv_impl=conv_impl, world=world,
                                     weight_decay=1.10, clip_norm=1.0, seed=1234 + rank)
        self.bytes = self.plan.bind(torch.device("cuda"), self.model.flat_params, self.model.flat_grads,
                                    norm_state=self.model.flat_norm_state)
        x0, y, mk = synth_rollout_batch(B, K, R, seed=1234 + rank)
        self.plan.x0.tensor.copy_(torch.from_numpy(x0))
        self.plan.y.tensor.copy_(torch.from_numpy(y))
        self.plan.masks.tensor.copy_(torch.from_numpy(mk))
        self.t = 0
        # batch formation inside the step: a new batch per step from the HBM-resident store (K = 1 plans take the masks of frame t)
        self.loader = synthetic_store(R, K, world, rank, B) if loader else None
        self._batches, self._epoch = [], 0
        self.probe = None
        if world > 1:
            import torch.distributed as dist
            self.dist = dist
            self.probe = ExchangeProbe(use_events=dist.get_backend() == "nccl")
            self.all_reduce = torch_all_reduce(dist, self.probe)
            if dist.get_backend() == "gloo":            # rehearsal mode: through the host
                h = self.model.flat_params.detach().cpu()
                dist.broadcast(h, 0)
                self.model.flat_params.data.copy_(h)
            else:
                dist.broadcast(self.model.flat_params, 0)

    def next_batch(self):
        if not self._batches:
            self.loader.set_epoch(self._epoch)
            self._batches = list(range(self.loader.upload_epoch()))          # one index / flip-flag upload per epoch
            self._epoch += 1
        self.loader.gather_uploaded(self._batches.pop(), self.plan.x0.tensor, self.plan.y.tensor, self.plan.masks.tensor)

    def step(self):
        self.t += 1
        p = self.plan
        if self.loader is not None:
            self.next_batch()
        p.set_adam_scalars(self.lr, self.t)
        p.set_dropout_seed(step_seed(self.t, self.rank))
        dp_train_step(p, self.world, self.all_reduce if self.world > 1 else None, probe=self.probe)

    def exchange_alone_ms(self, iters=10):
        """The step's exchange by itself, nothing to hide behind: the K fluid counts + the gradient buckets all-reduced `iters` times
        between two events on the current stream (ranks aligned by a barrier first).  Next to `allreduce_exposed_ms` it says how much
        of the exchange the backward pass covered."""
        p, g = self.plan, self.plan.pb.grads.tensor
        keep = g.clone()
        ar = torch_all_reduce(self.dist)
        barrier(self.world)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            hs = [ar(p.gcnt.tensor, "sum")]
            for _end, (lo, hi), _lane in p.grad_buckets:
                if hi > lo:
                    hs.append(ar(g[lo:hi], "avg"))
            for h in hs:
                if h is not None:
                    h.wait()
        e1.record()
        torch.cuda.synchronize()
        host = (time.perf_counter() - t0) * 1e3 / iters
        g.copy_(keep)
        return (e0.elapsed_time(e1) / iters) if self.dist.get_backend() == "nccl" else host


def profile_ops(plan_arrays, recs_by_section, dump_path=None):
    """One extra instrumented step: HIP events (on the launch stream) around every op -> per-kernel-kind time and FLOPs."""
    agg = {}
    tm = [rt.Timer() for _ in range(2)]
    rows = []
    # an empty start/stop bracket still shows a few microseconds between the two events: take its minimum as the timer's own
    # cost and subtract it, so that the per-launch averages agree with rocprofv3's kernel durations
    empty = []
    for _ in range(20):
        tm[1].start()
        tm[1].stop()
        empty.append(tm[1].elapsed_ms())
    overhead = min(empty)
    # a unit's statistics + normalise / activate records, and its two adjoint passes, are timed TOGETHER (one fnet_plan_run call): the
    # executor issues such a pair as one fused launch on small images, which a record-by-record timing would never see.  The pair's
    # algorithmic bytes are those of its second record (+ a materialised G): what an ideal fusion moves.
    PAIRS = {("IN_STATS", "APPLY"): "NORM_ACT_FWD", ("INACT_BWD_REDUCE", "INACT_BWD_APPLY"): "NORM_ACT_BWD",
             ("HEAD_FWD", "LOSS_REDUCE"): "HEAD_LOSS_FWD", ("LOSS_GRAD", "HEAD_BWD"): "LOSS_HEAD_BWD"}
    for name, arr in plan_arrays.items():
        recs = recs_by_section[name]
        k = -1
        while k + 1 < len(arr):
            k += 1
            t = tm[0]
            r = recs[k]
            kind = rt.OPNAME[r.code]
            pair = PAIRS.get((kind, rt.OPNAME[recs[k + 1].code])) if k + 1 < len(arr) else None
            one = arr[k:k + (2 if pair else 1)].copy()
            one["i"][:, rt.LANE_SLOT] = 0          # per-op timing: side-lane launches are timed on the caller's stream too
            t.start()
            rt.run_ops(one)
            t.stop()
            ms = max(t.elapsed_ms() - overhead, 0.0)
            if kind.startswith("EVENT_"):          # lane bookkeeping, not a kernel
                continue
            if pair:
                r2 = recs[k + 1]
                rows.append((name, k, pair, r2.tag, ms, 0.0, list(r.i[:12])))
                esz = 2 if r2.dtype == rt.BF16 else 4
                a = agg.setdefault(pair, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0})
                a["ms"] += ms
                a["n"] += 1
                if pair in ("HEAD_LOSS_FWD", "LOSS_HEAD_BWD"):
                    # head + loss in one pass: both records' tensors, the prediction (forward) / dpred (backward) not read back
                    e1 = 2 if r.dtype == rt.BF16 else 4
                    a["bytes"] += hbm_bytes(kind, r, e1) + hbm_bytes(rt.OPNAME[r2.code], r2, esz) - 3 * 4 * r.i[0] * r.i[1] * r.i[2]
                else:
                    a["bytes"] += hbm_bytes(rt.OPNAME[r2.code], r2, esz)
                if pair == "NORM_ACT_BWD" and len(r.p) > 6 and r.p[6] is not None:
                    a["bytes"] += r.i[0] * r.i[1] * r.i[2] * r.i[3] * esz          # G out
                k += 1
                continue
            rows.append((name, k, kind, r.tag, ms, r.flops, list(r.i[:12])))
            if kind == "CONV3X3":
                kind = "CONV3X3_" + ("MFMA" if r.i[8] == rt.CONV_MFMA else "REF")
            if kind == "WGRAD":
                kind = "WGRAD_" + ("MFMA" if r.i[8] == rt.CONV_MFMA else "REF")
            a = agg.setdefault(kind, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0})
            a["ms"] += ms
            a["n"] += 1
            a["flops"] += r.flops
            esz = 2 if r.dtype == rt.BF16 else 4
            if kind.startswith("CONV3X3"):
                # algorithmic bytes of one launch: the input pixels once, the output pixels once, the 9 weight slices once
                a["bytes"] += esz * (r.i[0] * (r.i[2] + r.i[5]) + 9 * r.i[2] * r.i[5])
            else:
                a["bytes"] += hbm_bytes(kind, r, esz)
    if dump_path:
        os.makedirs(os.path.dirname(dump_path) or ".", exist_ok=True)
        with open(dump_path, "w") as f:
            f.write("section,idx,op,tag,ms,gflop,tflops,i\n")
            for (name, k, kind, tag, ms, fl, iv) in rows:
                f.write(f"{name},{k},{kind},{tag},{ms:.4f},{fl / 1e9:.3f},{(fl / (ms * 1e-3) / 1e12) if ms > 0 else 0:.1f},{' '.join(map(str, iv))}\n")
    return agg


def hbm_bytes(kind, r, esz):
    """Algorithmic HBM bytes of one launch of a bandwidth-bound op: every tensor it must read once + every tensor it must write once
    (record layouts: include/fluidnet_ops.h); 0 for ops that are not accounted."""
    def has(k):
        return k < len(r.p) and r.p[k] is not None
    if kind in ("IN_STATS", "APPLY", "INACT_BWD_REDUCE", "INACT_BWD_APPLY", "FOLD_SUM", "BIAS_GRAD"):
        B, H, W, C = r.i[0], r.i[1], r.i[2], r.i[3]
        inner, padded = B * H * W * C * esz, B * (H + 2) * (W + 2) * C * esz
        if kind == "IN_STATS":
            return inner                                          # raw once
        if kind == "APPLY":
            return padded * (2 + (1 if has(4) else 0))            # raw (+ residual) in, activation out (ring included)
        if kind == "INACT_BWD_REDUCE":
            return inner * (2 + (1 if has(1) else 0) + (1 if has(6) else 0))      # g1 (+ g2), raw in (+ G out)
        if kind == "INACT_BWD_APPLY":
            return 2 * inner + padded                             # G, raw in; d_raw out (ring zeroed)
        if kind == "FOLD_SUM":
            return inner * (1 + (1 if has(1) else 0)) + padded
        return inner                                              # BIAS_GRAD
    if kind in ("UPSAMPLE2X", "UPSAMPLE2X_BWD"):
        B, h, w, C = r.i[0], r.i[1], r.i[2], r.i[3]
        return B * C * esz * (h * w + 4 * h * w)                  # the coarse tensor on one side, the fine one on the other
    if kind in ("LOSS_REDUCE", "LOSS_GRAD"):
        B, H, W = r.i[0], r.i[1], r.i[2]
        planes = 3 + 3 + 1 + 2 + (3 if kind == "LOSS_GRAD" else 0)    # pred, target, density_t, two masks (+ d pred out), fp32 planes
        return B * H * W * 4 * planes
    if kind == "HEAD_FWD":
        B, H, W, C = r.i[0], r.i[1], r.i[2], r.i[3]
        return B * H * W * (C * esz + 2 * 3 * 4)                  # the activation in; z and pred out (fp32 planes)
    if kind == "HEAD_BWD":
        B, H, W, C = r.i[0], r.i[1], r.i[2], r.i[3]
        return B * H * W * (3 * 3 * 4 + 2 * C * esz)              # dpred, z, pred in; the activation in, its gradient out
    if kind == "ADAM":
        n = r.i[0] + (r.i[1] << 31)
        return n * 4 * 7                                          # p, g, m, v in; p, m, v out
    return 0


def hbm_roofline(agg):
    """The bandwidth-bound families (norm / activation passes, loss stencils, resampling, optimiser): algorithmic bytes per step over the
    HIP-event time of their launches, against the HBM data-sheet figure (MI355X_MICROARCH.md: ~8 TB/s; a read-mostly streaming kernel
    reaches ~5.2 TB/s on these boxes)."""
    hb = {k: v for k, v in agg.items() if v["bytes"] > 0 and v["ms"] > 0 and not k.startswith("CONV3X3")}
    if not hb:
        return None
    tb, tms = sum(v["bytes"] for v in hb.values()), sum(v["ms"] for v in hb.values())
    return {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": round(tb / (tms * 1e-3) / 1e9, 1),
            "frac": round(tb / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_gb_per_step": round(tb / 1e9, 2), "ms_per_step": round(tms, 3),
            "kernels": {k: {"gb_per_s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "ms_per_step": round(v["ms"], 3), "launches": v["n"]}
                        for k, v in sorted(hb.items(), key=lambda kv: -kv[1]["ms"])}}


def reference_gemm():
    """For reading `roofline.frac`: what the vendor-tuned dense bf16 GEMM (torch.matmul -> hipBLASLt) reaches on THIS box, on a large square
    problem and on the plain-GEMM shapes of the step's three heaviest convolution classes (M = padded pixels of 64 images, N = couts,
    K = 9 * cin, no im2col cost counted; k_conv3x3_dma runs those layers at 817 / 935 / 923 TFLOP/s, halo staging and epilogue included:
    profiles/r3y_op_classes.txt).  Informational: `peak` stays the data-sheet figure."""
    import torch
    res = {}
    for key, (m, n, k) in (("square_8192", (8192, 8192, 8192)), ("conv_64to64_128x128_shape", (1081600, 64, 576)),
                           ("conv_128to128_64x64_shape", (278784, 128, 1152)), ("conv_256to256_32x32_shape", (73984, 256, 2304))):
        a_ = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
        b_ = torch.randn(k, n, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            c_ = a_ @ b_
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            c_ = a_ @ b_
        e1.record()
        torch.cuda.synchronize()
        res[key + "_tflops"] = round(2.0 * m * n * k / (e0.elapsed_time(e1) / 10) / 1e9, 1)
        del a_, b_, c_
    res["kind"] = "torch.matmul bf16 (hipBLASLt), measured in this run"
    return res


def reference_stream():
    """For reading `hbm_roofline.frac`: what plain streaming reaches on THIS box in the same run -- a device-to-device copy of 1 GiB
    (read + write) and a fill of the same buffer through torch (the runtime's blit / vendor kernels).  Informational:
    `peak` stays the 8 TB/s data-sheet figure."""
    import torch
    n = 1 << 28                                    # int32 elements = 1 GiB
    a_ = torch.ones(n, device="cuda", dtype=torch.int32)
    b_ = torch.empty_like(a_)
    res = {}
    for key, fn, nbytes in (("copy", lambda: b_.copy_(a_), 8.0 * n), ("fill", lambda: b_.zero_(), 4.0 * n)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[key + "_gb_per_s"] = round(nbytes / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e9, 1)
    del a_, b_
    res["kind"] = "torch copy_ / zero_ over 1 GiB, measured in this run"
    return res


def pmc_traffic(kernel):
    """Per-launch HBM bytes of `kernel` from the newest committed PMC summary (tools/pmc_traffic.py; rocprofv3 cannot run inside
    the timed process, so the counters come from a separate pass of the same workload), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            k = json.load(f)["kernels"].get(kernel)
        return (k["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)) if k else (None, None)
    except (OSError, ValueError, KeyError):
        return None, None


def cpu_baseline_train(K, R):
    """The CPU oracle (oracle/fluidnet_oracle.py, kind="port") timed on this host's cores, SURVEY.md section 8d protocol: batch 8, fp32,
    K = 1 and the benchmarked K, 3 warm-up + 5 timed optimisation steps each, median.  `value` is the benchmarked K's samples/s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fluidnet_oracle as orc
    host_cpus = os.cpu_count() or 1
    threads = min(host_cpus, 16) if os.environ.get("OMP_NUM_THREADS") is None else int(os.environ["OMP_NUM_THREADS"])
    torch.set_num_threads(threads)
    B = 8
    cfg = orc.medium_config(dropout=0.1)
    lc = orc.LossConfig(1.0, 0.01, 0.001, 0.01, 2.0 / R, True, True, True, "replicate", "forward")
    res = {}
    for k in sorted({1, K}):
        p = orc.make_params(cfg, seed=7)
        m = {n: torch.zeros_like(v) for n, v in p.items()}
        v = {n: torch.zeros_like(t) for n, t in p.items()}
        x0, y, mk = orc.synth_batch(B, k, R, seed=1234)
        args = (torch.from_numpy(x0), torch.from_numpy(y), torch.from_numpy(mk))
        ts = []
        for step in range(1, 9):
            t0 = time.perf_counter()
            orc.train_step(p, m, v, step, cfg, lc, *args, lr=3e-4, clip_norm=1.0)
            ts.append(time.perf_counter() - t0)
        res[k] = float(np.median(ts[3:]))
    out = {"value": round(B / res[K], 4), "unit": "samples/s", "cores": threads, "host_cpus": host_cpus, "kind": "port",
           "sample": f"oracle/fluidnet_oracle.py train_step, UNet_medium {R}x{R}, batch {B}, fp32, torch CPU with {threads} threads on a "
                     f"{host_cpus}-CPU host; per K: 3 warm-up + 5 timed steps, median; K{K}: {res[K]:.2f} s/step"}
    if K != 1:
        out["k1_samples_per_s"] = round(B / res[1], 4)
        out["sample"] += f", K1: {res[1]:.2f} s/step"
    return out


def cpu_baseline_rollout(R, budget_s=15.0):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import fluidnet_oracle as orc
    threads = min(os.cpu_count() or 1, 16) if os.environ.get("OMP_NUM_THREADS") is None else int(os.environ["OMP_NUM_THREADS"])
    torch.set_num_threads(threads)
    cfg = orc.medium_config(dropout=0.0)
    p = orc.make_params(cfg, seed=7)
    em, co = orc.simple_infer_masks(R, scale=R / 128.0)
    B = 2
    e = torch.from_numpy(em)[None].repeat(B, 1, 1)
    c = torch.from_numpy(co)[None].repeat(B, 1, 1)
    orc.rollout_infer(p, cfg, e, c, 1)
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < budget_s and n < 20):
        orc.rollout_infer(p, cfg, e, c, 1)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle rollout_infer, UNet_medium {R}x{R}, {B} scenes, fp32, {n} timed frames of {dt:.2f}s"}


def measure_rollout(a, R, batch, frames, warm, world, rank, profile=True, dump=None):
    """BASELINE config 5: `frames` hipGraph replays of one autoregressive frame over `batch` scenes (ml/scripts/simple_infer.py:226-260)."""
    torch.manual_seed(42)
    model = UNet(arch_cfg(a.arch, dropout=0.0, depth=a.depth), precision=a.precision, conv_impl=a.conv_impl).cuda().eval()
    plan = RolloutInferPlan(model.cfg, batch, R, R, params_flat=model.flat_params, dtype=a.precision, conv_impl=a.conv_impl,
                            norm_state=model.flat_norm_state)
    em, co = simple_infer_masks(R, scale=R / 128.0)
    e = torch.from_numpy(em).cuda()[None].repeat(batch, 1, 1)
    c = torch.from_numpy(co).cuda()[None].repeat(batch, 1, 1)
    plan.reset(e, c)
    plan.step(warm)
    barrier(world)
    t0 = time.perf_counter()
    plan.step(frames)
    barrier(world)
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    value = world * batch * frames / dt
    fl = sum(r.flops for r in plan.frame)
    out = {"metric": "rollout frames/sec", "value": round(value, 1), "unit": "scene-frames/s", "n_gpus": world, "steps": frames,
           "warmup": warm, "ms_per_step": round(dt / frames * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
           "config": {"workload": f"{frames}-frame autoregressive rollout, UNet_{a.arch} {R}x{R}, {batch} scenes/GPU, hipGraph-captured frame",
                      "steps_per_s": round(frames / dt, 2), "parallelism": f"replicas{world}",
                      "end_to_end_tflops": round(fl * frames / dt / 1e12, 2), "state_finite": bool(torch.isfinite(plan.cur.tensor).all())}}
    if batch == 1 and R == 128 and world == 1 and a.arch in PUBLISHED_B1_128_FPS:
        # the reference's own published configuration for this metric (other hardware, other runtime: see PUBLISHED_B1_128_FPS)
        pub = PUBLISHED_B1_128_FPS[a.arch]
        out["vs_baseline"] = round(value / pub, 2)
        out["config"]["published"] = f"{1e3 / pub:g} ms/frame = {pub:g} frames/s, UNet_{a.arch} 128x128 batch 1, ORT CUDA EP on RTX 3070 (README.md:159-161)"
    if rank == 0 and profile:
        agg = profile_ops({"frame": plan._frame}, {"frame": plan.frame}, dump)
        conv = agg.get("CONV3X3_MFMA") or agg.get("CONV3X3_REF")
        ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "k_conv3x3_dma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                           "launches_per_step": conv["n"], "avg_launch_ms": round(conv["ms"] / conv["n"], 4)}
        hb = hbm_roofline(agg)
        if hb:
            out["hbm_roofline"] = hb
        out["breakdown_ms_per_step"] = {k: round(v["ms"], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    return out


def measure_train(a, arch, depth, B, R, K, steps, warm, world, rank, loader=True, profile=True, dump=None, headline=False):
    """`steps` timed optimisation steps of one training configuration (after `warm` untimed ones); returns (JSON object, stepper)."""
    st = TrainStepper(B, R, K, a.precision, world, rank, conv_impl=a.conv_impl, arch=arch, depth=depth, loader=loader)
    for _ in range(warm):
        st.step()
    if st.probe is not None:
        st.probe = st.probe.__class__(st.probe.use_events)          # count the exchange over the timed steps only
        st.all_reduce = torch_all_reduce(st.dist, st.probe)
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step()
    barrier(world)
    dt_rank = time.perf_counter() - t0
    dt = dt_rank
    if world > 1:
        tmax = torch.tensor([dt], device="cuda" if st.dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        st.dist.all_reduce(tmax, op=st.dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss = st.plan.loss_dict()
    value = world * B * steps / dt
    out = {"metric": "train samples/sec (K4 UNet 128^2)" if (K == 4 and R == 128 and arch == "medium" and depth is None) else
           f"train samples/sec (K{K} UNet_{arch} {R}^2)",
           "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warm,
           "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": a.precision, "data": "synthetic",
           "config": {"workload": f"UNet_{arch}{'' if depth is None else f' depth {depth}'} K{K} rollout training {R}x{R}, batch {B}/GPU, MSE+divergence+gradient+emitter, "
                                  f"BPTT through all K, clip 1.0 + Adam, dropout {st.model.cfg.dropout}; "
                                  + ("a new batch per step gathered (flip augmentation included) from the HBM-resident store" if st.loader is not None
                                     else "one static resident batch"), "global_batch": B * world,
                      "parallelism": f"dp{world}", "final_loss": round(loss["total"], 6),
                      "hbm_bytes_allocated": st.bytes}}
    flops_step = sum(r.flops for s in st.plan.sections.values() for r in s)
    out["config"]["algorithmic_tflop_per_step"] = round(flops_step * world / 1e12, 4)          # whole job (all ranks), like `value`
    out["config"]["end_to_end_tflops"] = round(flops_step * world * steps / dt / 1e12, 2)
    if world > 1:
        out["exchange"] = exchange_report(st, B, steps, dt_rank, rank, world)
    if rank == 0 and profile:
        agg = profile_ops(st.plan._lowered, st.plan.sections, dump)
        tot = sum(v["ms"] for v in agg.values())
        conv = agg.get("CONV3X3_MFMA") or agg.get("CONV3X3_REF")
        ach = conv["flops"] / (conv["ms"] * 1e-3) / 1e12
        traffic, src = pmc_traffic("k_conv3x3_dma") if headline else (None, None)
        out["roofline"] = {"bound": "mfma", "kernel": "k_conv3x3_dma (forward + data-gradient launches)",
                           "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": src,
                           "launches_per_step": conv["n"], "avg_launch_ms": round(conv["ms"] / conv["n"], 4),
                           "algorithmic_gflop_per_launch": round(conv["flops"] / conv["n"] / 1e9, 3),
                           "algorithmic_bytes_per_launch": round(conv["bytes"] / conv["n"])}
        if headline:
            out["roofline"]["reference_gemm"] = reference_gemm()
        hb = hbm_roofline(agg)
        if hb:
            if headline:
                hb["reference_stream"] = reference_stream()
            out["hbm_roofline"] = hb
        out["breakdown_ms_per_step"] = {k: round(v["ms"], 3) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        out["breakdown_ms_per_step"]["_sum"] = round(tot, 3)
        wg = agg.get("WGRAD_MFMA")
        if wg:
            out["wgrad_tflops"] = round(wg["flops"] / (wg["ms"] * 1e-3) / 1e12, 2)
    return out, st


def exchange_report(st, B, steps, dt_rank, rank, world):
    """What lets a reader of an N > 1 line check that the collective library really saw N ranks and what the exchange cost: backend and
    world size as torch.distributed reports them, every rank's device and own samples/s, the exchange timed alone and the part of it
    that the compute stream had to wait for in the timed steps.  Collective on all ranks; rank 0 keeps the result."""
    dist = st.dist
    me = {"rank": rank, "device_index": torch.cuda.current_device(), "device_name": torch.cuda.get_device_name(),
          "pid": os.getpid(), "samples_per_s": round(B * steps / dt_rank, 2),
          "allreduce_exposed_ms": round(st.probe.exposed_ms() / max(st.probe.steps, 1), 4)}
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    alone = st.exchange_alone_ms()
    g = st.plan.pb.grads.tensor
    sps = [r["samples_per_s"] for r in ranks]
    return {"backend": dist.get_backend(), "library": "RCCL (torch.distributed backend nccl on ROCm)" if dist.get_backend() == "nccl"
            else "gloo through the host (rehearsal mode, FNET_DIST_BACKEND=gloo: ranks may share a device; not a scaling figure)",
            "rccl_world": dist.get_world_size() if dist.get_backend() == "nccl" else None, "world": dist.get_world_size(),
            "buckets": [{"elements": hi - lo, "lane": lane} for _e, (lo, hi), lane in st.plan.grad_buckets if hi > lo],
            "gradient_bytes": int(g.numel() * g.element_size()),
            "allreduce_ms_per_step": round(alone, 4), "allreduce_exposed_ms": max(r["allreduce_exposed_ms"] for r in ranks),
            "rank_samples_per_s_min": min(sps), "rank_samples_per_s_max": max(sps), "ranks": ranks}


C4_LEGS = (("K2 256x256 depth 4", dict(K=2, R=256, depth=4, B=32)), ("K2 512x512 depth 5", dict(K=2, R=512, depth=5, B=8)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=["train", "rollout"], default="train")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--res", type=int, default=None)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--arch", choices=list(ARCHS), default="medium", help="model architecture (default: the headline's UNet_medium)")
    ap.add_argument("--depth", type=int, default=None, help="override the architecture's encoder depth (BASELINE config 4: deeper encoders at 256^2 / 512^2)")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--conv-impl", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true", help="skip the 600-frame 256^2 x 64-scene rollout leg of the default training line")
    ap.add_argument("--rollout-frames", type=int, default=600, help="frames of the default line's rollout leg (BASELINE config 5: 600)")
    ap.add_argument("--no-config4", action="store_true", help="skip the two short K2 256^2 / 512^2 legs (BASELINE config 4) of the default line")
    ap.add_argument("--static-batch", action="store_true", help="reuse one resident batch instead of gathering a new one per step from the HBM-resident store")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--dump-ops", default=None, help="write per-op HIP-event timings of the instrumented step to this CSV")
    ap.add_argument("--spawn-timeout", type=float, default=1500.0, help="time limit (s) for the ranks this script starts itself")
    a = ap.parse_args()
    if launch.needs_spawn(a.gpus):
        # un-launched N-rank run: this process only starts the ranks (it has made no torch.cuda / HIP call and makes none)
        sys.exit(launch.main_spawn(os.path.abspath(__file__), sys.argv[1:], a.gpus, a.spawn_timeout))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); fluidnet_amd has no CPU path")
    rank, world, local = dist_setup(a.gpus)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: a launcher must start exactly --gpus ranks")
    rt.lib()

    if a.mode == "train":
        R = a.res or 128
        steps = a.steps if a.steps is not None else 10
        warm = a.warmup if a.warmup is not None else 3
        default_line = a.k == 4 and R == 128 and a.batch == 64 and a.precision == "bf16" and a.arch == "medium" and a.depth is None
        out, st = measure_train(a, a.arch, a.depth, a.batch, R, a.k, steps, warm, world, rank, loader=not a.static_batch,
                                profile=not a.no_profile, dump=a.dump_ops, headline=default_line)
        del st
        torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not a.no_cpu_baseline and a.arch == "medium":
            out["cpu_baseline"] = cpu_baseline_train(a.k, R)
        if default_line and not a.no_rollout:
            # the second half of BASELINE.json's metric ("rollout frames/sec", config 5) in the same line: 64 scenes at 256^2, the
            # hipGraph-captured frame replayed 600 times after 10 warm-up frames
            ro = measure_rollout(a, 256, 64, a.rollout_frames, 10, world, rank, profile=not a.no_profile)
            if rank == 0:
                out["rollout"] = {"metric": ro["metric"], "value": ro["value"], "unit": ro["unit"], "frames": ro["steps"], "scenes_per_gpu": 64,
                                  "resolution": 256, "ms_per_frame_step": ro["ms_per_step"], "state_finite": ro["config"]["state_finite"],
                                  "conv_tflops": ro.get("roofline", {}).get("achieved"), "conv_frac": ro.get("roofline", {}).get("frac"),
                                  "hbm_gb_per_s": ro.get("hbm_roofline", {}).get("achieved"), "hbm_frac": ro.get("hbm_roofline", {}).get("frac")}
            torch.cuda.empty_cache()
        if default_line and not a.no_config4:
            # BASELINE config 4 under the same clock: K2 at 256^2 / 512^2 with the deeper encoders, 3 warm-up + 5 timed steps each
            legs = {}
            for name, c in C4_LEGS:
                o4, st4 = measure_train(a, "medium", c["depth"], c["B"], c["R"], c["K"], 5, 3, world, rank, loader=True,
                                        profile=not a.no_profile)
                del st4
                torch.cuda.empty_cache()
                if rank == 0:
                    legs[name] = {"metric": o4["metric"], "value": o4["value"], "unit": o4["unit"], "ms_per_step": o4["ms_per_step"],
                                  "steps": 5, "warmup": 3, "batch_per_gpu": c["B"], "workload": o4["config"]["workload"],
                                  "end_to_end_tflops": o4["config"]["end_to_end_tflops"],
                                  "conv_tflops": o4.get("roofline", {}).get("achieved"), "conv_frac": o4.get("roofline", {}).get("frac"),
                                  "hbm_gb_per_s": o4.get("hbm_roofline", {}).get("achieved"), "final_loss": o4["config"]["final_loss"]}
            if rank == 0:
                out["config4"] = legs
    else:
        R = a.res or 256
        frames = a.steps if a.steps is not None else 600
        warm = a.warmup if a.warmup is not None else 10
        out = measure_rollout(a, R, a.batch, frames, warm, world, rank, profile=not a.no_profile, dump=a.dump_ops)
        if rank == 0 and world == 1 and not a.no_cpu_baseline and a.arch == "medium":
            out["cpu_baseline"] = cpu_baseline_rollout(R)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
