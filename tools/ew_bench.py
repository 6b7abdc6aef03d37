#!/usr/bin/env python3
"""Single-op micro-benchmark for the norm/activation kernels (development tool): times APPLY, INACT_BWD_REDUCE,
INACT_BWD_APPLY and IN_STATS on the layer shapes of UNet_medium K4 B64 128^2, rotating through enough buffer sets that a
repetition never finds its operands in the 256 MiB Infinity Cache (cold) or re-using one set (warm).

    python tools/ew_bench.py            # GPU box;  FNET_CG_BLOCKS=<n> changes the launch's block target
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

LAYERS = [(128, 32), (128, 64), (64, 64), (64, 128), (32, 128), (16, 256)]  # (H, C)
ACTN = os.environ.get("FNET_EW_ACT", "gelu")          # activation of the timed kernels
if os.environ.get("FNET_EW_LAYERS"):                     # e.g. "128x64,64x128"
    LAYERS = [tuple(int(v) for v in t.split("x")) for t in os.environ["FNET_EW_LAYERS"].split(",")]


def run(B=64, reps=12):
    import torch
    from fluidnet_amd.engine import runtime as rt
    from fluidnet_amd.engine.plan import PlanBuilder, Ref, lower
    from fluidnet_amd.models.unet import UNetConfig
    cfg = UNetConfig(in_channels=6, base_channels=32, depth=3, norm="instance", act="gelu", upsample="bilinear", padding_mode="replicate",
                     use_residual=True)
    print("lib:", rt.LIB_PATH, "FNET_CG_BLOCKS=", os.environ.get("FNET_CG_BLOCKS"), "act", ACTN)
    for (H, C) in LAYERS:
        tensor_mb = B * (H + 2) * (H + 2) * C * 2 / 1e6
        for mode in ("warm", "cold"):
            nset = 1 if mode == "warm" else max(2, int(600 / (4 * tensor_mb)) + 1)
            pb = PlanBuilder(cfg, B, 128, 128, dtype="bf16", conv_impl="mfma")
            pb.bufs.clear()
            sets = []
            for s in range(nset):
                bufs = {n: pb.new_buf(f"{n}{s}", (B, H + 2, H + 2, C), "T") for n in ("raw", "act", "g", "G", "draw")}
                bufs["stats"] = pb.new_buf(f"stats{s}", (B, C, 2), "f32")
                bufs["S"] = pb.new_buf(f"S{s}", (B, C, 2), "f32")
                sets.append(bufs)
            gamma = pb.new_buf("gamma", (C,), "f32")
            beta = pb.new_buf("beta", (C,), "f32")
            sc = pb.stats_scratch
            for b_ in list(pb.bufs) + [sc]:
                if b_.tensor is None:
                    b_.tensor = (torch.randn(b_.shape, device="cuda") * 0.5).to(torch.bfloat16 if b_.kind == "T" else torch.float32)
            for bufs in sets:
                bufs["stats"].tensor[..., 1] = 1.0
            per = {}
            for bufs in sets:
                ops = []
                pb.op(ops, "IN_STATS", [B, H, H, C, C, 0, sc.numel], f=[1e-5], p=[Ref(bufs["raw"]), Ref(bufs["stats"]), Ref(sc)], tag="stats")
                pb.op(ops, "APPLY", [B, H, H, C, C, 0, 0, 0, C, 0, rt.ACT[ACTN], rt.PAD["replicate"], 3, 1234], f=[0.1],
                      p=[Ref(bufs["raw"]), Ref(bufs["stats"]), Ref(gamma), Ref(beta), None, Ref(bufs["act"])], tag="apply")
                pb.op(ops, "INACT_BWD_REDUCE", [B, H, H, C, C, 0, 0, 0, C, 0, C, 0, 1, 0, rt.ACT[ACTN], rt.PAD["replicate"], 3, 1234], f=[0.1],
                      p=[Ref(bufs["g"]), None, Ref(bufs["raw"]), Ref(bufs["stats"]), Ref(gamma), Ref(beta), Ref(bufs["G"]), Ref(bufs["S"]), None, None],
                      tag="bwd_reduce")
                pb.op(ops, "INACT_BWD_APPLY", [B, H, H, C, C, 0, C, 0, C, 0, rt.ACT[ACTN], 3, 1234], f=[0.1],
                      p=[Ref(bufs["G"]), Ref(bufs["raw"]), Ref(bufs["stats"]), Ref(gamma), Ref(beta), Ref(bufs["S"]), Ref(bufs["draw"])], tag="bwd_apply")
                # the single-source form of pass 1 (no G written: the common case in the training plan)
                pb.op(ops, "INACT_BWD_REDUCE", [B, H, H, C, C, 0, 0, 0, C, 0, C, 0, 1, 0, rt.ACT[ACTN], rt.PAD["replicate"], 3, 1234], f=[0.1],
                      p=[Ref(bufs["g"]), None, Ref(bufs["raw"]), Ref(bufs["stats"]), Ref(gamma), Ref(beta), None, Ref(bufs["S"]), None, None],
                      tag="bwd_reduce1")
                arr = lower(pb, ops)
                for k, name in enumerate(("stats", "apply", "bwd_reduce", "bwd_apply", "bwd_reduce1")):
                    per.setdefault(name, []).append(arr[k:k + 1])
            traffic = {"stats": 1, "apply": 2, "bwd_reduce": 3, "bwd_apply": 3, "bwd_reduce1": 2}
            out = []
            for name, arrs in per.items():
                for a in arrs:
                    rt.run_ops(a)
                t = rt.Timer()
                t.start()
                for r in range(reps):
                    rt.run_ops(arrs[r % len(arrs)])
                t.stop()
                us = t.elapsed_ms() / reps * 1e3
                out.append(f"{name} {us:6.1f} us {traffic[name] * tensor_mb / us:5.2f} TB/s")
            print(f"H={H:4d} C={C:4d} ({tensor_mb:6.1f} MB/tensor) {mode}: " + " | ".join(out), flush=True)
            del sets, pb
            torch.cuda.empty_cache()


if __name__ == "__main__":
    run()
