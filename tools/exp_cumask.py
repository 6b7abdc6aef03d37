#!/usr/bin/env python3
"""Experiment: do an MFMA-bound convolution stream and an HBM-bound norm / activation stream run side by side at (nearly) full speed when
each gets its own share of the CUs (hipExtStreamCreateWithCUMask)?  The idea under test: HBM-bound kernels need far fewer than 256 CUs
to saturate HBM, so two half-batch chains whose convolutions run on ~160 CUs while the other chain's elementwise passes run on ~96 could
overlap where the single-stream step serialises them.

    python tools/exp_cumask.py [ew_cus_per_32]      (GPU box)   prints: each stream alone (unmasked / masked) and both together
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidnet_amd.engine import runtime as rt  # noqa: E402
from fluidnet_amd.engine.plan import PlanBuilder, Ref, lower  # noqa: E402
from fluidnet_amd.models.unet import UNetConfig  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(pred):
    """stream restricted to the CUs i (0..255) with pred(i)"""
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return s.value, sum(bin(w).count("1") for w in words)


def main():
    ew_share = int(sys.argv[1]) if len(sys.argv) > 1 else 12           # CUs of every 32 given to the elementwise stream
    B = 32
    cfg = UNetConfig(in_channels=6, base_channels=32, depth=3, norm="instance", act="gelu", upsample="bilinear", padding_mode="replicate", use_residual=True)
    pb = PlanBuilder(cfg, B, 128, 128, dtype="bf16", conv_impl="mfma")
    pb.bufs.clear()
    conv_ops, ew_ops = [], []
    # convolution stream: the three big layer classes, a few launches each (B = 32: one half-batch chain)
    for (H, c) in ((128, 64), (64, 128), (32, 256)):
        src = pb.new_buf(f"src{H}", (B, H + 2, H + 2, c), "T")
        wk = pb.new_buf(f"wk{H}", (9, c, c), "T")
        dst = pb.new_buf(f"dst{H}", (B, H + 2, H + 2, c), "T")
        for _ in range(4):
            pb.op(conv_ops, "CONV3X3", [B * (H + 2) * (H + 2), H + 2, c, c, 0, c, c, 0, rt.CONV_MFMA, 0], p=[Ref(src), Ref(wk), None, Ref(dst), Ref(pb.stats_scratch)])
    # elementwise stream: statistics + apply + the two adjoint passes on the same classes
    gam = {}
    for (H, c) in ((128, 64), (64, 128), (32, 256)):
        bufs = {n: pb.new_buf(f"{n}{H}", (B, H + 2, H + 2, c), "T") for n in ("raw", "act", "g", "draw")}
        st = pb.new_buf(f"stats{H}", (B, c, 2), "f32")
        S = pb.new_buf(f"S{H}", (B, c, 2), "f32")
        ga = pb.new_buf(f"gamma{H}", (c,), "f32")
        be = pb.new_buf(f"beta{H}", (c,), "f32")
        gam[H] = st
        for _ in range(2):
            pb.op(ew_ops, "IN_STATS", [B, H, H, c, c, 0, pb.stats_scratch.numel], f=[1e-5], p=[Ref(bufs["raw"]), Ref(st), Ref(pb.stats_scratch)])
            pb.op(ew_ops, "APPLY", [B, H, H, c, c, 0, 0, 0, c, 0, rt.ACT["gelu"], rt.PAD["replicate"], 3, 1234], f=[0.1],
                  p=[Ref(bufs["raw"]), Ref(st), Ref(ga), Ref(be), None, Ref(bufs["act"])])
            pb.op(ew_ops, "INACT_BWD_REDUCE", [B, H, H, c, c, 0, 0, 0, c, 0, c, 0, 1, 0, rt.ACT["gelu"], rt.PAD["replicate"], 3, 1234], f=[0.1],
                  p=[Ref(bufs["g"]), None, Ref(bufs["raw"]), Ref(st), Ref(ga), Ref(be), None, Ref(S), None, None])
            pb.op(ew_ops, "INACT_BWD_APPLY", [B, H, H, c, c, 0, c, 0, c, 0, rt.ACT["gelu"], 3, 1234, 1, rt.PAD["replicate"]], f=[0.1],
                  p=[Ref(bufs["g"]), Ref(bufs["raw"]), Ref(st), Ref(ga), Ref(be), Ref(S), Ref(bufs["draw"])])
    for b_ in list(pb.bufs) + [pb.stats_scratch]:
        if b_.tensor is None:
            b_.tensor = (torch.randn(b_.shape, device="cuda") * 0.5).to(torch.bfloat16 if b_.kind == "T" else torch.float32)
    for st in gam.values():
        st.tensor[..., 1] = 1.0
    ca, ea = lower(pb, conv_ops), lower(pb, ew_ops)
    rt.knob = getattr(rt, "set_knob")
    rt.set_knob("sa_minitems", 1 << 20)       # no fused small-image launches: keep the four kernels separate and comparable

    def run(streams, reps=6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for arr, s in streams:
                rt.run_ops(arr, s)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    s_full1 = torch.cuda.Stream().cuda_stream
    s_full2 = torch.cuda.Stream().cuda_stream
    s_conv, n_conv = masked_stream(lambda i: (i % 32) >= ew_share)
    s_ew, n_ew = masked_stream(lambda i: (i % 32) < ew_share)
    for arr, s in ((ca, s_full1), (ea, s_full2), (ca, s_conv), (ea, s_ew)):
        rt.run_ops(arr, s)
    res = {
        "conv_alone_256": run([(ca, s_full1)]), "ew_alone_256": run([(ea, s_full2)]),
        f"conv_alone_{n_conv}": run([(ca, s_conv)]), f"ew_alone_{n_ew}": run([(ea, s_ew)]),
        "both_unmasked_two_streams": run([(ca, s_full1), (ea, s_full2)]),
        f"both_masked_{n_conv}+{n_ew}": run([(ca, s_conv), (ea, s_ew)]),
    }
    res["sequential_256"] = res["conv_alone_256"] + res["ew_alone_256"]
    print({k: round(v, 3) for k, v in res.items()})


if __name__ == "__main__":
    main()
