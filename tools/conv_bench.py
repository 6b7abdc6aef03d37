#!/usr/bin/env python3
"""Single-op micro-benchmark for the MFMA convolution / weight-gradient kernels (development tool).

    python tools/conv_bench.py build            # CPU box: build ablation variants of the library into fluidnet_amd/lib/abl<N>/
    python tools/conv_bench.py run [variant]    # GPU box: time the layer shapes of UNet_medium K4 B64 128^2 with one variant
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ABL = tuple(int(x) for x in os.environ.get('ABL', '1,2,3').split(','))
LAYERS = [  # (H, Cin, Cout)
    (128, 64, 64), (128, 32, 64), (128, 128, 32), (128, 32, 32), (64, 128, 128), (64, 256, 64), (32, 256, 256), (32, 512, 128),
    (16, 256, 256),
]


def build():
    from fluidnet_amd import build as b
    b.build(verbose=False)
    for v in ABL:
        d = os.path.join(b.LIBDIR, f"abl{v}")
        os.makedirs(d, exist_ok=True)
        obj = os.path.join(d, "fnet_conv_mfma.o")
        flag = "-DFNET_CONV_STAMPS=1" if v == 9 else f"-DFNET_CONV_ABLATE={v}"       # variant 9 = in-kernel cycle stamps
        subprocess.run([b._hipcc(), *b.FLAGS, flag, "-c", os.path.join(b.CSRC, "fnet_conv_mfma.hip"), "-o", obj], check=True)
        objs = [obj if s == "fnet_conv_mfma.hip" else os.path.join(b.LIBDIR, s.replace(".hip", ".o")) for s in b.SOURCES]
        subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libfluidnet_hip.so"), *objs], check=True)
        print("built", d)


def run(B=64, reps=20):
    import torch
    from fluidnet_amd.engine import runtime as rt
    from fluidnet_amd.engine.plan import OpRec, PlanBuilder, PV, Ref, lower
    from fluidnet_amd.models.unet import UNetConfig
    cfg = UNetConfig(in_channels=6, base_channels=32, depth=3, norm="instance", act="gelu", upsample="bilinear", padding_mode="replicate",
                     use_residual=True)
    print("lib:", rt.LIB_PATH)
    for (H, cin, cout) in LAYERS:
        pb = PlanBuilder(cfg, B, 128, 128, dtype="bf16", conv_impl="mfma")
        pb.bufs.clear()
        src = pb.new_buf("src", (B, H + 2, H + 2, cin), "T")
        wk = pb.new_buf("wk", (9, cout, cin), "T")
        dst = pb.new_buf("dst", (B, H + 2, H + 2, cout), "T")
        dwk = pb.new_buf("dwk", (9, cout, cin), "f32")
        slab = pb.new_buf("slab", (2048 * 9 * 32 * 32,), "f32")
        sc = pb.stats_scratch
        for b_ in (src, wk, dst, dwk, slab, sc):
            b_.tensor = torch.randn(b_.shape, device="cuda").to(torch.bfloat16 if b_.kind == "T" else torch.float32)
        npix = B * (H + 2) * (H + 2)
        ops = []
        pb.op(ops, "CONV3X3", [npix, H + 2, cin, cin, 0, cout, cout, 0, rt.CONV_MFMA, 0, H], p=[Ref(src), Ref(wk), None, Ref(dst), None], tag="conv")
        pb.op(ops, "CONV3X3", [npix, H + 2, cin, cin, 0, cout, cout, 0, rt.CONV_MFMA, sc.numel, H], p=[Ref(src), Ref(wk), None, Ref(dst), Ref(sc)], tag="conv+stats")
        pb.op(ops, "WGRAD", [npix, H + 2, cin, cin, 0, cout, cout, 0, rt.CONV_MFMA, cin, B, H, 2048], p=[Ref(src), Ref(dst), Ref(dwk), Ref(slab)], tag="wgrad")
        arr = lower(pb, ops)
        fl = 2.0 * B * H * H * 9 * cin * cout
        out = []
        for k, name in enumerate(("conv", "conv+stats", "wgrad")):
            one = arr[k:k + 1]
            for _ in range(3):
                rt.run_ops(one)
            t = rt.Timer()
            t.start()
            for _ in range(reps):
                rt.run_ops(one)
            t.stop()
            ms = t.elapsed_ms() / reps
            out.append(f"{name} {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF")
            if name == "conv+stats" and os.environ.get("FNET_STAMPS"):
                # library built with -DFNET_CONV_STAMPS (tools/conv_bench.py build with STAMPS=1): cycles of wave 0 per block
                sc.tensor.zero_()
                rt.run_ops(one)
                torch.cuda.synchronize()
                c = sc.tensor[:16].view(torch.int64).cpu().numpy()
                nb = max(1, int(c[4]))
                out.append("cycles/block barrier-wait %d taps %d epilogue %d total %d (blocks %d)" % (c[0] / nb, c[1] / nb, c[2] / nb, c[3] / nb, nb))
        print(f"H={H:4d} {cin:4d}->{cout:4d}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run()
