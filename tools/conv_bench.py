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
LAYERS = [  # (H, Cin, Cout): the stride-1 classes of a UNet_medium K4 B64 128^2 step
    (128, 64, 64), (128, 32, 32), (128, 128, 32), (128, 32, 128), (128, 32, 64), (64, 128, 128), (64, 64, 64), (64, 256, 64), (64, 64, 256),
    (32, 256, 256), (32, 128, 128), (32, 512, 128), (32, 128, 512), (16, 256, 256),
]


def build():
    from fluidnet_amd import build as b
    b.build(verbose=False)
    for v in ABL:
        d = os.path.join(b.LIBDIR, f"abl{v}")
        os.makedirs(d, exist_ok=True)
        obj = os.path.join(d, "fnet_conv_mfma.o")
        # variant 9 = in-kernel cycle stamps of the convolution; 11-13 = weight-gradient ablations (FNET_WG_ABLATE 1-3)
        flag = "-DFNET_CONV_STAMPS=1" if v == 9 else (f"-DFNET_WG_ABLATE={v - 10}" if v > 10 else f"-DFNET_CONV_ABLATE={v}")
        subprocess.run([b._hipcc(), *b.FLAGS, flag, "-c", os.path.join(b.CSRC, "fnet_conv_mfma.hip"), "-o", obj], check=True)
        objs = [obj if s == "fnet_conv_mfma.hip" else os.path.join(b.LIBDIR, s.replace(".hip", ".o")) for s in b.SOURCES]
        subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libfluidnet_hip.so"), *objs], check=True)
        print("built", d)


def run(B=64, reps=20, layers=None):
    import torch
    from fluidnet_amd.engine import runtime as rt
    from fluidnet_amd.engine.plan import OpRec, PlanBuilder, PV, Ref, lower
    from fluidnet_amd.models.unet import UNetConfig
    cfg = UNetConfig(in_channels=6, base_channels=32, depth=3, norm="instance", act="gelu", upsample="bilinear", padding_mode="replicate",
                     use_residual=True)
    print("lib:", rt.LIB_PATH)
    for (H, cin, cout) in (layers or LAYERS):
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
        pb.op(ops, "CONV3X3", [npix, H + 2, cin, cin, 0, cout, cout, 0, rt.CONV_MFMA, 0], p=[Ref(src), Ref(wk), None, Ref(dst), Ref(sc)], tag="conv")
        pb.op(ops, "WGRAD", [npix, H + 2, cin, cin, 0, cout, cout, 0, rt.CONV_MFMA, cin, B, H, 2048], p=[Ref(src), Ref(dst), Ref(dwk), Ref(slab)], tag="wgrad")
        arr = lower(pb, ops)
        fl = 2.0 * B * H * H * 9 * cin * cout
        out = []
        for k, name in enumerate(("conv", "wgrad")):
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
            if name == "conv" and os.environ.get("FNET_STAMPS"):
                # library built with -DFNET_CONV_STAMPS (tools/conv_bench.py build with STAMPS=1): cycles of wave 0 per block
                sc.tensor.zero_()
                rt.run_ops(one)
                torch.cuda.synchronize()
                c = sc.tensor[:16].view(torch.int64).cpu().numpy()
                nb = max(1, int(c[4]))
                out.append("cycles/block barrier-wait %d (of which at a tile's first stage %d) taps %d epilogue %d total %d (blocks %d)" % (c[0] / nb, c[5] / nb, c[1] / nb, c[2] / nb, c[3] / nb, nb))
        print(f"B={B:3d} H={H:4d} {cin:4d}->{cout:4d}: " + " | ".join(out), flush=True)


def run_s2(B=64, reps=20):
    """The three stride-2 convolutions of UNet_medium at 128^2 in their 2x2 space-to-depth form (forward, data gradient, weight gradient)
    beside the full-resolution 3x3 launches they replace."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from fluidnet_amd.engine import runtime as rt
    from fluidnet_amd.engine.plan import lower
    from test_gpu_conv import _mk, _mk_s2
    for (h, C) in ((64, 64), (32, 128), (16, 256)):
        pb, ops = _mk_s2(B, h, h, C, C, 2 * C, C)
        pb2, ops2 = _mk(B, 2 * h, 2 * h, C, C, 2 * C, C, C, 0, "bf16", "mfma")
        out = []
        for pbx, opsx, names in ((pb, ops, ("s2 fwd", "s2 dgrad", "s2 wgrad")), (pb2, ops2, ("3x3 full-res conv", "3x3 full-res wgrad"))):
            for b_ in pbx.bufs:
                b_.tensor = b_.tensor.cuda()
            arr = lower(pbx, opsx)
            for k, name in enumerate(names):
                one = arr[k:k + 1]
                for _ in range(3):
                    rt.run_ops(one)
                t = rt.Timer()
                t.start()
                for _ in range(reps):
                    rt.run_ops(one)
                t.stop()
                out.append(f"{name} {t.elapsed_ms() / reps * 1e3:7.1f} us")
        print(f"B={B} half-res {h}x{h} C={C}: " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "s2":
        run_s2()
    elif sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "tail":
        # how much of a launch is the partial last round of the persistent grid?  the same layer at batch sizes that give a whole number of
        # rounds (items / blocks) and at B = 64
        for (H, cin, cout, bs) in ((32, 256, 256, (56, 64, 85)), (64, 128, 128, (60, 64, 75)), (128, 64, 64, (62, 64, 70)), (32, 128, 128, (56, 64)),
                                   (64, 64, 64, (45, 64))):
            for b_ in bs:
                run(B=b_, layers=[(H, cin, cout)])
    else:
        run()
