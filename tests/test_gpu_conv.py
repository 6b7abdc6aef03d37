"""MFMA convolution / weight-gradient kernels on multi-tile shapes (several pixel tiles, N tiles, channel slices of wider
buffers, partial tiles, images straddling tile boundaries) against the CPU twin, through the C ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from fluidnet_amd.engine import runtime as rt
from fluidnet_amd.engine.plan import OpRec, PlanBuilder, PV, Ref, lower
from helpers import tiny_cfg
from lockstep import lockstep

pytestmark = pytest.mark.gpu

SHAPES = [
    # B, H,  W,  Cin, Cout, ldc_in, coff_in, ldc_out, coff_out
    (2, 64, 64, 64, 128, 64, 0, 128, 0),
    (3, 20, 12, 32, 32, 96, 32, 64, 16),
    (1, 128, 128, 16, 32, 16, 0, 32, 0),
    (2, 16, 16, 256, 64, 256, 0, 64, 0),
    (2, 30, 34, 48, 24, 48, 0, 24, 0),
    (1, 8, 8, 128, 256, 256, 128, 256, 0),
    (1, 6, 520, 32, 128, 32, 0, 128, 0),       # wide image: the 512-pixel tile's stage buffers would not fit LDS -> small-tile fallback
]


def _mk(B, H, W, cin, cout, ldi, coi, ldo, coo, dtype, impl):
    pb = PlanBuilder(tiny_cfg(base_channels=16), 1, 16, 16, dtype=dtype, conv_impl=impl)
    pb.bufs.clear()
    pb.B = B
    g = torch.Generator().manual_seed(B * 1000 + H + cin)
    T = torch.bfloat16 if dtype == "bf16" else torch.float32
    src = pb.new_buf("src", (B, H + 2, W + 2, ldi), "T")
    src.tensor = torch.randn(src.shape, generator=g).to(T)
    wk = pb.new_buf("wk", (9, cout, cin), "T")
    wk.tensor = (torch.randn(wk.shape, generator=g) / (3 * cin ** 0.5)).to(T)
    bias = pb.new_buf("bias", (cout,), "f32")
    bias.tensor = torch.randn(cout, generator=g)
    dst = pb.new_buf("dst", (B, H + 2, W + 2, ldo), "T")
    dst.tensor = torch.full(dst.shape, 7.0).to(T)
    dout = pb.new_buf("dout", (B, H + 2, W + 2, ldo), "T")
    d = torch.randn(dout.shape, generator=g)
    d[:, 0] = 0; d[:, -1] = 0; d[:, :, 0] = 0; d[:, :, -1] = 0
    dout.tensor = d.to(T)
    dwk = pb.new_buf("dwk", (9, cout, cin), "f32")
    dwk.tensor = torch.randn(dwk.shape, generator=g) * 0.01
    ops = []
    pb.e_conv(ops, PV(src, cin, coi), wk, Ref(bias), PV(dst, cout, coo), cin, cout, tag="conv")
    npix = B * (H + 2) * (W + 2)
    slab = pb.new_buf("slab", (2048 * 9 * 32 * 32,), "f32")
    slab.tensor = torch.zeros(slab.shape)
    pb.op(ops, "WGRAD", [npix, W + 2, cin, ldi, coi, cout, ldo, coo, pb.impl_for(cin, cout), cin, B, H, 2048],
          p=[Ref(src), Ref(dout), Ref(dwk), Ref(slab) if impl == "mfma" else None], tag="wgrad")
    return pb, ops


@pytest.mark.parametrize("shape", SHAPES, ids=[f"B{s[0]}_{s[1]}x{s[2]}_{s[3]}to{s[4]}" for s in SHAPES])
def test_mfma_conv_and_wgrad(shape):
    pb, ops = _mk(*shape, "bf16", "mfma")
    failures = lockstep(pb, ops)
    assert not failures, "\n".join(failures)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_ref_conv_and_wgrad(dtype):
    pb, ops = _mk(2, 20, 12, 32, 40, 96, 32, 64, 16, dtype, "ref")
    failures = lockstep(pb, ops)
    assert not failures, "\n".join(failures)


def test_big_tile_kernels_on_every_shape():
    """The 512-pixel x 64/128-cout tile kernels (8 waves, one block per CU) are chosen by the launcher only for layers with
    >= 3 rounds of big tiles; FNET_CONV_TILE=big (read once per process) forces them, so the shapes above -- partial tiles,
    channel slices, images straddling tiles -- run again in a child process with that setting."""
    if os.environ.get("FNET_CONV_TILE"):
        pytest.skip("already inside the forced-tile child")
    env = dict(os.environ, FNET_CONV_TILE="big")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_mfma_conv_and_wgrad",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert f"{len(SHAPES)} passed" in r.stdout, r.stdout[-2000:]


def test_big_tile_selected_at_full_size():
    # 24 images of 128x128 with 64 -> 128 channels: 793 big tiles >= 3 x 256 CUs, so the launcher takes the 512 x 128 kernel itself
    pb, ops = _mk(24, 128, 128, 64, 128, 64, 0, 128, 0, "bf16", "mfma")
    failures = lockstep(pb, ops[:1])
    assert not failures, "\n".join(failures)


# Weight gradient with several 64-pixel chunks per block (the three-deep DMA pipeline: prologue of two chunks, steady state, drain) and a
# ragged last block: chunks = B * ceil(W/32) * ceil(H/2) for W > 16, one block per CU (256) over the launch
WG_PIPE_SHAPES = [
    (20, 32, 32, 32, 32, 32, 0, 32, 0),      # 320 chunks -> 2 per block
    (40, 32, 32, 64, 64, 64, 0, 64, 0),      # 640 chunks -> 3 per block, 214 blocks, the last with 1
    (5, 128, 128, 32, 64, 32, 0, 64, 0),     # 1280 chunks -> 5 per block
    (9, 64, 64, 64, 32, 128, 64, 32, 0),     # 576 chunks -> 3 per block; channel slice of a wider source
]


@pytest.mark.parametrize("px", ["64", "128"])
@pytest.mark.parametrize("shape", WG_PIPE_SHAPES, ids=[f"B{s[0]}_{s[1]}x{s[2]}_{s[3]}to{s[4]}" for s in WG_PIPE_SHAPES])
def test_wgrad_multi_chunk_pipeline(shape, px, knobs):
    # both chunk sizes of k_wgrad_dma (64 and 128 pixels), forced through the wg_px knob
    knobs("wg_px", px)
    pb, ops = _mk(*shape, "bf16", "mfma")
    failures = lockstep(pb, ops[1:])
    assert not failures, "\n".join(failures)


@pytest.mark.parametrize("px", ["64", "128"])
def test_wgrad_chunk_sizes_on_every_shape(px, knobs):
    """The weight gradient of every shape above (ragged images, channel slices, W <= 16 and W > 16 chunk shapes, the 2x2 stride-2 form)
    with the chunk size forced, whatever the launcher would pick by itself."""
    knobs("wg_px", px)
    for shape in SHAPES:
        pb, ops = _mk(*shape, "bf16", "mfma")
        failures = lockstep(pb, ops[1:])
        assert not failures, f"{shape}: " + "\n".join(failures)
    for shape in S2_SHAPES:
        pb, ops = _mk_s2(*shape)
        failures = lockstep(pb, ops[2:])
        assert not failures, f"s2 {shape}: " + "\n".join(failures)


@pytest.mark.parametrize("xcd", ["0", "1"])
def test_wgrad_chunk_dealing_contiguous_and_per_xcd(xcd, knobs):
    """k_wgrad_dma deals its chunks to blocks either as contiguous ranges or interleaved per XCD (knob wg_xcd, default on where the
    block count along the chunk axis is a multiple of 8): every chunk exactly once either way, ragged last ranges and blocks without a
    chunk included."""
    knobs("wg_xcd", xcd)
    for shape in WG_PIPE_SHAPES + [(64, 16, 16, 64, 64, 64, 0, 64, 0), (3, 48, 40, 128, 64, 128, 0, 64, 0)]:
        pb, ops = _mk(*shape, "bf16", "mfma")
        failures = lockstep(pb, ops[1:])
        assert not failures, f"{shape}: " + "\n".join(failures)
    for shape in S2_SHAPES[:3]:
        pb, ops = _mk_s2(*shape)
        failures = lockstep(pb, ops[2:])
        assert not failures, f"s2 {shape}: " + "\n".join(failures)


def test_wgrad_128_pixel_chunks_selected_at_full_size():
    # 16 images of 128x128, 64 -> 64 channels: 2048 chunks of 128 pixels >= 3 per block, so the launcher takes the 128-pixel form itself
    assert "FNET_WG_PX" not in os.environ
    pb, ops = _mk(16, 128, 128, 64, 64, 64, 0, 64, 0, "bf16", "mfma")
    failures = lockstep(pb, ops[1:])
    assert not failures, "\n".join(failures)


S2_SHAPES = [
    # B, h,  w,  C (per phase), Cout, ldc_full, coff_full     (full-resolution image 2h x 2w)
    (2, 16, 16, 32, 32, 64, 32),        # the cat-buffer skip slice of the training plan
    (3, 20, 12, 64, 64, 64, 0),         # several images per tile, partial tiles
    (1, 64, 64, 32, 64, 96, 32),
    (2, 8, 8, 128, 128, 256, 128),
    (1, 6, 130, 32, 32, 32, 0),         # wide rows
    (5, 3, 5, 64, 128, 64, 0),          # tiny images: a tile spans several of them
]


def _mk_s2(B, h, w, C, cout, ldf, cof):
    """Stride-2 convolution in its 2x2 space-to-depth form: forward (CONV3X3 mode 1), data gradient (mode 2), weight gradient (WGRAD mode 1)."""
    pb = PlanBuilder(tiny_cfg(base_channels=16), 1, 16, 16, dtype="bf16", conv_impl="mfma")
    pb.bufs.clear()
    pb.B = B
    g = torch.Generator().manual_seed(B * 1000 + h + C)
    T = torch.bfloat16
    full = pb.new_buf("full", (B, 2 * h + 2, 2 * w + 2, ldf), "T")
    full.tensor = torch.randn(full.shape, generator=g).to(T)
    wf = pb.new_buf("wf", (4, cout, 4 * C), "T")
    wf.tensor = (torch.randn(wf.shape, generator=g) / (4 * C) ** 0.5).to(T)
    wd = pb.new_buf("wd", (4, 4 * C, (cout + 15) // 16 * 16), "T")
    wd.tensor = (torch.randn(wd.shape, generator=g) / (4 * cout) ** 0.5).to(T)
    bias = pb.new_buf("bias", (cout,), "f32")
    bias.tensor = torch.randn(cout, generator=g)
    half = pb.new_buf("half", (B, h + 2, w + 2, cout), "T")
    half.tensor = torch.full(half.shape, 7.0).to(T)
    dout = pb.new_buf("dout", (B, h + 2, w + 2, cout), "T")
    d = torch.randn(dout.shape, generator=g)
    d[:, 0] = 0; d[:, -1] = 0; d[:, :, 0] = 0; d[:, :, -1] = 0
    dout.tensor = d.to(T)
    dfull = pb.new_buf("dfull", (B, 2 * h + 2, 2 * w + 2, ldf), "T")
    dfull.tensor = torch.full(dfull.shape, 5.0).to(T)
    dwk = pb.new_buf("dwk", (4, cout, 4 * C), "f32")
    dwk.tensor = torch.randn(dwk.shape, generator=g) * 0.01
    slab = pb.new_buf("slab", (2048 * 9 * 32 * 32,), "f32")
    slab.tensor = torch.zeros(slab.shape)
    ops = []
    pb.e_conv_s2(ops, PV(full, C, cof), wf, Ref(bias), PV(half, cout), 1, tag="s2 forward")
    pb.e_conv_s2(ops, PV(dout, cout), wd, None, PV(dfull, C, cof), 2, tag="s2 dgrad")
    npix = B * (h + 2) * (w + 2)
    pb.op(ops, "WGRAD", [npix, w + 2, 4 * C, ldf, cof, cout, cout, 0, rt.CONV_MFMA, 4 * C, B, h, 2048, 1, C],
          p=[Ref(full), Ref(dout), Ref(dwk), Ref(slab)], tag="s2 wgrad")
    return pb, ops


@pytest.mark.parametrize("shape", S2_SHAPES, ids=[f"B{s[0]}_{s[1]}x{s[2]}_{s[3]}to{s[4]}" for s in S2_SHAPES])
def test_stride2_space_to_depth_kernels(shape):
    pb, ops = _mk_s2(*shape)
    failures = lockstep(pb, ops)
    assert not failures, "\n".join(failures)


def test_stride2_big_tile_kernels():
    """The same shapes through the 8-wave tiles (FNET_CONV_TILE=big, read once per process: a child process)."""
    if os.environ.get("FNET_CONV_TILE"):
        pytest.skip("already inside the forced-tile child")
    env = dict(os.environ, FNET_CONV_TILE="big")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "test_stride2_space_to_depth_kernels",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert f"{len(S2_SHAPES)} passed" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("name", ["medium-bf16", "large-depth4-bf16", "tiny-transpose-fp32", "small-medium-base20-bf16"])
def test_prep_weights_batch_equals_record_by_record(name):
    """fnet_plan_run hands a run of consecutive PREP_WEIGHTS records over as one launch per 32 records (k_prep_weights_multi); every
    re-laid-out weight tensor (3x3, stride-2 2x2 form, transposed 2x2; zero-padded channel counts; > 32 records) must equal, bit for bit,
    what the records write when executed one at a time."""
    import fluidnet_oracle as orc
    from fluidnet_amd.engine.rollout_plan import RolloutInferPlan
    from fluidnet_amd.models.unet import UNet, UNetConfig
    cfg, prec = {
        "medium-bf16": (orc.medium_config(dropout=0.0), "bf16"),
        "large-depth4-bf16": (orc.medium_config(dropout=0.0, base_channels=48, depth=4, downsample="avgpool"), "bf16"),
        "tiny-transpose-fp32": (tiny_cfg(upsample="transpose"), "fp32"),
        "small-medium-base20-bf16": (orc.medium_config(dropout=0.0, base_channels=20, downsample="avgpool"), "bf16"),
    }[name]
    m = UNet(UNetConfig(**cfg.__dict__), precision=prec)
    m.load_state_dict(orc.make_params(cfg, seed=11))
    m = m.cuda().eval()
    plan = RolloutInferPlan(m.cfg, 1, 32, 32, params_flat=m.flat_params, dtype=prec)
    bufs = [c[k] for c in plan.pb.convs.values() for k in ("Wf", "Wd") if c.get(k) is not None and getattr(c[k], "tensor", None) is not None]
    assert len(plan._prep) >= 8

    def snapshot():
        torch.cuda.synchronize()
        return [b.tensor.clone() for b in bufs]

    for b in bufs:
        b.tensor.fill_(3.0)
    for k in range(len(plan._prep)):
        rt.run_ops(plan._prep[k:k + 1])          # one record per call: the single-record kernels
    one = snapshot()
    for b in bufs:
        b.tensor.fill_(5.0)
    rt.run_ops(plan._prep)                       # the whole run: batched
    many = snapshot()
    for a, b_ in zip(one, many):
        assert torch.equal(a.view(torch.uint8), b_.view(torch.uint8))
    # a bad record inside a run: the records before it are executed, the error names its index
    bad = plan._prep.copy()
    bad[3]["i"][0] = 0                           # Cout = 0
    for b in bufs:
        b.tensor.fill_(7.0)
    with pytest.raises(rt.FluidNetHipError) as ei:
        rt.run_ops(bad)
    assert "at op #3 (PREP_WEIGHTS)" in str(ei.value)
    rt.run_ops(plan._prep)
    for a, b_ in zip(one, snapshot()):
        assert torch.equal(a.view(torch.uint8), b_.view(torch.uint8))


@pytest.mark.parametrize("name", ["medium-bf16", "tiny-transpose-fp32", "large-depth4-bf16"])
def test_unprep_wgrad_batch_equals_record_by_record(name):
    """fnet_plan_run folds a run of consecutive UNPREP_WGRAD records of one lane in one launch per 32 records (k_unprep_wgrad_multi):
    the OIHW gradients must equal, bit for bit, what the records write one at a time (3x3, stride-2 2x2 form, transposed 2x2; > 32
    records), and a bad record inside a run must stop there and name its index."""
    import fluidnet_oracle as orc
    from fluidnet_amd.engine.train_plan import RolloutTrainPlan
    from fluidnet_amd.models.unet import UNet, UNetConfig
    from fluidnet_amd.training.physics_loss import PhysicsAwareLoss
    cfg, prec = {
        "medium-bf16": (orc.medium_config(dropout=0.0), "bf16"),
        "large-depth4-bf16": (orc.medium_config(dropout=0.0, base_channels=48, depth=4, downsample="avgpool"), "bf16"),
        "tiny-transpose-fp32": (tiny_cfg(upsample="transpose"), "fp32"),
    }[name]
    m = UNet(UNetConfig(**cfg.__dict__), precision=prec)
    m.load_state_dict(orc.make_params(cfg, seed=11))
    m = m.cuda()
    crit = PhysicsAwareLoss(1.0, 0.01, 0.001, 0.01, 2.0 / 32, True, True, True, "replicate", "forward")
    plan = RolloutTrainPlan(m.cfg, crit, 1, 32, 32, 1, dtype=prec)
    plan.bind(torch.device("cuda"), m.flat_params, m.flat_grads)
    arr = plan._lowered["bwd"]
    sel = np.nonzero(arr["op"] == rt.OPC["UNPREP_WGRAD"])[0]
    assert len(sel) >= 8 and np.all(np.diff(sel) == 1), sel          # one run at the end of the backward section
    run = arr[sel[0]:sel[-1] + 1].copy()
    run["i"][:, rt.LANE_SLOT] = 0
    g = torch.Generator(device="cuda").manual_seed(3)
    for c in plan.pb.convs.values():
        c["dWk"].tensor.copy_(torch.randn(c["dWk"].tensor.shape, device="cuda", generator=g))
    grads = m.flat_grads
    grads.fill_(3.0)
    for k in range(len(run)):
        rt.run_ops(run[k:k + 1])
    torch.cuda.synchronize()
    one = grads.clone()
    grads.fill_(5.0)
    rt.run_ops(run)
    torch.cuda.synchronize()
    many = grads.clone()
    touched = one != 3.0
    assert int(touched.sum()) > 1000
    assert torch.equal(one[touched].view(torch.int32), many[touched].view(torch.int32))
    assert torch.all(many[~touched] == 5.0)                              # nothing outside the weight tensors was written
    bad = run.copy()
    bad[3]["i"][0] = 0                                                   # Cout = 0
    grads.fill_(7.0)
    with pytest.raises(rt.FluidNetHipError) as ei:
        rt.run_ops(bad)
    assert "op #3" in str(ei.value)
    torch.cuda.synchronize()
    lo = int(bad[2]["p"][1] - grads.data_ptr()) // 4
    assert float(grads[lo]) != 7.0                                       # the records before the bad one ran
