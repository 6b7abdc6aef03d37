"""Head + clamp + loss in one pass (ml/models/unet.py:294-303 + ml/training/physics_loss.py:210-278):
fnet_plan_run executes HEAD_FWD immediately followed by the LOSS_REDUCE of its prediction as ONE launch (k_head_loss_fwd), and LOSS_GRAD
immediately followed by the HEAD_BWD that consumes its dpred as ONE launch (k_loss_head_bwd).  Each pair executed in one fnet_plan_run
call (fused) against the same two records executed one call at a time (the separate kernels, which the lock-step tests pin to the CPU
twin and the G1-G4 fixtures to the reference): z, pred, dpred and the activation gradient bit-identical; the fp64 loss sums and the
head weight / bias gradients to summation order.  The knob "fuse_head" = 0 must give the separate launches inside one call too."""
import numpy as np
import pytest
import torch

from fluidnet_amd.engine import runtime as rt
from fluidnet_amd.engine.plan import PlanBuilder, Ref, lower
from helpers import tiny_cfg

pytestmark = pytest.mark.gpu

CASES = [
    # B, H,   W,   C,  ldc, coff, dtype,  pm, stencil, flags, outact, dnext, world
    (4, 128, 128, 32, 32, 0, "bf16", 1, 0, 7, 0, True, 1),          # the benchmarked shape (smaller batch)
    (3, 20, 12, 8, 24, 8, "bf16", 0, 1, 7, 0, False, 1),            # ragged image, channel slice, zeros + central
    (2, 33, 40, 16, 16, 0, "fp32", 2, 0, 5, 1, True, 2),            # reflect, sigmoid_tanh, global fluid count (data parallel)
    (2, 16, 16, 64, 64, 0, "bf16", 3, 1, 3, 0, True, 1),            # circular + central: halo rows wrap around the image
    (1, 9, 7, 24, 32, 0, "fp32", 1, 1, 6, 0, False, 1),             # C = 24 of 32 (zero-padded tail), tiny odd image
    (70, 32, 32, 32, 32, 0, "bf16", 1, 0, 7, 0, False, 1),          # many images: several rows per block, more blocks than CUs
]


def _mk(B, H, W, C, ldc, coff, dtype, pm, stencil, flags, outact, dnext, world):
    pb = PlanBuilder(tiny_cfg(base_channels=16), 1, 16, 16, dtype=dtype, conv_impl="mfma" if dtype == "bf16" else "ref")
    pb.bufs.clear()
    pb.B = B
    g = torch.Generator().manual_seed(100 * B + H + C)
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
    HW = H * W
    Cz = (C + 7) // 8 * 8

    def buf(name, shape, kind, t):
        b = pb.new_buf(name, shape, kind)
        b.tensor = t.cuda()
        return b
    act = buf("act", (B, H + 2, W + 2, ldc), "T", torch.randn(B, H + 2, W + 2, ldc, generator=g).to(tdt))
    Wh = buf("Wh", (3 * C,), "f32", 0.4 * torch.randn(3 * C, generator=g))
    bh = buf("bh", (8,), "f32", 0.2 * torch.randn(8, generator=g))
    z = buf("z", (B, 3, H, W), "f32", torch.zeros(B, 3, H, W))
    pred = buf("pred", (B, 3, H, W), "f32", torch.zeros(B, 3, H, W))
    tgt = buf("tgt", (B, 3, H, W), "f32", torch.randn(B, 3, H, W, generator=g))
    x = torch.rand(B, 6, H, W, generator=g)
    x[:, 4] = (x[:, 4] > 0.85).float()
    x[:, 5] = (x[:, 5] > 0.9).float()
    xin = buf("xin", (B, 6, H, W), "f32", x)
    sums = buf("sums", (8,), "f64", torch.zeros(8, dtype=torch.float64))
    gcnt = buf("gcnt", (1,), "f64", torch.tensor([1.7 * float(((x[:, 4] == 0) & (x[:, 5] == 0)).sum())], dtype=torch.float64))
    dn = buf("dnext", (B, 4, H, W), "f32", 1e-3 * torch.randn(B, 4, H, W, generator=g))
    dpred = buf("dpred", (B, 3, H, W), "f32", torch.zeros(B, 3, H, W))
    dact = buf("dact", (B, H + 2, W + 2, Cz), "T", torch.zeros(B, H + 2, W + 2, Cz).to(tdt))
    dWh = buf("dWh", (3 * C,), "f32", torch.zeros(3 * C))
    dbh = buf("dbh", (8,), "f32", torch.zeros(8))
    li = [B, H, W, 3 * HW, 6 * HW, 6 * HW, 6 * HW, pm, stencil, flags]
    lp = [Ref(pred), Ref(tgt), Ref(xin, 0), Ref(xin, 4 * HW), Ref(xin, 5 * HW)]
    fwd, bwd = [], []
    pb.op(fwd, "HEAD_FWD", [B, H, W, C, ldc, coff, 3, outact], p=[Ref(act), Ref(Wh), Ref(bh), Ref(z), Ref(pred)])
    pb.op(fwd, "LOSS_REDUCE", li, f=[2.0 / W], p=lp + [Ref(sums)])
    pb.op(bwd, "LOSS_GRAD", li + [world, 4 * HW if dnext else 0], f=[2.0 / W, 1.0, 0.01, 0.001, 0.01, 0.4],
          p=lp + [Ref(sums), Ref(gcnt) if world > 1 else None, Ref(dn) if dnext else None, None, Ref(dpred)])
    pb.op(bwd, "HEAD_BWD", [B, H, W, C, ldc, coff, Cz, 0, 3, outact, Cz],
          p=[Ref(dpred), Ref(z), Ref(pred), Ref(act), Ref(Wh), Ref(dact), Ref(dWh), Ref(dbh)])
    outs = dict(z=z, pred=pred, sums=sums, dpred=dpred, dact=dact, dWh=dWh, dbh=dbh)
    return pb, fwd, bwd, outs


def _run(pb, fwd, bwd, outs, together):
    for b in outs.values():
        b.tensor.zero_()
    for ops in (fwd, bwd):
        arr = lower(pb, ops)
        if together:
            rt.run_ops(arr)
        else:
            for k in range(len(arr)):
                rt.run_ops(arr[k:k + 1])
    torch.cuda.synchronize()
    return {k: b.tensor.clone() for k, b in outs.items()}


@pytest.mark.parametrize("case", CASES, ids=[f"B{c[0]}_{c[1]}x{c[2]}_C{c[3]}_{c[6]}_pm{c[7]}_st{c[8]}_f{c[9]}_oa{c[10]}" for c in CASES])
def test_fused_head_loss_pairs_equal_separate_launches(case, knobs):
    pb, fwd, bwd, outs = _mk(*case)
    one = _run(pb, fwd, bwd, outs, together=False)
    many = _run(pb, fwd, bwd, outs, together=True)
    assert float(one["sums"][5]) > 0 and float(one["dpred"].abs().max()) > 0 and float(one["dact"].float().abs().max()) > 0
    for k in ("z", "pred", "dpred"):
        assert torch.equal(one[k], many[k]), k
    assert torch.equal(one["dact"].view(torch.uint8), many["dact"].view(torch.uint8))
    np.testing.assert_allclose(many["sums"][:6].cpu().numpy(), one["sums"][:6].cpu().numpy(), rtol=1e-11, atol=1e-12)
    for k in ("dWh", "dbh"):
        a, b = one[k].cpu().numpy(), many[k].cpu().numpy()
        np.testing.assert_allclose(b, a, rtol=2e-4, atol=2e-5 * float(np.abs(a).max()))
    # with the knob off the pairs run as their separate launches inside one call: bit-identical up to the atomics' order
    knobs("fuse_head", 0)
    off = _run(pb, fwd, bwd, outs, together=True)
    for k in ("z", "pred", "dpred"):
        assert torch.equal(one[k], off[k]), k


def test_pairs_that_do_not_match_are_not_fused():
    """A LOSS_REDUCE that reads ANOTHER prediction buffer than the HEAD_FWD in front of it wrote must run as written."""
    pb, fwd, bwd, outs = _mk(*CASES[1])
    other = pb.new_buf("pred2", outs["pred"].shape, "f32")
    other.tensor = torch.randn(outs["pred"].shape).cuda()
    fwd[1].p[0] = Ref(other)
    for b in outs.values():
        b.tensor.zero_()
    arr = lower(pb, fwd)
    rt.run_ops(arr)
    torch.cuda.synchronize()
    s_together = outs["sums"].tensor.clone()
    outs["sums"].tensor.zero_()
    rt.run_ops(arr[1:2])
    torch.cuda.synchronize()
    np.testing.assert_allclose(s_together.cpu().numpy(), outs["sums"].tensor.cpu().numpy(), rtol=1e-11)
    assert float(outs["pred"].tensor.abs().max()) > 0          # the head still ran
