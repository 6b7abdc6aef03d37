"""k_stats_apply: fnet_plan_run executes an IN_STATS record that is immediately followed by the APPLY of the same unit as ONE launch when the
image is small (<= 64^2) -- statistics, their finalisation and the normalise / activate pass of ml/models/unet.py:65-69 in one kernel.  The
pair executed in one fnet_plan_run call (fused) against the same two records executed one call at a time (k_in_stats_part + k_in_finalize,
then k_apply): statistics to fp32 summation order, outputs to one bf16 ulp where a statistic's last bit moved (the pixel walk and its
arithmetic are shared code), ring included; larger images and other pairings must not fuse (bit-identical either way)."""
import numpy as np
import pytest
import torch

from fluidnet_amd.engine import runtime as rt
from fluidnet_amd.engine.plan import PlanBuilder, Ref, lower
from helpers import tiny_cfg

pytestmark = pytest.mark.gpu
ACTS = {"relu": 0, "leaky": 1, "gelu": 2, "silu": 3}
CASES = [
    # B, H,  W,  C,  ldc_raw, coff_raw, residual, act,     pm, drop_p
    (4, 64, 64, 64, 64, 0, False, "gelu", 1, 0.0),
    (3, 32, 32, 256, 256, 0, True, "gelu", 1, 0.1),
    (5, 16, 16, 256, 256, 0, False, "silu", 0, 0.0),
    (2, 20, 12, 32, 96, 32, True, "relu", 2, 0.0),          # ragged image, channel slice of a wider buffer
    (2, 8, 8, 128, 128, 0, False, "leaky", 3, 0.2),
    (9, 64, 64, 128, 128, 0, True, "gelu", 1, 0.0),           # more items than 8 XCD slots x 1 round
]


def _mk(B, H, W, C, ldc, coff, residual, act, pm, drop_p):
    pb = PlanBuilder(tiny_cfg(base_channels=16), 1, 16, 16, dtype="bf16", conv_impl="mfma")
    pb.bufs.clear()
    pb.B = B
    g = torch.Generator().manual_seed(1000 * B + H + C)
    raw = pb.new_buf("raw", (B, H + 2, W + 2, ldc), "T")
    raw.tensor = (2.0 * torch.randn(raw.shape, generator=g) + 0.7).to(torch.bfloat16).cuda()
    res = pb.new_buf("res", (B, H + 2, W + 2, C), "T")
    res.tensor = torch.randn(res.shape, generator=g).to(torch.bfloat16).cuda()
    dst = pb.new_buf("dst", (B, H + 2, W + 2, C), "T")
    dst.tensor = torch.full(dst.shape, 7.0).to(torch.bfloat16).cuda()
    st = pb.new_buf("stats", (B, C, 2), "f32")
    st.tensor = torch.zeros(st.shape).cuda()
    gam = pb.new_buf("gamma", (C,), "f32")
    gam.tensor = (1.0 + 0.3 * torch.randn(C, generator=g)).cuda()
    bet = pb.new_buf("beta", (C,), "f32")
    bet.tensor = (0.2 * torch.randn(C, generator=g)).cuda()
    sc = pb.new_buf("scratch", (B * 64 * C * 2,), "f32")
    sc.tensor = torch.zeros(sc.shape).cuda()
    ops = []
    pb.op(ops, "IN_STATS", [B, H, W, C, ldc, coff, sc.numel], f=[1e-5], p=[Ref(raw), Ref(st), Ref(sc)])
    pb.op(ops, "APPLY", [B, H, W, C, ldc, coff, C if residual else 8, 0, C, 0, ACTS[act], pm, 5, 1234], f=[drop_p],
          p=[Ref(raw), Ref(st), Ref(gam), Ref(bet), Ref(res) if residual else None, Ref(dst)])
    return pb, ops, st, dst


def _run(arr, st, dst, together):
    st.tensor.zero_()
    dst.tensor.fill_(7.0)
    if together:
        rt.run_ops(arr)
    else:
        for k in range(len(arr)):
            rt.run_ops(arr[k:k + 1])
    torch.cuda.synchronize()
    return st.tensor.clone(), dst.tensor.float().clone()


@pytest.mark.parametrize("cs", ["16", "32", "64"])
@pytest.mark.parametrize("case", CASES, ids=[f"B{c[0]}_{c[1]}x{c[2]}_C{c[3]}_{c[7]}_pm{c[8]}" for c in CASES])
def test_fused_stats_apply_equals_separate_launches(case, cs, monkeypatch):
    monkeypatch.setenv("FNET_SA_CS", cs)       # every slice width of the kernel (64 falls back where C % 64 != 0) (the launcher picks by block count otherwise)
    pb, ops, st, dst = _mk(*case)
    arr = lower(pb, ops)
    s1, d1 = _run(arr, st, dst, together=False)
    s2, d2 = _run(arr, st, dst, together=True)
    assert bool(torch.isfinite(d2).all())
    # statistics: the same sums in another (fixed) order
    assert float((s2[..., 0] - s1[..., 0]).abs().max()) <= 2e-6 * max(1.0, float(s1[..., 0].abs().max()))
    assert float(((s2[..., 1] - s1[..., 1]) / s1[..., 1]).abs().max()) <= 5e-6
    # outputs: one bf16 ulp at most, and only where a statistic's last bits moved the rounding
    diff = (d2 - d1).abs()
    # (near an output's zero the statistics' fp32-level difference, ~5e-6 of a pre-activation of order 1 - 10, shows un-rounded: 5e-5 floor)
    ulp = torch.maximum(d1.abs(), d2.abs()) * 2.0 ** -7 + 5e-5
    assert bool((diff <= ulp).all()), float((diff / ulp).max())
    assert float((diff > 0).float().mean()) < 2e-2
    # run to run: the fused launch is deterministic
    s3, d3 = _run(arr, st, dst, together=True)
    assert torch.equal(s2, s3) and torch.equal(d2, d3)


def test_large_images_and_foreign_pairs_do_not_fuse():
    # 128 x 128: over the size limit -> the pair runs as two launches inside one call too: bit-identical to one call per record
    pb, ops, st, dst = _mk(2, 128, 128, 32, 32, 0, False, "gelu", 1, 0.0)
    arr = lower(pb, ops)
    s1, d1 = _run(arr, st, dst, together=False)
    s2, d2 = _run(arr, st, dst, together=True)
    assert torch.equal(s1, s2) and torch.equal(d1, d2)
    # an APPLY that normalises with OTHER statistics than the preceding IN_STATS wrote must not be fused with it
    pb, ops, st, dst = _mk(2, 16, 16, 32, 32, 0, False, "gelu", 1, 0.0)
    other = pb.new_buf("other_stats", (2, 32, 2), "f32")
    other.tensor = torch.stack([torch.full((2, 32), 0.25), torch.full((2, 32), 2.0)], dim=-1).cuda()
    ops[1].p[1] = Ref(other)
    arr = lower(pb, ops)
    s1, d1 = _run(arr, st, dst, together=False)
    s2, d2 = _run(arr, st, dst, together=True)
    assert torch.equal(s1, s2) and torch.equal(d1, d2)
