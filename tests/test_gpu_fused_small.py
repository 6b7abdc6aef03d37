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


@pytest.fixture(autouse=True)
def _fuse_small_batches(knobs):
    # the library fuses only where B * C / 16 >= 96 (enough blocks); these kernel tests use small batches
    knobs("sa_minitems", 1)


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
def test_fused_stats_apply_equals_separate_launches(case, cs, knobs):
    knobs("sa_cs", cs)       # every slice width of the kernel (64 falls back where C % 64 != 0) (the launcher picks by block count otherwise)
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


def test_small_batches_do_not_fuse(knobs):
    # B * C / 16 = 2 * 4 < 96: the pair runs as its separate launches inside one call too -> bit-identical to one call per record
    knobs("sa_minitems", 96)
    pb, ops, st, dst = _mk(2, 32, 32, 64, 64, 0, False, "gelu", 1, 0.0)
    arr = lower(pb, ops)
    s1, d1 = _run(arr, st, dst, together=False)
    s2, d2 = _run(arr, st, dst, together=True)
    assert torch.equal(s1, s2) and torch.equal(d1, d2)


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


# ---- the adjoint pair: INACT_BWD_REDUCE + INACT_BWD_APPLY of a single-source unit (k_inact_bwd_fused)
BWD_CASES = [
    # B, H,  W,  C,  ldc_g, coff_g, fold, act,    pm, drop_p
    (4, 64, 64, 64, 64, 0, 1, "gelu", 1, 0.0),
    (3, 32, 32, 256, 256, 0, 0, "gelu", 1, 0.1),
    (5, 16, 16, 256, 256, 0, 1, "silu", 2, 0.0),
    (2, 20, 12, 32, 96, 32, 1, "relu", 3, 0.0),          # ragged image, gradient source = a channel slice of a wider buffer
    (2, 8, 8, 128, 128, 0, 1, "leaky", 0, 0.2),          # zero padding: nothing to fold
    (9, 64, 64, 128, 128, 0, 1, "gelu", 1, 0.0),
]


def _mk_bwd(B, H, W, C, ldc, coff, fold, act, pm, drop_p, two_sources=False):
    pb = PlanBuilder(tiny_cfg(base_channels=16), 1, 16, 16, dtype="bf16", conv_impl="mfma")
    pb.bufs.clear()
    pb.B = B
    g = torch.Generator().manual_seed(77 * B + H + C)
    T = torch.bfloat16
    g1 = pb.new_buf("g1", (B, H + 2, W + 2, ldc), "T")
    g1.tensor = torch.randn(g1.shape, generator=g).to(T).cuda()
    g2 = pb.new_buf("g2", (B, H + 2, W + 2, C), "T")
    g2.tensor = torch.randn(g2.shape, generator=g).to(T).cuda()
    G = pb.new_buf("G", (B, H + 2, W + 2, C), "T")
    G.tensor = torch.zeros(G.shape).to(T).cuda()
    raw = pb.new_buf("raw", (B, H + 2, W + 2, C), "T")
    raw.tensor = (2.0 * torch.randn(raw.shape, generator=g) + 0.7).to(T).cuda()
    draw = pb.new_buf("draw", (B, H + 2, W + 2, C), "T")
    draw.tensor = torch.full(draw.shape, 7.0).to(T).cuda()
    st = pb.new_buf("stats", (B, C, 2), "f32")
    x = raw.tensor[:, 1:-1, 1:-1].float()
    mean, var = x.mean(dim=(1, 2)), x.var(dim=(1, 2), unbiased=False)
    st.tensor = torch.stack([mean, torch.rsqrt(var + 1e-5)], dim=-1).contiguous()
    gam = pb.new_buf("gamma", (C,), "f32")
    gam.tensor = (1.0 + 0.3 * torch.randn(C, generator=g)).cuda()
    bet = pb.new_buf("beta", (C,), "f32")
    bet.tensor = (0.2 * torch.randn(C, generator=g)).cuda()
    S = pb.new_buf("S", (B, C, 2), "f32")
    S.tensor = torch.zeros(S.shape).cuda()
    ops = []
    pb.op(ops, "INACT_BWD_REDUCE", [B, H, W, C, ldc, coff, C if two_sources else 8, 0, C, 0, C, 0, fold, 0, ACTS[act], pm, 5, 1234], f=[drop_p],
          p=[Ref(g1), Ref(g2) if two_sources else None, Ref(raw), Ref(st), Ref(gam), Ref(bet), Ref(G) if two_sources else None, Ref(S)])
    if two_sources:
        pb.op(ops, "INACT_BWD_APPLY", [B, H, W, C, C, 0, C, 0, C, 0, ACTS[act], 5, 1234, 0, pm, 0], f=[drop_p],
              p=[Ref(G), Ref(raw), Ref(st), Ref(gam), Ref(bet), Ref(S), Ref(draw)])
    else:
        pb.op(ops, "INACT_BWD_APPLY", [B, H, W, C, ldc, coff, C, 0, C, 0, ACTS[act], 5, 1234, fold, pm, 0], f=[drop_p],
              p=[Ref(g1), Ref(raw), Ref(st), Ref(gam), Ref(bet), Ref(S), Ref(draw)])
    return pb, ops, S, draw


def _run_bwd(arr, S, draw, together):
    S.tensor.zero_()
    draw.tensor.fill_(7.0)
    if together:
        rt.run_ops(arr)
    else:
        for k in range(len(arr)):
            rt.run_ops(arr[k:k + 1])
    torch.cuda.synchronize()
    return S.tensor.clone(), draw.tensor.float().clone()


@pytest.mark.parametrize("cs", ["16", "32", "64"])
@pytest.mark.parametrize("case", BWD_CASES, ids=[f"B{c[0]}_{c[1]}x{c[2]}_C{c[3]}_{c[7]}_pm{c[8]}_fold{c[6]}" for c in BWD_CASES])
def test_fused_adjoint_equals_separate_launches(case, cs, knobs):
    # (the library fuses the adjoint pair up to 32^2 by default -- at 64^2 it measured the same as the two launches; the sb_maxhw knob
    # raises the limit so that the 64^2 cases above exercise the kernel too)
    knobs("sa_cs", cs)
    knobs("sb_maxhw", 64)
    pb, ops, S, draw = _mk_bwd(*case)
    arr = lower(pb, ops)
    s1, d1 = _run_bwd(arr, S, draw, together=False)
    s2, d2 = _run_bwd(arr, S, draw, together=True)
    assert bool(torch.isfinite(d2).all())
    # the sums: same terms, another order (the separate kernel adds block partials with float atomics: not bit-reproducible itself)
    scale = float(s1.abs().max())
    assert float((s2 - s1).abs().max()) <= 2e-5 * scale + 1e-6
    # d_raw: the ring is exactly zero in both; the interior to one bf16 ulp (+ the fp32-level difference of the sums where d_raw is near 0)
    assert torch.equal(d2[:, 0], torch.zeros_like(d2[:, 0])) and torch.equal(d2[:, :, -1], torch.zeros_like(d2[:, :, -1]))
    diff = (d2 - d1).abs()
    tol = torch.maximum(d1.abs(), d2.abs()) * 2.0 ** -7 + 2e-5 * float(d1.abs().max())
    assert bool((diff <= tol).all()), float((diff / tol).max())
    assert float((diff > 0).float().mean()) < 2e-2


def test_two_source_units_keep_the_two_launches():
    pb, ops, S, draw = _mk_bwd(3, 16, 16, 64, 64, 0, 1, "gelu", 1, 0.0, two_sources=True)
    arr = lower(pb, ops)
    s1, d1 = _run_bwd(arr, S, draw, together=False)
    s2, d2 = _run_bwd(arr, S, draw, together=True)
    # (float atomics in pass 1: the sums agree to rounding, d_raw to an ulp -- and the launches are the same two kernels either way)
    assert float((s2 - s1).abs().max()) <= 2e-5 * float(s1.abs().max()) + 1e-6
    diff = (d2 - d1).abs()
    assert bool((diff <= torch.maximum(d1.abs(), d2.abs()) * 2.0 ** -7 + 2e-5 * float(d1.abs().max())).all())
