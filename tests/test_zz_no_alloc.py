"""fnet_physics_loss_terms_noalloc (include/fluidnet_hip.h): the allocation-free variant of the ZERO + LOSS_REDUCE + LOSS_FINALIZE op
sequence of fnet_plan_run, whose fp64 sums live in a static __device__ array of the library.  Called through the C ABI (ctypes) and
compared with fnet_plan_run on the same inputs -- two calls in a row, the second with a smaller input (the array keeps the first
call's sums unless the entry point clears them).  Tolerance: the sums are fp64 atomics in both paths, the terms fp32: 1e-6 relative.
Both shapes are also checked against the CPU oracle's physics_loss (PhysicsAwareLoss.forward, ml/training/physics_loss.py:210-278) on the
same inputs: fp32 torch reductions against fp64 device sums, 2e-5 relative."""
import ctypes

import numpy as np
import pytest
import torch

import fluidnet_oracle as orc
from fluidnet_amd.engine import runtime as rt

pytestmark = pytest.mark.gpu
W_ = (1.0, 0.01, 0.001, 0.01)
PAD_NAME = {v: k for k, v in rt.PAD.items()}
STENCIL_NAME = {v: k for k, v in rt.STENCIL.items()}


def _inputs(B, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(B, 3, H, W, generator=g).cuda()
    tgt = torch.randn(B, 3, H, W, generator=g).cuda()
    x = torch.rand(B, 6, H, W, generator=g)
    x[:, 4] = (x[:, 4] > 0.9).float()
    x[:, 5] = (x[:, 5] > 0.92).float()
    return pred, tgt, x.cuda()


def _via_plan(pred, tgt, x, pm, stencil, flags, dx):
    B, _, H, W = pred.shape
    HW = H * W
    sums = torch.zeros(8, dtype=torch.float64, device="cuda")
    terms = torch.zeros(8, dtype=torch.float32, device="cuda")
    ops = np.zeros(3, dtype=rt.OP_DTYPE)
    ops[0]["op"] = rt.OPC["ZERO"]
    ops[0]["i"][0] = 64
    ops[0]["p"][0] = sums.data_ptr()
    ops[1]["op"] = rt.OPC["LOSS_REDUCE"]
    ops[1]["i"][:10] = [B, H, W, 3 * HW, 6 * HW, 6 * HW, 6 * HW, pm, stencil, flags]
    ops[1]["f"][0] = dx
    ops[1]["p"][:6] = [pred.data_ptr(), tgt.data_ptr(), x.data_ptr(), x[:, 4].data_ptr(), x[:, 5].data_ptr(), sums.data_ptr()]
    ops[2]["op"] = rt.OPC["LOSS_FINALIZE"]
    ops[2]["i"][:5] = [B, H, W, flags, 1]
    ops[2]["f"][:5] = [*W_, 1.0]
    ops[2]["p"][0], ops[2]["p"][2] = sums.data_ptr(), terms.data_ptr()
    rt.run_ops(ops)
    return terms.cpu().numpy()


def _via_noalloc(pred, tgt, x, pm, stencil, flags, dx, terms):
    B, _, H, W = pred.shape
    HW = H * W
    vp = ctypes.c_void_p
    return rt.lib().fnet_physics_loss_terms_noalloc(
        vp(pred.data_ptr()), vp(tgt.data_ptr()), vp(x.data_ptr()), vp(x[:, 4].data_ptr()), vp(x[:, 5].data_ptr()), B, H, W, 3 * HW, 6 * HW,
        6 * HW, 6 * HW, pm, stencil, flags, dx, *W_, vp(terms.data_ptr()), vp(rt.current_stream_ptr()))


def test_noalloc_matches_plan_run_twice_second_smaller():
    terms = torch.zeros(8, dtype=torch.float32, device="cuda")             # the caller's output buffer, allocated before the calls
    for (B, H, W, seed, pm, stencil, flags) in ((16, 128, 128, 1, 1, 0, 7), (2, 24, 40, 2, 3, 1, 3)):
        pred, tgt, x = _inputs(B, H, W, seed)
        want = _via_plan(pred, tgt, x, pm, stencil, flags, 2.0 / W)
        before = torch.cuda.memory_allocated()
        rc = _via_noalloc(pred, tgt, x, pm, stencil, flags, 2.0 / W, terms)
        assert rc == 0, rt.lib().fnet_strerror(rc)
        assert torch.cuda.memory_allocated() == before
        got = terms.cpu().numpy()
        assert np.all(np.isfinite(got[:6])) and want[5] > 0
        np.testing.assert_allclose(got[:6], want[:6], rtol=1e-6, atol=1e-9)
        # the oracle on the same inputs: terms[0..4] = mse, divergence, gradient, emitter, total (disabled terms are absent / 0)
        lc = orc.LossConfig(*W_, 2.0 / W, bool(flags & 1), bool(flags & 2), bool(flags & 4), PAD_NAME[pm], STENCIL_NAME[stencil])
        total, d = orc.physics_loss(pred.cpu(), tgt.cpu(), x.cpu(), lc)
        ref = [float(d.get(k, 0.0)) for k in ("mse", "divergence", "gradient", "emitter")] + [float(total)]
        np.testing.assert_allclose(got[:5], ref, rtol=2e-5, atol=1e-7)


def test_noalloc_rejects_oversize_and_bad_arguments():
    pred, tgt, x = _inputs(1, 8, 8, 3)
    terms = torch.full((8,), -7.0, device="cuda")
    vp = ctypes.c_void_p
    L = rt.lib()
    args = lambda B, H, W: (vp(pred.data_ptr()), vp(tgt.data_ptr()), vp(x.data_ptr()), vp(x[:, 4].data_ptr()), vp(x[:, 5].data_ptr()), B, H, W,  # noqa: E731
                            3 * H * W, 6 * H * W, 6 * H * W, 6 * H * W, 1, 0, 7, 0.1, *W_, vp(terms.data_ptr()), vp(rt.current_stream_ptr()))
    rc = L.fnet_physics_loss_terms_noalloc(*args(4, 32768, 32768))        # 2^32 cells > 2^30: refused before anything is launched
    assert rc == -4 and b"maximum" in L.fnet_strerror(rc)
    assert L.fnet_physics_loss_terms_noalloc(*args(0, 8, 8)) == -1
    torch.cuda.synchronize()
    assert float(terms[0]) == -7.0
