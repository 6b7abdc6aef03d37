"""CPU twin of every FnetOp opcode -- TEST INFRASTRUCTURE ONLY.

Interprets the OpRec records produced by fluidnet_amd.engine.plan on CPU torch tensors with exactly the
buffer semantics of the HIP kernels (P-tensor layout, flattened convolution incl. ring outputs, ring folding,
dropout hash).  Used (a) on CPU to check the graph wiring of the plan builder against the oracle's autograd,
(b) on the GPU box as the per-kernel expected value.  Never imported by fluidnet_amd/.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from fluidnet_amd.engine import runtime as rt
from fluidnet_amd.engine.dropout import drop_scale_table

OPC = rt.OPC


def pad_src(i, n, pm):
    if 0 <= i < n:
        return i
    if pm == 1:
        return 0 if i < 0 else n - 1
    if pm == 2:
        return 1 if i < 0 else n - 2
    if pm == 3:
        return n - 1 if i < 0 else 0
    return -1


def pad_index(n, pm):
    """index tensor of length n+2 with the interior source of each padded coordinate (-1 -> zero)."""
    return torch.tensor([pad_src(i - 1, n, pm) for i in range(n + 2)], dtype=torch.long)


def pad_from_interior(x, pm):
    """x [B,H,W,C] -> [B,H+2,W+2,C] by the ring map."""
    B, H, W, C = x.shape
    iy, ix = pad_index(H, pm), pad_index(W, pm)
    out = x[:, iy.clamp(min=0)][:, :, ix.clamp(min=0)]
    out = out * (iy >= 0).float()[None, :, None, None] * (ix >= 0).float()[None, None, :, None]
    return out


def fold(g, pm, do_fold=1):
    """g [B,H+2,W+2,C] padded gradient -> [B,H,W,C] interior gradient."""
    B, Hp, Wp, C = g.shape
    H, W = Hp - 2, Wp - 2
    out = g[:, 1:-1, 1:-1].clone()
    if not do_fold or pm == 0:
        return out
    # rows first, then columns (separable)
    rows = g[:, :, :, :].clone()
    acc = rows[:, 1:-1].clone()
    for ring, src in ((0, pad_src(-1, H, pm)), (Hp - 1, pad_src(H, H, pm))):
        acc[:, src] += rows[:, ring]
    out = acc[:, :, 1:-1].clone()
    for ring, src in ((0, pad_src(-1, W, pm)), (Wp - 1, pad_src(W, W, pm))):
        out[:, :, src] += acc[:, :, ring]
    return out


def act_fwd(u, act):
    if act == 0:
        return torch.relu(u)
    if act == 1:
        return torch.nn.functional.leaky_relu(u, 0.1)
    if act == 2:
        return 0.5 * u * (1 + torch.erf(u * 0.70710678118654752))
    if act == 3:
        return u * torch.sigmoid(u)
    return u


def act_bwd(u, act):
    if act == 0:
        return (u > 0).float()
    if act == 1:
        return torch.where(u > 0, torch.ones_like(u), torch.full_like(u, 0.1))
    if act == 2:
        return 0.5 * (1 + torch.erf(u * 0.70710678118654752)) + u * 0.3989422804014327 * torch.exp(-0.5 * u * u)
    if act == 3:
        s = torch.sigmoid(u)
        return s * (1 + u * (1 - s))
    return torch.ones_like(u)


class Sim:
    def __init__(self, plan_dtype):
        self.T = torch.bfloat16 if plan_dtype == rt.BF16 else torch.float32

    # ---- accessors
    def pv(self, ref, C, ldc, coff):
        t = ref.buf.tensor
        assert t.shape[3] == ldc, (ref.buf.name, t.shape, ldc)
        return t[..., coff:coff + C]

    def rd(self, ref, C, ldc, coff):
        return self.pv(ref, C, ldc, coff).float()

    def wr(self, ref, C, ldc, coff, val, interior_only=False):
        dst = self.pv(ref, C, ldc, coff)
        if interior_only:
            dst[:, 1:-1, 1:-1] = val.to(dst.dtype)
        else:
            dst.copy_(val.to(dst.dtype))

    def round(self, v):
        """fp32 value as it would read back after a store in the plan's element type"""
        return v.to(self.T).float()

    def flat(self, ref):
        return ref.buf.tensor.view(-1)[ref.off:]

    def planes(self, ref, B, bstride, count):
        f = self.flat(ref)
        return torch.stack([f[n * bstride: n * bstride + count] for n in range(B)], 0)

    # ---- interpreter
    def run(self, recs):
        for r in recs:
            getattr(self, "op_" + rt.OPNAME[r.code])(r)

    def op_NOP(self, r):
        pass

    # lanes (fluidnet_hip.h): the twin executes the records in plan order on one thread, which is one of the orders the events allow
    def op_EVENT_RECORD(self, r):
        pass

    def op_EVENT_WAIT(self, r):
        pass

    def op_PACK_INPUT(self, r):
        B, H, W, Cpad, pm = r.i[:5]
        st = r.i[5:11]
        n = r.i[15] if len(r.i) > 15 and r.i[15] else 0
        if n:                                                   # one contiguous NCHW tensor of n channels
            x = self.planes(r.p[0], B, st[0], n * H * W).view(B, n, H, W).permute(0, 2, 3, 1).float()
        else:
            n = 6
            x = torch.stack([self.planes(r.p[c], B, st[c], H * W).view(B, H, W) for c in range(6)], -1).float()
        x = torch.cat([x, torch.zeros(B, H, W, Cpad - n)], -1)
        r.p[6].buf.tensor.copy_(pad_from_interior(x, pm).to(self.T))

    def _flat_conv_inputs(self, src, Wp):
        B, Hp, Wp_, C = src.shape
        npix = B * Hp * Wp_
        z = torch.zeros(Wp + 1, C)
        return torch.cat([z, src.reshape(npix, C), z], 0), npix

    @staticmethod
    def _s2d(src, h, w):
        """[B][2h+2][2w+2][C] -> the space-to-depth view X' [B][h+2][w+2][4C] over PADDED coordinates (row h+1 / column w+1 are zero):
        X'[Y][X][(py, px, c)] = src[2Y + py][2X + px][c]  (fnet_conv_mfma.hip, MODE 1)"""
        B, _, _, C = src.shape
        x = src.reshape(B, h + 1, 2, w + 1, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, h + 1, w + 1, 4 * C)
        out = torch.zeros(B, h + 2, w + 2, 4 * C)
        out[:, :h + 1, :w + 1] = x
        return out

    def _conv_s2(self, r):
        npix, Wp, cin, ldc_in, coff_in, cout, ldc_out, coff_out, impl, mode, h, C = r.i[:12]
        w = Wp - 2
        Wk = r.p[1].buf.tensor.float()                     # [4][cout][K], K-blocked [K/16][4][cout][16]
        K = Wk.shape[2]
        assert impl == rt.CONV_MFMA and tuple(Wk.shape[:2]) == (4, cout) and cin <= K < cin + 16
        Wk = Wk.reshape(K // 16, 4, cout, 16).permute(1, 2, 0, 3).reshape(4, cout, K)[:, :, :cin]
        if mode == 1:
            assert cin == 4 * C
            src = self._s2d(self.rd(r.p[0], C, ldc_in, coff_in), h, w)
            first = -(Wp + 1)                               # the 2x2 window reaches up / left
        else:
            assert cout == 4 * C
            src = self.rd(r.p[0], cin, ldc_in, coff_in)
            first = 0
        B = src.shape[0]
        xs, n = self._flat_conv_inputs(src, Wp)
        assert n == npix
        out = torch.zeros(npix, cout)
        for t in range(4):
            off = first + (t // 2) * Wp + (t % 2)
            out += xs[Wp + 1 + off: Wp + 1 + off + npix] @ Wk[t].t()
        if r.p[2] is not None:
            out += self.flat(r.p[2])[:cout].float()
        if mode == 1:
            self.wr(r.p[3], cout, ldc_out, coff_out, out.view(B, h + 2, w + 2, cout))
        else:
            g = out.view(B, h + 2, w + 2, 2, 2, C)[:, :h + 1, :w + 1].permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h + 2, 2 * w + 2, C)
            self.wr(r.p[3], C, ldc_out, coff_out, g)

    def op_CONV3X3(self, r):
        npix, Wp, cin, ldc_in, coff_in, cout, ldc_out, coff_out, impl = r.i[:9]
        if len(r.i) > 9 and r.i[9]:
            return self._conv_s2(r)
        src = self.rd(r.p[0], cin, ldc_in, coff_in)
        Wk = r.p[1].buf.tensor.float()                     # [9][cout][K], K = cin (K = cin rounded up to 16 for the MFMA kernel)
        K = Wk.shape[2]
        assert tuple(Wk.shape[:2]) == (9, cout) and cin <= K < cin + 16, (r.tag, Wk.shape, cout, cin)
        if impl == rt.CONV_MFMA:                           # the MFMA kernel reads the K-blocked layout [K/16][9][cout][16]
            Wk = Wk.reshape(K // 16, 9, cout, 16).permute(1, 2, 0, 3).reshape(9, cout, K)
        assert float(Wk[:, :, cin:].abs().max()) == 0.0 if K > cin else True      # what the kernel's K overrun multiplies by
        Wk = Wk[:, :, :cin]
        xs, n = self._flat_conv_inputs(src, Wp)
        assert n == npix
        out = torch.zeros(npix, cout)
        for t in range(9):
            off = (t // 3 - 1) * Wp + (t % 3 - 1)
            out += xs[Wp + 1 + off: Wp + 1 + off + npix] @ Wk[t].t()
        if r.p[2] is not None:
            out += self.flat(r.p[2])[:cout].float()
        self.wr(r.p[3], cout, ldc_out, coff_out, out.view(src.shape[0], src.shape[1], src.shape[2], cout))

    def op_WGRAD(self, r):
        npix, Wp, cin, ldc_in, coff_in, cout, ldc_do, coff_do, impl, cin_w = r.i[:10]
        if len(r.i) > 13 and r.i[13]:                       # stride-2 convolution in its 2x2 space-to-depth form
            h, C = r.i[11], r.i[14]
            src = self._s2d(self.rd(r.p[0], C, ldc_in, coff_in), h, Wp - 2)
            dout = self.rd(r.p[1], cout, ldc_do, coff_do).reshape(npix, cout)
            xs, _ = self._flat_conv_inputs(src, Wp)
            dWk = r.p[2].buf.tensor
            assert cin == 4 * C and tuple(dWk.shape[:2]) == (4, cout)
            for t in range(4):
                off = -(Wp + 1) + (t // 2) * Wp + (t % 2)
                dWk[t, :, :cin] += dout.t() @ xs[Wp + 1 + off: Wp + 1 + off + npix]
            return
        src = self.rd(r.p[0], cin, ldc_in, coff_in)
        dout = self.rd(r.p[1], cout, ldc_do, coff_do).reshape(npix, cout)
        xs, _ = self._flat_conv_inputs(src, Wp)
        dWk = r.p[2].buf.tensor
        for t in range(9):
            off = (t // 3 - 1) * Wp + (t % 3 - 1)
            dWk[t, :, :cin] += dout.t() @ xs[Wp + 1 + off: Wp + 1 + off + npix]

    def op_IN_STATS(self, r):
        B, H, W, C, ldc, coff = r.i[:6]
        x = self.rd(r.p[0], C, ldc, coff)[:, 1:-1, 1:-1]
        mean = x.mean((1, 2))
        var = ((x - mean[:, None, None]) ** 2).mean((1, 2))
        r.p[1].buf.tensor.copy_(torch.stack([mean, torch.rsqrt(var + r.f[0])], -1))

    def _u(self, raw_int, stats, gamma, beta):
        if stats is None:
            return raw_int, None
        xh = (raw_int - stats[:, None, None, :, 0]) * stats[:, None, None, :, 1]
        return xh * gamma + beta, xh

    def _drop(self, B, C, site, seed, p):
        return torch.from_numpy(drop_scale_table(seed, site, B, C, p))[:, None, None, :]

    def op_APPLY(self, r):
        B, H, W, C, ldc_raw, coff_raw, ldc_res, coff_res, ldc_dst, coff_dst, act, pm, site, seed = r.i[:14]
        raw = self.rd(r.p[0], C, ldc_raw, coff_raw)[:, 1:-1, 1:-1]
        stats = self.flat(r.p[1])[: raw.shape[0] * C * 2].view(raw.shape[0], C, 2) if r.p[1] is not None else None
        gamma = self.flat(r.p[2])[:C] if r.p[2] is not None else None
        beta = self.flat(r.p[3])[:C] if r.p[3] is not None else None
        u, _ = self._u(raw, stats, gamma, beta)
        o = act_fwd(u, act) * self._drop(B, C, site, seed, r.f[0])
        if r.p[4] is not None:
            o = o + self.rd(r.p[4], C, ldc_res, coff_res)[:, 1:-1, 1:-1]
        # the kernel evaluates the ring from the SOURCE pixel in fp32 and rounds once: same here
        self.wr(r.p[5], C, ldc_dst, coff_dst, pad_from_interior(o, pm))

    def op_INACT_BWD_REDUCE(self, r):
        (B, H, W, C, l1, c1, l2, c2, lr_, cr, lG, cG, fold1, fold2, act, pm, site, seed) = r.i[:18]
        g = fold(self.rd(r.p[0], C, l1, c1), pm, fold1)
        if r.p[1] is not None:
            g = g + fold(self.rd(r.p[1], C, l2, c2), pm, fold2)
        if r.p[6] is not None:
            self.wr(r.p[6], C, lG, cG, g, interior_only=True)
            g = self.rd(r.p[6], C, lG, cG)[:, 1:-1, 1:-1]       # as stored (rounded in bf16 mode)
        else:                                                   # single source, no G: pass 2 recomputes the same rounded value
            g = self.round(g)
        raw = self.rd(r.p[2], C, lr_, cr)[:, 1:-1, 1:-1]
        u, xh = self._u(raw, self.flat(r.p[3])[: B * C * 2].view(B, C, 2), self.flat(r.p[4])[:C], self.flat(r.p[5])[:C])
        dz = g * self._drop(B, C, site, seed, r.f[0]) * act_bwd(u, act)
        S = torch.stack([dz.sum((1, 2)), (dz * xh).sum((1, 2))], -1)
        self.flat(r.p[7])[: B * C * 2] += S.reshape(-1)
        if len(r.p) > 8 and r.p[8] is not None:
            self.flat(r.p[8])[:C] += S[:, :, 1].sum(0)
            self.flat(r.p[9])[:C] += S[:, :, 0].sum(0)

    def op_INACT_BWD_APPLY(self, r):
        B, H, W, C, lG, cG, lr_, cr, ld, cd, act, site, seed = r.i[:13]
        do_fold, pm = (r.i[13], r.i[14]) if len(r.i) > 14 else (0, 0)
        if do_fold:                                             # p[0] is the unit's single gradient source, ring not yet folded
            g = self.round(fold(self.rd(r.p[0], C, lG, cG), pm, 1))
        else:
            g = self.rd(r.p[0], C, lG, cG)[:, 1:-1, 1:-1]
        raw = self.rd(r.p[1], C, lr_, cr)[:, 1:-1, 1:-1]
        stats = self.flat(r.p[2])[: B * C * 2].view(B, C, 2)
        gamma = self.flat(r.p[3])[:C]
        u, xh = self._u(raw, stats, gamma, self.flat(r.p[4])[:C])
        dz = g * self._drop(B, C, site, seed, r.f[0]) * act_bwd(u, act)
        S = self.flat(r.p[5])[: B * C * 2].view(B, C, 2)
        inv = 1.0 / (H * W)
        rstd = stats[:, None, None, :, 1]
        if len(r.i) > 15 and r.i[15]:                           # S = NORM_COMBINE_BWD's gamma-weighted group means
            d = gamma * rstd * dz - rstd * (S[:, None, None, :, 0] * inv + xh * S[:, None, None, :, 1] * inv)
        else:
            d = gamma * rstd * (dz - S[:, None, None, :, 0] * inv - xh * S[:, None, None, :, 1] * inv)
        out = torch.zeros(B, H + 2, W + 2, C)
        out[:, 1:-1, 1:-1] = d
        self.wr(r.p[6], C, ld, cd, out)

    def op_NORM_COMBINE(self, r):
        B, C, kind, gs, HW = r.i[:5]
        cr = r.i[5] if len(r.i) > 5 and r.i[5] else C                # groups cover the real channels only
        eps, mom = r.f[0], r.f[1]
        st = r.p[0].buf.tensor.view(-1)[r.p[0].off: r.p[0].off + B * C * 2].view(B, C, 2)
        if kind == 3:
            st[:, :, 0] = self.flat(r.p[1])[:C]
            st[:, :, 1] = torch.rsqrt(self.flat(r.p[2])[:C] + eps)
            return
        mean = st[:, :, 0].clone()
        e2 = (1.0 / st[:, :, 1] ** 2 - eps).clamp_min(0) + mean ** 2
        if kind == 1:
            m, q = mean.mean(0), e2.mean(0)
            var = (q - m * m).clamp_min(0)
            st[:, :, 0] = m
            st[:, :, 1] = torch.rsqrt(var + eps)
            if r.p[1] is not None:
                n = float(B * HW)
                rm, rv = self.flat(r.p[1]), self.flat(r.p[2])
                rm[:C] = (1 - mom) * rm[:C] + mom * m
                rv[:C] = (1 - mom) * rv[:C] + mom * var * (n / max(n - 1.0, 1.0))
        else:
            m = mean[:, :cr].reshape(B, cr // gs, gs).mean(2, keepdim=True)
            q = e2[:, :cr].reshape(B, cr // gs, gs).mean(2, keepdim=True)
            var = (q - m * m).clamp_min(0)
            st[:, :cr, 0] = m.expand(B, cr // gs, gs).reshape(B, cr)
            st[:, :cr, 1] = torch.rsqrt(var + eps).expand(B, cr // gs, gs).reshape(B, cr)

    def op_NORM_COMBINE_BWD(self, r):
        B, C, kind, gs = r.i[:4]
        cr = r.i[4] if len(r.i) > 4 and r.i[4] else C
        S = self.flat(r.p[0])[: B * C * 2].view(B, C, 2)
        gamma = self.flat(r.p[1])[:C]
        out = self.flat(r.p[2])[: B * C * 2].view(B, C, 2)
        if kind == 1:
            out.copy_((gamma[:, None] * S.mean(0)).expand(B, C, 2))
        elif kind == 2:
            w = (gamma[None, :cr, None] * S[:, :cr]).reshape(B, cr // gs, gs, 2).mean(2, keepdim=True)
            out.zero_()
            out[:, :cr] = w.expand(B, cr // gs, gs, 2).reshape(B, cr, 2)
        else:
            out.zero_()

    def op_AFFINE_GRAD(self, r):
        B, C = r.i[:2]
        S = self.flat(r.p[0])[: B * C * 2].view(B, C, 2)
        self.flat(r.p[2])[:C] += S[:, :, 0].sum(0)
        self.flat(r.p[1])[:C] += S[:, :, 1].sum(0)

    def op_AFFINE_GRAD_MULTI(self, r):
        B, n = r.i[:2]
        slab, grads = self.flat(r.p[0]), self.flat(r.p[1])
        for j in range(n):
            s_off, C, dg, db = r.i[2 + 4 * j: 6 + 4 * j]
            S = slab[s_off: s_off + B * C * 2].view(B, C, 2)
            grads[db: db + C] += S[:, :, 0].sum(0)
            if dg >= 0:
                grads[dg: dg + C] += S[:, :, 1].sum(0)

    def op_FOLD_SUM(self, r):
        B, H, W, C, l1, c1, l2, c2, ld, cd, fold1, fold2, pm = r.i[:13]
        g = fold(self.rd(r.p[0], C, l1, c1), pm, fold1)
        if r.p[1] is not None:
            g = g + fold(self.rd(r.p[1], C, l2, c2), pm, fold2)
        out = torch.zeros(B, H + 2, W + 2, C)
        out[:, 1:-1, 1:-1] = g
        self.wr(r.p[2], C, ld, cd, out)

    def op_BIAS_GRAD(self, r):
        B, H, W, C, ldc, coff = r.i[:6]
        self.flat(r.p[1])[:C] += self.rd(r.p[0], C, ldc, coff)[:, 1:-1, 1:-1].sum((0, 1, 2))

    @staticmethod
    def _up_matrix(h, mode):
        M = torch.zeros(2 * h, h)
        for Y in range(2 * h):
            a = Y // 2
            if mode == 0:
                M[Y, a] = 1
            elif Y % 2:
                M[Y, a] += 0.75
                M[Y, min(a + 1, h - 1)] += 0.25
            else:
                M[Y, max(a - 1, 0)] += 0.25
                M[Y, a] += 0.75
        return M

    def op_UPSAMPLE2X(self, r):
        B, h, w, C, ls, cs, ld, cd, mode, pm = r.i[:10]
        x = self.rd(r.p[0], C, ls, cs)[:, 1:-1, 1:-1]
        My, Mx = self._up_matrix(h, mode), self._up_matrix(w, mode)
        y = torch.einsum("Yh,bhwc->bYwc", My, x)
        y = torch.einsum("Xw,bYwc->bYXc", Mx, y)
        self.wr(r.p[1], C, ld, cd, pad_from_interior(y, pm))

    def op_UPSAMPLE2X_BWD(self, r):
        B, h, w, C, lg, cg, ld, cd, do_fold, mode, pm = r.i[:11]
        g = fold(self.rd(r.p[0], C, lg, cg), pm, do_fold)
        My, Mx = self._up_matrix(h, mode), self._up_matrix(w, mode)
        y = torch.einsum("Yh,bYXc->bhXc", My, g)
        y = torch.einsum("Xw,bhXc->bhwc", Mx, y)
        self.wr(r.p[1], C, ld, cd, y, interior_only=True)

    def op_SUBSAMPLE_BIAS(self, r):
        B, H, W, C, ls, cs, ld, cd, pm = r.i[:9]
        x = self.rd(r.p[0], C, ls, cs)[:, 1:-1:2, 1:-1:2]
        if r.p[1] is not None:
            x = x + self.flat(r.p[1])[:C]
        self.wr(r.p[2], C, ld, cd, pad_from_interior(x, pm))

    def op_POOL2X(self, r):
        B, h, w, C, ls, cs, ld, cd, mode, pm = r.i[:10]
        x = self.rd(r.p[0], C, ls, cs)[:, 1:-1, 1:-1]
        q = [x[:, 0::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 0::2], x[:, 1::2, 1::2]]
        y = torch.maximum(torch.maximum(q[0], q[1]), torch.maximum(q[2], q[3])) if mode else 0.25 * ((q[0] + q[1]) + (q[2] + q[3]))
        self.wr(r.p[1], C, ld, cd, pad_from_interior(y, pm))

    def op_POOL2X_BWD(self, r):
        B, h, w, C, lg, cg, lx, cx, ld, cd, do_fold, mode, pm = r.i[:13]
        g = fold(self.rd(r.p[0], C, lg, cg), pm, do_fold)
        out = torch.zeros(B, 2 * h + 2, 2 * w + 2, C)
        inner = out[:, 1:-1, 1:-1]
        sl = [(0, 0), (0, 1), (1, 0), (1, 1)]
        if not mode:
            for (a, b) in sl:
                inner[:, a::2, b::2] = 0.25 * g
        else:
            x = self.rd(r.p[1], C, lx, cx)[:, 1:-1, 1:-1]
            q = torch.stack([x[:, a::2, b::2] for (a, b) in sl])
            arg = torch.zeros_like(q[0], dtype=torch.long)
            m = q[0].clone()
            for k in range(1, 4):                         # first maximum in row-major order wins (torch max_pool2d)
                better = q[k] > m
                arg = torch.where(better, torch.full_like(arg, k), arg)
                m = torch.where(better, q[k], m)
            for k, (a, b) in enumerate(sl):
                inner[:, a::2, b::2] = torch.where(arg == k, g, torch.zeros_like(g))
        self.wr(r.p[2], C, ld, cd, out)

    def op_ZERO_INSERT(self, r):
        B, h, w, C, ls, cs, ld, cd = r.i[:8]
        x = self.rd(r.p[0], C, ls, cs)[:, 1:-1, 1:-1]
        out = torch.zeros(B, 2 * h + 2, 2 * w + 2, C)
        out[:, 1:-1:2, 1:-1:2] = x
        self.wr(r.p[1], C, ld, cd, out)

    def op_PAD_RING(self, r):
        B, H, W, C, ldc, coff, pm = r.i[:7]
        x = self.pv(r.p[0], C, ldc, coff)
        x.copy_(pad_from_interior(x[:, 1:-1, 1:-1].float(), pm).to(x.dtype))

    def op_UNPACK_GRAD(self, r):
        B, H, W, ldc, nch, pm = r.i[:6]
        g = fold(r.p[0].buf.tensor.float(), pm, 1)[..., :nch]
        self.flat(r.p[1])[: B * nch * H * W].copy_(g.permute(0, 3, 1, 2).reshape(-1))

    def op_COPY_PLANES(self, r):
        B, count, ss, ds = r.i[:4]
        s, d = self.flat(r.p[0]), self.flat(r.p[1])
        for n in range(B):
            d[n * ds: n * ds + count] = s[n * ss: n * ss + count]

    def op_ZERO(self, r):
        nbytes = r.i[0] + (r.i[1] << 31)
        t = r.p[0].buf.tensor.view(-1)
        n = nbytes // t.element_size()
        t[r.p[0].off: r.p[0].off + n] = 0

    def op_PREP_WEIGHTS(self, r):
        cout, cin, cin_pad, cout_d = r.i[:4]
        blocked = r.i[4] if len(r.i) > 4 else 0
        if len(r.i) > 7 and r.i[7] == 1:
            return self._prep_transposed(r)
        if len(r.i) > 7 and r.i[7] == 2:
            return self._prep_s2(r)
        cop = r.i[5] if len(r.i) > 5 and r.i[5] else cout          # padded rows / columns are zeros
        kd = r.i[6] if len(r.i) > 6 and r.i[6] else cop
        W = self.flat(r.p[0])[: cout * cin * 9].view(cout, cin, 9).float()
        Wf = torch.zeros(9, cop, cin_pad)
        Wf[:, :cout, :cin] = W.permute(2, 0, 1)
        if blocked & 1:
            Wf = Wf.reshape(9, cop, cin_pad // 16, 16).permute(2, 0, 1, 3).reshape(9, cop, cin_pad)
        r.p[1].buf.tensor.copy_(Wf.to(self.T))
        if r.p[2] is not None:
            Wd = torch.zeros(9, cout_d, kd)
            Wd[:, :cin, :cout] = W.flip(2).permute(2, 1, 0)
            if blocked & 2:
                Wd = Wd.reshape(9, cout_d, kd // 16, 16).permute(2, 0, 1, 3).reshape(9, cout_d, kd)
            r.p[2].buf.tensor.copy_(Wd.to(self.T))

    def _prep_s2(self, r):
        """Conv2d(C, cout, 3, stride 2) as a 2x2-tap convolution over the space-to-depth view: tap (dy, dx), K index (py, px, c) carries
        W[co][c][2dy + py][2dx + px] (zero where an index would be 3); Wd[(ty, tx)][k][co] = Wf[(1 - ty, 1 - tx)][co][k]"""
        cout, C = r.i[:2]
        cop, kd = r.i[5], r.i[6]
        W = self.flat(r.p[0])[: cout * C * 9].view(cout, C, 3, 3).float()
        Wf = torch.zeros(4, cop, 4 * C)
        for dy in range(2):
            for dx in range(2):
                for py in range(2):
                    for px in range(2):
                        ky, kx = 2 * dy + py, 2 * dx + px
                        if ky <= 2 and kx <= 2:
                            ph = py * 2 + px
                            Wf[dy * 2 + dx, :cout, ph * C:(ph + 1) * C] = W[:, :, ky, kx]
        K = 4 * C
        r.p[1].buf.tensor.copy_(Wf.reshape(4, cop, K // 16, 16).permute(2, 0, 1, 3).reshape(4, cop, K).to(self.T))
        if r.p[2] is not None:
            Wd = torch.zeros(4, K, kd)
            for td in range(4):
                Wd[td, :, :cop] = Wf[3 - td].t()
            r.p[2].buf.tensor.copy_(Wd.reshape(4, K, kd // 16, 16).permute(2, 0, 1, 3).reshape(4, K, kd).to(self.T))

    def _prep_transposed(self, r):
        """ConvTranspose2d(k=2, s=2) weight [ci][co][a][b] as a centre-tap convolution into 4*cout channels, phase q = 2a+b outermost"""
        cout, cin, kf, cout_d, blocked = r.i[:5]
        kd = r.i[6] if r.i[6] else 4 * cout
        W = self.flat(r.p[0])[: cin * cout * 4].view(cin, cout, 4).float()
        rows = W.permute(2, 1, 0).reshape(4 * cout, cin)            # [(q, co)][ci]
        Wf = torch.zeros(9, 4 * cout, kf)
        Wf[4, :, :cin] = rows
        if blocked & 1:
            Wf = Wf.reshape(9, 4 * cout, kf // 16, 16).permute(2, 0, 1, 3).reshape(9, 4 * cout, kf)
        r.p[1].buf.tensor.copy_(Wf.to(self.T))
        if r.p[2] is not None:
            Wd = torch.zeros(9, cout_d, kd)
            Wd[4, :cin, :4 * cout] = rows.t()
            if blocked & 2:
                Wd = Wd.reshape(9, cout_d, kd // 16, 16).permute(2, 0, 1, 3).reshape(9, cout_d, kd)
            r.p[2].buf.tensor.copy_(Wd.to(self.T))

    def op_DEPTH2SPACE(self, r):
        B, h, w, C, ls, cs, ld, cd, pm = r.i[:9]
        x = self.rd(r.p[0], 4 * C, ls, cs)[:, 1:-1, 1:-1].reshape(B, h, w, 2, 2, C)
        y = x.permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, C) + self.flat(r.p[1])[:C]
        self.wr(r.p[2], C, ld, cd, pad_from_interior(y, pm))

    def op_SPACE2DEPTH(self, r):
        B, h, w, C, lg, cg, ld, cd, do_fold, pm = r.i[:10]
        g = fold(self.rd(r.p[0], C, lg, cg), pm, do_fold)          # [B, 2h, 2w, C]
        d = g.reshape(B, h, 2, w, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, h, w, 4 * C)
        out = torch.zeros(B, h + 2, w + 2, 4 * C)
        out[:, 1:-1, 1:-1] = d
        self.wr(r.p[1], 4 * C, ld, cd, out)

    def op_UNPREP_WGRAD(self, r):
        cout, cin, cin_w, accumulate = r.i[:4]
        if len(r.i) > 5 and r.i[5] == 1:                            # transposed 2x2: dW[ci][co][q] from the centre tap
            v = r.p[0].buf.tensor[4, :, :cin].reshape(4, cout, cin).permute(2, 1, 0).reshape(-1)
            d = self.flat(r.p[1])[: cin * cout * 4]
            if accumulate:
                d += v
            else:
                d.copy_(v)
            return
        if len(r.i) > 5 and r.i[5] == 2:                            # stride-2 3x3 in 2x2 form: dW[co][c][ky][kx] from tap (ky>>1, kx>>1), phase (ky&1, kx&1)
            t = r.p[0].buf.tensor
            v = torch.zeros(cout, cin, 3, 3)
            for ky in range(3):
                for kx in range(3):
                    ph = (ky & 1) * 2 + (kx & 1)
                    v[:, :, ky, kx] = t[(ky >> 1) * 2 + (kx >> 1), :cout, ph * cin:(ph + 1) * cin]
            d = self.flat(r.p[1])[: cout * cin * 9]
            if accumulate:
                d += v.reshape(-1)
            else:
                d.copy_(v.reshape(-1))
            return
        dWk = r.p[0].buf.tensor[:, :cout, :cin]
        v = dWk.permute(1, 2, 0).reshape(-1)
        d = self.flat(r.p[1])[: cout * cin * 9]
        if accumulate:
            d += v
        else:
            d.copy_(v)

    def op_GRAD_NORM(self, r):
        n = r.i[0] + (r.i[1] << 31)
        self.flat(r.p[1])[0] += (self.flat(r.p[0])[:n].double() ** 2).sum()

    def op_ADAM(self, r):
        n = r.i[0] + (r.i[1] << 31)
        lr, b1, b2, eps, bc1, bc2s, max_norm, gscale = r.f[:8]
        p, g, m, v = (self.flat(r.p[k])[:n] for k in range(4))
        coef = gscale
        if max_norm > 0 and r.p[4] is not None:
            total = float(torch.sqrt(self.flat(r.p[4])[0])) * gscale
            coef *= min(1.0, max_norm / (total + 1e-6))
        gi = g * coef
        m.mul_(b1).add_(gi, alpha=1 - b1)
        v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
        p.sub_((lr / bc1) * m / (v.sqrt() / bc2s + eps))

    # ---- head / loss
    def op_HEAD_FWD(self, r):
        B, H, W, C, ldc, coff, cout, outact = r.i[:8]
        x = self.rd(r.p[0], C, ldc, coff)[:, 1:-1, 1:-1]
        Wh = self.flat(r.p[1])[: cout * C].view(cout, C)
        z = torch.einsum("bhwc,oc->bohw", x, Wh) + self.flat(r.p[2])[:cout][None, :, None, None]
        if outact == 1:
            pred = torch.cat([torch.sigmoid(z[:, :1]), torch.tanh(z[:, 1:])], 1)
        else:
            pred = torch.cat([z[:, :1].clamp(0, 1), z[:, 1:].clamp(-1, 1)], 1)
        self.flat(r.p[3])[: z.numel()].copy_(z.reshape(-1))
        self.flat(r.p[4])[: z.numel()].copy_(pred.reshape(-1))

    def op_HEAD_BWD(self, r):
        B, H, W, C, ldc, coff, ldd, cd, cout, outact = r.i[:10]
        n = B * cout * H * W
        dp = self.flat(r.p[0])[:n].view(B, cout, H, W)
        z = self.flat(r.p[1])[:n].view(B, cout, H, W)
        pred = self.flat(r.p[2])[:n].view(B, cout, H, W)
        if outact == 1:
            dz = torch.cat([dp[:, :1] * pred[:, :1] * (1 - pred[:, :1]), dp[:, 1:] * (1 - pred[:, 1:] ** 2)], 1)
        else:
            lo = torch.tensor([0.0] + [-1.0] * (cout - 1))[None, :, None, None]
            dz = dp * ((z >= lo) & (z <= 1.0)).float()
        x = self.rd(r.p[3], C, ldc, coff)[:, 1:-1, 1:-1]
        Wh = self.flat(r.p[4])[: cout * C].view(cout, C)
        cz = r.i[10] if len(r.i) > 10 and r.i[10] else (C + 7) // 8 * 8     # channels written: zero-padded tail included
        dact = torch.zeros(B, H, W, cz)
        dact[..., :C] = torch.einsum("bohw,oc->bhwc", dz, Wh)
        self.wr(r.p[5], cz, ldd, cd, dact, interior_only=True)
        self.flat(r.p[6])[: cout * C] += torch.einsum("bohw,bhwc->oc", dz, x).reshape(-1)
        self.flat(r.p[7])[:cout] += dz.sum((0, 2, 3))

    def _loss_inputs(self, r):
        B, H, W, ts, s0, se, sc, pm, stencil, flags = r.i[:10]
        HW = H * W
        pred = self.flat(r.p[0])[: B * 3 * HW].view(B, 3, H, W)
        tgt = self.planes(r.p[1], B, ts, 3 * HW).view(B, 3, H, W)
        in0 = self.planes(r.p[2], B, s0, HW).view(B, H, W)
        em = self.planes(r.p[3], B, se, HW).view(B, H, W)
        co = self.planes(r.p[4], B, sc, HW).view(B, H, W)
        return B, H, W, pm, stencil, flags, pred, tgt, in0, em, co

    @staticmethod
    def _dd(f, axis, pm, stencil, dx):
        """f [B,H,W]; derivative along W (axis=2) or H (axis=1) with the loss module's padding rules."""
        n = f.shape[axis]
        idx_p = torch.tensor([pad_src(i + 1, n, pm) for i in range(n)])
        fp = f.index_select(axis, idx_p.clamp(min=0)) * (idx_p >= 0).float().view([-1 if a == axis else 1 for a in range(3)])
        if stencil == 0:
            return (fp - f) / dx
        idx_m = torch.tensor([pad_src(i - 1, n, pm) for i in range(n)])
        fm = f.index_select(axis, idx_m.clamp(min=0)) * (idx_m >= 0).float().view([-1 if a == axis else 1 for a in range(3)])
        return (fp - fm) / (2.0 * dx)

    def _loss_terms(self, r, pred):
        B, H, W, pm, stencil, flags, _, tgt, in0, em, co = self._loss_inputs(r)
        dx = r.f[0]
        fluid = ((em == 0) & (co == 0)).float()
        s = [((pred - tgt) ** 2).sum(), 0, 0, 0, 0, fluid.sum()]
        if flags & 1:
            div = self._dd(pred[:, 1], 2, pm, stencil, dx) + self._dd(pred[:, 2], 1, pm, stencil, dx)
            s[1] = (div.abs() * fluid).sum()
        if flags & 2:
            s[2] = ((self._dd(pred[:, 0], 2, pm, stencil, dx) - self._dd(tgt[:, 0], 2, pm, stencil, dx)).abs() * fluid).sum()
            s[3] = ((self._dd(pred[:, 0], 1, pm, stencil, dx) - self._dd(tgt[:, 0], 1, pm, stencil, dx)).abs() * fluid).sum()
        if flags & 4:
            allowed = (em > 0.01) | (in0 > 0.01)
            s[4] = (pred[:, 0] * (~allowed).float()).sum()
        return s

    def op_LOSS_REDUCE(self, r):
        pred = self._loss_inputs(r)[6]
        s = self._loss_terms(r, pred)
        out = self.flat(r.p[5])
        for k in range(6):
            out[k] += float(s[k])

    def op_LOSS_FINALIZE(self, r):
        B, H, W, flags, world = r.i[:5]
        wm, wd, wg, we, wk = r.f[:5]
        sums = self.flat(r.p[0])
        cnt = float(self.flat(r.p[1])[0]) if r.p[1] is not None else float(sums[5])
        inv = 0.0 if cnt < 1e-8 else world / (cnt + 1e-8)
        mse = float(sums[0]) / (B * 3 * H * W)
        dv = float(sums[1]) * inv if flags & 1 else 0.0
        gr = (float(sums[2]) + float(sums[3])) * inv if flags & 2 else 0.0
        em = float(sums[4]) / (B * H * W) if flags & 4 else 0.0
        total = wm * mse + wd * dv + wg * gr + we * em
        t = self.flat(r.p[2])
        for k, v in enumerate((mse, dv, gr, em, total, float(sums[5]))):
            t[k] = v
        if r.p[3] is not None:
            a = self.flat(r.p[3])
            for k, v in enumerate((mse, dv, gr, em, total)):
                a[k] += wk * v

    def op_LOSS_GRAD(self, r):
        B, H, W, pm, stencil, flags, pred, tgt, in0, em, co = self._loss_inputs(r)
        world, dn_stride = r.i[10], r.i[11]
        dx, wm, wd, wg, we, wk = r.f[:6]
        sums = self.flat(r.p[5])
        cnt = float(self.flat(r.p[6])[0]) if r.p[6] is not None else float(sums[5])
        inv = 0.0 if cnt < 1e-8 else world / (cnt + 1e-8)
        gs = (float(self.flat(r.p[8])[0]) if r.p[8] is not None else 1.0) * wk
        p = pred.clone().requires_grad_(True)
        s = self._loss_terms(r, p)
        loss = wm * s[0] / (B * 3 * H * W)
        if flags & 1:
            loss = loss + wd * s[1] * inv
        if flags & 2:
            loss = loss + wg * (s[2] + s[3]) * inv
        if flags & 4:
            loss = loss + we * s[4] / (B * H * W)
        (g,) = torch.autograd.grad(loss * gs, p)
        g = g.clone()
        if r.p[7] is not None:
            D = self.planes(r.p[7], B, dn_stride, 4 * H * W).view(B, 4, H, W)
            g[:, 0] += D[:, 0] + D[:, 3]
            g[:, 1] += D[:, 1]
            g[:, 2] += D[:, 2]
        self.flat(r.p[9])[: g.numel()].copy_(g.reshape(-1))

    def op_STENCIL(self, r):
        B, H, W, pm, stencil, mode = r.i[:6]
        dx, dy = r.f[:2]
        f0 = self.flat(r.p[0])[: B * H * W].view(B, H, W)
        if mode == 0:
            self.flat(r.p[2])[: B * H * W].copy_(self._dd(f0, 2, pm, stencil, dx).reshape(-1))
            self.flat(r.p[3])[: B * H * W].copy_(self._dd(f0, 1, pm, stencil, dy).reshape(-1))
        else:
            f1 = self.flat(r.p[1])[: B * H * W].view(B, H, W)
            self.flat(r.p[2])[: B * H * W].copy_((self._dd(f0, 2, pm, stencil, dx) + self._dd(f1, 1, pm, stencil, dy)).reshape(-1))
