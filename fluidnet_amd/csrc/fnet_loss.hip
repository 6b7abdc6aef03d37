// Physics loss (ml/training/physics_loss.py) forward reduction + analytic gradient, the 1x1 head conv with
// its output activation (ml/models/unet.py:294-303), and the finite-difference stencils as free functions.
// All field planes are fp32 NCHW exactly like the reference's tensors; reductions are wavefront-64 shuffles,
// one double atomic per block.
#include <stdlib.h>

#include "fnet_common.h"

namespace {

constexpr int kThreads = 256;
constexpr float kLossEps = 1e-8f;  // physics_loss.py:101,136

struct Plane {
    const float* p;
    int bstride;  // elements between batch items
};

// value of the (conceptually padded) plane at (y,x), y in [-1,H], x in [-1,W]
__device__ __forceinline__ float pat(const float* pl, int y, int x, int H, int W, int pm) {
    const int sy = pad_src(y, H, pm), sx = pad_src(x, W, pm);
    if (sy < 0 || sx < 0) return 0.f;
    return pl[(size_t)sy * W + sx];
}
// d/dx and d/dy of plane pl at interior (y,x)   (physics_loss.py:35-44)
__device__ __forceinline__ float ddx(const float* pl, int y, int x, int H, int W, int pm, int stencil, float dx) {
    if (stencil == FNET_STENCIL_FORWARD) return (pat(pl, y, x + 1, H, W, pm) - pl[(size_t)y * W + x]) / dx;
    return (pat(pl, y, x + 1, H, W, pm) - pat(pl, y, x - 1, H, W, pm)) / (2.0f * dx);
}
__device__ __forceinline__ float ddy(const float* pl, int y, int x, int H, int W, int pm, int stencil, float dy) {
    if (stencil == FNET_STENCIL_FORWARD) return (pat(pl, y + 1, x, H, W, pm) - pl[(size_t)y * W + x]) / dy;
    return (pat(pl, y + 1, x, H, W, pm) - pat(pl, y - 1, x, H, W, pm)) / (2.0f * dy);
}
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

struct LossArgs {
    const float* pred;    // [B,3,H,W] contiguous
    Plane tgt;            // 3 planes per item, contiguous within the item
    Plane in0, emit, coll;
    int B, H, W, pm, stencil, flags;  // flags: bit0 divergence, bit1 gradient, bit2 emitter
    float dx;
};

__device__ __forceinline__ float fluid_at(const LossArgs& a, int n, int y, int x) {
    const size_t o = (size_t)y * a.W + x;
    return (a.emit.p[(size_t)n * a.emit.bstride + o] == 0.f && a.coll.p[(size_t)n * a.coll.bstride + o] == 0.f) ? 1.f : 0.f;
}

// sums[0]=sum (p-t)^2, [1]=sum |div| fluid, [2]=sum |gx_p-gx_t| fluid, [3]=same for y, [4]=sum d_pred*forbidden, [5]=sum fluid
__global__ __launch_bounds__(256) void k_loss_reduce(LossArgs a, double* sums) {
    __shared__ double red[4 * 6];
    const int HW = a.H * a.W, total = a.B * HW;
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int idx = blockIdx.x * kThreads + threadIdx.x; idx < total; idx += gridDim.x * kThreads) {
        const int n = idx / HW, r = idx - n * HW, y = r / a.W, x = r - y * a.W;
        const float* P = a.pred + (size_t)n * 3 * HW;
        const float* T = a.tgt.p + (size_t)n * a.tgt.bstride;
        float e = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = P[c * HW + r] - T[c * HW + r]; e += d * d; }
        s[0] += e;
        const float fl = fluid_at(a, n, y, x);
        s[5] += fl;
        if (a.flags & 1) {
            const float dv = ddx(P + HW, y, x, a.H, a.W, a.pm, a.stencil, a.dx) + ddy(P + 2 * HW, y, x, a.H, a.W, a.pm, a.stencil, a.dx);
            s[1] += fabsf(dv) * fl;
        }
        if (a.flags & 2) {
            const float gx = ddx(P, y, x, a.H, a.W, a.pm, a.stencil, a.dx) - ddx(T, y, x, a.H, a.W, a.pm, a.stencil, a.dx);
            const float gy = ddy(P, y, x, a.H, a.W, a.pm, a.stencil, a.dx) - ddy(T, y, x, a.H, a.W, a.pm, a.stencil, a.dx);
            s[2] += fabsf(gx) * fl;
            s[3] += fabsf(gy) * fl;
        }
        if (a.flags & 4) {
            const bool allowed = (a.emit.p[(size_t)n * a.emit.bstride + r] > 0.01f) || (a.in0.p[(size_t)n * a.in0.bstride + r] > 0.01f);
            s[4] += allowed ? 0.f : P[r];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = wave_sum_d(s[k]);
    if (lane == 0)
        for (int k = 0; k < 6; ++k) red[wave * 6 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        atomicAdd(sums + k, red[k] + red[6 + k] + red[12 + k] + red[18 + k]);
    }
}

// denominators of the masked means; DP passes the all-reduced global fluid count and its world size so that the
// plain gradient average over ranks equals the global-batch gradient.
__device__ __forceinline__ void masked_den(const double* sums, const double* gcnt, int world, float& inv_den, bool& zero) {
    const double cnt = gcnt ? *gcnt : sums[5];
    zero = cnt < (double)kLossEps;                       // physics_loss.py:119-121,153-155
    inv_den = zero ? 0.f : (float)((double)world / (cnt + (double)kLossEps));
}

// terms: [0] mse [1] divergence [2] gradient [3] emitter [4] total [5] fluid count;  accum[0..4] += wk * terms
__global__ __launch_bounds__(256) void k_loss_finalize(const double* sums, const double* gcnt, float* terms, float* accum, int B, int H, int W, int flags,
                                int world, float w_mse, float w_div, float w_grad, float w_emit, float wk) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float inv_den;
    bool zero;
    masked_den(sums, gcnt, world, inv_den, zero);
    const float mse = (float)(sums[0] / ((double)B * 3.0 * H * W));
    const float dv = (flags & 1) ? (float)(sums[1] * (double)inv_den) : 0.f;
    const float gr = (flags & 2) ? (float)(sums[2] * (double)inv_den) + (float)(sums[3] * (double)inv_den) : 0.f;
    const float em = (flags & 4) ? (float)(sums[4] / ((double)B * H * W)) : 0.f;
    const float total = w_mse * mse + w_div * dv + w_grad * gr + w_emit * em;
    terms[0] = mse; terms[1] = dv; terms[2] = gr; terms[3] = em; terms[4] = total; terms[5] = (float)sums[5];
    if (accum) {
        accum[0] += wk * mse; accum[1] += wk * dv; accum[2] += wk * gr; accum[3] += wk * em; accum[4] += wk * total;
    }
}

// masked sign fields S(y,x) whose adjoint stencil gives the gradient
__device__ __forceinline__ float s_div(const LossArgs& a, const float* P, int n, int y, int x, int HW) {
    const float fl = fluid_at(a, n, y, x);
    if (fl == 0.f) return 0.f;
    return sgn(ddx(P + HW, y, x, a.H, a.W, a.pm, a.stencil, a.dx) + ddy(P + 2 * HW, y, x, a.H, a.W, a.pm, a.stencil, a.dx));
}
__device__ __forceinline__ float s_gx(const LossArgs& a, const float* P, const float* T, int n, int y, int x) {
    const float fl = fluid_at(a, n, y, x);
    if (fl == 0.f) return 0.f;
    return sgn(ddx(P, y, x, a.H, a.W, a.pm, a.stencil, a.dx) - ddx(T, y, x, a.H, a.W, a.pm, a.stencil, a.dx));
}
__device__ __forceinline__ float s_gy(const LossArgs& a, const float* P, const float* T, int n, int y, int x) {
    const float fl = fluid_at(a, n, y, x);
    if (fl == 0.f) return 0.f;
    return sgn(ddy(P, y, x, a.H, a.W, a.pm, a.stencil, a.dx) - ddy(T, y, x, a.H, a.W, a.pm, a.stencil, a.dx));
}

// Adjoint of the 1-D difference along one axis: returns sum_{i in plus_pre(j)} S(i) - sum_{i in minus_pre(j)} S(i),
// where the stencil at i reads index pad_src(i+1) with + and (forward: i | central: pad_src(i-1)) with -.
template <typename F>
__device__ __forceinline__ float adj1d(int j, int n, int pm, int stencil, F S) {
    float acc = 0.f;
    if (j >= 1) acc += S(j - 1);                                   // i+1 == j
    if (pad_src(n, n, pm) == j) acc += S(n - 1);                   // right/bottom pad copy
    if (stencil == FNET_STENCIL_FORWARD) {
        acc -= S(j);
    } else {
        if (j + 1 <= n - 1) acc -= S(j + 1);                       // i-1 == j
        if (pad_src(-1, n, pm) == j) acc -= S(0);                  // left/top pad copy
    }
    return acc;
}

// dpred[n,c,y,x] = gscale*wk*dL_k/dpred + (gradient arriving from the next rollout step's input)
struct LossGradArgs {
    const double* sums;
    const double* gcnt;
    const float* dnext;
    int dnext_bstride;
    const float* gscale_ptr;
    int world;
    float w_mse, w_div, w_grad, w_emit, wk;
};
// the three gradient values of pixel (n, y, x): shared by k_loss_grad and k_loss_head_bwd
__device__ __forceinline__ void loss_grad_px(const LossArgs& a, const LossGradArgs& q, int n, int y, int x, float& g0, float& g1, float& g2) {
    const int HW = a.H * a.W, r = y * a.W + x;
    const float* P = a.pred + (size_t)n * 3 * HW;
    const float* T = a.tgt.p + (size_t)n * a.tgt.bstride;
    float inv_den;
    bool zero;
    masked_den(q.sums, q.gcnt, q.world, inv_den, zero);
    const float gs = (q.gscale_ptr ? *q.gscale_ptr : 1.f) * q.wk;
    const float cm = gs * q.w_mse * 2.f / ((float)a.B * 3.f * (float)HW);
    g0 = cm * (P[r] - T[r]);
    g1 = cm * (P[HW + r] - T[HW + r]);
    g2 = cm * (P[2 * HW + r] - T[2 * HW + r]);
    const float sc = (a.stencil == FNET_STENCIL_FORWARD) ? 1.f / a.dx : 0.5f / a.dx;
    if ((a.flags & 1) && !zero) {
        const float c = gs * q.w_div * inv_den * sc;
        g1 += c * adj1d(x, a.W, a.pm, a.stencil, [&](int i) { return s_div(a, P, n, y, i, HW); });
        g2 += c * adj1d(y, a.H, a.pm, a.stencil, [&](int i) { return s_div(a, P, n, i, x, HW); });
    }
    if ((a.flags & 2) && !zero) {
        const float c = gs * q.w_grad * inv_den * sc;
        g0 += c * adj1d(x, a.W, a.pm, a.stencil, [&](int i) { return s_gx(a, P, T, n, y, i); });
        g0 += c * adj1d(y, a.H, a.pm, a.stencil, [&](int i) { return s_gy(a, P, T, n, i, x); });
    }
    if (a.flags & 4) {
        const bool allowed = (a.emit.p[(size_t)n * a.emit.bstride + r] > 0.01f) || (a.in0.p[(size_t)n * a.in0.bstride + r] > 0.01f);
        if (!allowed) g0 += gs * q.w_emit / ((float)a.B * (float)HW);
    }
    if (q.dnext) {
        // next input = cat[pred(3), pred[:,0:1], masks]  (trainer.py:374-379): channels 0 and 3 both hit density
        const float* D = q.dnext + (size_t)n * q.dnext_bstride;
        g0 += D[r] + D[3 * HW + r];
        g1 += D[HW + r];
        g2 += D[2 * HW + r];
    }
}
__global__ __launch_bounds__(256) void k_loss_grad(LossArgs a, LossGradArgs q, float* dpred) {
    const int HW = a.H * a.W, total = a.B * HW;
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= total) return;
    const int n = idx / HW, r = idx - n * HW, y = r / a.W, x = r - y * a.W;
    float g0, g1, g2;
    loss_grad_px(a, q, n, y, x, g0, g1, g2);
    float* O = dpred + (size_t)n * 3 * HW;
    O[r] = g0; O[HW + r] = g1; O[2 * HW + r] = g2;
}

// ------------------------------------------------------------------ STENCIL free functions
__global__ __launch_bounds__(256) void k_stencil(const float* f0, const float* f1, float* o0, float* o1, int B, int H, int W, int pm, int stencil,
                          float dx, float dy, int mode) {
    const int HW = H * W, total = B * HW;
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= total) return;
    const int n = idx / HW, r = idx - n * HW, y = r / W, x = r - y * W;
    const float* a = f0 + (size_t)n * HW;
    if (mode == 0) {  // compute_spatial_gradients
        o0[idx] = ddx(a, y, x, H, W, pm, stencil, dx);
        o1[idx] = ddy(a, y, x, H, W, pm, stencil, dy);
    } else {  // compute_divergence
        const float* b = f1 + (size_t)n * HW;
        o0[idx] = ddx(a, y, x, H, W, pm, stencil, dx) + ddy(b, y, x, H, W, pm, stencil, dy);
    }
}

// ------------------------------------------------------------------ HEAD (1x1 conv + output activation)
template <typename T, int MAXO>
__global__ __launch_bounds__(256) void k_head_fwd(const T* src, const float* Wh, const float* bh, float* z, float* pred, int B, int H, int W, int C,
                           int ldc, int coff, int Cout, int outact) {
    extern __shared__ float sw[];  // [Cout][C] + [Cout]
    for (int i = threadIdx.x; i < Cout * C; i += kThreads) sw[i] = Wh[i];
    for (int i = threadIdx.x; i < Cout; i += kThreads) sw[Cout * C + i] = bh[i];
    __syncthreads();
    const int HW = H * W, total = B * HW;
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= total) return;
    const int n = idx / HW, r = idx - n * HW, y = r / W, x = r - y * W;
    const T* s = src + (((size_t)n * (H + 2) + y + 1) * (W + 2) + x + 1) * ldc + coff;
    float acc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) acc[o] = o < Cout ? sw[Cout * C + o] : 0.f;
    for (int c0 = 0; c0 < C; c0 += 8) {
        Vec8<T> v;
        v.load(s + c0);
#pragma unroll
        for (int o = 0; o < MAXO; ++o)
            if (o < Cout) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[o] += (c0 + k < C) ? v.v[k] * sw[o * C + c0 + k] : 0.f;   // C % 8 != 0: zero-padded tail
            }
    }
#pragma unroll
    for (int o = 0; o < MAXO; ++o)
        if (o < Cout) {
            const float zz = acc[o];
            float pv;
            if (outact == FNET_OUT_SIGMOID_TANH) pv = (o == 0) ? 1.f / (1.f + __expf(-zz)) : tanhf(zz);
            else pv = (o == 0) ? fminf(fmaxf(zz, 0.f), 1.f) : fminf(fmaxf(zz, -1.f), 1.f);
            const size_t oi = ((size_t)n * Cout + o) * HW + r;
            z[oi] = zz;
            pred[oi] = pv;
        }
}

// backward: dz = dpred * gate; d_act[c] = sum_o Wh[o][c] dz[o]; dWh[o][c] += sum dz[o] act[c]; dbh[o] += sum dz[o]
template <typename T, int MAXO, int MAXC>
__global__ __launch_bounds__(256) void k_head_bwd(const float* dpred, const float* z, const float* pred, const T* act, const float* Wh, T* dact,
                           float* dWh, float* dbh, int B, int H, int W, int C, int ldc, int coff, int ldc_d, int coff_d, int Cout,
                           int outact, int Cz /* channels of dact written: C rounded up to the plan's channel padding */) {
    extern __shared__ float sw[];  // [Cout][C] weights, then [4 waves][Cout*C + Cout] reduction scratch
    for (int i = threadIdx.x; i < Cout * C; i += kThreads) sw[i] = Wh[i];
    __syncthreads();
    float* red = sw + Cout * C;
    const int HW = H * W, total = B * HW;
    float gw[MAXO][MAXC];
    float gb[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
        gb[o] = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) gw[o][c] = 0.f;
    }
    for (int idx = blockIdx.x * kThreads + threadIdx.x; idx < total; idx += gridDim.x * kThreads) {
        const int n = idx / HW, r = idx - n * HW, y = r / W, x = r - y * W;
        float dz[MAXO];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            dz[o] = 0.f;
            if (o < Cout) {
                const size_t oi = ((size_t)n * Cout + o) * HW + r;
                const float g = dpred[oi], zz = z[oi], pv = pred[oi];
                if (outact == FNET_OUT_SIGMOID_TANH) dz[o] = (o == 0) ? g * pv * (1.f - pv) : g * (1.f - pv * pv);
                else {
                    const float lo = (o == 0) ? 0.f : -1.f;
                    dz[o] = (zz >= lo && zz <= 1.f) ? g : 0.f;       // torch.clamp backward: inclusive bounds
                }
                gb[o] += dz[o];
            }
        }
        const size_t pix = ((size_t)n * (H + 2) + y + 1) * (W + 2) + x + 1;
#pragma unroll
        for (int cg = 0; cg < MAXC / 8; ++cg) {
            if (cg * 8 < Cz) {
                Vec8<T> v, d;
                if (cg * 8 < C) v.load(act + pix * ldc + coff + cg * 8);
                else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v.v[k] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float acc = 0.f;
                    const bool real = cg * 8 + k < C;       // zero-padded tail channels get a zero gradient (they must stay zero)
#pragma unroll
                    for (int o = 0; o < MAXO; ++o)
                        if (o < Cout && real) {
                            acc += sw[o * C + cg * 8 + k] * dz[o];
                            gw[o][cg * 8 + k] += dz[o] * v.v[k];
                        }
                    d.v[k] = acc;
                }
                d.store(dact + pix * ldc_d + coff_d + cg * 8);
            }
        }
    }
    // block reduction of gw/gb -> atomics
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nred = Cout * C + Cout;
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
        if (o < Cout) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
                    const float s = wave_sum(gw[o][c]);
                    if (lane == 0) red[wave * nred + o * C + c] = s;
                }
            const float s = wave_sum(gb[o]);
            if (lane == 0) red[wave * nred + Cout * C + o] = s;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nred; i += kThreads) {
        const float s = red[i] + red[nred + i] + red[2 * nred + i] + red[3 * nred + i];
        if (i < Cout * C) atomicAdd(dWh + i, s);
        else atomicAdd(dbh + (i - Cout * C), s);
    }
}

// Channel-group form of k_head_bwd for CG = Cz / 8 in {1, 2, 4, 8}: CG consecutive lanes share a pixel and each owns 8 channels of it,
// so a wave reads and writes 64 / CG whole pixels as one contiguous run (16 B per lane) and a thread carries Cout x 8 weight-gradient
// accumulators instead of Cout x Cz (the pixel-per-thread form above ran 2 blocks per CU at ~1.3 TB/s).  The three (dpred, z, pred)
// planes are read by all CG lanes of a pixel (same address: one request).
template <typename T, int MAXO, int CG, bool EXACT /* Cout == MAXO: no run-time channel-count tests in the unrolled loops */>
__global__ __launch_bounds__(256) void k_head_bwd_cg(const float* dpred, const float* z, const float* pred, const T* act, const float* Wh, T* dact,
                                                      float* dWh, float* dbh, int B, int H, int W, int C, int ldc, int coff, int ldc_d, int coff_d,
                                                      int Cout, int outact) {
    extern __shared__ float sw[];  // [4 waves][Cout * CG*8 + Cout] reduction scratch
    const int cg = threadIdx.x & (CG - 1);
    const int HW = H * W;
    const long long total = (long long)B * HW * CG;
    float w8[MAXO][8], gw[MAXO][8], gb[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
        gb[o] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            gw[o][k] = 0.f;
            w8[o][k] = ((EXACT || o < Cout) && cg * 8 + k < C) ? Wh[o * C + cg * 8 + k] : 0.f;   // zero-padded tail channels: zero gradient
        }
    }
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (long long)gridDim.x * kThreads) {
        const int p = (int)(idx / CG);
        const int n = p / HW, r = p - n * HW, y = r / W, x = r - y * W;
        float dz[MAXO];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
            dz[o] = 0.f;
            if (EXACT || o < Cout) {
                const size_t oi = ((size_t)n * Cout + o) * HW + r;
                const float g = dpred[oi], zz = z[oi], pv = pred[oi];
                if (outact == FNET_OUT_SIGMOID_TANH) dz[o] = (o == 0) ? g * pv * (1.f - pv) : g * (1.f - pv * pv);
                else {
                    const float lo = (o == 0) ? 0.f : -1.f;
                    dz[o] = (zz >= lo && zz <= 1.f) ? g : 0.f;       // torch.clamp backward: inclusive bounds
                }
                gb[o] += dz[o];
            }
        }
        const size_t pix = ((size_t)n * (H + 2) + y + 1) * (W + 2) + x + 1;
        Vec8<T> v, d;
        if (cg * 8 < C) v.load(act + pix * ldc + coff + cg * 8);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v.v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (EXACT || o < Cout) {
                    acc += w8[o][k] * dz[o];
                    gw[o][k] += dz[o] * v.v[k];
                }
            d.v[k] = acc;
        }
        d.store(dact + pix * ldc_d + coff_d + cg * 8);
    }
    // lanes l, l + CG, l + 2 CG, ... own the same channel group: butterfly over those lane bits, then lanes 0..CG-1 hold the wave's sums
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nred = Cout * C + Cout;
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
        if (EXACT || o < Cout) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float s = gw[o][k];
#pragma unroll
                for (int off = CG; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
                if (lane < CG && lane * 8 + k < C) sw[wave * nred + o * C + lane * 8 + k] = s;
            }
            float s = gb[o];                 // every lane of a pixel counted dz once: keep group 0 only
#pragma unroll
            for (int off = CG; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
            if (lane == 0) sw[wave * nred + Cout * C + o] = s;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nred; i += kThreads) {
        const float s = sw[i] + sw[nred + i] + sw[2 * nred + i] + sw[3 * nred + i];
        if (i < Cout * C) atomicAdd(dWh + i, s);
        else atomicAdd(dbh + (i - Cout * C), s);
    }
}

// ------------------------------------------------------------------ head + clamp + loss in one pass
// HEAD_FWD immediately followed by the LOSS_REDUCE that reads its prediction, as ONE launch: a block owns R rows of one image, computes
// the 1x1 head conv + output activation for those rows and the two halo rows the stencils reach (through the loss's padding map: a
// replicated / reflected / wrapped row is the head of its source row, a zero-padded one is zero), keeps the three prediction planes
// of the R + 2 rows in LDS, writes z / pred of its own rows and accumulates the loss sums from LDS: the prediction planes are not
// read back from memory and two launches become one.  Per pixel the arithmetic is that of k_head_fwd and k_loss_reduce (same order),
// so z and pred are bit-identical and the fp64 sums differ only in summation order.  Replaces ml/models/unet.py:294-303 +
// ml/training/physics_loss.py:210-267 for the training plan's (prediction, loss) pair.
template <typename T>
__global__ __launch_bounds__(256) void k_head_loss_fwd(const T* __restrict__ src, const float* __restrict__ Wh, const float* __restrict__ bh,
                                                       float* __restrict__ z, float* __restrict__ pred, LossArgs a, double* sums, int C, int ldc,
                                                       int coff, int outact, int R) {
    extern __shared__ float sm[];          // [3][(R + 2) * W] prediction tile, [3 * C + 3] weights and bias
    __shared__ double red[4 * 6];
    const int W = a.W, H = a.H, HW = H * W;
    const int nrb = (H + R - 1) / R;
    const int n = blockIdx.x / nrb, y0 = (blockIdx.x - n * nrb) * R;
    const int rows = min(R, H - y0);
    const int TW = (R + 2) * W;
    float* tp = sm;
    float* sw = sm + 3 * TW;
    for (int i = threadIdx.x; i < 3 * C + 3; i += kThreads) sw[i] = i < 3 * C ? Wh[i] : bh[i - 3 * C];
    __syncthreads();
    for (int idx = threadIdx.x; idx < (rows + 2) * W; idx += kThreads) {
        const int lr = idx / W, x = idx - lr * W;
        const int y = y0 - 1 + lr;                        // -1 .. H
        const int sy = pad_src(y, H, a.pm);
        float pv[3] = {0.f, 0.f, 0.f};
        if (sy >= 0) {
            const T* sp = src + (((size_t)n * (H + 2) + sy + 1) * (W + 2) + x + 1) * ldc + coff;
            float acc[3] = {sw[3 * C], sw[3 * C + 1], sw[3 * C + 2]};
            for (int c0 = 0; c0 < C; c0 += 8) {
                Vec8<T> v;
                v.load(sp + c0);
#pragma unroll
                for (int o = 0; o < 3; ++o) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[o] += v.v[k] * sw[o * C + c0 + k];
                }
            }
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float zz = acc[o];
                if (outact == FNET_OUT_SIGMOID_TANH) pv[o] = (o == 0) ? 1.f / (1.f + __expf(-zz)) : tanhf(zz);
                else pv[o] = (o == 0) ? fminf(fmaxf(zz, 0.f), 1.f) : fminf(fmaxf(zz, -1.f), 1.f);
                if (lr >= 1 && lr <= rows) {
                    const size_t oi = ((size_t)n * 3 + o) * HW + (size_t)y * W + x;
                    z[oi] = zz;
                    pred[oi] = pv[o];
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) tp[o * TW + idx] = pv[o];
    }
    __syncthreads();
    // the loss sums of this block's rows: P(c, local row, x) from LDS; local row lr <-> image row y0 - 1 + lr, already padding-mapped
    auto Px = [&](int c, int lr, int xx) -> float {      // x through the padding map, -1 <= xx <= W
        const int sx = pad_src(xx, W, a.pm);
        return sx < 0 ? 0.f : tp[c * TW + lr * W + sx];
    };
    auto dxp = [&](int c, int lr, int x) -> float {
        if (a.stencil == FNET_STENCIL_FORWARD) return (Px(c, lr, x + 1) - tp[c * TW + lr * W + x]) / a.dx;
        return (Px(c, lr, x + 1) - Px(c, lr, x - 1)) / (2.0f * a.dx);
    };
    auto dyp = [&](int c, int lr, int x) -> float {
        if (a.stencil == FNET_STENCIL_FORWARD) return (tp[c * TW + (lr + 1) * W + x] - tp[c * TW + lr * W + x]) / a.dx;
        return (tp[c * TW + (lr + 1) * W + x] - tp[c * TW + (lr - 1) * W + x]) / (2.0f * a.dx);
    };
    double s[6] = {0, 0, 0, 0, 0, 0};
    const float* Tg = a.tgt.p + (size_t)n * a.tgt.bstride;
    for (int idx = threadIdx.x; idx < rows * W; idx += kThreads) {
        const int ly = idx / W, x = idx - ly * W, y = y0 + ly, lr = ly + 1, r = y * W + x;
        float e = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = tp[c * TW + lr * W + x] - Tg[c * HW + r]; e += d * d; }
        s[0] += e;
        const float fl = fluid_at(a, n, y, x);
        s[5] += fl;
        if (a.flags & 1) s[1] += fabsf(dxp(1, lr, x) + dyp(2, lr, x)) * fl;
        if (a.flags & 2) {
            const float gx = dxp(0, lr, x) - ddx(Tg, y, x, H, W, a.pm, a.stencil, a.dx);
            const float gy = dyp(0, lr, x) - ddy(Tg, y, x, H, W, a.pm, a.stencil, a.dx);
            s[2] += fabsf(gx) * fl;
            s[3] += fabsf(gy) * fl;
        }
        if (a.flags & 4) {
            const bool allowed = (a.emit.p[(size_t)n * a.emit.bstride + r] > 0.01f) || (a.in0.p[(size_t)n * a.in0.bstride + r] > 0.01f);
            s[4] += allowed ? 0.f : tp[lr * W + x];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] = wave_sum_d(s[k]);
    if (lane == 0)
        for (int k = 0; k < 6; ++k) red[wave * 6 + k] = s[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        atomicAdd(sums + k, red[k] + red[6 + k] + red[12 + k] + red[18 + k]);
    }
}

// LOSS_GRAD immediately followed by the HEAD_BWD that consumes its dpred, as ONE launch: a block owns R rows of one image; phase A forms
// the loss gradient of its pixels (loss_grad_px, as k_loss_grad), writes dpred (kept: the record says it is written) and leaves the gated
// dz = dpred * clamp' in LDS; phase B is k_head_bwd_cg's channel-group walk over the same pixels with dz from LDS.  dpred, dz and the
// activation gradient are bit-identical to the separate launches; the head weight / bias gradients differ in fp32 summation order.
template <typename T, int CG>
__global__ __launch_bounds__(256) void k_loss_head_bwd(LossArgs a, LossGradArgs q, float* __restrict__ dpred, const float* __restrict__ z,
                                                       const T* __restrict__ act, const float* __restrict__ Wh, T* __restrict__ dact, float* dWh,
                                                       float* dbh, int C, int ldc, int coff, int ldc_d, int coff_d, int outact, int R) {
    extern __shared__ float sm[];          // [3][R * W] dz tile, then [4 waves][3 * C + 3] reduction scratch
    const int W = a.W, H = a.H, HW = H * W;
    const int nrb = (H + R - 1) / R;
    const int n = blockIdx.x / nrb, y0 = (blockIdx.x - n * nrb) * R;
    const int rows = min(R, H - y0);
    const int TP = R * W;
    float* dzt = sm;
    float* sw = sm + 3 * TP;
    for (int idx = threadIdx.x; idx < rows * W; idx += kThreads) {
        const int ly = idx / W, x = idx - ly * W, y = y0 + ly, r = y * W + x;
        float g[3];
        loss_grad_px(a, q, n, y, x, g[0], g[1], g[2]);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const size_t oi = ((size_t)n * 3 + o) * HW + r;
            dpred[oi] = g[o];
            const float zz = z[oi], pv = a.pred[oi];
            float dz;
            if (outact == FNET_OUT_SIGMOID_TANH) dz = (o == 0) ? g[o] * pv * (1.f - pv) : g[o] * (1.f - pv * pv);
            else {
                const float lo = (o == 0) ? 0.f : -1.f;
                dz = (zz >= lo && zz <= 1.f) ? g[o] : 0.f;       // torch.clamp backward: inclusive bounds
            }
            dzt[o * TP + idx] = dz;
        }
    }
    __syncthreads();
    const int cg = threadIdx.x & (CG - 1);
    float w8[3][8], gw[3][8], gb[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        gb[o] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            gw[o][k] = 0.f;
            w8[o][k] = (cg * 8 + k < C) ? Wh[o * C + cg * 8 + k] : 0.f;   // zero-padded tail channels: zero gradient
        }
    }
    for (int item = threadIdx.x; item < rows * W * CG; item += kThreads) {
        const int p = item / CG;
        const int ly = p / W, x = p - ly * W, y = y0 + ly;
        float dz[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            dz[o] = dzt[o * TP + p];
            gb[o] += dz[o];
        }
        const size_t pix = ((size_t)n * (H + 2) + y + 1) * (W + 2) + x + 1;
        Vec8<T> v, d;
        if (cg * 8 < C) v.load(act + pix * ldc + coff + cg * 8);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v.v[k] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                acc += w8[o][k] * dz[o];
                gw[o][k] += dz[o] * v.v[k];
            }
            d.v[k] = acc;
        }
        d.store(dact + pix * ldc_d + coff_d + cg * 8);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nred = 3 * C + 3;
#pragma unroll
    for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float sv = gw[o][k];
#pragma unroll
            for (int off = CG; off < 64; off <<= 1) sv += __shfl_xor(sv, off, 64);
            if (lane < CG && lane * 8 + k < C) sw[wave * nred + o * C + lane * 8 + k] = sv;
        }
        float sv = gb[o];                 // every lane of a pixel counted dz once: keep group 0 only
#pragma unroll
        for (int off = CG; off < 64; off <<= 1) sv += __shfl_xor(sv, off, 64);
        if (lane == 0) sw[wave * nred + 3 * C + o] = sv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nred; i += kThreads) {
        const float sv = sw[i] + sw[nred + i] + sw[2 * nred + i] + sw[3 * nred + i];
        if (i < 3 * C) atomicAdd(dWh + i, sv);
        else atomicAdd(dbh + (i - 3 * C), sv);
    }
}

// ------------------------------------------------------------------ validation metrics (ml/training/metrics.py)
// One pass over (pred, target, masks) for every sum the reference's metric functions need:
//   m[0..2] sum (p_c - t_c)^2 per channel   (compute_per_channel_mse, metrics.py:7-12)
//   m[3], m[4] sum |d/dx (p0 - t0)|, |d/dy ...|  (compute_gradient_l1, :24-38; unmasked means)
//   m[5] sum div(vx,vy)^2                     (compute_divergence_norm, :41-51)
//   m[6] sum vx^2 + vy^2                      (compute_kinetic_energy, :54-61)
//   m[7] sum collider, m[8] sum density*collider   (compute_collider_violation, :64-78)
//   m[9] sum emitter,  m[10] sum density*emitter   (compute_emitter_density_accuracy, :81-97)
struct MetricArgs {
    const float* pred;
    Plane tgt, emit, coll;
    int B, H, W, pm, stencil;
    float dx;
};
__global__ __launch_bounds__(256) void k_metrics_reduce(MetricArgs a, double* sums) {
    constexpr int NM = 11;
    __shared__ double red[4 * NM];
    const int HW = a.H * a.W, total = a.B * HW;
    double s[NM];
#pragma unroll
    for (int k = 0; k < NM; ++k) s[k] = 0.0;
    for (int idx = blockIdx.x * kThreads + threadIdx.x; idx < total; idx += gridDim.x * kThreads) {
        const int n = idx / HW, r = idx - n * HW, y = r / a.W, x = r - y * a.W;
        const float* P = a.pred + (size_t)n * 3 * HW;
        const float* T = a.tgt.p + (size_t)n * a.tgt.bstride;
#pragma unroll
        for (int c = 0; c < 3; ++c) { const float d = P[c * HW + r] - T[c * HW + r]; s[c] += d * d; }
        s[3] += fabsf(ddx(P, y, x, a.H, a.W, a.pm, a.stencil, a.dx) - ddx(T, y, x, a.H, a.W, a.pm, a.stencil, a.dx));
        s[4] += fabsf(ddy(P, y, x, a.H, a.W, a.pm, a.stencil, a.dx) - ddy(T, y, x, a.H, a.W, a.pm, a.stencil, a.dx));
        const float dv = ddx(P + HW, y, x, a.H, a.W, a.pm, a.stencil, a.dx) + ddy(P + 2 * HW, y, x, a.H, a.W, a.pm, a.stencil, a.dx);
        s[5] += dv * dv;
        const float vx = P[HW + r], vy = P[2 * HW + r];
        s[6] += vx * vx + vy * vy;
        const float co = a.coll.p[(size_t)n * a.coll.bstride + r], em = a.emit.p[(size_t)n * a.emit.bstride + r];
        s[7] += co; s[8] += P[r] * co;
        s[9] += em; s[10] += P[r] * em;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NM; ++k) s[k] = wave_sum_d(s[k]);
    if (lane == 0)
        for (int k = 0; k < NM; ++k) red[wave * NM + k] = s[k];
    __syncthreads();
    if (threadIdx.x < NM) {
        const int k = threadIdx.x;
        atomicAdd(sums + k, red[k] + red[NM + k] + red[2 * NM + k] + red[3 * NM + k]);
    }
}

// SSIM of the density channel (compute_ssim_density, metrics.py:15-21 = torchmetrics' structural_similarity_index_measure with
// its defaults: 11x11 Gaussian window, sigma 1.5, k1 0.01, k2 0.03, data_range 1; torchmetrics is NOT in this image, so this
// follows its published algorithm and parity is unpinned).  The library reflect-pads by 5, filters, and then crops 5 pixels
// per side before averaging, so only windows that lie fully inside the image count: m[11] += sum of the SSIM map over
// y in [5, H-5), x in [5, W-5); the host divides by B*(H-10)*(W-10).
__global__ __launch_bounds__(256) void k_ssim_density(const float* pred, Plane tgt, int B, int H, int W, double* sums) {
    __shared__ double red[4];
    __shared__ float g[11];
    if (threadIdx.x < 11) {
        float tot = 0.f;
        for (int k = 0; k < 11; ++k) { const float d = (float)(k - 5) / 1.5f; tot += expf(-0.5f * d * d); }
        const float d = (float)((int)threadIdx.x - 5) / 1.5f;
        g[threadIdx.x] = expf(-0.5f * d * d) / tot;
    }
    __syncthreads();
    const int h = H - 10, w = W - 10, hw = h * w, total = B * hw;
    double acc = 0.0;
    for (int idx = blockIdx.x * kThreads + threadIdx.x; idx < total; idx += gridDim.x * kThreads) {
        const int n = idx / hw, r = idx - n * hw, y = r / w, x = r - y * w;      // window rows y..y+10, columns x..x+10
        const float* P = pred + (size_t)n * 3 * H * W;
        const float* T = tgt.p + (size_t)n * tgt.bstride;
        float mp = 0.f, mt = 0.f, pp = 0.f, tt = 0.f, pt = 0.f;
        for (int j = 0; j < 11; ++j) {
            float rp = 0.f, rt = 0.f, rpp = 0.f, rtt = 0.f, rpt = 0.f;
            const size_t o = (size_t)(y + j) * W + x;
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                const float p = P[o + i], t = T[o + i], wgt = g[i];
                rp = fmaf(wgt, p, rp); rt = fmaf(wgt, t, rt);
                rpp = fmaf(wgt, p * p, rpp); rtt = fmaf(wgt, t * t, rtt); rpt = fmaf(wgt, p * t, rpt);
            }
            const float wj = g[j];
            mp = fmaf(wj, rp, mp); mt = fmaf(wj, rt, mt); pp = fmaf(wj, rpp, pp); tt = fmaf(wj, rtt, tt); pt = fmaf(wj, rpt, pt);
        }
        const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
        const float mpp = mp * mp, mtt = mt * mt, mpt = mp * mt;
        const float sp = fmaxf(pp - mpp, 0.f), st = fmaxf(tt - mtt, 0.f), spt = pt - mpt;
        acc += (double)(((2.f * mpt + c1) * (2.f * spt + c2)) / ((mpp + mtt + c1) * (sp + st + c2)));
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sums + 11, red[0] + red[1] + red[2] + red[3]);
}

inline LossArgs make_loss_args(const FnetOp& op) {
    LossArgs a;
    a.pred = (const float*)op.p[0];
    a.tgt = {(const float*)op.p[1], op.i[3]};
    a.in0 = {(const float*)op.p[2], op.i[4]};
    a.emit = {(const float*)op.p[3], op.i[5]};
    a.coll = {(const float*)op.p[4], op.i[6]};
    a.B = op.i[0]; a.H = op.i[1]; a.W = op.i[2];
    a.pm = op.i[7]; a.stencil = op.i[8]; a.flags = op.i[9];
    a.dx = op.f[0];
    return a;
}
inline bool loss_args_ok(const FnetOp& op) {
    const int B = op.i[0], H = op.i[1], W = op.i[2];
    if (!(B > 0 && H >= 2 && W >= 2)) return false;
    if ((long long)B * H * W >= (1ll << 31)) return false;
    for (int k = 0; k < 5; ++k) if (!op.p[k]) return false;
    if (op.i[3] < 3 * H * W || op.i[4] < H * W || op.i[5] < H * W || op.i[6] < H * W) return false;
    if (op.i[7] < 0 || op.i[7] > 3 || op.i[8] < 0 || op.i[8] > 1) return false;
    if (!(op.f[0] > 0.f)) return false;
    return true;
}

}  // namespace

// LOSS_REDUCE  i: 0 B,1 H,2 W,3 tgt bstride,4 in0 bstride,5 emit bstride,6 coll bstride,7 pm,8 stencil,9 flags; f: 0 dx
//              p: 0 pred [B,3,H,W],1 target,2 in0 (density_t plane),3 emitter,4 collider,5 sums double[8] (zeroed by the caller)
int fnet_op_loss_reduce(const FnetOp& op, hipStream_t st) {
    FNET_CHECK_ARG(loss_args_ok(op) && op.p[5]);
    const LossArgs a = make_loss_args(op);
    int blocks = (a.B * a.H * a.W + kThreads * 4 - 1) / (kThreads * 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_loss_reduce, dim3(blocks), dim3(kThreads), 0, st, a, (double*)op.p[5]);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// LOSS_FINALIZE  i: 0 B,1 H,2 W,3 flags,4 world; f: 0 w_mse,1 w_div,2 w_grad,3 w_emit,4 wk
//                p: 0 sums,1 global fluid count double*|0,2 terms float[8],3 accum float[8]|0
int fnet_op_loss_finalize(const FnetOp& op, hipStream_t st) {
    FNET_CHECK_ARG(op.i[0] > 0 && op.i[1] > 0 && op.i[2] > 0 && op.i[4] >= 1 && op.p[0] && op.p[2]);
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(64), 0, st, (const double*)op.p[0], (const double*)op.p[1], (float*)op.p[2],
                       (float*)op.p[3], op.i[0], op.i[1], op.i[2], op.i[3], op.i[4], op.f[0], op.f[1], op.f[2], op.f[3], op.f[4]);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// LOSS_GRAD  i: 0..9 as LOSS_REDUCE,10 world,11 dnext bstride; f: 0 dx,1 w_mse,2 w_div,3 w_grad,4 w_emit,5 wk
//            p: 0..4 as LOSS_REDUCE,5 sums,6 global count|0,7 dnext [B,>=4,H,W]|0,8 gscale float*|0,9 dpred out [B,3,H,W]
static bool loss_grad_ok(const FnetOp& op) {
    return loss_args_ok(op) && op.p[5] && op.p[9] && op.i[10] >= 1 && (op.p[7] == 0 || op.i[11] >= 4 * op.i[1] * op.i[2]);
}
static LossGradArgs make_loss_grad_args(const FnetOp& op) {
    return LossGradArgs{(const double*)op.p[5], (const double*)op.p[6], (const float*)op.p[7], op.i[11], (const float*)op.p[8], op.i[10],
                        op.f[1], op.f[2], op.f[3], op.f[4], op.f[5]};
}
int fnet_op_loss_grad(const FnetOp& op, hipStream_t st) {
    FNET_CHECK_ARG(loss_grad_ok(op));
    const LossArgs a = make_loss_args(op);
    const int total = a.B * a.H * a.W;
    hipLaunchKernelGGL(k_loss_grad, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, st, a, make_loss_grad_args(op), (float*)op.p[9]);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// METRICS  i: 0 B,1 H,2 W,3 tgt bstride,4 emit bstride,5 coll bstride,6 pm,7 stencil,8 flags (bit0: SSIM); f: 0 dx
//          p: 0 pred [B,3,H,W],1 target,2 emitter,3 collider,4 sums double[16] (zeroed by the caller; slots above)
int fnet_op_metrics(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && (long long)B * H * W < (1ll << 31));
    FNET_CHECK_ARG(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4]);
    FNET_CHECK_ARG(op.i[3] >= 3 * H * W && op.i[4] >= H * W && op.i[5] >= H * W);
    FNET_CHECK_ARG(op.i[6] >= 0 && op.i[6] <= 3 && op.i[7] >= 0 && op.i[7] <= 1 && op.f[0] > 0.f);
    FNET_CHECK_ARG(!(op.i[8] & 1) || (H > 10 && W > 10));   // SSIM: the library's crop leaves nothing otherwise (checked before any launch)
    MetricArgs a;
    a.pred = (const float*)op.p[0];
    a.tgt = {(const float*)op.p[1], op.i[3]};
    a.emit = {(const float*)op.p[2], op.i[4]};
    a.coll = {(const float*)op.p[3], op.i[5]};
    a.B = B; a.H = H; a.W = W; a.pm = op.i[6]; a.stencil = op.i[7]; a.dx = op.f[0];
    int blocks = (B * H * W + kThreads * 4 - 1) / (kThreads * 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_metrics_reduce, dim3(blocks), dim3(kThreads), 0, st, a, (double*)op.p[4]);
    if (op.i[8] & 1) {
        int sb = (B * (H - 10) * (W - 10) + kThreads - 1) / kThreads;
        if (sb > 4096) sb = 4096;
        hipLaunchKernelGGL(k_ssim_density, dim3(sb), dim3(kThreads), 0, st, a.pred, a.tgt, B, H, W, (double*)op.p[4]);
    }
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// STENCIL  i: 0 B,1 H,2 W,3 pm,4 stencil,5 mode (0 gradients,1 divergence); f: 0 dx,1 dy; p: 0 f0,1 f1|0,2 out0,3 out1|0
int fnet_op_stencil(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2];
    FNET_CHECK_ARG(B > 0 && H >= 1 && W >= 1 && (long long)B * H * W < (1ll << 31));
    FNET_CHECK_ARG(op.i[3] >= 0 && op.i[3] <= 3 && op.i[4] >= 0 && op.i[4] <= 1 && (op.i[5] == 0 || op.i[5] == 1));
    FNET_CHECK_ARG(op.i[3] != FNET_PAD_REFLECT || (H >= 2 && W >= 2));
    FNET_CHECK_ARG(op.p[0] && op.p[2] && (op.i[5] == 0 ? op.p[3] != 0 : op.p[1] != 0));
    FNET_CHECK_ARG(op.f[0] != 0.f && op.f[1] != 0.f);
    const int total = B * H * W;
    hipLaunchKernelGGL(k_stencil, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, st, (const float*)op.p[0],
                       (const float*)op.p[1], (float*)op.p[2], (float*)op.p[3], B, H, W, op.i[3], op.i[4], op.f[0], op.f[1], op.i[5]);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// HEAD_FWD  i: 0 B,1 H,2 W,3 C,4 ldc,5 coff,6 Cout,7 outact; p: 0 src (P),1 Wh fp32 [Cout][C],2 bh,3 z [B,Cout,H,W],4 pred
int fnet_op_head_fwd(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3], Cout = op.i[6];
    // C need not be a multiple of 8: channels C .. roundup8(C)-1 of the activation are zero padding inside ldc
    FNET_CHECK_ARG(B > 0 && H >= 1 && W >= 1 && C > 0 && C <= 1024 && Cout >= 1 && Cout <= 4);
    FNET_CHECK_ARG(op.i[4] % 8 == 0 && op.i[5] % 8 == 0 && op.i[5] + (C + 7) / 8 * 8 <= op.i[4]);
    FNET_CHECK_ARG(op.p[0] && op.p[1] && op.p[2] && op.p[3] && op.p[4] && (op.i[7] == 0 || op.i[7] == 1));
    const int total = B * H * W;
    const size_t lds = (size_t)(Cout * C + Cout) * sizeof(float);
    DISPATCH_T(op.dtype, hipLaunchKernelGGL((k_head_fwd<T, 4>), dim3((total + kThreads - 1) / kThreads), dim3(kThreads), lds, st,
                                            (const T*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], (float*)op.p[3],
                                            (float*)op.p[4], B, H, W, C, op.i[4], op.i[5], Cout, op.i[7]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// HEAD_BWD  i: 0 B,1 H,2 W,3 C,4 ldc,5 coff,6 ldc_d,7 coff_d,8 Cout,9 outact,10 Cz (channels of dact written, 0 = roundup8(C); >= C: zeros)
//           p: 0 dpred,1 z,2 pred,3 act (P),4 Wh,5 dact out (P interior),6 dWh (+=),7 dbh (+=)
int fnet_op_head_bwd(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3], Cout = op.i[8];
    FNET_CHECK_ARG(B > 0 && H >= 1 && W >= 1 && C > 0 && C <= 64 && Cout >= 1 && Cout <= 4);
    FNET_CHECK_ARG(op.i[4] % 8 == 0 && op.i[5] % 8 == 0 && op.i[6] % 8 == 0 && op.i[7] % 8 == 0);
    const int Cz = op.i[10] ? op.i[10] : (C + 7) / 8 * 8;
    FNET_CHECK_ARG(Cz % 8 == 0 && Cz >= C && Cz <= 64);
    FNET_CHECK_ARG(op.i[5] + (C + 7) / 8 * 8 <= op.i[4] && op.i[7] + Cz <= op.i[6]);
    for (int k = 0; k < 8; ++k) FNET_CHECK_ARG(op.p[k] != 0);
    const int total = B * H * W;
    int blocks = (total + kThreads * 8 - 1) / (kThreads * 8);
    if (blocks > 1024) blocks = 1024;
    const size_t lds = (size_t)(Cout * C + 4 * (Cout * C + Cout)) * sizeof(float);
    if (Cz == 8 || Cz == 16 || Cz == 32 || Cz == 64) {
        // channel-group form: Cz / 8 lanes per pixel
        const long long items = (long long)total * (Cz / 8);
        long long nb = (items + kThreads * 4 - 1) / (kThreads * 4);
        if (nb > 768) nb = 768;          // every block ends in Cout * C + Cout same-address atomics: keep them few
        const size_t lds_cg = (size_t)(4 * (Cout * C + Cout)) * sizeof(float);
#define HEAD_BWD_CG(CGN)                                                                                                   \
    if (Cout == 3) {                                                                                                       \
        DISPATCH_T(op.dtype, hipLaunchKernelGGL((k_head_bwd_cg<T, 3, CGN, true>), dim3((unsigned)nb), dim3(kThreads), lds_cg, st, \
                                                (const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2],       \
                                                (const T*)op.p[3], (const float*)op.p[4], (T*)op.p[5], (float*)op.p[6],    \
                                                (float*)op.p[7], B, H, W, C, op.i[4], op.i[5], op.i[6], op.i[7], Cout, op.i[9])); \
    } else {                                                                                                               \
        DISPATCH_T(op.dtype, hipLaunchKernelGGL((k_head_bwd_cg<T, 4, CGN, false>), dim3((unsigned)nb), dim3(kThreads), lds_cg, st, \
                                                (const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2],       \
                                                (const T*)op.p[3], (const float*)op.p[4], (T*)op.p[5], (float*)op.p[6],    \
                                                (float*)op.p[7], B, H, W, C, op.i[4], op.i[5], op.i[6], op.i[7], Cout, op.i[9])); \
    }
        if (Cz == 8) { HEAD_BWD_CG(1); }
        else if (Cz == 16) { HEAD_BWD_CG(2); }
        else if (Cz == 32) { HEAD_BWD_CG(4); }
        else { HEAD_BWD_CG(8); }
#undef HEAD_BWD_CG
        FNET_LAUNCH_CHECK();
        return FNET_OK;
    }
#define HEAD_BWD_LAUNCH(MAXC)                                                                                              \
    DISPATCH_T(op.dtype, hipLaunchKernelGGL((k_head_bwd<T, 4, MAXC>), dim3(blocks), dim3(kThreads), lds, st,              \
                                            (const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2],           \
                                            (const T*)op.p[3], (const float*)op.p[4], (T*)op.p[5], (float*)op.p[6],        \
                                            (float*)op.p[7], B, H, W, C, op.i[4], op.i[5], op.i[6], op.i[7], Cout, op.i[9], Cz))
    if (Cz <= 8) { HEAD_BWD_LAUNCH(8); }
    else if (Cz <= 16) { HEAD_BWD_LAUNCH(16); }
    else if (Cz <= 32) { HEAD_BWD_LAUNCH(32); }
    else { HEAD_BWD_LAUNCH(64); }
#undef HEAD_BWD_LAUNCH
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// ---- executor-level fusion of the head / loss pairs (fnet_plan_run; knob "fuse_head", FNET_FUSE_HEAD=0 disables)
static int head_rows(int B, int H, int W, int per_row_floats, int min_rows, int max_rows) {
    // rows per block: ~512-1024 blocks over the launch, the LDS tile within 48 KB
    int R = (int)(((long long)B * H + 1023) / 1024);
    if (R < min_rows) R = min_rows;
    if (R > max_rows) R = max_rows;
    while (R > 1 && (long long)(R + 2) * W * per_row_floats * 4 > 46 * 1024) --R;
    if (R > H) R = H;
    return R;
}
bool fnet_can_fuse_head_loss_fwd(const FnetOp& ho, const FnetOp& lo) {
    if (!fnet_knob(FNET_KNOB_FUSE_HEAD) || ho.i[FNET_LANE_SLOT] != lo.i[FNET_LANE_SLOT]) return false;
    const int B = ho.i[0], H = ho.i[1], W = ho.i[2], C = ho.i[3];
    if (B != lo.i[0] || H != lo.i[1] || W != lo.i[2] || ho.i[6] != 3 || !ho.p[4] || ho.p[4] != lo.p[0]) return false;
    if (C <= 0 || C > 64 || C % 8 != 0 || ho.i[4] % 8 != 0 || ho.i[5] % 8 != 0 || ho.i[5] + C > ho.i[4]) return false;
    if (!ho.p[0] || !ho.p[1] || !ho.p[2] || !ho.p[3] || (ho.i[7] != 0 && ho.i[7] != 1)) return false;
    if (!loss_args_ok(lo) || !lo.p[5] || W > 900) return false;
    return true;
}
int fnet_op_head_loss_fwd(const FnetOp& ho, const FnetOp& lo, hipStream_t st) {
    FNET_CHECK_ARG(fnet_can_fuse_head_loss_fwd(ho, lo));
    const LossArgs a = make_loss_args(lo);
    const int C = ho.i[3];
    const int R = head_rows(a.B, a.H, a.W, 3, 4, 16);
    const int nrb = (a.H + R - 1) / R;
    const size_t lds = ((size_t)3 * (R + 2) * a.W + 3 * C + 3) * sizeof(float);
    FNET_CHECK_ARG(lds <= 64 * 1024);
    DISPATCH_T(ho.dtype, hipLaunchKernelGGL((k_head_loss_fwd<T>), dim3(a.B * nrb), dim3(kThreads), lds, st, (const T*)ho.p[0],
                                            (const float*)ho.p[1], (const float*)ho.p[2], (float*)ho.p[3], (float*)ho.p[4], a,
                                            (double*)lo.p[5], C, ho.i[4], ho.i[5], ho.i[7], R));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}
bool fnet_can_fuse_loss_head_bwd(const FnetOp& lg, const FnetOp& hb) {
    if (!fnet_knob(FNET_KNOB_FUSE_HEAD) || lg.i[FNET_LANE_SLOT] != hb.i[FNET_LANE_SLOT]) return false;
    const int B = hb.i[0], H = hb.i[1], W = hb.i[2], C = hb.i[3];
    if (B != lg.i[0] || H != lg.i[1] || W != lg.i[2] || hb.i[8] != 3) return false;
    if (!loss_grad_ok(lg) || lg.p[9] != hb.p[0] || lg.p[0] != hb.p[2]) return false;
    for (int k = 0; k < 8; ++k)
        if (!hb.p[k]) return false;
    const int Cz = hb.i[10] ? hb.i[10] : (C + 7) / 8 * 8;
    if (C <= 0 || C > 64 || !(Cz == 8 || Cz == 16 || Cz == 32 || Cz == 64) || Cz < C) return false;
    if (hb.i[4] % 8 != 0 || hb.i[5] % 8 != 0 || hb.i[6] % 8 != 0 || hb.i[7] % 8 != 0) return false;
    if (hb.i[5] + (C + 7) / 8 * 8 > hb.i[4] || hb.i[7] + Cz > hb.i[6] || (hb.i[9] != 0 && hb.i[9] != 1) || W > 900) return false;
    return true;
}
int fnet_op_loss_head_bwd(const FnetOp& lg, const FnetOp& hb, hipStream_t st) {
    FNET_CHECK_ARG(fnet_can_fuse_loss_head_bwd(lg, hb));
    const LossArgs a = make_loss_args(lg);
    const LossGradArgs q = make_loss_grad_args(lg);
    const int C = hb.i[3], Cz = hb.i[10] ? hb.i[10] : (C + 7) / 8 * 8;
    // ~1024 blocks (swept on MI355X, B = 64 at 128^2: 512 / 1024 / 2048 / 4096 blocks -> 171 / 120 / 138 / 175 us per launch; the
    // separate LOSS_GRAD + HEAD_BWD launches take 129): phase A wants many blocks, the same-address atomics at the end few
    int R = (int)(((long long)a.B * a.H + 1023) / 1024);
    if (R < 1) R = 1;
    if (R > 16) R = 16;
    while (R > 1 && (long long)R * a.W * 3 * 4 > 40 * 1024) --R;
    if (R > a.H) R = a.H;
    const int nrb = (a.H + R - 1) / R;
    const size_t lds = ((size_t)3 * R * a.W + 4 * (3 * C + 3)) * sizeof(float);
    FNET_CHECK_ARG(lds <= 64 * 1024);
#define LOSS_HEAD_BWD(CGN)                                                                                                              \
    DISPATCH_T(hb.dtype, hipLaunchKernelGGL((k_loss_head_bwd<T, CGN>), dim3(a.B * nrb), dim3(kThreads), lds, st, a, q, (float*)lg.p[9],   \
                                            (const float*)hb.p[1], (const T*)hb.p[3], (const float*)hb.p[4], (T*)hb.p[5],               \
                                            (float*)hb.p[6], (float*)hb.p[7], C, hb.i[4], hb.i[5], hb.i[6], hb.i[7], hb.i[9], R))
    if (Cz == 8) { LOSS_HEAD_BWD(1); }
    else if (Cz == 16) { LOSS_HEAD_BWD(2); }
    else if (Cz == 32) { LOSS_HEAD_BWD(4); }
    else { LOSS_HEAD_BWD(8); }
#undef LOSS_HEAD_BWD
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

extern "C" int fnet_spatial_gradients(const float* field, float* grad_x, float* grad_y, int32_t B, int32_t H, int32_t W, float dx,
                                      float dy, int32_t padding_mode, int32_t stencil_mode, void* stream) {
    FnetOp op = {};
    op.op = FNET_OP_STENCIL;
    op.i[0] = B; op.i[1] = H; op.i[2] = W; op.i[3] = padding_mode; op.i[4] = stencil_mode; op.i[5] = 0;
    op.f[0] = dx; op.f[1] = dy;
    op.p[0] = (uint64_t)field; op.p[2] = (uint64_t)grad_x; op.p[3] = (uint64_t)grad_y;
    return fnet_op_stencil(op, (hipStream_t)stream);
}
extern "C" int fnet_divergence(const float* velx, const float* vely, float* div, int32_t B, int32_t H, int32_t W, float dx, float dy,
                               int32_t padding_mode, int32_t stencil_mode, void* stream) {
    FnetOp op = {};
    op.op = FNET_OP_STENCIL;
    op.i[0] = B; op.i[1] = H; op.i[2] = W; op.i[3] = padding_mode; op.i[4] = stencil_mode; op.i[5] = 1;
    op.f[0] = dx; op.f[1] = dy;
    op.p[0] = (uint64_t)velx; op.p[1] = (uint64_t)vely; op.p[2] = (uint64_t)div;
    return fnet_op_stencil(op, (hipStream_t)stream);
}

// ---- allocation-free physics-loss terms (no counterpart in the reference) ----------------------------------------------------------
// For callers that cannot call hipMalloc: code under hipGraph capture, the viewer's per-frame loop (live divergence / loss read-outs),
// harnesses that forbid allocation.  Same kernels as the ZERO + LOSS_REDUCE + LOSS_FINALIZE ops of fnet_plan_run; the only scratch
// those need -- the eight fp64 sums the reduce kernel adds into -- lives in this file-scope device array instead of a caller
// buffer.  Every GPU the library runs on holds its own 256-byte copy until the process ends.  All calls on one GPU share it: it is
// cleared at the start of every call, and calls that overlap on one GPU (two streams, two threads) race on it.
static __device__ __align__(256) double scratch[8];
static double* g_scratch_ptr[16];          // per device: the array's DEVICE address (the name above is only a host-side stand-in)

#define FNET_NOALLOC_MAX_CELLS (1ll << 30)

extern "C" int fnet_physics_loss_terms_noalloc(const float* pred, const float* target, const float* density_in, const float* emitter,
                                               const float* collider, int32_t B, int32_t H, int32_t W, int32_t target_bstride,
                                               int32_t density_bstride, int32_t emitter_bstride, int32_t collider_bstride,
                                               int32_t padding_mode, int32_t stencil_mode, int32_t flags, float grid_spacing, float w_mse,
                                               float w_div, float w_grad, float w_emit, float* terms_out, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || !terms_out) return FNET_E_BADARG;
    if ((long long)B * H * W > FNET_NOALLOC_MAX_CELLS) return FNET_E_TOO_LARGE;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 16) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    if (!g_scratch_ptr[dev]) {             // once per device, with that device current
        double* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
        g_scratch_ptr[dev] = p;
    }
    double* sums = g_scratch_ptr[dev];
    hipStream_t st = (hipStream_t)stream;
    FnetOp r = {};
    r.op = FNET_OP_LOSS_REDUCE;
    r.i[0] = B; r.i[1] = H; r.i[2] = W; r.i[3] = target_bstride; r.i[4] = density_bstride; r.i[5] = emitter_bstride; r.i[6] = collider_bstride;
    r.i[7] = padding_mode; r.i[8] = stencil_mode; r.i[9] = flags;
    r.f[0] = grid_spacing;
    r.p[0] = (uint64_t)pred; r.p[1] = (uint64_t)target; r.p[2] = (uint64_t)density_in; r.p[3] = (uint64_t)emitter; r.p[4] = (uint64_t)collider;
    r.p[5] = (uint64_t)sums;
    if (!loss_args_ok(r)) return FNET_E_BADARG;        // before anything is queued
    e = hipMemsetAsync(sums, 0, sizeof(double) * 8, st);            // the last call's sums
    if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    int rc = fnet_op_loss_reduce(r, st);
    if (rc != FNET_OK) return rc;
    FnetOp f = {};
    f.op = FNET_OP_LOSS_FINALIZE;
    f.i[0] = B; f.i[1] = H; f.i[2] = W; f.i[3] = flags; f.i[4] = 1;
    f.f[0] = w_mse; f.f[1] = w_div; f.f[2] = w_grad; f.f[3] = w_emit; f.f[4] = 1.f;
    f.p[0] = (uint64_t)sums; f.p[2] = (uint64_t)terms_out;
    return fnet_op_loss_finalize(f, st);
}
