// Bandwidth kernels of the UNet hot path: input packing, InstanceNorm statistics, the fused
// norm+activation(+dropout)(+residual) "apply" pass and its two-pass adjoint, bilinear/nearest
// upsampling and adjoint, stride-2 subsample / zero-insert, ring fill, gradient folding.
//
// All P-tensor kernels move 8 channels (16 B bf16 / 32 B fp32) per lane; a row of the padded
// image is walked by consecutive lanes so global accesses coalesce (cdna guide G2/G13).
// Grid convention for "row kernels": blockIdx.x = n*(H+2)+y (one padded row), blockIdx.y*256+tid
// = x*(C/8)+cg within the row.
#include <cstdlib>
#include "fnet_common.h"

int g_fnet_last_hip_error = 0;

// pixels per loop trip of the channel-group kernels (forward/backward apply, backward reduce, statistics).  A launch has 768 blocks
// (cg_pixels_per_block), i.e. 3 blocks = 3 waves per SIMD on every CU whatever the register budget allows, so the trips are as deep
// as 512 / 3 VGPRs permit: more loads in flight per thread (measured: 4/3 vs 2/2 takes 1 % off the training step).
#ifndef FNET_EW_U
#define FNET_EW_U 4
#endif
#ifndef FNET_EW_UR
#define FNET_EW_UR 3
#endif
#ifndef FNET_EW_US
#define FNET_EW_US 4
#endif
namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ void block_sum8(float (&v)[8], float* lds /* >= 8*4 floats */) {
    // sum v[0..7] over the 256 threads of the block; result valid in all threads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) lds[wave * 8 + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = lds[k] + lds[8 + k] + lds[16 + k] + lds[24 + k];
}

// ------------------------------------------------------------------ PACK_INPUT
// dst[n,y,x,c] = plane_c[n, src(y), src(x)] for c < 6, 0 for c >= 6  (ring per padding mode)
struct PackArgs {
    const float* src[6];
    int bstride[6];
    int nplanes;           // 6 separate planes, or (plane_stride > 0) nplanes <= 16 planes of one NCHW tensor at src[0]
    int plane_stride;
};
// One thread per (padded pixel, 8-channel group): the planes are read coalesced along x, a group leaves as one 16-B (bf16) / 32-B
// (fp32) store (a 2-byte store per channel, with half of every 256-thread row block idle at W + 2 = 130, ran at 1.2 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void k_pack_input(PackArgs a, T* dst, int B, int H, int W, int Cpad, int pm) {
    const int Hp = H + 2, Wp = W + 2, groups = Cpad >> 3;
    const long long item = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (item >= (long long)B * Hp * Wp * groups) return;
    const int cg = (int)(item % groups);
    const long long pix = item / groups;
    const int x = (int)(pix % Wp);
    const int row = (int)(pix / Wp);
    const int n = row / Hp, y = row - n * Hp;
    const int sy = pad_src(y - 1, H, pm), sx = pad_src(x - 1, W, pm);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = cg * 8 + k;
        float v = 0.f;
        if (c < a.nplanes && sy >= 0 && sx >= 0)
            v = a.plane_stride ? a.src[0][(size_t)n * a.bstride[0] + (size_t)c * a.plane_stride + (size_t)sy * W + sx]
                               : a.src[c < 6 ? c : 0][(size_t)n * a.bstride[c < 6 ? c : 0] + (size_t)sy * W + sx];
        o.v[k] = v;
    }
    o.store(dst + (size_t)pix * Cpad + cg * 8);
}

// ------------------------------------------------------------------ GATHER_BATCH
// Builds the dataset's (x0, y_seq, masks) tuples (ml/dataset/npz_sequence.py:349-433) for B samples straight from the
// HBM-resident sequence store, with the flip augmentation of ml/dataset/augmentations.py:20-35 (mirror the width axis,
// negate velx) applied on the fly.  grid = (pixel blocks, output planes, B); plane p of a sample:
//   p < 6            x0: density_t, velx_t, velz_t, density_{t-1}, emitter_t, collider_t
//   p < 6 + 3K       y_seq[k][c] = field c at t+1+k
//   else             masks[j][m] = emitter/collider at t + mask_first + j
__global__ __launch_bounds__(256) void k_gather_batch(const float* __restrict__ frames, const int* __restrict__ index, const int* __restrict__ flip,
                                                      float* __restrict__ x0, float* __restrict__ y, float* __restrict__ masks, int H, int W, int K,
                                                      int Km, int mask_first, int Tt) {
    const int HW = H * W, r = blockIdx.x * kThreads + threadIdx.x;
    if (r >= HW) return;
    const int n = blockIdx.z, p = blockIdx.y, t = index[n], fl = flip[n];
    int field, dt;
    float* dst;
    if (p < 6) {
        field = p < 3 ? p : (p == 3 ? 0 : p - 1);         // 3 -> density (of t-1), 4 -> emitter (field 3), 5 -> collider (field 4)
        dt = p == 3 ? -1 : 0;
        dst = x0 + ((size_t)n * 6 + p) * HW;
    } else if (p < 6 + 3 * K) {
        const int q = p - 6, k = q / 3;
        field = q - 3 * k;
        dt = 1 + k;
        dst = y + ((size_t)n * K * 3 + q) * HW;
    } else {
        const int q = p - 6 - 3 * K, j = q >> 1;
        field = 3 + (q & 1);
        dt = mask_first + j;
        dst = masks + ((size_t)n * Km * 2 + q) * HW;
    }
    int f = t + dt;
    f = f < 0 ? 0 : (f >= Tt ? Tt - 1 : f);                // the loader only hands out valid (sequence, t) pairs; never fault
    const int yy = r / W, xx = r - yy * W;
    const int sx = fl ? W - 1 - xx : xx;
    float v = frames[((size_t)field * Tt + f) * HW + (size_t)yy * W + sx];
    if (fl && field == 1) v = -v;
    dst[r] = v;
}

// ------------------------------------------------------------------ channel-group thread mapping ("cg kernels")
// The norm/activation kernels below give every thread a FIXED group of 8 channels (cg = tid % groups) and let it walk
// pixels with stride ppi = 256 / groups, so the per-channel constants (mean, rstd, gamma, beta, dropout multiplier, the
// backward sums) are loaded and folded into two FMA coefficients once per thread instead of once per 16 bytes.  A wave
// still covers 64 consecutive 16-B pieces of the pixel-major buffer, so global accesses stay fully coalesced.
// grid = (chunks per image, B); a block owns a contiguous pixel range of one image.
struct CgMap {
    int groups, ppi, cg, px;
    bool active;
    __device__ __forceinline__ CgMap(int C) {
        groups = C >> 3;
        ppi = kThreads / groups;
        cg = threadIdx.x % groups;
        px = threadIdx.x / groups;
        active = px < ppi;
    }
    // This thread's pixel sequence first, first+stride, ... < end: the block owns pixels [blockIdx.x*per_block, +per_block) of
    // an image of n_pix pixels.  (Interleaving an image's blocks so that they sweep one window together was measured and
    // made no difference; fewer, longer blocks did: see cg_pixels_per_block.)
    __device__ __forceinline__ void range(int n_pix, int per_block, int& first, int& stride, int& end) const {
        const int p0 = blockIdx.x * per_block;
        first = p0 + px; stride = ppi; end = min(n_pix, p0 + per_block);
    }
};

// Walks pixel indices p, p+stride, p+2*stride, ... of rows of length L as (y, x) without a division or a branch per step
// (the stride is split into whole rows + columns once).  The cg kernels issue the loads of U pixels back to back, so the
// address arithmetic between them must not contain branches, or hipcc waits for each load before the next.
struct PixWalk {
    int y, x, sy, sx, L;
    __device__ __forceinline__ PixWalk(int p, int stride, int L_) : L(L_) {
        y = p / L; x = p - y * L;
        sy = stride / L; sx = stride - sy * L;
    }
    __device__ __forceinline__ void step() {
        x += sx; y += sy;
        const bool w = x >= L;
        x = w ? x - L : x;
        y = w ? y + 1 : y;
    }
};
template <int N> struct IntC { static constexpr int value = N; };
// Element offset of pixel `pix` of an image with `ldc` channels per pixel, in 32-bit arithmetic on the full-rate 24-bit multiplier
// (v_mul_lo_u32 and the 64-bit forms hipcc picks for size_t indices run at quarter rate, and the channel-group kernels form two or
// three such addresses per pixel).  Launchers check (H+2)(W+2) < 2^24, ldc < 2^24 and (H+2)(W+2)*ldc < 2^31.
__device__ __forceinline__ uint32_t pix_off(uint32_t pix, int ldc) { return __umul24(pix, (uint32_t)ldc); }
__device__ __forceinline__ uint32_t pix_of(int yp, int xp, int Wp) { return __umul24((uint32_t)yp, (uint32_t)Wp) + (uint32_t)xp; }
// v or +0.0 by bit mask (keep = 0xffffffff / 0): a select that the compiler cannot turn back into a branch around the
// arithmetic that produced v, and that also kills a NaN/Inf computed from an unwritten ring pixel
__device__ __forceinline__ float keep_bits(float v, uint32_t keep) { return __uint_as_float(__float_as_uint(v) & keep); }
// pad_src for i in [-1, n] with the out-of-range results hoisted (lo = source of -1, hi = source of n; -1 = zero padding)
__device__ __forceinline__ int pad_lo(int n, int pm) { return pm == FNET_PAD_REPLICATE ? 0 : pm == FNET_PAD_REFLECT ? 1 : pm == FNET_PAD_CIRCULAR ? n - 1 : -1; }
__device__ __forceinline__ int pad_hi(int n, int pm) { return pm == FNET_PAD_REPLICATE ? n - 1 : pm == FNET_PAD_REFLECT ? n - 2 : pm == FNET_PAD_CIRCULAR ? 0 : -1; }

// ------------------------------------------------------------------ IN_STATS
// fp32 plans: one block per (n, 8-channel group), exact two-pass mean / biased variance over the interior H*W
template <typename T>
__global__ __launch_bounds__(256) void k_in_stats(const T* raw, float2* stats, int B, int H, int W, int C, int ldc, int coff, float eps) {
    __shared__ float red[32];
    const int groups = C >> 3;
    const int n = blockIdx.x / groups, cg = blockIdx.x % groups;
    const int Wp = W + 2;
    const size_t img = (size_t)n * (H + 2) * Wp;
    const T* base = raw + coff + cg * 8;
    const int HW = H * W;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    for (int idx = threadIdx.x; idx < HW; idx += kThreads) {
        const int y = idx / W, x = idx - y * W;
        Vec8<T> v;
        v.load(base + (img + (size_t)(y + 1) * Wp + (x + 1)) * ldc);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v.v[k];
    }
    block_sum8(s, red);
    float mean[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { mean[k] = s[k] / (float)HW; q[k] = 0.f; }
    for (int idx = threadIdx.x; idx < HW; idx += kThreads) {
        const int y = idx / W, x = idx - y * W;
        Vec8<T> v;
        v.load(base + (img + (size_t)(y + 1) * Wp + (x + 1)) * ldc);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v.v[k] - mean[k]; q[k] += d * d; }
    }
    block_sum8(q, red);
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        float m = 0.f, qq = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j == k) { m = mean[j]; qq = q[j]; }
        stats[(size_t)n * C + cg * 8 + k] = make_float2(m, rsqrtf(qq / (float)HW + eps));
    }
}

// bf16 plans: one coalesced pass; every block writes its per-channel (sum, sumsq) partials to part[n][chunk][C][2] and
// k_in_finalize adds them in chunk order (no atomics: the forward pass stays bit-reproducible run to run).
template <typename T>
__global__ __launch_bounds__(256) void k_in_stats_part(const T* raw, float* part, int H, int W, int C, int ldc, int coff, int q_per_block) {
    __shared__ float red[kThreads * 16];
    const CgMap m(C);
    const int n = blockIdx.y, HW = H * W, Wp = W + 2;
    int q, qs, q1;
    m.range(HW, q_per_block, q, qs, q1);
    float s[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; t[k] = 0.f; }
    if (m.active) {
        const T* base = raw + (size_t)n * (H + 2) * Wp * ldc + coff + m.cg * 8;
        PixWalk w(q, qs, W);
        auto trip = [&](auto uc) {
            constexpr int UU = decltype(uc)::value;
            Vec8<T> v[UU];
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                v[u].load(base + ((size_t)(w.y + 1) * Wp + (w.x + 1)) * ldc);
                w.step();
            }
#pragma unroll
            for (int u = 0; u < UU; ++u) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { s[k] += v[u].v[k]; t[k] = fmaf(v[u].v[k], v[u].v[k], t[k]); }
            }
        };
        constexpr int U = FNET_EW_US;
        for (; q + (U - 1) * qs < q1; q += U * qs) trip(IntC<U>{});
        for (; q < q1; q += qs) trip(IntC<1>{});
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[threadIdx.x * 16 + k] = s[k]; red[threadIdx.x * 16 + 8 + k] = t[k]; }
    __syncthreads();
    float* dst = part + ((size_t)n * gridDim.x + blockIdx.x) * C * 2;
    for (int idx = threadIdx.x; idx < m.groups * 16; idx += kThreads) {
        const int cg = idx >> 4, k = idx & 15;
        float a = 0.f;
        for (int p = 0; p < m.ppi; ++p) a += red[(p * m.groups + cg) * 16 + k];
        dst[(cg * 8 + (k & 7)) * 2 + (k >> 3)] = a;
    }
}
__global__ __launch_bounds__(256) void k_in_finalize(const float2* part, float2* stats, int B, int C, int chunks, float inv_hw, float eps) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= B * C) return;
    const int n = i / C, c = i - n * C;
    float sx = 0.f, sq = 0.f;
#pragma unroll 8
    for (int j = 0; j < chunks; ++j) {
        const float2 v = part[((size_t)n * chunks + j) * C + c];
        sx += v.x; sq += v.y;
    }
    const float mean = sx * inv_hw;
    const float var = fmaxf(sq * inv_hw - mean * mean, 0.f);
    stats[i] = make_float2(mean, rsqrtf(var + eps));
}

// ------------------------------------------------------------------ NORM_COMBINE (BatchNorm2d / GroupNorm on top of the per-(n,c) statistics)
// IN_STATS leaves (mean, rstd) per (n, c).  With var = 1/rstd^2 - eps and E[x^2] = var + mean^2 those pool exactly: over n
// (BatchNorm2d in training mode, unet.py:20-21; also updates the running statistics, momentum 0.1, unbiased variance) or over the
// channels of a group (GroupNorm, unet.py:24-28).  kind 3 = BatchNorm2d in eval mode: statistics are the running ones.
// The pooled (mean, rstd) is written back to every (n, c) it covers, so APPLY and the adjoint kernels are unchanged.
__global__ __launch_bounds__(256) void k_norm_combine(float2* st, float* rmean, float* rvar, int B, int C, int kind, int gs, float eps,
                                                      float momentum, float count, int Cr /* real channels: groups cover [0, Cr) */) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (kind == 1) {
        if (i >= C) return;
        float m = 0.f, e2 = 0.f;
        for (int n = 0; n < B; ++n) {
            const float2 v = st[(size_t)n * C + i];
            const float var = fmaxf(1.f / (v.y * v.y) - eps, 0.f);
            m += v.x; e2 += var + v.x * v.x;
        }
        m /= (float)B; e2 /= (float)B;
        const float var = fmaxf(e2 - m * m, 0.f);
        const float2 o = make_float2(m, rsqrtf(var + eps));
        for (int n = 0; n < B; ++n) st[(size_t)n * C + i] = o;
        if (rmean) {
            rmean[i] = (1.f - momentum) * rmean[i] + momentum * m;
            rvar[i] = (1.f - momentum) * rvar[i] + momentum * var * (count / fmaxf(count - 1.f, 1.f));
        }
    } else if (kind == 2) {
        const int ng = Cr / gs;
        if (i >= B * ng) return;
        const int n = i / ng, g = i - n * ng;
        float2* s = st + (size_t)n * C + g * gs;
        float m = 0.f, e2 = 0.f;
        for (int c = 0; c < gs; ++c) {
            const float2 v = s[c];
            const float var = fmaxf(1.f / (v.y * v.y) - eps, 0.f);
            m += v.x; e2 += var + v.x * v.x;
        }
        m /= (float)gs; e2 /= (float)gs;
        const float2 o = make_float2(m, rsqrtf(fmaxf(e2 - m * m, 0.f) + eps));
        for (int c = 0; c < gs; ++c) s[c] = o;
    } else {
        if (i >= B * C) return;
        const int c = i % C;
        st[i] = make_float2(rmean[c], rsqrtf(rvar[c] + eps));
    }
}
// adjoint side: Sw[n][c] = the gamma-weighted mean of S = (sum dz, sum dz*xhat) over the normalisation group of (n, c):
// kind 1 gamma_c * mean_n S[n][c]; kind 2 mean over the group's channels c' of gamma_c' S[n][c']; kind 3 zero (eval-mode
// BatchNorm2d is an affine map: nothing to centre).  S itself stays per (n, c) for AFFINE_GRAD.
__global__ __launch_bounds__(256) void k_norm_combine_bwd(const float2* S, const float* gamma, float2* Sw, int B, int C, int kind, int gs, int Cr) {
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (kind == 1) {
        if (i >= C) return;
        float a = 0.f, b = 0.f;
        for (int n = 0; n < B; ++n) { const float2 v = S[(size_t)n * C + i]; a += v.x; b += v.y; }
        const float g = gamma[i] / (float)B;
        const float2 o = make_float2(a * g, b * g);
        for (int n = 0; n < B; ++n) Sw[(size_t)n * C + i] = o;
    } else if (kind == 2) {
        const int ng = Cr / gs, npad = C - Cr;
        if (i >= B * ng) {                       // zero-padded channels: nothing to centre
            const int j = i - B * ng;
            if (j < B * npad) Sw[(size_t)(j / npad) * C + Cr + j % npad] = make_float2(0.f, 0.f);
            return;
        }
        const int n = i / ng, g = i - n * ng;
        float a = 0.f, b = 0.f;
        for (int c = 0; c < gs; ++c) { const float2 v = S[(size_t)n * C + g * gs + c]; const float w = gamma[g * gs + c]; a += w * v.x; b += w * v.y; }
        const float2 o = make_float2(a / (float)gs, b / (float)gs);
        for (int c = 0; c < gs; ++c) Sw[(size_t)n * C + g * gs + c] = o;
    } else {
        if (i >= B * C) return;
        Sw[i] = make_float2(0.f, 0.f);
    }
}

// The pixel walk of APPLY for one thread's 8 channels: padded positions p, p + ps, ... < p1 of one image (shared by k_apply and k_stats_apply, so
// that both produce the same bits from the same constants)
template <typename T, int ACT>
__device__ __forceinline__ void apply_walk(const T* __restrict__ rbase, const T* __restrict__ sbase, T* __restrict__ dbase, const float (&ca)[8],
                                           const float (&cb)[8], const float (&dsc)[8], int H, int W, int ldc_raw, int ldc_res, int ldc_dst, int pm,
                                           int p, int ps, int p1) {
    const int Wp = W + 2;
    PixWalk w(p, ps, Wp);
    const int ylo = pad_lo(H, pm), yhi = pad_hi(H, pm), xlo = pad_lo(W, pm), xhi = pad_hi(W, pm);
    auto trip = [&](auto uc, auto rc) {
        constexpr int UU = decltype(uc)::value;
        constexpr bool RES = decltype(rc)::value != 0;
        Raw8<T> r[UU], q[UU];
        uint32_t keep[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            // source pixel of padded position (y, x): itself in the interior, the padding mode's source on the ring
            int sy = w.y - 1, sx = w.x - 1;
            sy = sy < 0 ? ylo : (sy >= H ? yhi : sy);
            sx = sx < 0 ? xlo : (sx >= W ? xhi : sx);
            keep[u] = (sy | sx) < 0 ? 0u : 0xffffffffu;    // zero padding: load pixel (0,0), discard it below
            sy = max(sy, 0); sx = max(sx, 0);
            const uint32_t sp = pix_of(sy + 1, sx + 1, Wp);
            r[u].load(rbase + pix_off(sp, ldc_raw));
            if (RES) q[u].load(sbase + pix_off(sp, ldc_res));
            w.step();
        }
        __builtin_amdgcn_sched_barrier(0);     // every load of the trip is out before the first unpack (see Raw8)
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            Vec8<T> o;
            float rv[8], qv[8];
            r[u].unpack(rv);
            if (RES) q[u].unpack(qv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float v = act_fwd_t<ACT>(fmaf(rv[k], ca[k], cb[k])) * dsc[k];
                if (RES) v += qv[k];
                o.v[k] = keep_bits(v, keep[u]);
            }
            o.store_nt(dbase + pix_off((uint32_t)(p + u * ps), ldc_dst));
        }
    };
    constexpr int U = FNET_EW_U;       // pixels per trip, all loads first
    if (sbase) {
        for (; p + (U - 1) * ps < p1; p += U * ps) trip(IntC<U>{}, IntC<1>{});
        for (; p < p1; p += ps) trip(IntC<1>{}, IntC<1>{});
    } else {
        for (; p + (U - 1) * ps < p1; p += U * ps) trip(IntC<U>{}, IntC<0>{});
        for (; p < p1; p += ps) trip(IntC<1>{}, IntC<0>{});
    }
}

// ------------------------------------------------------------------ APPLY
// dst[n,y,x,c] = act(gamma*(raw[src]-mean)*rstd+beta) * drop(n,c) + res[src], over the padded domain
template <typename T, int ACT>
__global__ __launch_bounds__(256) void k_apply(const T* __restrict__ raw, const float2* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, const T* __restrict__ res, T* __restrict__ dst,
                        int H, int W, int C, int ldc_raw, int coff_raw, int ldc_res, int coff_res, int ldc_dst,
                        int coff_dst, int act, int pm, uint32_t site, uint32_t seed, float drop_p, int p_per_block) {
    const CgMap m(C);
    if (!m.active) return;
    const int n = blockIdx.y, Hp = H + 2, Wp = W + 2, c0 = m.cg * 8;
    int p, ps, p1;
    m.range(Hp * Wp, p_per_block, p, ps, p1);
    float ca[8], cb[8], dsc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (stats) {
            const float2 ms = stats[(size_t)n * C + c0 + k];
            ca[k] = ms.y * gamma[c0 + k];
            cb[k] = fmaf(-ms.x, ca[k], beta[c0 + k]);
        } else { ca[k] = 1.f; cb[k] = 0.f; }
        dsc[k] = drop_scale(seed, site, n, c0 + k, drop_p);
    }
    const size_t img = (size_t)n * Hp * Wp;
    const T* rbase = raw + img * ldc_raw + coff_raw + c0;
    const T* sbase = res ? res + img * ldc_res + coff_res + c0 : nullptr;
    T* dbase = dst + img * ldc_dst + coff_dst + c0;
    apply_walk<T, ACT>(rbase, sbase, dbase, ca, cb, dsc, H, W, ldc_raw, ldc_res, ldc_dst, pm, p, ps, p1);
}

// ------------------------------------------------------------------ IN_STATS + APPLY in one launch (small images)
// InstanceNorm statistics, their finalisation and the normalise / activate pass of one unit as ONE kernel, for images whose 32-channel
// slab one block can walk twice: block = (image n, 32-channel slice), 1024 threads = 4 channel groups x 256 pixel lanes.  Pass 1 sums x
// and x^2 over the interior (fp32, fixed order: per thread, across the 16 lanes of a wave that share a channel group, across the 16 waves);
// (mean, rstd) go to `stats` for the backward pass and stay in LDS for pass 2, which is APPLY's own pixel walk (apply_walk) reading the
// slab a second time from L2.  Replaces three launches (k_in_stats_part, k_in_finalize, k_apply) whose fixed cost (~10 us each: launch
// ramp, constant loads, a first load round trip, drain) exceeds their traffic time on the <= 64^2 layers: 72 of a step's 104 units.
// Blocks of one image sit on one XCD back to back (item = (b % 8) * ceil(items / 8) + b / 8): its slices share 128-B lines.
// SA_CS = channels per slice: 64, 32 or 16 -- the widest that still leaves a block per CU (UNet_medium at B = 64: exactly 256 blocks on every layer).
constexpr int SA_THREADS = 1024;
#ifndef FNET_SA_U1
#define FNET_SA_U1 4
#endif
template <typename T, int ACT, int SA_CS>
__global__ __launch_bounds__(SA_THREADS) void k_stats_apply(const T* __restrict__ raw, float2* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const T* __restrict__ res, T* __restrict__ dst, int B, int H,
                                                            int W, int C, int ldc_raw, int coff_raw, int ldc_res, int coff_res, int ldc_dst, int coff_dst,
                                                            int pm, uint32_t site, uint32_t seed, float drop_p, float eps) {
    constexpr int SA_GROUPS = SA_CS / 8, SA_PPI = SA_THREADS / SA_GROUPS;
    __shared__ float red[SA_THREADS / 64][SA_GROUPS][16];
    __shared__ float2 cst[SA_CS];
    const int nslices = C / SA_CS, items = nslices * B, per = (items + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= items) return;            // (whole block: no barrier has been reached)
    const int n = item / nslices, slice = item - n * nslices;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & (SA_GROUPS - 1), px = tid / SA_GROUPS;
    const int c0 = slice * SA_CS + cg * 8;
    const int Hp = H + 2, Wp = W + 2, HW = H * W;
    const size_t img = (size_t)n * Hp * Wp;
    const T* rbase = raw + img * ldc_raw + coff_raw + c0;
    float s[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; t[k] = 0.f; }
    {
        int q = px;
        PixWalk w(q, SA_PPI, W);
        auto trip = [&](auto uc) {
            constexpr int UU = decltype(uc)::value;
            Raw8<T> r[UU];
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                r[u].load(rbase + pix_off(pix_of(w.y + 1, w.x + 1, Wp), ldc_raw));
                w.step();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UU; ++u) {
                float v[8];
                r[u].unpack(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s[k] += v[k]; t[k] = fmaf(v[k], v[k], t[k]); }
            }
        };
        constexpr int U = FNET_SA_U1;
        for (; q + (U - 1) * SA_PPI < HW; q += U * SA_PPI) trip(IntC<U>{});
        for (; q < HW; q += SA_PPI) trip(IntC<1>{});
    }
    // lanes l, l + 4, l + 8, ... of a wave share a channel group
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int d = SA_GROUPS; d < 64; d <<= 1) {
            s[k] += __shfl_xor(s[k], d);
            t[k] += __shfl_xor(t[k], d);
        }
    }
    if (lane < SA_GROUPS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[wave][lane][k] = s[k]; red[wave][lane][8 + k] = t[k]; }
    }
    __syncthreads();
    if (tid < SA_CS) {
        const int g = tid >> 3, k = tid & 7;
        float sx = 0.f, sq = 0.f;
#pragma unroll
        for (int wv = 0; wv < SA_THREADS / 64; ++wv) { sx += red[wv][g][k]; sq += red[wv][g][8 + k]; }
        const float inv_hw = 1.f / (float)HW;
        const float mean = sx * inv_hw;
        const float var = fmaxf(sq * inv_hw - mean * mean, 0.f);
        const float2 ms = make_float2(mean, rsqrtf(var + eps));
        cst[tid] = ms;
        stats[(size_t)n * C + slice * SA_CS + tid] = ms;
    }
    __syncthreads();
    float ca[8], cb[8], dsc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 ms = cst[cg * 8 + k];
        ca[k] = ms.y * gamma[c0 + k];
        cb[k] = fmaf(-ms.x, ca[k], beta[c0 + k]);
        dsc[k] = drop_scale(seed, site, n, c0 + k, drop_p);
    }
    const T* sbase = res ? res + img * ldc_res + coff_res + c0 : nullptr;
    T* dbase = dst + img * ldc_dst + coff_dst + c0;
    apply_walk<T, ACT>(rbase, sbase, dbase, ca, cb, dsc, H, W, ldc_raw, ldc_res, ldc_dst, pm, px, SA_PPI, Hp * Wp);
}

// ------------------------------------------------------------------ gradient read with ring fold
// Gradient of an interior pixel of a padded buffer = sum over all padded positions that copied it.
template <typename T>
__device__ __forceinline__ void fold_read8(const T* g, int n, int y, int x, int H, int W, int ldc, int pm, int fold,
                                           float (&acc)[8]) {
    const int Hp = H + 2, Wp = W + 2;
    // ring copies fold only onto the two outermost rows / columns (pad_lo / pad_hi of every padding mode): elsewhere a plain read
    if (!fold || (y >= 2 && y <= H - 3 && x >= 2 && x <= W - 3)) {
        Vec8<T> v;
        v.load(g + (((size_t)n * Hp + y + 1) * Wp + x + 1) * ldc);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v.v[k];
        return;
    }
    int ys[3], xs[3];
    const int ny = fold_list(y, H, pm, ys), nx = fold_list(x, W, pm, xs);
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            Vec8<T> v;
            v.load(g + (((size_t)n * Hp + ys[a]) * Wp + xs[b]) * ldc);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v.v[k];
        }
}
// the ring copies of interior pixel (y, x) other than its own position (border pixels only): the rare half of fold_read8,
// kept out of the way of the unconditional centre loads of the pass-1 kernel.  gimg = image base + channel offset.
template <typename T>
__device__ __forceinline__ void fold_extras8(const T* gimg, int y, int x, int H, int W, int ldc, int pm, float (&acc)[8]) {
    const int Wp = W + 2;
    int ys[3], xs[3];
    const int ny = fold_list(y, H, pm, ys), nx = fold_list(x, W, pm, xs);
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) {
            if (a == 0 && b == 0) continue;               // (y+1, x+1) itself
            Vec8<T> v;
            v.load(gimg + ((size_t)ys[a] * Wp + xs[b]) * ldc);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v.v[k];
        }
}

// Pass 1 of the norm / activation adjoint for one thread's 8 channels: interior pixels q, q + qs, ... < q1 of one image; adds dz to s and
// dz * xhat to t, writes G where Gbase is given (shared by k_inact_bwd_reduce and k_inact_bwd_fused)
// LEAN (the bf16 kernels): the caller passes xa := rstd * gamma, xb := beta - mean * rstd * gamma (u = raw * xa + xb in one FMA), the walk
// sums g * act'(u) into s and g * act'(u) * raw into t, and the caller restores S1 = drop * s, S2 = drop * rstd * (t - mean * s): 16
// per-channel constants in registers instead of 40 (110 VGPRs in the single-source form: two waves per SIMD fit beside a weight-gradient
// wave of the other lane instead of one; capping it at 96 for three spills and is slower)
template <typename T, int ACT, bool LEAN = false, bool ONE = false>     // ONE: g2img == Gbase == nullptr known at compile time
__device__ __forceinline__ void inact_reduce_walk(const T* __restrict__ g1img, const T* __restrict__ g2img, const T* __restrict__ rbase, T* __restrict__ Gbase,
                                                  const float (&xa)[8], const float (&xb)[8], const float (&gam)[8], const float (&bet)[8],
                                                  const float (&dsc)[8], float (&s)[8], float (&t)[8], int H, int W, int ldc_g1, int ldc_g2, int ldc_raw,
                                                  int ldc_G, int fold1, int fold2, int pm, int q, int qs, int q1) {
    const int Wp = W + 2;
    const bool f1 = fold1 && pm != FNET_PAD_ZEROS, f2 = fold2 && pm != FNET_PAD_ZEROS;
    const int fylo = pad_lo(H, pm), fyhi = pad_hi(H, pm), fxlo = pad_lo(W, pm), fxhi = pad_hi(W, pm);
    PixWalk w(q, qs, W);
    auto trip = [&](auto uc, auto gc) {
        constexpr int UU = decltype(uc)::value;
        constexpr bool G2 = decltype(gc)::value != 0;
        Raw8<T> ra1[UU], ra2[UU], rr[UU];
        uint32_t pix[UU];
        int yy[UU], xx[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            yy[u] = w.y; xx[u] = w.x;
            pix[u] = pix_of(w.y + 1, w.x + 1, Wp);
            ra1[u].load(g1img + pix_off(pix[u], ldc_g1));
            if (G2) ra2[u].load(g2img + pix_off(pix[u], ldc_g2));
            rr[u].load(rbase + pix_off(pix[u], ldc_raw));
            w.step();
        }
        __builtin_amdgcn_sched_barrier(0);     // every load of the trip is out before the first unpack (see Raw8)
        Vec8<T> a1[UU], a2[UU], r[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            ra1[u].unpack(a1[u].v);
            if (G2) ra2[u].unpack(a2[u].v);
            rr[u].unpack(r[u].v);
            float acc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = G2 ? a1[u].v[k] + a2[u].v[k] : a1[u].v[k];
            if ((f1 || f2) && (yy[u] == fylo || yy[u] == fyhi || xx[u] == fxlo || xx[u] == fxhi)) {
                // a pixel the ring copies (row/column 0 and n-1 for replicate and circular, 1 and n-2 for reflect): add the
                // ring positions that are copies of it (same order as fold_read8: g1's, then g2's)
                if (G2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = a1[u].v[k];
                }
                if (f1) fold_extras8<T>(g1img, yy[u], xx[u], H, W, ldc_g1, pm, acc);
                if (G2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] += a2[u].v[k];
                    if (f2) fold_extras8<T>(g2img, yy[u], xx[u], H, W, ldc_g2, pm, acc);
                } else if (sizeof(T) == 2) {
                    // single source: only a folded pixel is a sum that needs rounding to the stored type (see below)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = bf2f(f2bf(acc[k]));
                }
            }
            Vec8<T> go;
#pragma unroll
            for (int k = 0; k < 8; ++k) go.v[k] = acc[k];
            if (!ONE && Gbase) go.store_nt(Gbase + pix_off(pix[u], ldc_G));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // use the value as stored (bf16-rounded in bf16 mode) so that pass 2 sees the same G
                float gq = go.v[k];
                if (sizeof(T) == 2 && G2) gq = bf2f(f2bf(gq));   // (one source: a stored value, or rounded where it was folded)
                if (LEAN) {
                    const float dz = gq * act_bwd_t<ACT>(fmaf(r[u].v[k], xa[k], xb[k]));
                    s[k] += dz;
                    t[k] = fmaf(dz, r[u].v[k], t[k]);
                } else {
                    const float xh = fmaf(r[u].v[k], xa[k], xb[k]);
                    const float uu = fmaf(xh, gam[k], bet[k]);
                    const float dz = gq * dsc[k] * act_bwd_t<ACT>(uu);
                    s[k] += dz;
                    t[k] = fmaf(dz, xh, t[k]);
                }
            }
        }
    };
    constexpr int U = FNET_EW_UR;   // pixels per trip: the gradient and activation loads of all of them are issued first
    if (!ONE && g2img) {
        for (; q + (U - 1) * qs < q1; q += U * qs) trip(IntC<U>{}, IntC<1>{});
        for (; q < q1; q += qs) trip(IntC<1>{}, IntC<1>{});
    } else {
        for (; q + (U - 1) * qs < q1; q += U * qs) trip(IntC<U>{}, IntC<0>{});
        for (; q < q1; q += qs) trip(IntC<1>{}, IntC<0>{});
    }
}

// ------------------------------------------------------------------ INACT_BWD_REDUCE (pass 1)
// G = fold(g1) + fold(g2); dz = G*drop*act'(u); S1 = sum dz, S2 = sum dz*xhat per (n,c).  Writes G (interior) and adds the
// block's partial sums into S (zeroed by the plan before this op).
#ifndef FNET_P1_LEAN
#define FNET_P1_LEAN 1
#endif
template <typename T, int ACT, bool ONE = false>
__global__ __launch_bounds__(256) void k_inact_bwd_reduce(const T* __restrict__ g1, const T* __restrict__ g2, const T* __restrict__ raw, const float2* __restrict__ stats, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, T* __restrict__ G, float* __restrict__ S, float* dgamma, float* dbeta, int H, int W, int C, int ldc_g1,
                                   int coff_g1, int ldc_g2, int coff_g2, int ldc_raw, int coff_raw, int ldc_G, int coff_G,
                                   int fold1, int fold2, int act, int pm, uint32_t site, uint32_t seed, float drop_p, int q_per_block) {
    __shared__ float red[kThreads * 16];
    const CgMap m(C);
    const int n = blockIdx.y, Hp = H + 2, Wp = W + 2, HW = H * W, c0 = m.cg * 8;
    int q, qs, q1;
    m.range(HW, q_per_block, q, qs, q1);
    float s[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; t[k] = 0.f; }
    if (m.active) {
        constexpr bool LEAN = sizeof(T) == 2 && FNET_P1_LEAN;
        float xa[8], xb[8], gam[8], bet[8], dsc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float2 ms = stats[(size_t)n * C + c0 + k];
            xa[k] = ms.y; xb[k] = -ms.x * ms.y;          // xhat = raw*xa + xb
            gam[k] = gamma[c0 + k]; bet[k] = beta[c0 + k];
            dsc[k] = drop_scale(seed, site, n, c0 + k, drop_p);
        }
        const size_t img = (size_t)n * Hp * Wp;
        const T* rbase = raw + img * ldc_raw + coff_raw + c0;
        T* Gbase = G ? G + img * ldc_G + coff_G + c0 : nullptr;    // G == 0: single-source unit, pass 2 re-reads (and re-folds) g1
        const T* g1img = g1 + img * ldc_g1 + coff_g1 + c0;
        const T* g2img = g2 ? g2 + img * ldc_g2 + coff_g2 + c0 : nullptr;
        if (LEAN) {
            float ua[8], ub[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { ua[k] = xa[k] * gam[k]; ub[k] = fmaf(xb[k], gam[k], bet[k]); }
            inact_reduce_walk<T, ACT, true, ONE>(g1img, g2img, rbase, Gbase, ua, ub, ua, ub, ua, s, t, H, W, ldc_g1, ldc_g2, ldc_raw, ldc_G, fold1, fold2, pm, q, qs, q1);
            // the constants are fetched again here rather than kept live across the walk
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float2 ms = stats[(size_t)n * C + c0 + k];
                const float d = drop_scale(seed, site, n, c0 + k, drop_p);
                t[k] = d * ms.y * fmaf(-ms.x, s[k], t[k]);
                s[k] *= d;
            }
        } else {
            inact_reduce_walk<T, ACT, false, ONE>(g1img, g2img, rbase, Gbase, xa, xb, gam, bet, dsc, s, t, H, W, ldc_g1, ldc_g2, ldc_raw, ldc_G, fold1, fold2, pm, q, qs, q1);
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[threadIdx.x * 16 + k] = s[k]; red[threadIdx.x * 16 + 8 + k] = t[k]; }
    __syncthreads();
    for (int idx = threadIdx.x; idx < m.groups * 16; idx += kThreads) {
        const int cg = idx >> 4, k = idx & 15;
        float a = 0.f;
        for (int p = 0; p < m.ppi; ++p) a += red[(p * m.groups + cg) * 16 + k];
        const int c = cg * 8 + (k & 7);
        atomicAdd(S + ((size_t)n * C + c) * 2 + (k >> 3), a);
        // affine gradients: dbeta[c] += sum_n S1, dgamma[c] += sum_n S2  (ml/models/unet.py:22-23 InstanceNorm2d(affine=True))
        if (dgamma) atomicAdd((k >> 3) ? dgamma + c : dbeta + c, a);
    }
}

// Pass 2 for one thread's 8 channels: padded positions p, p + ps, ... < p1 of one image (shared by k_inact_bwd_apply and k_inact_bwd_fused)
template <typename T, int ACT, bool WS>
__device__ __forceinline__ void inact_apply_walk(const T* __restrict__ Gbase, const T* __restrict__ rbase, T* __restrict__ dbase, const float (&xa)[8],
                                                 const float (&xb)[8], const float (&gam)[8], const float (&bet)[8], const float (&dsc)[8],
                                                 const float (&k0)[8], const float (&k1)[8], const float (&k2)[8], int H, int W, int ldc_G,
                                                 int ldc_raw, int ldc_d, int fold, int pm, int p, int ps, int p1) {
    const int Hp = H + 2, Wp = W + 2;
    PixWalk w(p, ps, Wp);
    // fold != 0: `G` is the unit's single gradient source itself (pass 1 wrote no G): ring copies are folded onto the pixels they
    // copy here, exactly as pass 1 did (same order, same rounding), so both forms give the same bits
    const bool do_fold = fold && pm != FNET_PAD_ZEROS;
    const int fylo = pad_lo(H, pm), fyhi = pad_hi(H, pm), fxlo = pad_lo(W, pm), fxhi = pad_hi(W, pm);
    auto trip = [&](auto uc) {
        constexpr int UU = decltype(uc)::value;
        Raw8<T> rg[UU], rr[UU];
        uint32_t keep[UU];
        int yy[UU], xx[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            keep[u] = (w.y == 0 || w.y == Hp - 1 || w.x == 0 || w.x == Wp - 1) ? 0u : 0xffffffffu;   // ring: loaded all the same (valid memory), zeroed below
            yy[u] = w.y - 1; xx[u] = w.x - 1;
            const uint32_t pu = (uint32_t)(p + u * ps);
            rg[u].load(Gbase + pix_off(pu, ldc_G));
            rr[u].load(rbase + pix_off(pu, ldc_raw));
            w.step();
        }
        __builtin_amdgcn_sched_barrier(0);     // every load of the trip is out before the first unpack (see Raw8)
        Vec8<T> g[UU], r[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            rg[u].unpack(g[u].v);
            rr[u].unpack(r[u].v);
            if (do_fold && keep[u] && (yy[u] == fylo || yy[u] == fyhi || xx[u] == fxlo || xx[u] == fxhi)) {
                fold_extras8<T>(Gbase, yy[u], xx[u], H, W, ldc_G, pm, g[u].v);
                if (sizeof(T) == 2) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) g[u].v[k] = bf2f(f2bf(g[u].v[k]));
                }
            }
            Vec8<T> o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xh = fmaf(r[u].v[k], xa[k], xb[k]);
                const float uu = fmaf(xh, gam[k], bet[k]);
                const float dz = g[u].v[k] * dsc[k] * act_bwd_t<ACT>(uu);
                o.v[k] = keep_bits(WS ? fmaf(k0[k], dz, -fmaf(xh, k2[k], k1[k])) : k0[k] * (dz - k1[k] - xh * k2[k]), keep[u]);
            }
            o.store_nt(dbase + pix_off((uint32_t)(p + u * ps), ldc_d));
        }
    };
    constexpr int U = FNET_EW_U;       // pixels per trip, all loads first
    for (; p + (U - 1) * ps < p1; p += U * ps) trip(IntC<U>{});
    for (; p < p1; p += ps) trip(IntC<1>{});
}

// ------------------------------------------------------------------ INACT_BWD_APPLY (pass 2)
// d_raw = gamma*rstd*(dz - S1/HW - xhat*S2/HW) on the interior, 0 on the ring (so that d_raw can feed
// the flattened dgrad / wgrad convolutions directly).
// WS ("weighted sums", BatchNorm2d / GroupNorm): S holds what NORM_COMBINE_BWD made of the per-(n,c) sums -- their gamma-weighted
// mean over the normalisation group -- and d_raw = gamma*rstd*dz - rstd*(S1'/HW + xhat*S2'/HW).
template <typename T, int ACT, bool WS>
__global__ __launch_bounds__(256) void k_inact_bwd_apply(const T* __restrict__ G, const T* __restrict__ raw, const float2* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                  const float2* __restrict__ S, T* __restrict__ draw, int H, int W, int C, int ldc_G, int coff_G, int ldc_raw,
                                  int coff_raw, int ldc_d, int coff_d, int act, uint32_t site, uint32_t seed, float drop_p,
                                  int p_per_block, int fold, int pm) {
    const CgMap m(C);
    if (!m.active) return;
    const int n = blockIdx.y, Hp = H + 2, Wp = W + 2, c0 = m.cg * 8;
    int p, ps, p1;
    m.range(Hp * Wp, p_per_block, p, ps, p1);
    const float inv = 1.f / (float)(H * W);
    float xa[8], xb[8], gam[8], bet[8], dsc[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 ms = stats[(size_t)n * C + c0 + k];
        const float2 ss = S[(size_t)n * C + c0 + k];
        xa[k] = ms.y; xb[k] = -ms.x * ms.y;
        gam[k] = gamma[c0 + k]; bet[k] = beta[c0 + k];
        dsc[k] = drop_scale(seed, site, n, c0 + k, drop_p);
        k0[k] = gam[k] * ms.y;                            // out = k0*(dz - k1 - xh*k2); WS: k0*dz - (k1 + xh*k2)
        k1[k] = ss.x * inv * (WS ? ms.y : 1.f);
        k2[k] = ss.y * inv * (WS ? ms.y : 1.f);
    }
    const size_t img = (size_t)n * Hp * Wp;
    const T* Gbase = G + img * ldc_G + coff_G + c0;
    const T* rbase = raw + img * ldc_raw + coff_raw + c0;
    T* dbase = draw + img * ldc_d + coff_d + c0;
    inact_apply_walk<T, ACT, WS>(Gbase, rbase, dbase, xa, xb, gam, bet, dsc, k0, k1, k2, H, W, ldc_G, ldc_raw, ldc_d, fold, pm, p, ps, p1);
}

// ------------------------------------------------------------------ INACT_BWD_REDUCE + INACT_BWD_APPLY in one launch (small images)
// The two passes of the norm / activation adjoint of a single-source unit as ONE kernel, on the same decomposition as k_stats_apply:
// block = (image n, SA_CS-channel slice), 768 threads (the walks keep their 3- / 4-pixel trips: 140-170 registers).  Pass 1
// (inact_reduce_walk) leaves S = (sum dz, sum dz * xhat) of the block's channels -- the whole image, so the sums are final in the block: they
// are added to the S slab for AFFINE_GRAD_MULTI (one atomic per channel, onto the plan's zero) and stay in LDS for pass 2
// (inact_apply_walk), which reads the gradient source and raw a second time from L2.  Replaces two launches of ~10 us fixed cost each.
constexpr int SB_THREADS = 768;      // 12 waves = three per SIMD at 138 registers: what three co-resident blocks of the separate kernels reach
template <typename T, int ACT, int SA_CS>
__global__ __launch_bounds__(SB_THREADS) void k_inact_bwd_fused(const T* __restrict__ g1, const T* __restrict__ raw, const float2* __restrict__ stats,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ S,
                                                                T* __restrict__ draw, int B, int H, int W, int C, int ldc_g1, int coff_g1, int ldc_raw,
                                                                int coff_raw, int ldc_d, int coff_d, int fold, int pm, uint32_t site, uint32_t seed,
                                                                float drop_p) {
    constexpr int GROUPS = SA_CS / 8, PPI = SB_THREADS / GROUPS;
    __shared__ float red[SB_THREADS / 64][GROUPS][16];
    __shared__ float2 sums[SA_CS];
    const int nslices = C / SA_CS, items = nslices * B, per = (items + 7) >> 3;
    const int item = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per || item >= items) return;            // (whole block: no barrier has been reached)
    const int n = item / nslices, slice = item - n * nslices;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = tid & (GROUPS - 1), px = tid / GROUPS;
    const int c0 = slice * SA_CS + cg * 8;
    const int Hp = H + 2, Wp = W + 2, HW = H * W;
    float xa[8], xb[8], gam[8], bet[8], dsc[8], s[8], t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 ms = stats[(size_t)n * C + c0 + k];
        xa[k] = ms.y; xb[k] = -ms.x * ms.y;
        gam[k] = gamma[c0 + k]; bet[k] = beta[c0 + k];
        dsc[k] = drop_scale(seed, site, n, c0 + k, drop_p);
        s[k] = 0.f; t[k] = 0.f;
    }
    const size_t img = (size_t)n * Hp * Wp;
    const T* rbase = raw + img * ldc_raw + coff_raw + c0;
    const T* g1img = g1 + img * ldc_g1 + coff_g1 + c0;
    T* dbase = draw + img * ldc_d + coff_d + c0;
    inact_reduce_walk<T, ACT>(g1img, nullptr, rbase, nullptr, xa, xb, gam, bet, dsc, s, t, H, W, ldc_g1, 8, ldc_raw, 8, fold, 0, pm, px, PPI, HW);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int d = GROUPS; d < 64; d <<= 1) {
            s[k] += __shfl_xor(s[k], d);
            t[k] += __shfl_xor(t[k], d);
        }
    }
    if (lane < GROUPS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[wave][lane][k] = s[k]; red[wave][lane][8 + k] = t[k]; }
    }
    __syncthreads();
    if (tid < SA_CS) {
        const int g = tid >> 3, k = tid & 7;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int wv = 0; wv < SB_THREADS / 64; ++wv) { a += red[wv][g][k]; b += red[wv][g][8 + k]; }
        sums[tid] = make_float2(a, b);
        float* Sp = S + ((size_t)n * C + slice * SA_CS + tid) * 2;
        atomicAdd(Sp, a);
        atomicAdd(Sp + 1, b);
    }
    __syncthreads();
    const float inv = 1.f / (float)HW;
    float k0[8], k1[8], k2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 ss = sums[cg * 8 + k];
        k0[k] = gam[k] * xa[k];
        k1[k] = ss.x * inv;
        k2[k] = ss.y * inv;
    }
    inact_apply_walk<T, ACT, false>(g1img, rbase, dbase, xa, xb, gam, bet, dsc, k0, k1, k2, H, W, ldc_g1, ldc_raw, ldc_d, fold, pm, px, PPI, Hp * Wp);
}

// ------------------------------------------------------------------ AFFINE_GRAD: dgamma += sum_n S2, dbeta += sum_n S1
__global__ __launch_bounds__(256) void k_affine_grad(const float2* S, float* dgamma, float* dbeta, int B, int C) {
    // block = 64 channels x 4 batch groups; the batch loop is split 4 ways and unrolled so the loads overlap
    __shared__ float2 red[kThreads];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int n = g; n < B; n += 4) {
            const float2 s = S[(size_t)n * C + c];
            a += s.x; b += s.y;
        }
    }
    red[threadIdx.x] = make_float2(a, b);
    __syncthreads();
    if (g == 0 && c < C) {
        const float2 r1 = red[threadIdx.x + 64], r2 = red[threadIdx.x + 128], r3 = red[threadIdx.x + 192];
        dbeta[c] += (a + r1.x) + (r2.x + r3.x);
        dgamma[c] += (b + r1.y) + (r2.y + r3.y);
    }
}

// the same for up to 7 norm layers per launch (blockIdx.y = layer): a backward pass has 26 of these 4-microsecond kernels
struct AffineUnits {
    int n;
    int s_off[7], C[7], dg_off[7], db_off[7];
};
__global__ __launch_bounds__(256) void k_affine_grad_multi(const float* Sbase, float* grads, AffineUnits u, int B) {
    __shared__ float2 red[kThreads];
    const int j = blockIdx.y;
    const int C = u.C[j];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    if (blockIdx.x * 64 >= C) return;             // uniform per block
    const float2* S = reinterpret_cast<const float2*>(Sbase + u.s_off[j]);
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int n = g; n < B; n += 4) {
            const float2 s = S[(size_t)n * C + c];
            a += s.x; b += s.y;
        }
    }
    red[threadIdx.x] = make_float2(a, b);
    __syncthreads();
    if (g == 0 && c < C) {
        const float2 r1 = red[threadIdx.x + 64], r2 = red[threadIdx.x + 128], r3 = red[threadIdx.x + 192];
        grads[u.db_off[j] + c] += (a + r1.x) + (r2.x + r3.x);
        if (u.dg_off[j] >= 0) grads[u.dg_off[j] + c] += (b + r1.y) + (r2.y + r3.y);     // -1: no scale parameter (norm = "none")
    }
}

// ------------------------------------------------------------------ FOLD_SUM: dst = fold(g1)+fold(g2) interior, 0 ring
template <typename T>
__global__ __launch_bounds__(256) void k_fold_sum(const T* g1, const T* g2, T* dst, int B, int H, int W, int C, int ldc_g1, int coff_g1,
                           int ldc_g2, int coff_g2, int ldc_d, int coff_d, int fold1, int fold2, int pm) {
    const int Hp = H + 2, Wp = W + 2, groups = C >> 3;
    const int row = blockIdx.x;
    const int n = row / Hp, y = row % Hp;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= Wp * groups) return;
    const int x = item / groups, cg = item - x * groups, c0 = cg * 8;
    T* d = dst + ((size_t)row * Wp + x) * ldc_d + coff_d + c0;
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    if (!(y == 0 || y == Hp - 1 || x == 0 || x == Wp - 1)) {
        fold_read8<T>(g1 + coff_g1 + c0, n, y - 1, x - 1, H, W, ldc_g1, pm, fold1, o.v);
        if (g2) fold_read8<T>(g2 + coff_g2 + c0, n, y - 1, x - 1, H, W, ldc_g2, pm, fold2, o.v);
    }
    o.store(d);
}

// ------------------------------------------------------------------ BIAS_GRAD: dbias[c] += sum over n, interior
template <typename T>
__global__ __launch_bounds__(256) void k_bias_grad(const T* d, float* dbias, int B, int H, int W, int C, int ldc, int coff) {
    __shared__ float red[32];
    const int cg = blockIdx.x, c0 = cg * 8;
    const int Hp = H + 2, Wp = W + 2, HW = H * W;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    const int total = B * HW;
    for (int idx = blockIdx.y * kThreads + threadIdx.x; idx < total; idx += gridDim.y * kThreads) {
        const int n = idx / HW, r = idx - n * HW, y = r / W, x = r - y * W;
        Vec8<T> v;
        v.load(d + (((size_t)n * Hp + y + 1) * Wp + x + 1) * ldc + coff + c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v.v[k];
    }
    block_sum8(s, red);
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j == k) a = s[j];
        atomicAdd(dbias + c0 + k, a);
    }
}

// Channel-group form (C <= 1024): the C/8 groups of a pixel sit on consecutive threads, so a block reads whole pixels as contiguous
// runs; per-thread sums meet in LDS (ds_add_f32), one global atomic per channel and block.
template <typename T>
__global__ __launch_bounds__(256) void k_bias_grad_cg(const T* d, float* dbias, int B, int H, int W, int C, int ldc, int coff) {
    __shared__ float red[1024];
    const int groups = C >> 3, ppb = kThreads / groups;        // pixels per block trip
    const int cg = threadIdx.x % groups, pl = threadIdx.x / groups;
    for (int i = threadIdx.x; i < C; i += kThreads) red[i] = 0.f;
    __syncthreads();
    const int Hp = H + 2, Wp = W + 2, HW = H * W;
    const long long total = (long long)B * HW;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    if (pl < ppb) {
        for (long long p = (long long)blockIdx.x * ppb + pl; p < total; p += (long long)gridDim.x * ppb) {
            const int n = (int)(p / HW), r = (int)(p - (long long)n * HW), y = r / W, x = r - y * W;
            Vec8<T> v;
            v.load(d + (((size_t)n * Hp + y + 1) * Wp + x + 1) * ldc + coff + cg * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += v.v[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&red[cg * 8 + k], s[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += kThreads) atomicAdd(dbias + i, red[i]);
}

// ------------------------------------------------------------------ UPSAMPLE2X (nn.Upsample(scale_factor=2), unet.py:152-156)
// bilinear, align_corners=False: out[2a] = .25 in[a-1] + .75 in[a], out[2a+1] = .75 in[a] + .25 in[a+1], edges clamped.
__device__ __forceinline__ void up_taps(int Y, int h, int mode, int& i0, int& i1, float& w0, float& w1) {
    const int a = Y >> 1;
    if (mode == 0) { i0 = i1 = a; w0 = 1.f; w1 = 0.f; return; }
    if (Y & 1) { i0 = a; i1 = min(a + 1, h - 1); w0 = 0.75f; w1 = 0.25f; }
    else { i0 = max(a - 1, 0); i1 = a; w0 = 0.25f; w1 = 0.75f; }
}
template <typename T>
__global__ __launch_bounds__(256) void k_upsample2x(const T* src, T* dst, int B, int h, int w, int C, int ldc_s, int coff_s, int ldc_d, int coff_d,
                             int mode, int pm) {
    const int H = 2 * h, W = 2 * w, Hp = H + 2, Wp = W + 2, groups = C >> 3;
    const int row = blockIdx.x;
    const int n = row / Hp, y = row % Hp;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= Wp * groups) return;
    const int x = item / groups, cg = item - x * groups, c0 = cg * 8;
    T* d = dst + ((size_t)row * Wp + x) * ldc_d + coff_d + c0;
    const int Y = pad_src(y - 1, H, pm), X = pad_src(x - 1, W, pm);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    if (Y >= 0 && X >= 0) {
        int y0, y1, x0, x1;
        float wy0, wy1, wx0, wx1;
        up_taps(Y, h, mode, y0, y1, wy0, wy1);
        up_taps(X, w, mode, x0, x1, wx0, wx1);
        const int hp = h + 2, wp = w + 2;
        const T* s = src + coff_s + c0;
        const size_t img = (size_t)n * hp;
        Vec8<T> a, b, c, e;
        a.load(s + ((img + y0 + 1) * wp + x0 + 1) * ldc_s);
        b.load(s + ((img + y0 + 1) * wp + x1 + 1) * ldc_s);
        c.load(s + ((img + y1 + 1) * wp + x0 + 1) * ldc_s);
        e.load(s + ((img + y1 + 1) * wp + x1 + 1) * ldc_s);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            o.v[k] = wy0 * (wx0 * a.v[k] + wx1 * b.v[k]) + wy1 * (wx0 * c.v[k] + wx1 * e.v[k]);
    }
    o.store(d);
}

// adjoint: dst[n,i,j] = sum_{Y,X} wy(Y,i) wx(X,j) fold(g)[Y,X]   (dst interior only)
__device__ __forceinline__ int up_adj(int i, int h, int mode, int Ys[4], float ws[4]) {
    int k = 0;
    if (mode == 0) { Ys[0] = 2 * i; ws[0] = 1.f; Ys[1] = 2 * i + 1; ws[1] = 1.f; return 2; }
    Ys[k] = 2 * i; ws[k++] = (i == 0) ? 1.0f : 0.75f;
    Ys[k] = 2 * i + 1; ws[k++] = (i == h - 1) ? 1.0f : 0.75f;
    if (i >= 1) { Ys[k] = 2 * i - 1; ws[k++] = 0.25f; }
    if (i + 1 <= h - 1) { Ys[k] = 2 * i + 2; ws[k++] = 0.25f; }
    return k;
}
// Strip form: a block owns UB_S consecutive coarse rows of one image, a thread one coarse column j and 8 channels.  The adjoint is
// separable, dst[i][j] = sum_a wy_a(i) * hx[2i-1+a], hx[Y] = sum_b wx_b(j) * G[Y][2j-1+b] (a, b = 0..3; a weight is 0 where the fine
// index does not exist, and the load then hits the zero-weighted ring, which is valid memory), so a thread walks down the fine rows of
// its strip, forms hx of two new rows per coarse row (2 x 4 loads, issued together) and keeps the two it still needs in registers: every
// fine row is fetched once per strip (2S + 2 rows per S coarse rows) instead of twice, and 8 loads serve an output instead of 16 (the
// one-row-per-block form fetched 194 MB for 98 MB of work at 128^2 x 64, PMC).  G = the gradient with its ring copies folded in
// (fold != 0); ring copies fold only onto the two outermost fine rows / columns, where the taps go through fold_read8.
constexpr int UB_S = 8;
template <typename T>
__global__ __launch_bounds__(256) void k_upsample2x_bwd(const T* g, T* dst, int B, int h, int w, int C, int ldc_g, int coff_g, int ldc_d, int coff_d,
                                 int fold, int mode, int pm, int S, int strips) {
    const int hp = h + 2, wp = w + 2, groups = C >> 3;
    const int n = blockIdx.x / strips, i0 = (blockIdx.x - n * strips) * S;
    const int i1 = min(i0 + S, h);                      // coarse rows [i0, i1)
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= w * groups) return;
    const int j = item / groups, cg = item - j * groups, c0 = cg * 8;
    const int H2 = 2 * h, W2 = 2 * w, Wp2 = W2 + 2;
    // column weights of fine columns 2j-1 .. 2j+2 (bilinear, align_corners=False, edges clamped; nearest: 0 1 1 0)
    float wx[4];
    wx[0] = (mode && j >= 1) ? 0.25f : 0.f;
    wx[1] = (mode && j > 0) ? 0.75f : 1.f;
    wx[2] = (mode && j < w - 1) ? 0.75f : 1.f;
    wx[3] = (mode && j + 1 <= w - 1) ? 0.25f : 0.f;
    const bool do_fold = fold && pm != FNET_PAD_ZEROS;
    const T* gimg = g + (size_t)n * (H2 + 2) * Wp2 * ldc_g + coff_g + c0;
    // hx of fine row Y (-1 <= Y <= H2): 4 taps
    auto row_loads = [&](int Y, Raw8<T> (&r)[4]) {
        const T* p = gimg + ((size_t)(Y + 1) * Wp2 + 2 * j) * ldc_g;         // padded column of fine column 2j-1 is 2j
#pragma unroll
        for (int b = 0; b < 4; ++b) r[b].load(p + (size_t)b * ldc_g);
    };
    // ring copies fold only onto the two outermost fine rows / columns: taps there go through fold_read8 (whole row of taps of an
    // edge-column thread, every thread on an edge row); everything else is plain loads
    const bool edge_col = do_fold && (2 * j - 1 < 2 || 2 * j + 2 > W2 - 3);
    auto row_sum = [&](int Y, const Raw8<T> (&r)[4], float (&hx)[8]) {
        const bool slow = do_fold && (edge_col || Y < 2 || Y > H2 - 3);
#pragma unroll
        for (int k = 0; k < 8; ++k) hx[k] = 0.f;
        if (!slow) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float v[8];
                r[b].unpack(v);
#pragma unroll
                for (int k = 0; k < 8; ++k) hx[k] += wx[b] * (0.f + v[k]);
            }
        } else if (Y >= 0 && Y < H2) {
            for (int b = 0; b < 4; ++b) {
                const int X = 2 * j - 1 + b;
                if (X < 0 || X >= W2) continue;
                float acc[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] = 0.f;
                fold_read8<T>(g + coff_g + c0, n, Y, X, H2, W2, ldc_g, pm, fold, acc);
#pragma unroll
                for (int k = 0; k < 8; ++k) hx[k] += wx[b] * acc[k];
            }
        }
    };
    float ha[8], hb[8], hc[8], hd[8];                   // hx of fine rows 2i-1, 2i, 2i+1, 2i+2
    {
        Raw8<T> ra[4], rb[4];
        row_loads(2 * i0 - 1, ra);
        row_loads(2 * i0, rb);
        __builtin_amdgcn_sched_barrier(0);
        row_sum(2 * i0 - 1, ra, ha);
        row_sum(2 * i0, rb, hb);
    }
    for (int i = i0; i < i1; ++i) {
        Raw8<T> rc[4], rd[4];
        row_loads(2 * i + 1, rc);
        row_loads(2 * i + 2, rd);
        __builtin_amdgcn_sched_barrier(0);              // the eight loads of the two new rows are out before the first unpack
        row_sum(2 * i + 1, rc, hc);
        row_sum(2 * i + 2, rd, hd);
        const float wa = (mode && i >= 1) ? 0.25f : 0.f, wb = (mode && i > 0) ? 0.75f : 1.f;
        const float wc = (mode && i < h - 1) ? 0.75f : 1.f, wd = (mode && i + 1 <= h - 1) ? 0.25f : 0.f;
        Vec8<T> o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o.v[k] = (wa * ha[k] + wb * hb[k]) + (wc * hc[k] + wd * hd[k]);
        o.store(dst + (((size_t)n * hp + i + 1) * wp + j + 1) * ldc_d + coff_d + c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) { ha[k] = hc[k]; hb[k] = hd[k]; }
    }
}

// ------------------------------------------------------------------ POOL2X (nn.AvgPool2d(2) / nn.MaxPool2d(2), unet.py:93-94)
// dst (h, w) whole padded domain: interior = the 2x2 window's mean or max, ring = the padding values of the conv that follows.
template <typename T>
__global__ __launch_bounds__(256) void k_pool2x(const T* src, T* dst, int B, int h, int w, int C, int ldc_s, int coff_s, int ldc_d, int coff_d,
                                                int mode, int pm) {
    const int hp = h + 2, wp = w + 2, Wp = 2 * w + 2, groups = C >> 3;
    const int row = blockIdx.x;
    const int n = row / hp, y = row % hp;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= wp * groups) return;
    const int x = item / groups, cg = item - x * groups, c0 = cg * 8;
    const int i = pad_src(y - 1, h, pm), j = pad_src(x - 1, w, pm);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    if (i >= 0 && j >= 0) {
        const T* s = src + (((size_t)n * (2 * h + 2) + 2 * i + 1) * Wp + 2 * j + 1) * ldc_s + coff_s + c0;
        Vec8<T> a, b, c, e;
        a.load(s); b.load(s + ldc_s); c.load(s + (size_t)Wp * ldc_s); e.load(s + (size_t)(Wp + 1) * ldc_s);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            o.v[k] = mode ? fmaxf(fmaxf(a.v[k], b.v[k]), fmaxf(c.v[k], e.v[k])) : 0.25f * ((a.v[k] + b.v[k]) + (c.v[k] + e.v[k]));
    }
    o.store(dst + ((size_t)row * wp + x) * ldc_d + coff_d + c0);
}

// adjoint: dst (2h, 2w) whole padded domain, ring 0.  avg: a quarter of the (folded) window gradient to each of its four
// pixels; max: all of it to the FIRST maximum of the window in row-major order (torch's max_pool2d keeps the first index
// whose value is greater than the running maximum), found again from the forward input x.
template <typename T>
__global__ __launch_bounds__(256) void k_pool2x_bwd(const T* g, const T* xin, T* dst, int B, int h, int w, int C, int ldc_g, int coff_g, int ldc_x,
                                                    int coff_x, int ldc_d, int coff_d, int fold, int mode, int pm) {
    const int H = 2 * h, W = 2 * w, Hp = H + 2, Wp = W + 2, groups = C >> 3;
    const int row = blockIdx.x;
    const int n = row / Hp, y = row % Hp;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= Wp * groups) return;
    const int x = item / groups, cg = item - x * groups, c0 = cg * 8;
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    if (y >= 1 && y <= H && x >= 1 && x <= W) {
        const int i = (y - 1) >> 1, j = (x - 1) >> 1;
        float gv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gv[k] = 0.f;
        fold_read8<T>(g + coff_g + c0, n, i, j, h, w, ldc_g, pm, fold, gv);
        if (!mode) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o.v[k] = 0.25f * gv[k];
        } else {
            const T* s = xin + (((size_t)n * Hp + 2 * i + 1) * Wp + 2 * j + 1) * ldc_x + coff_x + c0;
            Vec8<T> q[4];
            q[0].load(s); q[1].load(s + ldc_x); q[2].load(s + (size_t)Wp * ldc_x); q[3].load(s + (size_t)(Wp + 1) * ldc_x);
            const int me = ((y - 1) & 1) * 2 + ((x - 1) & 1);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                int arg = 0;
                float m = q[0].v[k];
#pragma unroll
                for (int r = 1; r < 4; ++r)
                    if (q[r].v[k] > m || q[r].v[k] != q[r].v[k]) { m = q[r].v[k]; arg = r; }
                o.v[k] = arg == me ? gv[k] : 0.f;
            }
        }
    }
    o.store(dst + ((size_t)row * Wp + x) * ldc_d + coff_d + c0);
}

// ------------------------------------------------------------------ DEPTH2SPACE / SPACE2DEPTH (nn.ConvTranspose2d(k=2, s=2), unet.py:150)
// The transposed convolution is computed as a centre-tap convolution into 4*C channels (phase q = 2a + b outermost, PREP_WEIGHTS kind 1);
// DEPTH2SPACE writes out[2i+a][2j+b][c] = src[i][j][q*C + c] + bias[c] over the whole padded domain of dst (ring per pm).
template <typename T>
__global__ __launch_bounds__(256) void k_depth2space(const T* src, const float* bias, T* dst, int B, int h, int w, int C, int ldc_s, int coff_s,
                                                     int ldc_d, int coff_d, int pm) {
    const int H = 2 * h, W = 2 * w, Hp = H + 2, Wp = W + 2, groups = C >> 3;
    const int row = blockIdx.x;
    const int n = row / Hp, y = row % Hp;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= Wp * groups) return;
    const int x = item / groups, cg = item - x * groups, c0 = cg * 8;
    const int Y = pad_src(y - 1, H, pm), X = pad_src(x - 1, W, pm);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    if (Y >= 0 && X >= 0) {
        const int q = (Y & 1) * 2 + (X & 1);
        o.load(src + (((size_t)n * (h + 2) + (Y >> 1) + 1) * (w + 2) + (X >> 1) + 1) * ldc_s + coff_s + q * C + c0);
#pragma unroll
        for (int k = 0; k < 8; ++k) o.v[k] += bias[c0 + k];
    }
    o.store(dst + ((size_t)row * Wp + x) * ldc_d + coff_d + c0);
}
// adjoint: dst[i][j][q*C + c] = fold(g)[2i+a][2j+b][c] on the interior of the (h, w) grid, 0 on its ring
template <typename T>
__global__ __launch_bounds__(256) void k_space2depth(const T* g, T* dst, int B, int h, int w, int C, int ldc_g, int coff_g, int ldc_d, int coff_d,
                                                     int fold, int pm) {
    const int hp = h + 2, wp = w + 2, groups = (4 * C) >> 3;
    const int row = blockIdx.x;
    const int n = row / hp, y = row % hp;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= wp * groups) return;
    const int x = item / groups, cg = item - x * groups, k0 = cg * 8;
    const int q = k0 / C, c0 = k0 - q * C;
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    if (y >= 1 && y <= h && x >= 1 && x <= w)
        fold_read8<T>(g + coff_g + c0, n, 2 * (y - 1) + (q >> 1), 2 * (x - 1) + (q & 1), 2 * h, 2 * w, ldc_g, pm, fold, o.v);
    o.store(dst + ((size_t)row * wp + x) * ldc_d + coff_d + k0);
}

// ------------------------------------------------------------------ SUBSAMPLE_BIAS (stride-2 conv = stride-1 conv sampled at even pixels)
template <typename T>
__global__ __launch_bounds__(256) void k_subsample_bias(const T* src, const float* bias, T* dst, int B, int H, int W, int C, int ldc_s, int coff_s,
                                 int ldc_d, int coff_d, int pm) {
    const int h = H / 2, w = W / 2, hp = h + 2, wp = w + 2, groups = C >> 3;
    const int row = blockIdx.x;
    const int n = row / hp, y = row % hp;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= wp * groups) return;
    const int x = item / groups, cg = item - x * groups, c0 = cg * 8;
    T* d = dst + ((size_t)row * wp + x) * ldc_d + coff_d + c0;
    const int i = pad_src(y - 1, h, pm), j = pad_src(x - 1, w, pm);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    if (i >= 0 && j >= 0) {
        o.load(src + (((size_t)n * (H + 2) + 2 * i + 1) * (W + 2) + 2 * j + 1) * ldc_s + coff_s + c0);
        if (bias) {
#pragma unroll
            for (int k = 0; k < 8; ++k) o.v[k] += bias[c0 + k];
        }
    }
    o.store(d);
}

// ------------------------------------------------------------------ ZERO_INSERT (adjoint of the even-pixel sampling)
template <typename T>
__global__ __launch_bounds__(256) void k_zero_insert(const T* src, T* dst, int B, int h, int w, int C, int ldc_s, int coff_s, int ldc_d, int coff_d) {
    const int H = 2 * h, W = 2 * w, Hp = H + 2, Wp = W + 2, groups = C >> 3;
    const int row = blockIdx.x;
    const int n = row / Hp, y = row % Hp;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= Wp * groups) return;
    const int x = item / groups, cg = item - x * groups, c0 = cg * 8;
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    // padded (y,x) = (2i+1, 2j+1) <-> half-res interior (i,j)
    if ((y & 1) && (x & 1) && y < Hp - 1 && x < Wp - 1) {
        const int i = (y - 1) >> 1, j = (x - 1) >> 1;
        o.load(src + (((size_t)n * (h + 2) + i + 1) * (w + 2) + j + 1) * ldc_s + coff_s + c0);
    }
    o.store(dst + ((size_t)row * Wp + x) * ldc_d + coff_d + c0);
}

// ------------------------------------------------------------------ PAD_RING (in place)
template <typename T>
__global__ __launch_bounds__(256) void k_pad_ring(T* buf, int B, int H, int W, int C, int ldc, int coff, int pm) {
    const int Hp = H + 2, Wp = W + 2, groups = C >> 3;
    const int ring = 2 * Wp + 2 * H;  // ring pixels per image
    const int n = blockIdx.x;
    const int item = blockIdx.y * kThreads + threadIdx.x;
    if (item >= ring * groups) return;
    const int r = item / groups, cg = item - r * groups, c0 = cg * 8;
    int y, x;
    if (r < Wp) { y = 0; x = r; }
    else if (r < 2 * Wp) { y = Hp - 1; x = r - Wp; }
    else if (r < 2 * Wp + H) { y = r - 2 * Wp + 1; x = 0; }
    else { y = r - 2 * Wp - H + 1; x = Wp - 1; }
    const int sy = pad_src(y - 1, H, pm), sx = pad_src(x - 1, W, pm);
    Vec8<T> o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.v[k] = 0.f;
    T* base = buf + coff + c0;
    if (sy >= 0 && sx >= 0) o.load(base + (((size_t)n * Hp + sy + 1) * Wp + sx + 1) * ldc);
    o.store(base + (((size_t)n * Hp + y) * Wp + x) * ldc);
}

// ------------------------------------------------------------------ UNPACK_GRAD: dx[n,c,y,x] = fold(d_in_pad)[n,y,x,c]
template <typename T>
__global__ __launch_bounds__(256) void k_unpack_grad(const T* g, float* dx, int B, int H, int W, int ldc, int nch, int pm) {
    const int row = blockIdx.x;  // n*H + y
    const int n = row / H, y = row % H;
    const int x = blockIdx.y * kThreads + threadIdx.x;
    if (x >= W) return;
    float acc[8];
    for (int c0 = 0; c0 < nch; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        fold_read8<T>(g + c0, n, y, x, H, W, ldc, pm, 1, acc);
        for (int k = 0; k < 8 && c0 + k < nch; ++k) dx[(((size_t)n * nch + c0 + k) * H + y) * W + x] = acc[k];
    }
}

// ------------------------------------------------------------------ COPY_PLANES: dst[n, :] = src[n, :] (count floats per n, strided batches)
__global__ __launch_bounds__(256) void k_copy_planes(const float* src, float* dst, int B, int count, int sstride, int dstride) {
    const int n = blockIdx.y;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < count; i += gridDim.x * kThreads)
        dst[(size_t)n * dstride + i] = src[(size_t)n * sstride + i];
}

inline dim3 row_grid(int rows, int items) { return dim3(rows, (items + kThreads - 1) / kThreads, 1); }
inline bool vec_ok(const void* p, int ldc, int coff, int C, int dtype) {
    const int esz = dtype == FNET_BF16 ? 2 : 4;
    return p != nullptr && (C % 8) == 0 && (ldc % 8) == 0 && (coff % 8) == 0 && C > 0 && coff + C <= ldc &&
           (((uintptr_t)p) % (8 * esz)) == 0;
}
inline bool vec_ok_opt(const void* p, int ldc, int coff, int C, int dtype) { return p == nullptr || vec_ok(p, ldc, coff, C, dtype); }
// cg kernels: pixels per block so that the launch has about `target` blocks but every thread still walks >= 4 pixels.
// Measured on MI355X (tools/ew_bench.py, B = 64): 768-1024 blocks (3-4 per CU) beat 2048 by 8-10 % on the 128^2 and 64^2
// layers -- every block pays a prologue of ~30 dependent scalar loads plus the dropout hash before its first pixel -- and
// 256-512 lose again to the tail; 1024 blocks of a power-of-two image (16 equal power-of-two ranges) was the one bad point
// above 512 in the backward reduce: every block then streams at the same address modulo a large power of two.
// FNET_CG_BLOCKS overrides the target (tuning knob).
inline int cg_pixels_per_block(int npix, int B, int C, int* chunks_out, int target) {
    const int ppi = kThreads / (C >> 3);
    static const int forced = [] { const char* e = getenv("FNET_CG_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    if (forced) target = forced;
    int chunks = (target + B - 1) / B;
    const int max_chunks = (npix + ppi * 4 - 1) / (ppi * 4);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    int per = (npix + chunks - 1) / chunks;
    per = (per + ppi - 1) / ppi * ppi;       // whole iterations: keeps a block's pixel range a multiple of the stride
    *chunks_out = (npix + per - 1) / per;
    return per;
}
inline bool cg_ok(int C, int B) { return C >= 8 && (C >> 3) <= kThreads && B > 0 && B < 65536; }
// pix_off / pix_of: image-local pixel indices fit 24 bits and element offsets 31 bits for every stream (ldc) of the launch
inline bool off_ok(int H, int W, int ldc_a, int ldc_b, int ldc_c = 8, int ldc_d = 8) {
    const long long px = (long long)(H + 2) * (W + 2);
    long long ld = ldc_a > ldc_b ? ldc_a : ldc_b;
    if (ldc_c > ld) ld = ldc_c;
    if (ldc_d > ld) ld = ldc_d;
    return px < (1ll << 24) && ld < (1ll << 24) && px * ld < (1ll << 31);
}

}  // namespace

// PACK_INPUT  i: 0 B,1 H,2 W,3 Cpad,4 pm,5..10 batch strides of the six planes (elements),15 n (0: six planes; n: p[0] is one NCHW tensor of n channels)
//             p: 0..5 source planes fp32 (plane n at src + n*stride, H*W contiguous), 6 dst P[B,H+2,W+2,Cpad]
int fnet_op_pack_input(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], Cpad = op.i[3], pm = op.i[4];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && Cpad >= 6 && Cpad % 8 == 0 && pm >= 0 && pm <= 3 && op.p[6]);
    PackArgs a = {};
    if (op.i[15]) {
        // i[15] = n: the n <= Cpad channels of ONE contiguous NCHW tensor at p[0] (batch stride i[5] >= n*H*W): the nn.Module path,
        // where in_channels need not be the rollout's six fields
        FNET_CHECK_ARG(op.i[15] >= 1 && op.i[15] <= Cpad && op.p[0] != 0 && op.i[5] >= op.i[15] * H * W);
        a.src[0] = (const float*)op.p[0];
        a.bstride[0] = op.i[5];
        a.nplanes = op.i[15];
        a.plane_stride = H * W;
    } else {
        for (int c = 0; c < 6; ++c) {
            FNET_CHECK_ARG(op.p[c] != 0 && op.i[5 + c] >= H * W);
            a.src[c] = (const float*)op.p[c];
            a.bstride[c] = op.i[5 + c];
        }
        a.nplanes = 6;
    }
    const long long items = (long long)B * (H + 2) * (W + 2) * (Cpad / 8);
    FNET_CHECK_ARG((items + kThreads - 1) / kThreads < (1ll << 31));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_pack_input<T>, dim3((unsigned)((items + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, a,
                                            (T*)op.p[6], B, H, W, Cpad, pm));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// IN_STATS  i: 0 B,1 H,2 W,3 C,4 ldc,5 coff,6 scratch floats; f: 0 eps; p: 0 raw, 1 stats float2[B][C] = (mean, rstd), 2 scratch|0
int fnet_op_in_stats(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && H > 0 && W > 0 && vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype) && op.p[1]);
    if (op.dtype == FNET_BF16 && cg_ok(C, B) && op.p[2]) {
        // one coalesced pass into per-block partials + ordered finalize (bf16 plans; p[2] = scratch of i[6] floats);
        // fp32 plans, or no scratch, keep the exact two-pass kernel below
        int chunks;
        const int per = cg_pixels_per_block(H * W, B, C, &chunks, 768);
        FNET_CHECK_ARG((long long)B * chunks * C * 2 <= (long long)op.i[6]);
        // (finalising inside the same launch -- last block of an image, agent-scope fences -- was built and measured: the
        // release fence in every block costs ~70 us per launch against 4.7 us for this second kernel)
        hipLaunchKernelGGL(k_in_stats_part<bf16_t>, dim3(chunks, B), dim3(kThreads), 0, st, (const bf16_t*)op.p[0], (float*)op.p[2], H, W, C,
                           op.i[4], op.i[5], per);
        hipLaunchKernelGGL(k_in_finalize, dim3((B * C + kThreads - 1) / kThreads), dim3(kThreads), 0, st, (const float2*)op.p[2],
                           (float2*)op.p[1], B, C, chunks, 1.f / (float)(H * W), op.f[0]);
        FNET_LAUNCH_CHECK();
        return FNET_OK;
    }
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_in_stats<T>, dim3(B * (C / 8)), dim3(kThreads), 0, st, (const T*)op.p[0],
                                            (float2*)op.p[1], B, H, W, C, op.i[4], op.i[5], op.f[0]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// APPLY  i: 0 B,1 H,2 W,3 C,4 ldc_raw,5 coff_raw,6 ldc_res,7 coff_res,8 ldc_dst,9 coff_dst,10 act,11 pm,12 drop site,13 seed
//        f: 0 drop_p; p: 0 raw,1 stats|0,2 gamma|0,3 beta|0,4 res|0,5 dst
int fnet_op_apply(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2);
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok_opt((void*)op.p[4], op.i[6], op.i[7], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[5], op.i[8], op.i[9], C, op.dtype));
    FNET_CHECK_ARG(op.i[10] >= 0 && op.i[10] <= 4 && op.i[11] >= 0 && op.i[11] <= 3);
    FNET_CHECK_ARG(op.p[1] == 0 || (op.p[2] != 0 && op.p[3] != 0));
    FNET_CHECK_ARG(op.f[0] >= 0.f && op.f[0] < 1.f && cg_ok(C, B) && off_ok(H, W, op.i[4], op.p[4] ? op.i[6] : 8, op.i[8]));
    int chunks;
    const int per = cg_pixels_per_block((H + 2) * (W + 2), B, C, &chunks, 768);
    DISPATCH_ACT(op.i[10], DISPATCH_T(op.dtype, hipLaunchKernelGGL((k_apply<T, ACT>), dim3(chunks, B), dim3(kThreads), 0, st,
                                            (const T*)op.p[0], (const float2*)op.p[1], (const float*)op.p[2],
                                            (const float*)op.p[3], (const T*)op.p[4], (T*)op.p[5], H, W, C, op.i[4],
                                            op.i[5], op.i[6], op.i[7], op.i[8], op.i[9], op.i[10], op.i[11],
                                            (uint32_t)op.i[12], (uint32_t)op.i[13], op.f[0], per)));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// IN_STATS immediately followed by the APPLY of the same unit (InstanceNorm: nothing pools the statistics in between), on a small image:
// one k_stats_apply launch instead of three.  fnet_plan_run asks fnet_can_fuse_stats_apply first; anything it does not cover -- and any
// record that would fail its own checks -- takes the two ordinary launchers (which then report the error).  FNET_FUSE_SMALL=0 disables it.
static long long sa_max_pixels() {       // padded pixels per image up to which the pair is fused (FNET_SA_MAXHW = the image side, tuning knob)
    static const long long v = [] { const char* e = getenv("FNET_SA_MAXHW"); const long long s = e ? atoll(e) : 64; return (s + 2) * (s + 2); }();
    return v;
}
static bool sa_enough_items(int B, int C) {
    // (96: K2 256^2 B = 32 gains with 128 blocks, 823 against 802 samples/s; a batch-1 frame -- 4 to 16 blocks -- loses; the kernel
    // tests, which use small batches, set the knob to 1)
    const long long need = fnet_knob(FNET_KNOB_SA_MINITEMS), items = (long long)B * (C / 16);
    return items >= need && items < (1ll << 30);
}
bool fnet_can_fuse_stats_apply(const FnetOp& so, const FnetOp& ao) {
    static const bool on = [] { const char* e = getenv("FNET_FUSE_SMALL"); return !e || e[0] != '0'; }();
    if (!on || so.dtype != FNET_BF16 || ao.dtype != FNET_BF16 || so.i[FNET_LANE_SLOT] != ao.i[FNET_LANE_SLOT]) return false;
    for (int k = 0; k < 6; ++k)
        if (so.i[k] != ao.i[k]) return false;                       // B, H, W, C, ldc, coff of raw
    const int B = ao.i[0], H = ao.i[1], W = ao.i[2], C = ao.i[3];
    if (so.p[0] != ao.p[0] || !so.p[1] || so.p[1] != ao.p[1] || !ao.p[2] || !ao.p[3] || !ao.p[5]) return false;
    if (B <= 0 || H < 2 || W < 2 || C % 16 != 0 || (long long)(H + 2) * (W + 2) > sa_max_pixels()) return false;
    if (!vec_ok((void*)ao.p[0], ao.i[4], ao.i[5], C, ao.dtype) || !vec_ok_opt((void*)ao.p[4], ao.i[6], ao.i[7], C, ao.dtype) ||
        !vec_ok((void*)ao.p[5], ao.i[8], ao.i[9], C, ao.dtype))
        return false;
    if (ao.i[10] < 0 || ao.i[10] > 4 || ao.i[11] < 0 || ao.i[11] > 3 || !(ao.f[0] >= 0.f && ao.f[0] < 1.f) || !cg_ok(C, B)) return false;
    // one block per (image, slice): worth it only with enough blocks (a batch-1 rollout frame keeps the separate launches: 0.83 against
    // 0.89 ms per frame)
    return off_ok(H, W, ao.i[4], ao.p[4] ? ao.i[6] : 8, ao.i[8]) && sa_enough_items(B, C);
}
int fnet_op_stats_apply(const FnetOp& so, const FnetOp& op, hipStream_t st) {
    FNET_CHECK_ARG(fnet_can_fuse_stats_apply(so, op));
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    const int forced_cs = fnet_knob(FNET_KNOB_SA_CS);       // 16 | 32 | 64 forces the slice width (the tests run all of them)
    // the widest slice (whole 128-B / 64-B pixel segments) that still gives every CU a block
    int cs = (C % 64 == 0 && (long long)B * (C / 64) >= 256) ? 64 : (C % 32 == 0 && (long long)B * (C / 32) >= 256) ? 32 : 16;
    if (forced_cs == 16 || (forced_cs == 32 && C % 32 == 0) || (forced_cs == 64 && C % 64 == 0)) cs = forced_cs;
    const int items = B * (C / cs);
    using T = bf16_t;
#define FNET_SA_LAUNCH(CS_)                                                                                                                    \
    DISPATCH_ACT(op.i[10], hipLaunchKernelGGL((k_stats_apply<T, ACT, CS_>), dim3((items + 7) / 8 * 8), dim3(SA_THREADS), 0, st,                  \
                                              (const T*)op.p[0], (float2*)op.p[1], (const float*)op.p[2], (const float*)op.p[3],               \
                                              (const T*)op.p[4], (T*)op.p[5], B, H, W, C, op.i[4], op.i[5], op.i[6], op.i[7], op.i[8], op.i[9], \
                                              op.i[11], (uint32_t)op.i[12], (uint32_t)op.i[13], op.f[0], so.f[0]))
    if (cs == 64) { FNET_SA_LAUNCH(64); } else if (cs == 32) { FNET_SA_LAUNCH(32); } else { FNET_SA_LAUNCH(16); }
#undef FNET_SA_LAUNCH
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// INACT_BWD_REDUCE  i: 0 B,1 H,2 W,3 C,4 ldc_g1,5 coff_g1,6 ldc_g2,7 coff_g2,8 ldc_raw,9 coff_raw,10 ldc_G,11 coff_G,
//                      12 fold1,13 fold2,14 act,15 pm,16 drop site,17 seed; f: 0 drop_p
//                   p: 0 g1,1 g2|0,2 raw,3 stats,4 gamma,5 beta,6 G out,7 S float2[B][C] (+=: the plan zeroes it first),
//                      8 dgamma|0 (+=), 9 dbeta|0 (+=)
int fnet_op_inact_bwd_reduce(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2);
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok_opt((void*)op.p[1], op.i[6], op.i[7], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[2], op.i[8], op.i[9], C, op.dtype));
    FNET_CHECK_ARG(vec_ok_opt((void*)op.p[6], op.i[10], op.i[11], C, op.dtype));
    FNET_CHECK_ARG(op.p[6] != 0 || op.p[1] == 0);          // G may be omitted only for a single gradient source
    FNET_CHECK_ARG(op.p[3] && op.p[4] && op.p[5] && op.p[7] && cg_ok(C, B) && ((op.p[8] == 0) == (op.p[9] == 0)));
    FNET_CHECK_ARG(off_ok(H, W, op.i[4], op.p[1] ? op.i[6] : 8, op.i[8], op.p[6] ? op.i[10] : 8));
    int chunks;
    // The single-source form (110 VGPRs) gets 6 blocks per CU instead of the family's 3: beside a 232-register weight-gradient wave only two
    // of its blocks are resident per CU, and 3 per CU then run as a full round and a half-empty one (same-box A/B, K4 B64 128^2: 768 ->
    // 34.08 ms, 512 -> 33.93, 1536 -> 33.84, 3072 -> 33.86; alone the launch is 2 % slower than with 768)
    static const int p1one = [] { const char* e = getenv("FNET_P1ONE_BLOCKS"); return e ? atoi(e) : 1536; }();
    // (the general form too: 768 -> 1536 blocks 35.12 -> 34.94 ms; 2304: 34.32 against 34.19; every channel-group kernel at 1536: +0.25 ms)
    static const int p1two = [] { const char* e = getenv("FNET_P1TWO_BLOCKS"); return e ? atoi(e) : 1536; }();
    const int per = cg_pixels_per_block(H * W, B, C, &chunks, (op.p[1] == 0 && op.p[6] == 0) ? p1one : p1two);
#define FNET_P1_LAUNCH(ONE_)                                                                                                                   \
    DISPATCH_ACT(op.i[14], DISPATCH_T(op.dtype,                                                                                              \
               hipLaunchKernelGGL((k_inact_bwd_reduce<T, ACT, ONE_>), dim3(chunks, B), dim3(kThreads), 0, st, (const T*)op.p[0],              \
                                  (const T*)op.p[1], (const T*)op.p[2], (const float2*)op.p[3], (const float*)op.p[4],                       \
                                  (const float*)op.p[5], (T*)op.p[6], (float*)op.p[7], (float*)op.p[8], (float*)op.p[9], H, W, C, op.i[4], op.i[5], \
                                  op.i[6], op.i[7], op.i[8], op.i[9], op.i[10], op.i[11], op.i[12], op.i[13], op.i[14],                      \
                                  op.i[15], (uint32_t)op.i[16], (uint32_t)op.i[17], op.f[0], per)))
    // the single-source form without a materialised G (most units of a UNet) is its own instantiation: fewer registers
    if (op.p[1] == 0 && op.p[6] == 0) { FNET_P1_LAUNCH(true); } else { FNET_P1_LAUNCH(false); }
#undef FNET_P1_LAUNCH
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// INACT_BWD_APPLY  i: 0 B,1 H,2 W,3 C,4 ldc_G,5 coff_G,6 ldc_raw,7 coff_raw,8 ldc_d,9 coff_d,10 act,11 drop site,12 seed,
//                     13 fold (p[0] is the unit's single gradient source, ring not yet folded),14 pm,
//                     15 weighted S (BatchNorm2d / GroupNorm: p[5] = output of NORM_COMBINE_BWD)
//                  f: 0 drop_p; p: 0 G,1 raw,2 stats,3 gamma,4 beta,5 S,6 d_raw out
int fnet_op_inact_bwd_apply(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2);
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[1], op.i[6], op.i[7], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[6], op.i[8], op.i[9], C, op.dtype));
    FNET_CHECK_ARG(op.p[2] && op.p[3] && op.p[4] && op.p[5] && cg_ok(C, B) && off_ok(H, W, op.i[4], op.i[6], op.i[8]));
    FNET_CHECK_ARG((op.i[13] == 0 || op.i[13] == 1) && op.i[14] >= 0 && op.i[14] <= 3);
    int chunks;
    const int per = cg_pixels_per_block((H + 2) * (W + 2), B, C, &chunks, 768);
#define INACT_APPLY_LAUNCH(WS_)                                                                                                     \
    DISPATCH_ACT(op.i[10], DISPATCH_T(op.dtype, hipLaunchKernelGGL((k_inact_bwd_apply<T, ACT, WS_>), dim3(chunks, B), dim3(kThreads), 0, \
                                            st, (const T*)op.p[0], (const T*)op.p[1], (const float2*)op.p[2],                        \
                                            (const float*)op.p[3], (const float*)op.p[4], (const float2*)op.p[5],                    \
                                            (T*)op.p[6], H, W, C, op.i[4], op.i[5], op.i[6], op.i[7], op.i[8], op.i[9],              \
                                            op.i[10], (uint32_t)op.i[11], (uint32_t)op.i[12], op.f[0], per, op.i[13], op.i[14])))
    FNET_CHECK_ARG(op.i[15] == 0 || op.i[15] == 1);
    if (op.i[15]) { INACT_APPLY_LAUNCH(true); } else { INACT_APPLY_LAUNCH(false); }
#undef INACT_APPLY_LAUNCH
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// NORM_COMBINE  i: 0 B,1 C,2 kind (1 batch/training, 2 group, 3 batch/eval),3 channels per group (kind 2),4 H*W,5 real C (0 = C)
//               f: 0 eps,1 momentum; p: 0 stats float2[B][C] (in place),1 running_mean|0,2 running_var|0 (kind 1: updated when given;
//               kind 3: read)
int fnet_op_norm_combine(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], C = op.i[1], kind = op.i[2], gs = op.i[3], Cr = op.i[5] ? op.i[5] : C;
    FNET_CHECK_ARG(B > 0 && C > 0 && kind >= 1 && kind <= 3 && op.p[0] && op.i[4] > 0 && Cr >= 1 && Cr <= C);
    FNET_CHECK_ARG(kind != 2 || (gs >= 1 && Cr % gs == 0));
    FNET_CHECK_ARG(kind != 3 || (op.p[1] && op.p[2]));
    FNET_CHECK_ARG((op.p[1] == 0) == (op.p[2] == 0));
    const int items = kind == 1 ? C : kind == 2 ? B * (Cr / gs) : B * C;
    hipLaunchKernelGGL(k_norm_combine, dim3((items + kThreads - 1) / kThreads), dim3(kThreads), 0, st, (float2*)op.p[0], (float*)op.p[1],
                       (float*)op.p[2], B, C, kind, gs, op.f[0], op.f[1], (float)B * (float)op.i[4], Cr);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// NORM_COMBINE_BWD  i: 0 B,1 C,2 kind,3 channels per group,4 real C (0 = C); p: 0 S float2[B][C],1 gamma,2 Sw float2[B][C] out
int fnet_op_norm_combine_bwd(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], C = op.i[1], kind = op.i[2], gs = op.i[3], Cr = op.i[4] ? op.i[4] : C;
    FNET_CHECK_ARG(B > 0 && C > 0 && kind >= 1 && kind <= 3 && op.p[0] && op.p[1] && op.p[2] && Cr >= 1 && Cr <= C);
    FNET_CHECK_ARG(kind != 2 || (gs >= 1 && Cr % gs == 0));
    const int items = kind == 1 ? C : kind == 2 ? B * (Cr / gs) + B * (C - Cr) : B * C;
    hipLaunchKernelGGL(k_norm_combine_bwd, dim3((items + kThreads - 1) / kThreads), dim3(kThreads), 0, st, (const float2*)op.p[0],
                       (const float*)op.p[1], (float2*)op.p[2], B, C, kind, gs, Cr);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// AFFINE_GRAD  i: 0 B,1 C; p: 0 S float2[B][C],1 dgamma (+=),2 dbeta (+=)
int fnet_op_affine_grad(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], C = op.i[1];
    FNET_CHECK_ARG(B > 0 && C > 0 && op.p[0] && op.p[1] && op.p[2]);
    hipLaunchKernelGGL(k_affine_grad, dim3((C + 63) / 64), dim3(kThreads), 0, st, (const float2*)op.p[0],
                       (float*)op.p[1], (float*)op.p[2], B, C);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// AFFINE_GRAD_MULTI  i: 0 B,1 n_units (<= 6), per unit u at 2+4u: S offset (floats),C,dgamma offset,dbeta offset (floats)
//                    p: 0 S slab base,1 flat gradient buffer (+=)
int fnet_op_affine_grad_multi(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], n = op.i[1];
    FNET_CHECK_ARG(B > 0 && n >= 1 && n <= 6 && op.p[0] && op.p[1] && (op.p[0] % 8) == 0);
    AffineUnits u;
    u.n = n;
    int cmax = 0;
    for (int j = 0; j < 7; ++j) {
        const int k = j < n ? j : 0;
        u.s_off[j] = op.i[2 + 4 * k]; u.C[j] = op.i[3 + 4 * k]; u.dg_off[j] = op.i[4 + 4 * k]; u.db_off[j] = op.i[5 + 4 * k];
        FNET_CHECK_ARG(u.s_off[j] >= 0 && u.s_off[j] % 2 == 0 && u.C[j] > 0 && u.dg_off[j] >= -1 && u.db_off[j] >= 0);
        if (u.C[j] > cmax) cmax = u.C[j];
    }
    hipLaunchKernelGGL(k_affine_grad_multi, dim3((cmax + 63) / 64, n), dim3(kThreads), 0, st, (const float*)op.p[0], (float*)op.p[1], u, B);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// INACT_BWD_REDUCE immediately followed by the INACT_BWD_APPLY of the same single-source InstanceNorm unit (no G, nothing pools S in
// between), on a small image: one k_inact_bwd_fused launch instead of two.  Same contract as fnet_can_fuse_stats_apply.
bool fnet_can_fuse_inact_bwd(const FnetOp& ro, const FnetOp& ao) {
    static const bool on = [] { const char* e = getenv("FNET_FUSE_SMALL"); return !e || e[0] != '0'; }();
    if (!on || ro.dtype != FNET_BF16 || ao.dtype != FNET_BF16 || ro.i[FNET_LANE_SLOT] != ao.i[FNET_LANE_SLOT]) return false;
    for (int k = 0; k < 4; ++k)
        if (ro.i[k] != ao.i[k]) return false;                       // B, H, W, C
    const int B = ao.i[0], H = ao.i[1], W = ao.i[2], C = ao.i[3];
    if (ro.p[1] || ro.p[6] || ro.p[8] || ro.p[9] || ao.i[15] != 0) return false;          // one source, no G, no in-kernel affine gradients, plain S
    if (!ro.p[0] || ro.p[0] != ao.p[0] || ro.i[4] != ao.i[4] || ro.i[5] != ao.i[5]) return false;      // pass 2 reads the same gradient source
    if (!ro.p[2] || ro.p[2] != ao.p[1] || ro.i[8] != ao.i[6] || ro.i[9] != ao.i[7]) return false;      // ... the same raw
    if (!ro.p[3] || ro.p[3] != ao.p[2] || !ro.p[4] || ro.p[4] != ao.p[3] || !ro.p[5] || ro.p[5] != ao.p[4] || !ro.p[7] || ro.p[7] != ao.p[5] || !ao.p[6])
        return false;
    if (ro.i[14] != ao.i[10] || ro.i[15] != ao.i[14] || ro.i[16] != ao.i[11] || ro.i[17] != ao.i[12] || ro.f[0] != ao.f[0]) return false;
    if ((ro.i[12] != 0) != (ao.i[13] != 0) || (ao.i[13] != 0 && ao.i[13] != 1)) return false;
    // (<= 32^2 only: at 64^2 the fused launch measured the same as the two -- 52.6 against 51.6 us, 86.6 against 82.8 -- these passes are
    // bound by their arithmetic there, not by launch costs; FNET_SB_MAXHW = the image side, tuning knob)
    const long long sbv = fnet_knob(FNET_KNOB_SB_MAXHW), sb_max = (sbv + 2) * (sbv + 2);      // (the kernel tests raise it to cover 64^2)
    if (B <= 0 || H < 2 || W < 2 || C % 16 != 0 || (long long)(H + 2) * (W + 2) > sb_max) return false;
    if (!vec_ok((void*)ao.p[0], ao.i[4], ao.i[5], C, ao.dtype) || !vec_ok((void*)ao.p[1], ao.i[6], ao.i[7], C, ao.dtype) ||
        !vec_ok((void*)ao.p[6], ao.i[8], ao.i[9], C, ao.dtype))
        return false;
    if (ao.i[10] < 0 || ao.i[10] > 4 || ao.i[14] < 0 || ao.i[14] > 3 || !(ao.f[0] >= 0.f && ao.f[0] < 1.f) || !cg_ok(C, B)) return false;
    return off_ok(H, W, ao.i[4], ao.i[6], ao.i[8]) && sa_enough_items(B, C);
}
int fnet_op_inact_bwd_fused(const FnetOp& ro, const FnetOp& op, hipStream_t st) {
    FNET_CHECK_ARG(fnet_can_fuse_inact_bwd(ro, op));
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    const int forced_cs = fnet_knob(FNET_KNOB_SA_CS);
    int cs = (C % 64 == 0 && (long long)B * (C / 64) >= 256) ? 64 : (C % 32 == 0 && (long long)B * (C / 32) >= 256) ? 32 : 16;
    if (forced_cs == 16 || (forced_cs == 32 && C % 32 == 0) || (forced_cs == 64 && C % 64 == 0)) cs = forced_cs;
    const int items = B * (C / cs);
    using T = bf16_t;
#define FNET_SB_LAUNCH(CS_)                                                                                                                    \
    DISPATCH_ACT(op.i[10], hipLaunchKernelGGL((k_inact_bwd_fused<T, ACT, CS_>), dim3((items + 7) / 8 * 8), dim3(SB_THREADS), 0, st,              \
                                              (const T*)op.p[0], (const T*)op.p[1], (const float2*)op.p[2], (const float*)op.p[3],             \
                                              (const float*)op.p[4], (float*)op.p[5], (T*)op.p[6], B, H, W, C, op.i[4], op.i[5], op.i[6],       \
                                              op.i[7], op.i[8], op.i[9], op.i[13], op.i[14], (uint32_t)op.i[11], (uint32_t)op.i[12], op.f[0]))
    if (cs == 64) { FNET_SB_LAUNCH(64); } else if (cs == 32) { FNET_SB_LAUNCH(32); } else { FNET_SB_LAUNCH(16); }
#undef FNET_SB_LAUNCH
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// FOLD_SUM  i: 0 B,1 H,2 W,3 C,4 ldc_g1,5 coff_g1,6 ldc_g2,7 coff_g2,8 ldc_d,9 coff_d,10 fold1,11 fold2,12 pm
//           p: 0 g1,1 g2|0,2 dst (interior = folded sum, ring = 0)
int fnet_op_fold_sum(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2);
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok_opt((void*)op.p[1], op.i[6], op.i[7], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[2], op.i[8], op.i[9], C, op.dtype));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_fold_sum<T>, row_grid(B * (H + 2), (W + 2) * (C / 8)), dim3(kThreads), 0, st,
                                            (const T*)op.p[0], (const T*)op.p[1], (T*)op.p[2], B, H, W, C, op.i[4], op.i[5],
                                            op.i[6], op.i[7], op.i[8], op.i[9], op.i[10], op.i[11], op.i[12]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// BIAS_GRAD  i: 0 B,1 H,2 W,3 C,4 ldc,5 coff; p: 0 d (P),1 dbias (+=, atomics)
int fnet_op_bias_grad(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && H > 0 && W > 0 && vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype) && op.p[1]);
    if (C <= 1024 && C / 8 <= kThreads) {
        const int ppb = kThreads / (C / 8);
        long long nb = ((long long)B * H * W + (long long)ppb * 8 - 1) / ((long long)ppb * 8);
        if (nb > 1024) nb = 1024;
        DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_bias_grad_cg<T>, dim3((unsigned)nb), dim3(kThreads), 0, st, (const T*)op.p[0],
                                                (float*)op.p[1], B, H, W, C, op.i[4], op.i[5]));
        FNET_LAUNCH_CHECK();
        return FNET_OK;
    }
    int split = (B * H * W + kThreads * 16 - 1) / (kThreads * 16);
    if (split < 1) split = 1;
    if (split > 128) split = 128;
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_bias_grad<T>, dim3(C / 8, split), dim3(kThreads), 0, st, (const T*)op.p[0],
                                            (float*)op.p[1], B, H, W, C, op.i[4], op.i[5]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// UPSAMPLE2X  i: 0 B,1 h,2 w,3 C,4 ldc_s,5 coff_s,6 ldc_d,7 coff_d,8 mode (0 nearest,1 bilinear),9 pm; p: 0 src (h,w),1 dst (2h,2w)
int fnet_op_upsample2x(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], h = op.i[1], w = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && h >= 1 && w >= 1 && (op.i[8] == 0 || op.i[8] == 1));
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[1], op.i[6], op.i[7], C, op.dtype));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_upsample2x<T>, row_grid(B * (2 * h + 2), (2 * w + 2) * (C / 8)), dim3(kThreads),
                                            0, st, (const T*)op.p[0], (T*)op.p[1], B, h, w, C, op.i[4], op.i[5], op.i[6],
                                            op.i[7], op.i[8], op.i[9]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// UPSAMPLE2X_BWD  i: 0 B,1 h,2 w,3 C,4 ldc_g,5 coff_g,6 ldc_d,7 coff_d,8 fold,9 mode,10 pm; p: 0 g (2h,2w),1 dst (h,w interior)
int fnet_op_upsample2x_bwd(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], h = op.i[1], w = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && h >= 1 && w >= 1 && (op.i[9] == 0 || op.i[9] == 1));
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[1], op.i[6], op.i[7], C, op.dtype));
    // strips of 8 coarse rows on the large images, 4 below 64 rows (swept on MI355X at B = 64: h = 64 / 32 / 16 take 80 / 42 / 31 us
    // with 8 / 4 / 4, against 100 / 54 / 29 us for the one-row-per-block form this kernel replaced), fewer where blocks get scarce
    const int colblocks = (w * (C / 8) + kThreads - 1) / kThreads;
    int S = h >= 64 ? UB_S : 4;
    while (S > 1 && (long long)B * ((h + S - 1) / S) * colblocks < 512) S >>= 1;
    const int strips = (h + S - 1) / S;
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_upsample2x_bwd<T>, row_grid(B * strips, w * (C / 8)), dim3(kThreads), 0, st,
                                            (const T*)op.p[0], (T*)op.p[1], B, h, w, C, op.i[4], op.i[5], op.i[6], op.i[7],
                                            op.i[8], op.i[9], op.i[10], S, strips));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// POOL2X  i: 0 B,1 h,2 w,3 C,4 ldc_s,5 coff_s,6 ldc_d,7 coff_d,8 mode (0 avg,1 max),9 pm; p: 0 src (2h,2w),1 dst (h,w, whole padded domain)
int fnet_op_pool2x(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], h = op.i[1], w = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && h >= 1 && w >= 1 && (op.i[8] == 0 || op.i[8] == 1));
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[1], op.i[6], op.i[7], C, op.dtype));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_pool2x<T>, row_grid(B * (h + 2), (w + 2) * (C / 8)), dim3(kThreads), 0, st,
                                            (const T*)op.p[0], (T*)op.p[1], B, h, w, C, op.i[4], op.i[5], op.i[6], op.i[7], op.i[8],
                                            op.i[9]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// POOL2X_BWD  i: 0 B,1 h,2 w,3 C,4 ldc_g,5 coff_g,6 ldc_x,7 coff_x,8 ldc_d,9 coff_d,10 fold,11 mode,12 pm
//             p: 0 g (h,w),1 x = forward input (2h,2w; max only, else 0),2 dst (2h,2w, whole padded domain, ring 0)
int fnet_op_pool2x_bwd(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], h = op.i[1], w = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && h >= 1 && w >= 1 && (op.i[11] == 0 || op.i[11] == 1) && (op.i[10] == 0 || op.i[10] == 1));
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    if (op.i[11]) FNET_CHECK_ARG(vec_ok((void*)op.p[1], op.i[6], op.i[7], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[2], op.i[8], op.i[9], C, op.dtype));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_pool2x_bwd<T>, row_grid(B * (2 * h + 2), (2 * w + 2) * (C / 8)), dim3(kThreads), 0, st,
                                            (const T*)op.p[0], (const T*)op.p[1], (T*)op.p[2], B, h, w, C, op.i[4], op.i[5], op.i[6],
                                            op.i[7], op.i[8], op.i[9], op.i[10], op.i[11], op.i[12]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// DEPTH2SPACE  i: 0 B,1 h,2 w,3 C,4 ldc_s,5 coff_s,6 ldc_d,7 coff_d,8 pm; p: 0 src (h,w; 4C channels),1 bias fp32 [C],2 dst (2h,2w; C channels)
int fnet_op_depth2space(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], h = op.i[1], w = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && h >= 1 && w >= 1 && op.p[1]);
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], 4 * C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[2], op.i[6], op.i[7], C, op.dtype));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_depth2space<T>, row_grid(B * (2 * h + 2), (2 * w + 2) * (C / 8)), dim3(kThreads), 0, st,
                                            (const T*)op.p[0], (const float*)op.p[1], (T*)op.p[2], B, h, w, C, op.i[4], op.i[5], op.i[6],
                                            op.i[7], op.i[8]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// SPACE2DEPTH  i: 0 B,1 h,2 w,3 C,4 ldc_g,5 coff_g,6 ldc_d,7 coff_d,8 fold,9 pm; p: 0 g (2h,2w; C channels),1 dst (h,w; 4C channels, ring 0)
int fnet_op_space2depth(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], h = op.i[1], w = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && h >= 1 && w >= 1 && (op.i[8] == 0 || op.i[8] == 1));
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[1], op.i[6], op.i[7], 4 * C, op.dtype));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_space2depth<T>, row_grid(B * (h + 2), (w + 2) * (4 * C / 8)), dim3(kThreads), 0, st,
                                            (const T*)op.p[0], (T*)op.p[1], B, h, w, C, op.i[4], op.i[5], op.i[6], op.i[7], op.i[8],
                                            op.i[9]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// SUBSAMPLE_BIAS  i: 0 B,1 H,2 W,3 C,4 ldc_s,5 coff_s,6 ldc_d,7 coff_d,8 pm; p: 0 src (H,W),1 bias|0,2 dst (H/2,W/2)
int fnet_op_subsample_bias(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0);
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[2], op.i[6], op.i[7], C, op.dtype));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_subsample_bias<T>, row_grid(B * (H / 2 + 2), (W / 2 + 2) * (C / 8)),
                                            dim3(kThreads), 0, st, (const T*)op.p[0], (const float*)op.p[1], (T*)op.p[2], B, H,
                                            W, C, op.i[4], op.i[5], op.i[6], op.i[7], op.i[8]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// ZERO_INSERT  i: 0 B,1 h,2 w,3 C,4 ldc_s,5 coff_s,6 ldc_d,7 coff_d; p: 0 src (h,w),1 dst (2h,2w, whole padded domain written)
int fnet_op_zero_insert(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], h = op.i[1], w = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && h >= 1 && w >= 1);
    FNET_CHECK_ARG(vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    FNET_CHECK_ARG(vec_ok((void*)op.p[1], op.i[6], op.i[7], C, op.dtype));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_zero_insert<T>, row_grid(B * (2 * h + 2), (2 * w + 2) * (C / 8)), dim3(kThreads),
                                            0, st, (const T*)op.p[0], (T*)op.p[1], B, h, w, C, op.i[4], op.i[5], op.i[6],
                                            op.i[7]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// PAD_RING  i: 0 B,1 H,2 W,3 C,4 ldc,5 coff,6 pm; p: 0 buf (in place)
int fnet_op_pad_ring(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && vec_ok((void*)op.p[0], op.i[4], op.i[5], C, op.dtype));
    const int items = (2 * (W + 2) + 2 * H) * (C / 8);
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_pad_ring<T>, dim3(B, (items + kThreads - 1) / kThreads), dim3(kThreads), 0, st,
                                            (T*)op.p[0], B, H, W, C, op.i[4], op.i[5], op.i[6]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// UNPACK_GRAD  i: 0 B,1 H,2 W,3 ldc,4 nch,5 pm; p: 0 d_in_pad (P),1 dx fp32 [B,nch,H,W]
int fnet_op_unpack_grad(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], ldc = op.i[3], nch = op.i[4];
    FNET_CHECK_ARG(B > 0 && H >= 2 && W >= 2 && ldc % 8 == 0 && nch > 0 && nch <= ldc && op.p[0] && op.p[1]);
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_unpack_grad<T>, row_grid(B * H, W), dim3(kThreads), 0, st, (const T*)op.p[0],
                                            (float*)op.p[1], B, H, W, ldc, nch, op.i[5]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// GATHER_BATCH  i: 0 B,1 H,2 W,3 K,4 Km,5 mask_first,6 Ttotal; p: 0 frames fp32 [5][Ttotal][H][W] (density, velx, velz, emitter,
//               collider),1 frame index int32[B],2 flip int32[B],3 x0 [B,6,H,W],4 y_seq [B,K,3,H,W],5 masks [B,Km,2,H,W]
int fnet_op_gather_batch(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], H = op.i[1], W = op.i[2], K = op.i[3], Km = op.i[4], Tt = op.i[6];
    FNET_CHECK_ARG(B > 0 && H > 0 && W > 0 && K >= 1 && Km >= 1 && Km <= K + 1 && (op.i[5] == 0 || op.i[5] == 1) && Tt >= 3);
    FNET_CHECK_ARG((long long)Tt * H * W < (1ll << 31) && (long long)(6 + 3 * K + 2 * Km) * H * W < (1ll << 31));
    for (int k = 0; k < 6; ++k) FNET_CHECK_ARG(op.p[k] != 0);
    const int planes = 6 + 3 * K + 2 * Km;
    hipLaunchKernelGGL(k_gather_batch, dim3((H * W + kThreads - 1) / kThreads, planes, B), dim3(kThreads), 0, st, (const float*)op.p[0],
                       (const int*)op.p[1], (const int*)op.p[2], (float*)op.p[3], (float*)op.p[4], (float*)op.p[5], H, W, K, Km, op.i[5], Tt);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// COPY_PLANES  i: 0 B,1 count,2 src batch stride,3 dst batch stride (floats); p: 0 src,1 dst
int fnet_op_copy_planes(const FnetOp& op, hipStream_t st) {
    const int B = op.i[0], count = op.i[1];
    FNET_CHECK_ARG(B > 0 && B < 65536 && count > 0 && op.p[0] && op.p[1]);
    int gx = (count + kThreads * 4 - 1) / (kThreads * 4);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_copy_planes, dim3(gx, B), dim3(kThreads), 0, st, (const float*)op.p[0], (float*)op.p[1], B, count,
                       op.i[2], op.i[3]);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// ZERO  i: 0 = bytes & 0x7fffffff, 1 = bytes >> 31; p: 0 ptr
int fnet_op_zero(const FnetOp& op, hipStream_t st) {
    const size_t bytes = (size_t)op.i[0] + ((size_t)op.i[1] << 31);
    FNET_CHECK_ARG(op.p[0] != 0 && bytes > 0);
    hipError_t e = hipMemsetAsync((void*)op.p[0], 0, bytes, st);
    if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    return FNET_OK;
}
