// Shared device/host helpers for libfluidnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fluidnet_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    // hardware conversion (v_cvt_pk_bf16_f32): round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t f2bf_pk(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// 8-channel vector load/store for P-tensors (C % 8 == 0, channel offsets % 8 == 0)
#ifndef FNET_NT_STORE
#define FNET_NT_STORE 0
#endif
template <typename T>
struct Vec8;
template <>
struct Vec8<float> {
    float v[8];
    __device__ __forceinline__ void load(const float* p) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    // streaming store: the consumer is a later kernel, so the lines need not stay in (and later be flushed from) this XCD's L2
    __device__ __forceinline__ void store_nt(float* p) const {
#if FNET_NT_STORE
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(p));
        __builtin_nontemporal_store((f32x4){v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4*>(p + 4));
#else
        store(p);
#endif
    }
};
template <>
struct Vec8<bf16_t> {
    float v[8];
    __device__ __forceinline__ void load(const bf16_t* p) {
        const uint4 a = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(w[k] << 16);
            v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void store(bf16_t* p) const {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = f2bf_pk(v[2 * k], v[2 * k + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __device__ __forceinline__ void store_nt(bf16_t* p) const {
#if FNET_NT_STORE
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = f2bf_pk(v[2 * k], v[2 * k + 1]);
        __builtin_nontemporal_store((u32x4){w[0], w[1], w[2], w[3]}, reinterpret_cast<u32x4*>(p));
#else
        store(p);
#endif
    }
};

// Eight channels as loaded (still packed): the multi-pixel trips of the channel-group kernels issue the loads of all their pixels into
// Raw8 registers, fence the instruction scheduler, and only then unpack.  With Vec8::load (load + unpack in one) hipcc hoisted the first
// pixel's unpack in front of the later pixels' loads and put `s_waitcnt vmcnt(1)` there: a wave had 2 of its 6 loads in flight, then 4.
template <typename T>
struct Raw8;
template <>
struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
template <>
struct Raw8<bf16_t> {
    uint4 a;
    __device__ __forceinline__ void load(const bf16_t* p) { a = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void unpack(float (&v)[8]) const {
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(w[k] << 16);
            v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
    }
};

// 16-byte streaming store of an already packed vector (convolution epilogue)
__device__ __forceinline__ void st_stream16(bf16_t* p, const uint4& v) {
#if FNET_NT_STORE
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p));
#else
    *reinterpret_cast<uint4*>(p) = v;
#endif
}

template <typename T>
__device__ __forceinline__ float ld1(const T* p);
template <>
__device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T>
__device__ __forceinline__ void st1(T* p, float v);
template <>
__device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// ---------------------------------------------------------------- padding-ring index maps
// Source interior index for padded-domain index i in [-1, n]; -1 means "constant zero".
__host__ __device__ __forceinline__ int pad_src(int i, int n, int pm) {
    if (i >= 0 && i < n) return i;
    switch (pm) {
        case FNET_PAD_REPLICATE: return i < 0 ? 0 : n - 1;
        case FNET_PAD_REFLECT: return i < 0 ? 1 : n - 2;
        case FNET_PAD_CIRCULAR: return i < 0 ? n - 1 : 0;
        default: return -1;
    }
}
// Padded coordinates (0..n+1) whose value is a copy of interior index i (the adjoint "fold" set).
__host__ __device__ __forceinline__ int fold_list(int i, int n, int pm, int out[3]) {
    int k = 0;
    out[k++] = i + 1;
    if (pm != FNET_PAD_ZEROS) {
        if (pad_src(-1, n, pm) == i) out[k++] = 0;
        if (pad_src(n, n, pm) == i) out[k++] = n + 1;
    }
    return k;
}

// ---------------------------------------------------------------- activations (ml/models/unet.py:31-39)
// Phi(u) = 0.5*(1+erf(u/sqrt2)) via Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, i.e. fp32 rounding level) evaluated
// without cancellation on the negative side; E returns exp(-u^2/2), which the GELU derivative needs as well.  Two
// transcendental issues (v_rcp, v_exp) + 9 FMAs instead of libm erff's ~40 instructions: the elementwise kernels are
// VALU-bound on GELU, not HBM-bound, with the libm version.
__device__ __forceinline__ float gelu_phi(float u, float& E) {
    // w = |u| * sqrt(log2(e) / 2): E = exp(-u^2/2) = exp2(-w^2) is one multiply + v_exp_f32; the A&S argument z = |u|/sqrt2 = w / sqrt(log2 e)
    // is folded into the rational's coefficient, the 0.5 of 0.5*erfc into the polynomial's
    const float w = fabsf(u) * 0.84932180028801904f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.2727374809f, w, 1.f));        // 0.3275911 / sqrt(log2 e)
    E = __builtin_amdgcn_exp2f(-w * w);
    float p = fmaf(0.5307027145f, t, -0.7265760135f);
    p = fmaf(p, t, 0.7107068705f);
    p = fmaf(p, t, -0.142248368f);
    p = fmaf(p, t, 0.127414796f);
    const float q = p * t * E;             // = 0.5*erfc(|u|/sqrt2)
    return u < 0.f ? q : 1.f - q;
}
template <int ACT>
__device__ __forceinline__ float act_fwd_t(float u) {
    if (ACT == FNET_ACT_RELU) return u > 0.f ? u : 0.f;
    if (ACT == FNET_ACT_LEAKY) return u > 0.f ? u : 0.1f * u;
    if (ACT == FNET_ACT_GELU) { float E; return u * gelu_phi(u, E); }
    if (ACT == FNET_ACT_SILU) return u * __builtin_amdgcn_rcpf(1.f + __expf(-u));
    return u;
}
template <int ACT>
__device__ __forceinline__ float act_bwd_t(float u) {
    if (ACT == FNET_ACT_RELU) return u > 0.f ? 1.f : 0.f;
    if (ACT == FNET_ACT_LEAKY) return u > 0.f ? 1.f : 0.1f;
    if (ACT == FNET_ACT_GELU) { float E; const float phi = gelu_phi(u, E); return fmaf(u * 0.3989422804014327f, E, phi); }
    if (ACT == FNET_ACT_SILU) {
        const float s = __builtin_amdgcn_rcpf(1.f + __expf(-u));
        return s * (1.f + u * (1.f - s));
    }
    return 1.f;
}
__device__ __forceinline__ float act_fwd(float u, int act) {
    switch (act) {
        case FNET_ACT_RELU: return u > 0.f ? u : 0.f;
        case FNET_ACT_LEAKY: return u > 0.f ? u : 0.1f * u;
        case FNET_ACT_GELU: { float E; return u * gelu_phi(u, E); }
        case FNET_ACT_SILU: return u / (1.f + __expf(-u));
        default: return u;
    }
}
__device__ __forceinline__ float act_bwd(float u, int act) {
    switch (act) {
        case FNET_ACT_RELU: return u > 0.f ? 1.f : 0.f;
        case FNET_ACT_LEAKY: return u > 0.f ? 1.f : 0.1f;
        case FNET_ACT_GELU: { float E; const float phi = gelu_phi(u, E); return fmaf(u * 0.3989422804014327f, E, phi); }
        case FNET_ACT_SILU: {
            const float s = 1.f / (1.f + __expf(-u));
            return s * (1.f + u * (1.f - s));
        }
        default: return 1.f;
    }
}

// ---------------------------------------------------------------- Dropout2d channel mask
// Counter-based: keep(n,c) depends only on (seed, site id, n, c) so that forward and backward agree and the
// host can reproduce the mask (fluidnet_amd/engine/dropout.py).  Returns the multiplier 0 or 1/(1-p).
__host__ __device__ __forceinline__ uint32_t fnet_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ float drop_scale(uint32_t seed, uint32_t site, uint32_t n, uint32_t c, float p) {
    if (p <= 0.f) return 1.f;
    uint32_t h = fnet_mix32(seed ^ fnet_mix32(site * 0x9E3779B1u + 0x7F4A7C15u));
    h = fnet_mix32(h ^ (n * 0x85EBCA77u + 0x165667B1u));
    h = fnet_mix32(h ^ (c * 0xC2B2AE3Du + 0x27D4EB2Fu));
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);  // [0,1)
    return u < p ? 0.f : 1.f / (1.f - p);
}

// ---------------------------------------------------------------- block reductions (256 threads = 4 waves)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// host-side launch checking
extern int g_fnet_last_hip_error;
// tuning knobs that tests change inside one process: read from the environment ONCE (first use), then only through fnet_set_knob
// (fnet_plan.hip; include/fluidnet_hip.h) -- no getenv on the launch path
enum { FNET_KNOB_SA_MINITEMS = 0, FNET_KNOB_SA_CS, FNET_KNOB_SB_MAXHW, FNET_KNOB_WG_PX, FNET_KNOB_FUSE_HEAD, FNET_KNOB_WG_XCD, FNET_KNOB_COUNT };
int fnet_knob(int id);
#define FNET_CHECK_ARG(cond)                     \
    do {                                         \
        if (!(cond)) return FNET_E_BADARG;       \
    } while (0)
#define FNET_LAUNCH_CHECK()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) {                 \
            g_fnet_last_hip_error = (int)e__;    \
            return FNET_E_HIP;                   \
        }                                        \
    } while (0)

#define DISPATCH_ACT(act, ...)                                                  \
    switch (act) {                                                              \
        case FNET_ACT_RELU: { constexpr int ACT = FNET_ACT_RELU; __VA_ARGS__; } break;       \
        case FNET_ACT_LEAKY: { constexpr int ACT = FNET_ACT_LEAKY; __VA_ARGS__; } break;     \
        case FNET_ACT_GELU: { constexpr int ACT = FNET_ACT_GELU; __VA_ARGS__; } break;       \
        case FNET_ACT_SILU: { constexpr int ACT = FNET_ACT_SILU; __VA_ARGS__; } break;       \
        case FNET_ACT_IDENTITY: { constexpr int ACT = FNET_ACT_IDENTITY; __VA_ARGS__; } break; \
        default: return FNET_E_BADARG;                                          \
    }

#define DISPATCH_T(dtype, ...)                                       \
    if ((dtype) == FNET_BF16) { using T = bf16_t; __VA_ARGS__; }      \
    else if ((dtype) == FNET_F32) { using T = float; __VA_ARGS__; }   \
    else return FNET_E_BADARG;

// op launchers (one per opcode), defined across the .hip files
#define FNET_DECL_OP(name) int name(const FnetOp& op, hipStream_t st)
FNET_DECL_OP(fnet_op_pack_input);
FNET_DECL_OP(fnet_op_conv3x3);
FNET_DECL_OP(fnet_op_wgrad);
FNET_DECL_OP(fnet_op_in_stats);
FNET_DECL_OP(fnet_op_apply);
FNET_DECL_OP(fnet_op_upsample2x);
FNET_DECL_OP(fnet_op_subsample_bias);
FNET_DECL_OP(fnet_op_pad_ring);
FNET_DECL_OP(fnet_op_head_fwd);
FNET_DECL_OP(fnet_op_loss_reduce);
FNET_DECL_OP(fnet_op_loss_finalize);
FNET_DECL_OP(fnet_op_loss_grad);
FNET_DECL_OP(fnet_op_head_bwd);
FNET_DECL_OP(fnet_op_inact_bwd_reduce);
FNET_DECL_OP(fnet_op_inact_bwd_apply);
FNET_DECL_OP(fnet_op_affine_grad);
FNET_DECL_OP(fnet_op_affine_grad_multi);
FNET_DECL_OP(fnet_op_metrics);
FNET_DECL_OP(fnet_op_gather_batch);
FNET_DECL_OP(fnet_op_pool2x);
FNET_DECL_OP(fnet_op_pool2x_bwd);
FNET_DECL_OP(fnet_op_norm_combine);
FNET_DECL_OP(fnet_op_norm_combine_bwd);
FNET_DECL_OP(fnet_op_depth2space);
FNET_DECL_OP(fnet_op_space2depth);
FNET_DECL_OP(fnet_op_fold_sum);
FNET_DECL_OP(fnet_op_bias_grad);
FNET_DECL_OP(fnet_op_zero_insert);
FNET_DECL_OP(fnet_op_upsample2x_bwd);
FNET_DECL_OP(fnet_op_unpack_grad);
FNET_DECL_OP(fnet_op_zero);
FNET_DECL_OP(fnet_op_prep_weights);
bool fnet_can_fuse_stats_apply(const FnetOp& stats_op, const FnetOp& apply_op);                          // IN_STATS + APPLY of one unit, small image
int fnet_op_stats_apply(const FnetOp& stats_op, const FnetOp& apply_op, hipStream_t st);
bool fnet_can_fuse_inact_bwd(const FnetOp& reduce_op, const FnetOp& apply_op);                             // INACT_BWD_REDUCE + INACT_BWD_APPLY, single source, small image
int fnet_op_inact_bwd_fused(const FnetOp& reduce_op, const FnetOp& apply_op, hipStream_t st);
bool fnet_can_fuse_head_loss_fwd(const FnetOp& head_op, const FnetOp& loss_op);                            // HEAD_FWD + the LOSS_REDUCE of its prediction
int fnet_op_head_loss_fwd(const FnetOp& head_op, const FnetOp& loss_op, hipStream_t st);
bool fnet_can_fuse_loss_head_bwd(const FnetOp& loss_grad_op, const FnetOp& head_bwd_op);                   // LOSS_GRAD + the HEAD_BWD that consumes its dpred
int fnet_op_loss_head_bwd(const FnetOp& loss_grad_op, const FnetOp& head_bwd_op, hipStream_t st);
int fnet_op_prep_weights_batch(const FnetOp* ops, int n, hipStream_t st, int* n_done);   // consecutive PREP_WEIGHTS records, one launch per 32
int fnet_op_unprep_wgrad_batch(const FnetOp* ops, int n, hipStream_t st, int* n_done);  // consecutive UNPREP_WGRAD records, one launch per 32
FNET_DECL_OP(fnet_op_grad_norm);
FNET_DECL_OP(fnet_op_adam);
FNET_DECL_OP(fnet_op_copy_planes);
FNET_DECL_OP(fnet_op_unprep_wgrad);
FNET_DECL_OP(fnet_op_stencil);

// conv implementations (fnet_conv_ref.hip / fnet_conv_mfma.hip)
int fnet_conv3x3_ref(const FnetOp& op, hipStream_t st);
int fnet_conv3x3_mfma(const FnetOp& op, hipStream_t st);
int fnet_wgrad_ref(const FnetOp& op, hipStream_t st);
int fnet_wgrad_mfma(const FnetOp& op, hipStream_t st);
