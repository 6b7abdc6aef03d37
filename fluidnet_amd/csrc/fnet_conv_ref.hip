// Plain FMA implementations of the flattened 3x3 convolution and its weight gradient.  fp32 accumulate, any
// element type.  These are the exact ("fp32") product path and the on-GPU cross-check for the MFMA kernels;
// they trade speed for being obviously right.
//
// Flattened convolution over a padded P-tensor with Npix = B*(H+2)*(W+2) pixels and row pitch Wp = W+2:
//     dst[f][co] = bias[co] + sum_{t=0..8} sum_ci  Wk[t][co][ci] * src[f + (t/3-1)*Wp + (t%3-1)][ci]
// with src taken as zero outside [0, Npix).  Interior outputs equal the 2-D convolution of the padded image;
// ring outputs are don't-care for forward convs and exact for the dgrad use.
#include "fnet_common.h"

namespace {

constexpr int kThreads = 256;

template <typename T>
__global__ __launch_bounds__(256) void k_conv3x3_ref(const T* src, const T* Wk, const float* bias, T* dst, int Npix, int Wp, int Cin,
                                                     int ldc_in, int coff_in, int Cout, int ldc_out, int coff_out) {
    const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
    if (idx >= (long long)Npix * Cout) return;
    const int co = (int)(idx % Cout);
    const int f = (int)(idx / Cout);
    float acc = bias ? bias[co] : 0.f;
    for (int t = 0; t < 9; ++t) {
        const int g = f + (t / 3 - 1) * Wp + (t % 3 - 1);
        if (g < 0 || g >= Npix) continue;
        const T* s = src + (size_t)g * ldc_in + coff_in;
        const T* w = Wk + ((size_t)t * Cout + co) * Cin;
        for (int c0 = 0; c0 < Cin; c0 += 8) {
            Vec8<T> a, b;
            a.load(s + c0);
            b.load(w + c0);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc = fmaf(a.v[k], b.v[k], acc);
        }
    }
    st1<T>(dst + (size_t)f * ldc_out + coff_out + co, acc);
}

// dWk[t][co][ci] += sum_f src[f+off_t][ci] * dout[f][co]     (fp32, atomics across the pixel splits)
template <typename T>
__global__ __launch_bounds__(256) void k_wgrad_ref(const T* src, const T* dout, float* dWk, int Npix, int Wp, int Cin, int ldc_in,
                                                   int coff_in, int Cout, int ldc_do, int coff_do, int Cin_w, int chunk) {
    const int idx = blockIdx.x * kThreads + threadIdx.x;
    if (idx >= 9 * Cout * Cin) return;
    const int ci = idx % Cin;
    const int co = (idx / Cin) % Cout;
    const int t = idx / (Cin * Cout);
    const int off = (t / 3 - 1) * Wp + (t % 3 - 1);
    const int f0 = blockIdx.y * chunk;
    const int f1 = min(Npix, f0 + chunk);
    float acc = 0.f;
    for (int f = f0; f < f1; ++f) {
        const int g = f + off;
        if (g < 0 || g >= Npix) continue;
        acc = fmaf(ld1<T>(src + (size_t)g * ldc_in + coff_in + ci), ld1<T>(dout + (size_t)f * ldc_do + coff_do + co), acc);
    }
    atomicAdd(dWk + ((size_t)t * Cout + co) * Cin_w + ci, acc);
}

}  // namespace

int fnet_conv3x3_ref(const FnetOp& op, hipStream_t st) {
    const int Npix = op.i[0], Wp = op.i[1], Cin = op.i[2], Cout = op.i[5];
    const long long total = (long long)Npix * Cout;
    FNET_CHECK_ARG((total + kThreads - 1) / kThreads < (1ll << 31));
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_conv3x3_ref<T>, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                                            st, (const T*)op.p[0], (const T*)op.p[1], (const float*)op.p[2], (T*)op.p[3], Npix, Wp,
                                            Cin, op.i[3], op.i[4], Cout, op.i[6], op.i[7]));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

int fnet_wgrad_ref(const FnetOp& op, hipStream_t st) {
    const int Npix = op.i[0], Wp = op.i[1], Cin = op.i[2], Cout = op.i[5], Cin_w = op.i[9];
    const int nw = 9 * Cout * Cin;
    // enough pixel splits to fill the chip when the weight tensor is small
    int splits = (256 * 8 * kThreads + nw - 1) / nw;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    int chunk = (Npix + splits - 1) / splits;
    if (chunk < 64) chunk = 64;
    splits = (Npix + chunk - 1) / chunk;
    DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_wgrad_ref<T>, dim3((nw + kThreads - 1) / kThreads, splits), dim3(kThreads), 0, st,
                                            (const T*)op.p[0], (const T*)op.p[1], (float*)op.p[2], Npix, Wp, Cin, op.i[3], op.i[4],
                                            Cout, op.i[6], op.i[7], Cin_w, chunk));
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// CONV3X3  i: 0 Npix,1 Wp,2 Cin,3 ldc_in,4 coff_in,5 Cout,6 ldc_out,7 coff_out,8 impl,9 mode (MFMA: stride-2 forms, see fluidnet_ops.h)
//          p: 0 src (P),1 Wk [9][Cout][Cin] (same element type as src),2 bias fp32|0,3 dst (P)
int fnet_op_conv3x3(const FnetOp& op, hipStream_t st) {
    const int Npix = op.i[0], Wp = op.i[1], Cin = op.i[2], Cout = op.i[5];
    const int esz = op.dtype == FNET_BF16 ? 2 : 4;
    FNET_CHECK_ARG(Npix > 0 && Wp >= 3 && Cin > 0 && Cin % 8 == 0 && Cout > 0);
    if (op.i[9] != 0) {                     // stride-2 forms (fnet_conv_mfma.hip): MFMA only, channel ranges checked there
        FNET_CHECK_ARG(op.i[8] == FNET_CONV_MFMA && op.i[4] >= 0 && op.i[7] >= 0);
        return fnet_conv3x3_mfma(op, st);
    }
    FNET_CHECK_ARG(op.i[3] % 8 == 0 && op.i[4] % 8 == 0 && op.i[4] + Cin <= op.i[3]);
    FNET_CHECK_ARG(op.i[7] >= 0 && op.i[7] + Cout <= op.i[6]);
    FNET_CHECK_ARG(op.p[0] && op.p[1] && op.p[3] && op.p[0] % (8 * esz) == 0 && op.p[1] % (8 * esz) == 0);
    FNET_CHECK_ARG((long long)Npix * (op.i[3] > op.i[6] ? op.i[3] : op.i[6]) < (1ll << 40));
    if (op.i[8] == FNET_CONV_MFMA) return fnet_conv3x3_mfma(op, st);
    FNET_CHECK_ARG(op.i[8] == FNET_CONV_REF);
    return fnet_conv3x3_ref(op, st);
}

// WGRAD  i: 0 Npix,1 Wp,2 Cin,3 ldc_in,4 coff_in,5 Cout,6 ldc_do,7 coff_do,8 impl,9 Cin_w
//        p: 0 src (P, the conv's input),1 dout (P, gradient of the conv's output, ring = 0),2 dWk fp32 [9][Cout][Cin_w] (+=)
int fnet_op_wgrad(const FnetOp& op, hipStream_t st) {
    const int Npix = op.i[0], Wp = op.i[1], Cin = op.i[2], Cout = op.i[5], Cin_w = op.i[9];
    FNET_CHECK_ARG(Npix > 0 && Wp >= 3 && Cin > 0 && Cout > 0 && Cin_w >= Cin);
    FNET_CHECK_ARG(op.i[4] >= 0 && (op.i[13] != 0 || op.i[4] + Cin <= op.i[3]) && op.i[7] >= 0 && op.i[7] + Cout <= op.i[6]);
    FNET_CHECK_ARG(op.p[0] && op.p[1] && op.p[2] && (op.i[13] == 0 || op.i[8] == FNET_CONV_MFMA));
    FNET_CHECK_ARG((long long)9 * Cout * Cin < (1ll << 31));
    if (op.i[8] == FNET_CONV_MFMA) return fnet_wgrad_mfma(op, st);
    FNET_CHECK_ARG(op.i[8] == FNET_CONV_REF);
    return fnet_wgrad_ref(op, st);
}
