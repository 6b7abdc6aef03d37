// bf16 MFMA (v_mfma_f32_32x32x16_bf16) implicit-GEMM kernels for gfx950: the flattened 3x3 convolution
// (forward and data-gradient share it) and its weight gradient.
//
// Convolution: GEMM with M = flattened padded pixels, N = Cout, K = 9 taps x Cin.  Because activations are
// stored spatially padded (P-tensor), the input of tap (ty,tx) for output pixel f is simply pixel
// f + (ty-1)*Wp + (tx-1): one contiguous run of TM + 2*Wp + 2 pixels staged in LDS serves all nine taps of a
// 256-pixel tile, and the A-fragment address of a tap is the base address plus a wave-uniform offset.
//   block  = 256 threads = 4 waves; tile = 256 pixels x TN couts; wave tile = 64 pixels x TN (2 x TN/32 MFMA tiles)
//   LDS    = input run [TM+2Wp+2][16 ch] + weights [9][TN][16 ch], rows padded 32 B -> 48 B: with a 48-B pitch the
//            16 lanes of a ds_read_b128 group start on 16 distinct 4-bank slots (conflict-free, guide section 2)
//   A frag = 8 consecutive channels of one pixel (ds_read_b128), B frag = 8 consecutive ci of one cout row.
//
// Weight gradient: dWk[t][co][ci] += sum_f dout[f][co] * src[f+off_t][ci]: GEMM with K = pixels.  Both operands are
// pixel-major in memory, so both fragments are read with the gfx950 transposed LDS read ds_read_b64_tr_b16 from
// [pixel][32-channel] panels (64-B rows: the four rows of a transposed block tile all 64 banks).  Each wave owns one
// (co-panel, ci-panel) pair and keeps all nine taps' 32x32 accumulators in registers (144 VGPRs); the pixel axis is
// split across blocks and reduced with fp32 atomics whose lanes run along ci (128-B segments).
#include "fnet_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kThreads = 256;
constexpr int TM = 256;     // pixels per block
constexpr int KC = 16;      // channels per LDS stage (= K of one MFMA)
constexpr int PS = 48;      // LDS pitch of a 16-channel row, bytes

template <int TN>
__global__ __launch_bounds__(256) void k_conv3x3_mfma(const bf16_t* __restrict__ src, const bf16_t* __restrict__ Wk,
                                                      const float* __restrict__ bias, bf16_t* __restrict__ dst, int Npix, int Wp,
                                                      int Cin, int ldc_in, int coff_in, int Cout, int ldc_out, int coff_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = TN / 32;
    const int rows_in = TM + 2 * Wp + 2;
    unsigned char* lds_in = smem;
    unsigned char* lds_w = smem + (size_t)rows_in * PS;  // rows_in*48 is a multiple of 16
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const long long f0 = (long long)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;

    f32x16 acc[2][NT];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int ns = 0; ns < NT; ++ns)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ms][ns][i] = 0.f;

    const long long gbase = f0 - Wp - 1;
    for (int c0 = 0; c0 < Cin; c0 += KC) {
        // stage the input run: 2 x 16-B pieces per pixel
        for (int p = tid; p < rows_in * 2; p += kThreads) {
            const int row = p >> 1, half = p & 1;
            const long long g = gbase + row;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (g >= 0 && g < Npix) v = *reinterpret_cast<const uint4*>(src + (size_t)g * ldc_in + coff_in + c0 + half * 8);
            *reinterpret_cast<uint4*>(lds_in + row * PS + half * 16) = v;
        }
        // stage the weight slab [9][TN][16]
        for (int p = tid; p < 9 * TN * 2; p += kThreads) {
            const int row = p >> 1, half = p & 1;
            const int t = row / TN, col = row - t * TN;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (n0 + col < Cout) v = *reinterpret_cast<const uint4*>(Wk + ((size_t)t * Cout + n0 + col) * Cin + c0 + half * 8);
            *reinterpret_cast<uint4*>(lds_w + row * PS + half * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = (t / 3) * Wp + (t % 3);
            bf16x8 a[2], b[NT];
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
                a[ms] = *reinterpret_cast<const bf16x8*>(lds_in + (wave * 64 + ms * 32 + r + toff) * PS + h * 16);
#pragma unroll
            for (int ns = 0; ns < NT; ++ns)
                b[ns] = *reinterpret_cast<const bf16x8*>(lds_w + (t * TN + ns * 32 + r) * PS + h * 16);
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int ns = 0; ns < NT; ++ns)
                    acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ms], b[ns], acc[ms][ns], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (i&3) + 8*(i>>2) + 4*(lane>>5)
#pragma unroll
    for (int ns = 0; ns < NT; ++ns) {
        const int co = n0 + ns * 32 + r;
        if (co >= Cout) continue;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                const long long f = f0 + wave * 64 + ms * 32 + row;
                if (f < Npix) dst[(size_t)f * ldc_out + coff_out + co] = f2bf(acc[ms][ns][i] + bv);
            }
    }
}

// ------------------------------------------------------------------------------------------- weight gradient
constexpr int KP = 128;        // pixels per staged chunk
constexpr int PANEL_ROW = 64;  // bytes: 32 bf16 channels

// transposed fragment: lane (r = lane&31, h = lane>>5) gets M[k0 + 8h + j][c = r], j = 0..7, from a [rows][32] bf16 panel
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* panel, int k0, int lane) {
    const int h = lane >> 5;
    const int i = lane & 15;                   // index inside the 16-lane group
    const int q = i >> 2, p = i & 3;
    const int colbase = ((lane >> 4) & 1) * 16;  // lanes 16..31 / 48..63 take channels 16..31
    const unsigned char* a0 = panel + (size_t)(k0 + 8 * h + q) * PANEL_ROW + (colbase + 4 * p) * 2;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * PANEL_ROW));
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8, v);
}

// stage `rows` pixels starting at global pixel g0 into NP panels of [rows][32] (channels c_base .. c_base + 32*NP), zero outside
template <int NP>
__device__ __forceinline__ void stage_panels(unsigned char* lds, int rows, const bf16_t* __restrict__ src, long long g0, int Npix,
                                             int ldc, int coff, int c_base, int C, int tid) {
    const int pieces = rows * NP * 4;  // 16-B pieces: 4 per 32-channel row
    for (int p = tid; p < pieces; p += kThreads) {
        const int piece = p & 3, rp = p >> 2;
        const int row = rp % rows, panel = rp / rows;
        const int c = c_base + panel * 32 + piece * 8;
        const long long g = g0 + row;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (g >= 0 && g < Npix && c < C) v = *reinterpret_cast<const uint4*>(src + (size_t)g * ldc + coff + c);
        *reinterpret_cast<uint4*>(lds + ((size_t)panel * rows + row) * PANEL_ROW + piece * 16) = v;
    }
}

template <int NCO, int NCI>
__global__ __launch_bounds__(256) void k_wgrad_mfma(const bf16_t* __restrict__ src, const bf16_t* __restrict__ dout,
                                                    float* __restrict__ dWk, int Npix, int Wp, int Cin, int ldc_in, int coff_in,
                                                    int Cout, int ldc_do, int coff_do, int Cin_w, int chunks_per_block) {
    constexpr int KS = 4 / (NCO * NCI);  // waves that share one (co,ci) panel pair and split the k-steps
    constexpr int SROWS = KP + 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds_do[NCO * KP * PANEL_ROW];
    __shared__ __attribute__((aligned(16))) unsigned char lds_src[NCI * SROWS * PANEL_ROW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cop = wave % NCO, cip = (wave / NCO) % NCI, ksl = wave / (NCO * NCI);
    const int co_base = blockIdx.y * (32 * NCO), ci_base = blockIdx.z * (32 * NCI);
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const long long fbeg = (long long)blockIdx.x * chunks_per_block * KP;
    for (int ch = 0; ch < chunks_per_block; ++ch) {
        const long long f0 = fbeg + (long long)ch * KP;
        if (f0 >= Npix) break;  // uniform across the block
        stage_panels<NCO>(lds_do, KP, dout, f0, Npix, ldc_do, coff_do, co_base, Cout, tid);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            stage_panels<NCI>(lds_src, SROWS, src, f0 + (long long)(ty - 1) * Wp - 1, Npix, ldc_in, coff_in, ci_base, Cin, tid);
            __syncthreads();
            const unsigned char* pa = lds_do + (size_t)cop * KP * PANEL_ROW;
            const unsigned char* pb = lds_src + (size_t)cip * SROWS * PANEL_ROW;
            for (int ks = ksl; ks < KP / 16; ks += KS) {
                const bf16x8 a = tr_frag(pa, ks * 16, lane);
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const bf16x8 b = tr_frag(pb, ks * 16 + tx, lane);
                    acc[ty * 3 + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ty * 3 + tx], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    // D[row = co][col = ci]; lanes run along ci -> 128-B atomic segments
    const int ci = ci_base + cip * 32 + r;
    if (ci < Cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co_base + cop * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                if (co < Cout) atomicAdd(dWk + ((size_t)t * Cout + co) * Cin_w + ci, acc[t][i]);
            }
    }
}

template <typename K>
int set_lds_limit(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return FNET_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    return FNET_OK;
}

}  // namespace

int fnet_conv3x3_mfma(const FnetOp& op, hipStream_t st) {
    const int Npix = op.i[0], Wp = op.i[1], Cin = op.i[2], Cout = op.i[5];
    FNET_CHECK_ARG(op.dtype == FNET_BF16);
    FNET_CHECK_ARG(Cin % KC == 0 && op.i[3] % 8 == 0 && op.i[4] % 8 == 0);
    const int TN = Cout > 32 ? 64 : 32;
    const size_t lds = (size_t)(TM + 2 * Wp + 2) * PS + (size_t)9 * TN * PS;
    FNET_CHECK_ARG(lds <= 160 * 1024);
    const dim3 grid((Npix + TM - 1) / TM, (Cout + TN - 1) / TN);
    const bf16_t* src = (const bf16_t*)op.p[0];
    const bf16_t* Wk = (const bf16_t*)op.p[1];
    const float* bias = (const float*)op.p[2];
    bf16_t* dst = (bf16_t*)op.p[3];
    if (TN == 64) {
        int rc = set_lds_limit(k_conv3x3_mfma<64>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_conv3x3_mfma<64>, grid, dim3(kThreads), lds, st, src, Wk, bias, dst, Npix, Wp, Cin, op.i[3], op.i[4], Cout,
                           op.i[6], op.i[7]);
    } else {
        int rc = set_lds_limit(k_conv3x3_mfma<32>, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(k_conv3x3_mfma<32>, grid, dim3(kThreads), lds, st, src, Wk, bias, dst, Npix, Wp, Cin, op.i[3], op.i[4], Cout,
                           op.i[6], op.i[7]);
    }
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

int fnet_wgrad_mfma(const FnetOp& op, hipStream_t st) {
    const int Npix = op.i[0], Wp = op.i[1], Cin = op.i[2], Cout = op.i[5], Cin_w = op.i[9];
    FNET_CHECK_ARG(op.dtype == FNET_BF16);
    FNET_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0 && op.i[3] % 8 == 0 && op.i[4] % 8 == 0 && op.i[6] % 8 == 0 && op.i[7] % 8 == 0);
    const int nco = Cout > 32 ? 2 : 1, nci = Cin > 32 ? 2 : 1;
    const int tiles = ((Cout + 32 * nco - 1) / (32 * nco)) * ((Cin + 32 * nci - 1) / (32 * nci));
    const int nchunks = (Npix + KP - 1) / KP;
    // aim for ~4 blocks per CU over the whole launch, at least 2 chunks per block to amortise the atomics
    int splits = (1024 + tiles - 1) / tiles;
    if (splits > nchunks) splits = nchunks;
    if (splits < 1) splits = 1;
    int cpb = (nchunks + splits - 1) / splits;
    if (cpb < 2 && nchunks >= 2) cpb = 2;
    splits = (nchunks + cpb - 1) / cpb;
    const dim3 grid(splits, (Cout + 32 * nco - 1) / (32 * nco), (Cin + 32 * nci - 1) / (32 * nci));
    const bf16_t* src = (const bf16_t*)op.p[0];
    const bf16_t* dout = (const bf16_t*)op.p[1];
    float* dWk = (float*)op.p[2];
#define WG_LAUNCH(A, B)                                                                                                       \
    hipLaunchKernelGGL((k_wgrad_mfma<A, B>), grid, dim3(kThreads), 0, st, src, dout, dWk, Npix, Wp, Cin, op.i[3], op.i[4], Cout, \
                       op.i[6], op.i[7], Cin_w, cpb)
    if (nco == 2 && nci == 2) WG_LAUNCH(2, 2);
    else if (nco == 2 && nci == 1) WG_LAUNCH(2, 1);
    else if (nco == 1 && nci == 2) WG_LAUNCH(1, 2);
    else WG_LAUNCH(1, 1);
#undef WG_LAUNCH
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}
