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
#include <stdlib.h>

#include <array>
#include <mutex>
#include <unordered_map>

#include "fnet_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kThreads = 256;
constexpr int TM = 256;     // pixels per block
constexpr int KC = 16;      // channels per LDS stage (= K of one MFMA)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// ---- convolution: persistent blocks, LDS-DMA double-buffered --------------------------------------------------
// LDS (one dynamic array) = two stage buffers of BUF bytes: [weights 9*TN rows x 32 B | input RINP pixels x 32 B].
//   input image  : RINP pixels (TM + 2Wp + 2 rounded up to 32) x 16 channels = 32 B per pixel, two 16-B slots
//   weight image : 9*TN rows (tap, cout) x 16 channels = 32 B per row, from the K-blocked global layout
//                  [Cin/16][9][Cout][16] (PREP_WEIGHTS), so one stage of weights is a contiguous slab
// Both images are filled by LDS-DMA (buffer_load ... lds: no VGPR round trip, no ds_write): one wave instruction moves
// 64 x 16 B = 32 rows, landing linearly in LDS; the bank swizzle therefore lives in the per-lane SOURCE address: row p
// keeps its two 16-B halves swapped when bit 3 of p is set, slot(p,h) = 2p + (h ^ ((p>>3)&1)), which makes every
// ds_read_b128 fragment read (16 lanes = 16 consecutive rows, same half) hit 16 distinct slots.  (Buffer form for both:
// a FLAT-encoded global_load_lds makes hipcc wait lgkmcnt(0) before every fragment use.)  The input goes through a buffer
// descriptor windowed on the tile's pixel run, so pixels before the start / past the end of the tensor read as zero (the
// flattened-convolution semantics of the P-tensor layout, see fluidnet_hip.h).
// Persistent: the grid is 2 blocks per CU; a block walks tiles L = blockIdx.x, +gridDim.x, ... and its stage pipeline
// runs ACROSS tiles (stage 0 of the next tile is in flight during the last stage and the epilogue of the current one):
// per-block prologues, which otherwise arrive as chip-wide DMA bursts with every wave idle, disappear.
// Per stage: barrier -> issue DMA of the next stage -> 9 taps x (2 + NT) fragment reads, software-pipelined one tap
// ahead, + 2*NT MFMAs.  MFMA orientation: A = weights (rows = cout), B = pixels, so D[cout][pixel] leaves 4 consecutive
// couts of one pixel in each accumulator quad = one 8-B LDS store into the pixel-major output tile (XOR-swizzled
// 16-B chunks, pitch TN*2), which then streams out 16 B per lane.
// FNET_CONV_ABLATE (diagnostic builds only, tools/conv_bench.py): 1 = no MFMA, 2 = no DMA after the first stage, 3 = weight pieces only for the first two stages of a block (what LDS-resident weights would save at most)
#ifndef FNET_CONV_ABLATE
#define FNET_CONV_ABLATE 0
#endif
#ifndef FNET_CONV_STAMPS
#define FNET_CONV_STAMPS 0
#endif
// ---- stride-2 convolutions as 2x2-tap convolutions over a space-to-depth view (MODE 1 / MODE 2) ----------------------------
// Conv2d(C, Co, 3, stride 2, padding 1) on an H x W image (ml/models/unet.py:90-91) reads, for output (y, x), the padded rows
// 2y, 2y+1, 2y+2 of the source P-tensor.  Pair the PADDED rows and columns, (0,1), (2,3), ...: X'[Y'][X'][(py, px, c)] =
// Xp[2Y' + py][2X' + px][c] is an (h+1) x (w+1) image of 4C channels (h = H/2), and the convolution becomes a 2x2-tap, stride-1
// convolution over X' whose tap (dy, dx) carries W[ky = 2dy + py][kx = 2dx + px] (zero where ky or kx would be 3): 16*C*Co MACs per
// output instead of the 36*C*Co of the full-resolution convolution + subsampling it replaces.  X' is never stored:
//   MODE 1 (forward): GEMM pixels = the half-resolution padded output space; the LDS-DMA of the input run gathers X' rows straight
//           from the full-resolution P-tensor (per-lane source offsets, computed once per tile);
//   MODE 2 (data gradient): dX'[Y'][X'] = sum_t Wd'[t] dOut[Y' + ty][X' + tx] (dOut = half-resolution gradient, ring zero), GEMM N
//           = 4C; the epilogue scatters channel block (py, px) of pixel (Y', X') to dXp[2Y' + py][2X' + px]: every cell of the
//           full-resolution padded gradient is written exactly once, ring included (a "folded" gradient source, like every dgrad).
// Both walk the 2x2 window {0, 1} x {0, 1} from the window origin: f0 - Wp - 1 (MODE 1: taps reach up/left) or f0 (MODE 2).
struct ConvTile {          // wave-uniform description of one (pixel tile, cout tile) work item
    long long f0;
    int n0;
    int in_lane_off;       // per lane (MODE 0, 2)
    __amdgpu_buffer_rsrc_t rs;
};
constexpr int S2_MAXP = 10; // DMA pieces per wave and stage whose per-lane source offsets are kept in registers (piece q = wave + NW*k <-> slot k)

// exact a / d for 0 <= a < 2^24, d > 0, inv = 1.f / d
__device__ __forceinline__ int fdiv24(int a, int d, float inv) {
    int q = (int)((float)a * inv);
    const int r = a - q * d;
    if (r < 0) --q;
    else if (r >= d) ++q;
    return q;
}

// Two block shapes: NW = 4 waves (256 pixels per tile, two blocks per CU) and NW = 8 waves (512 pixels per tile, one block
// per CU, TN up to 128).  The kernel is bound by the LDS-DMA issue rate (every 1-KB piece costs its wave 60-185 cycles,
// MI355X_MICROARCH.md), i.e. by bytes staged per FLOP: per 16-channel stage a tile stages 9*TN weight rows + (TM + 2Wp + 2)
// pixels for TM*TN*9 MACs, so the 512 x 128 tile stages 2.3x fewer bytes per FLOP than the 256 x 64 one; the small tile
// remains for layers with too few tiles to fill 256 big-tile blocks.
struct ConvS2 {            // MODE 1 / 2 geometry (half-resolution space: Hh rows, Wp - 2 columns; Cs channels per phase)
    int Hh, Cs, HpWp;
    float inv_hpwp, inv_wp, inv_cs;
};

template <int TN, int NW, int MODE>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_conv3x3_dma(const bf16_t* __restrict__ src, const bf16_t* __restrict__ Wk,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ dst, int Npix, int Wp,
                                                        int Cin, int ldc_in, int coff_in, int Cout, int ldc_out, int coff_out,
                                                        int n_mtiles, int n_ntiles, int rinp, int bufsz, float* __restrict__ stamps,
                                                        ConvS2 g2) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TMB = NW * 64;                    // pixels per tile
    constexpr int NTHR = NW * 64;
    constexpr int NT = TN / 32;
    constexpr int NTAPS = MODE ? 4 : 9;
    constexpr int TW = MODE ? 2 : 3;                // taps per window row
    constexpr int WB = NTAPS * TN * 32;             // weight image bytes
    constexpr int NPW = NTAPS * TN / 32;            // weight DMA pieces per stage
    const int npi = rinp >> 5;                      // input DMA pieces per stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // Cin need only be a multiple of 8: the last 16-channel stage then runs 8 channels into the next pixel (or past the window,
    // which reads 0); PREP_WEIGHTS zero-pads the K axis of the weights to the same 16, so those products vanish (finite data)
    const int nst = (Cin + KC - 1) / KC;

    // XCD-aware tile order: work items with equal L % 8 run on blocks that share an L2 (gridDim.x is a multiple of 8); give
    // each XCD a contiguous range of pixel tiles and run the cout tiles of one pixel tile back to back on it.
    const int cpx = (n_mtiles + 7) >> 3;
    const int n_items = 8 * cpx * n_ntiles;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)Wk, 0, NTAPS * Cout * nst * KC * 2, 0x00020000);
    const int swz = (lane >> 4) & 1;                // bit 3 of the row a DMA lane fills (row = 32*piece + lane/2)
    const int dh = (lane & 1) ^ swz;                // channel half this lane fetches
    const int in_piece_stride = 32 * ldc_in * 2;
    const int wrow = lane >> 1;                     // weight source: row = 32*piece + lane/2 -> (tap, col)
    const int win0 = MODE == 2 ? 0 : Wp + 1;        // the window starts this many pixels before the tile

    // MODE 1: the whole full-resolution source through one descriptor; a lane's gather offsets for the pieces its wave issues
    // (piece q = wave + NW * k <-> register k) are recomputed per tile, the stage adds a wave-uniform (phase, channel) offset
    const int Hps = 2 * g2.Hh + 2, Wps = 2 * (Wp - 2) + 2;
    __amdgpu_buffer_rsrc_t rsg = rsw;
    if (MODE == 1) {
        const long long bytes = (long long)(Npix / g2.HpWp) * Hps * Wps * ldc_in * 2;
        rsg = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
    }
    auto gather_off = [&](long long f0, int k) -> int {
        const int q = wave + NW * k;
        const long long P = f0 - win0 + q * 32 + (lane >> 1);      // pixel of X' (half-resolution padded, flattened)
        if (q >= npi || P < 0 || P >= Npix) return 0x7ffffff0;     // out of range: the DMA writes zeros
        const int n = fdiv24((int)P, g2.HpWp, g2.inv_hpwp);
        const int rem = (int)P - n * g2.HpWp;
        const int Y = fdiv24(rem, Wp, g2.inv_wp), X = rem - Y * Wp;
        if (Y > g2.Hh || X > Wp - 2) return 0x7ffffff0;            // X' has (h+1) x (w+1) cells
        return (((n * Hps + 2 * Y) * Wps + 2 * X) * ldc_in + coff_in + dh * 8) * 2;
    };
    auto stage_off = [&](int s) -> int {           // MODE 1: channel block 16s of the 4C axis = (phase, c0)
        const int ph = (int)((float)(s * KC) * g2.inv_cs + 0.001f);          // C is a multiple of 16: the quotient is far from a boundary
        const int c0 = s * KC - ph * g2.Cs;
        return (((ph >> 1) * Wps + (ph & 1)) * ldc_in + c0) * 2;
    };

    auto tile_of = [&](int L, ConvTile& T) -> bool {
        // returns false when L is past the end or maps to a padding slot of the XCD layout
        if (L >= n_items) return false;
        const int xcd = L & 7, j = L >> 3;
        const int mt = xcd * cpx + j / n_ntiles;
        if (mt >= n_mtiles) return false;
        T.f0 = (long long)mt * TMB;
        T.n0 = (j - (j / n_ntiles) * n_ntiles) * TN;
        if (MODE == 1) {
            T.rs = rsg;
            T.in_lane_off = 0;
            return true;
        }
        const long long gbase = T.f0 - win0;
        const long long wpix = gbase > 0 ? gbase : 0;
        long long wbytes = ((long long)Npix - wpix) * ldc_in * 2;
        const long long wmax = (long long)(rinp + 1) * ldc_in * 2;
        if (wbytes > wmax) wbytes = wmax;
        T.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)wpix * ldc_in), 0, (int)wbytes, 0x00020000);
        T.in_lane_off = (int)(((long long)(lane >> 1) + (gbase - wpix)) * ldc_in + coff_in + dh * 8) * 2;
        return true;
    };
    const int npieces = npi + NPW;                 // DMA pieces per stage
    const int wstage = NTAPS * Cout * KC * 2;      // bytes of one stage's weight slab in the K-blocked layout
    // Per-lane source offset of every piece this wave issues for tile T, stage 0 (slot k <-> piece q = wave + NW*k): computed once per tile,
    // so that a piece inside the tap loops costs one scalar compare and the DMA instruction itself; the stage adds a wave-uniform offset
    // (input: 32 B per stage, MODE 1: stage_off(s); weights: s * wstage) through the instruction's scalar offset.
    auto piece_off = [&](const ConvTile& T, int q, int k) -> int {
        if (q < npi) return MODE == 1 ? gather_off(T.f0, k) : T.in_lane_off + q * in_piece_stride;
        const int row = (q - npi) * 32 + wrow;
        const int t = row / TN;
        int co = T.n0 + (row - t * TN);
        co = co < Cout ? co : Cout - 1;
        return ((t * Cout + co) * 16 + dh * 8) * 2;
    };
    // (the 512 x 128 tile of the 3x3 form has no registers to spare -- 256 accumulators: it forms every offset where it is used)
    constexpr int MAXP = (TN == 128 && MODE == 0) ? 0 : S2_MAXP;
    auto piece_offsets = [&](const ConvTile& T, int (&po)[S2_MAXP]) {
#pragma unroll
        for (int k = 0; k < MAXP; ++k) po[k] = piece_off(T, wave + NW * k, k);
    };
    // one DMA piece (1 KB = 32 rows x 32 B) of stage s of tile T into the stage buffer at byte offset boff: pieces [0, npi) are the
    // input run, [npi, npi + NPW) the weight slab; voff = the piece's per-lane offset (po[k], or piece_off for the rare pieces beyond
    // the register slots), sin / sw = the stage's scalar offsets
    unsigned g = 0;                       // running stage counter: stage buffer = g & 1
    auto issue_piece = [&](const ConvTile& T, int voff, int sin, int sw, int boff, int q) {
        unsigned char* wdst = smem + boff;
        if (q < npi) __builtin_amdgcn_raw_ptr_buffer_load_lds(T.rs, (lds_void_t*)(wdst + WB + q * 1024), 16, voff, sin, 0, 0);
#if FNET_CONV_ABLATE == 3
        else if (g < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_t*)(wdst + (q - npi) * 1024), 16, voff, sw, 0, 0);   // weights staged once per buffer
#else
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_t*)(wdst + (q - npi) * 1024), 16, voff, sw, 0, 0);
#endif
    };

    // fragment read addresses (bytes, relative to a stage buffer): taps x 2 pixel sub-tiles; weights: lane base + immediates
    unsigned xaddr[NTAPS][2];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            const int p = wave * 64 + ms * 32 + r + (t / TW) * Wp + (t % TW);
            xaddr[t][ms] = WB + (unsigned)(2 * p + (h ^ ((p >> 3) & 1))) * 16;
        }
    const unsigned waddr = (unsigned)(2 * r + (h ^ ((r >> 3) & 1))) * 16;

    f32x16 acc[2][NT];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int ns = 0; ns < NT; ++ns)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ms][ns][i] = 0.f;

    ConvTile cur, nxt;
    int po[S2_MAXP];
    bool have = tile_of(blockIdx.x, cur);
    if (!have) return;
    piece_offsets(cur, po);
    int L = blockIdx.x;
#if FNET_CONV_STAMPS
    // diagnostic build (tools/conv_bench.py): wave 0 of every block adds its cycles spent waiting at the stage barrier, in the
    // tap loops, in the epilogue and in total to the first 5 int64 of `stamps`
    long long st_bar0 = 0, st_bar = 0, st_loop = 0, st_epi = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_a, st_b;
#define STAMP(x) x = __builtin_amdgcn_s_memtime()
#else
#define STAMP(x)
#endif
    {
        const int sin = MODE == 1 ? stage_off(0) : 0;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int q = wave + NW * k;
            if (q < npieces) issue_piece(cur, po[k], sin, 0, 0, q);
        }
        for (int q = wave + NW * MAXP; q < npieces; q += NW) issue_piece(cur, piece_off(cur, q, 0), sin, 0, 0, q);   // wide images, MODE 0 / 2
    }
    while (true) {
        bool have_next = false;
        for (int s = 0; s < nst; ++s, ++g) {
            STAMP(st_a);
            __syncthreads();  // this stage has landed (vmcnt(0) + barrier); every wave is done with the other buffer
            STAMP(st_b);
#if FNET_CONV_STAMPS
            st_bar += st_b - st_a;
            if (s == 0) st_bar0 += st_b - st_a;
#endif
            // The next stage's DMA (of this tile, or stage 0 of the block's next tile: it then lands during the epilogue) is
            // issued one piece per tap INSIDE the tap loop below: a piece costs the issuing wave 60-185 cycles, which hide behind
            // the four MFMAs of a tap instead of delaying every wave's first MFMA after the barrier.
            const bool last = s + 1 == nst;
            ConvTile Ti = cur;             // by value (a reference to one of two structs would put both in scratch)
            if (last) {
                // every piece of this tile has been issued: the offset registers now describe the block's next tile, whose stage 0
                // is issued during this stage
                L += gridDim.x;
                have_next = tile_of(L, nxt);
                if (have_next) piece_offsets(nxt, po);
                Ti = nxt;
            }
            const int si = last ? 0 : s + 1;
            const int sin = MODE == 1 ? stage_off(si) : si * (KC * 2);
            const int sw = si * wstage;
            const int nbuf = ((g + 1) & 1) ? bufsz : 0;
#if FNET_CONV_ABLATE == 2
            const int qlim = 0;
#else
            const int qlim = (!last || have_next) ? npieces : 0;
#endif
            const unsigned bb = (g & 1) ? (unsigned)bufsz : 0u;
            // fragment reads run one tap ahead of the MFMAs (two register sets, static indexing after unrolling)
            bf16x8 xb[2][2], wa[2][NT];
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) xb[0][ms] = *reinterpret_cast<const bf16x8*>(smem + xaddr[0][ms] + bb);
#pragma unroll
            for (int ns = 0; ns < NT; ++ns) wa[0][ns] = *reinterpret_cast<const bf16x8*>(smem + waddr + bb + (ns * 32) * 32);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int c = t & 1, n = c ^ 1;
                if (t + 1 < NTAPS) {
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms) xb[n][ms] = *reinterpret_cast<const bf16x8*>(smem + xaddr[t + 1][ms] + bb);
#pragma unroll
                    for (int ns = 0; ns < NT; ++ns)
                        wa[n][ns] = *reinterpret_cast<const bf16x8*>(smem + waddr + bb + ((t + 1) * TN + ns * 32) * 32);
                }
                {
                    const int q = wave + NW * t;
                    if (q < qlim) issue_piece(Ti, t < MAXP ? po[t < MAXP ? t : 0] : piece_off(Ti, q, 0), sin, sw, nbuf, q);
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this tap's MFMAs (hipcc otherwise re-serialises them)
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int ns = 0; ns < NT; ++ns)
#if FNET_CONV_ABLATE == 1
                        { asm volatile("" ::"v"(wa[c][ns]), "v"(xb[c][ms])); }
#else
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[c][ns], xb[c][ms], acc[ms][ns], 0, 0, 0);
#endif
            }
            // more pieces per wave than taps (wide images; always in the 2x2 modes)
#pragma unroll
            for (int k = NTAPS; k < MAXP; ++k) {
                const int q = wave + NW * k;
                if (q < qlim) issue_piece(Ti, po[k], sin, sw, nbuf, q);
            }
            for (int q = wave + NW * (MAXP > NTAPS ? MAXP : NTAPS); q < qlim; q += NW) issue_piece(Ti, piece_off(Ti, q, 0), sin, sw, nbuf, q);   // MODE 0 / 2
#if FNET_CONV_STAMPS
            STAMP(st_a);
            st_loop += st_a - st_b;
#endif
        }
        STAMP(st_a);
        __syncthreads();  // all fragment reads of the last stage done: its buffer now holds the output tile

        // D[row = cout][col = pixel]: lane (r, h) holds pixel r, couts (i&3) + 8*(i>>2) + 4*h.  The output goes through LDS in
        // slices of TNE <= 64 couts: slice tile [TMB px][TNE] bf16, pitch TNE*2, 16-B chunk index XOR-swizzled with the pixel
        // index, then streams out 16 B per lane.
        unsigned char* ot = smem + ((g - 1) & 1) * bufsz;
        constexpr int TNE = TN > 64 ? 64 : TN; // couts per output slice
        constexpr int NTE = TNE / 32;
        constexpr int PO = TNE * 2;
        constexpr int CH = TNE / 8;            // 16-B chunks per pixel
        constexpr int PPI = NTHR / CH;         // pixels per pass
        const long long f0 = cur.f0;
        const int chunk = tid % CH, px0 = tid / CH;
#pragma unroll
        for (int sl = 0; sl < TN / TNE; ++sl) {
            const int n0 = cur.n0 + sl * TNE;
            if (sl > 0) __syncthreads();       // the previous slice has been read out of the tile
#pragma unroll
            for (int nl = 0; nl < NTE; ++nl) {
                const int ns = sl * NTE + nl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = nl * 32 + 8 * q + 4 * h;
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (bias) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) bv[k] = (n0 + col + k < Cout) ? bias[n0 + col + k] : 0.f;
                    }
#pragma unroll
                    for (int ms = 0; ms < 2; ++ms) {
                        const int px = wave * 64 + ms * 32 + r;
                        uint2 v;
                        v.x = f2bf_pk(acc[ms][ns][4 * q + 0] + bv[0], acc[ms][ns][4 * q + 1] + bv[1]);
                        v.y = f2bf_pk(acc[ms][ns][4 * q + 2] + bv[2], acc[ms][ns][4 * q + 3] + bv[3]);
                        *reinterpret_cast<uint2*>(ot + px * PO + ((((col >> 3) ^ px) & (CH - 1)) << 4) + (col & 4) * 2) = v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[ms][ns][4 * q + k] = 0.f;
                    }
                }
            }
            __syncthreads();
            const bool cvalid = n0 + chunk * 8 < Cout;
            if (cvalid) {
                // MODE 2: this thread's 8 channels are channels c .. c+7 of phase (py, px) of the full-resolution gradient
                int ph_off = 0;
                if (MODE == 2) {
                    const int chn = n0 + chunk * 8;
                    const int ph = (int)((float)chn * g2.inv_cs + 0.001f);       // C is a multiple of 8
                    ph_off = ((ph >> 1) * Wps + (ph & 1)) * ldc_out + (chn - ph * g2.Cs);
                }
#pragma unroll
                for (int k = 0; k < TMB / PPI; ++k) {
                    const int px = px0 + k * PPI;
                    const long long f = f0 + px;
                    if (f < Npix) {
                        const uint4 v = *reinterpret_cast<const uint4*>(ot + px * PO + (((chunk ^ px) & (CH - 1)) << 4));
                        if (MODE == 2) {
                            const int n = fdiv24((int)f, g2.HpWp, g2.inv_hpwp);
                            const int rem = (int)f - n * g2.HpWp;
                            const int Y = fdiv24(rem, Wp, g2.inv_wp), X = rem - Y * Wp;
                            if (Y <= g2.Hh && X <= Wp - 2)
                                st_stream16(dst + (size_t)((n * Hps + 2 * Y) * Wps + 2 * X) * ldc_out + coff_out + ph_off, v);
                        } else {
                            st_stream16(dst + (size_t)f * ldc_out + coff_out + n0 + chunk * 8, v);
                        }
                    }
                }
            }
        }
#if FNET_CONV_STAMPS
        STAMP(st_b);
        st_epi += st_b - st_a;
#endif
        if (!have_next) break;
        cur = nxt;
    }
#if FNET_CONV_STAMPS
    if (tid == 0 && stamps) {
        atomicAdd((unsigned long long*)stamps + 0, (unsigned long long)st_bar);
        atomicAdd((unsigned long long*)stamps + 1, (unsigned long long)st_loop);
        atomicAdd((unsigned long long*)stamps + 2, (unsigned long long)st_epi);
        atomicAdd((unsigned long long*)stamps + 3, (unsigned long long)(__builtin_amdgcn_s_memtime() - st_t0));
        atomicAdd((unsigned long long*)stamps + 4, 1ull);
        atomicAdd((unsigned long long*)stamps + 5, (unsigned long long)st_bar0);
    }
#endif
#undef STAMP
}

// ------------------------------------------------------------------------------------------- weight gradient
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

// ---- weight gradient, LDS-DMA double-buffered, 2-D pixel chunks, slab reduction ------------------------------------
// K axis = interior pixels only (the ring of dout is zero by construction of every producer: INACT_BWD_APPLY, FOLD_SUM,
// ZERO_INSERT), walked in chunks of 64 pixels shaped CH x CW (2 x 32 or 4 x 16) so that the source halo is
// (CH+2) x (CW+2) <= 144 pixels whatever the image width.  Per stage the block DMA-loads NCO dout panels [64 px][32 co] and
// NCI source panels [144 px][32 ci] (64-B rows, the layout ds_read_b64_tr_b16 reads conflict-free), double-buffered
// exactly like the convolution.  Each wave keeps the nine 32x32 tap accumulators of one (co, ci) panel pair in
// registers for its whole pixel range and finally writes them, with plain coalesced stores, as one 36-KB partial into
// the slab; k_wgrad_reduce adds the partials into dWk.  (Accumulating with float atomics instead ran at the chip-wide
// atomic rate: ~140 MB of adds per launch at 1.3 TB/s was a third of the kernel.)
// Pixels per chunk, PX: 64 (2 x 32 / 4 x 16) or 128 (4 x 32 / 8 x 16).  The kernel is bound by the bytes its LDS-DMA moves (ablation builds,
// 64 -> 64 channels at 128^2: no MFMA 122 of 129 us, no DMA 97): a 64-pixel chunk stages a 144-pixel source halo per panel (2.25x), a
// 128-pixel chunk 204 (1.6x) -- 20 % fewer staged bytes per pixel; the launcher takes 128 where the launch still has enough chunks.
// source rows staged per panel: >= (CH+2)*(CW+2), in the 2x2 (stride-2) form >= (CH+1)*(CW+1), a multiple of 16
__host__ __device__ constexpr int wg_srows(int px, bool s2) { return px == 64 ? (s2 ? 112 : 144) : (s2 ? 176 : 208); }
constexpr int WG_PART = 9 * 32 * 32;      // floats per wave partial
#ifndef FNET_WG_STAGES
#define FNET_WG_STAGES 3
#endif
constexpr int WG_STAGES = FNET_WG_STAGES; // LDS stage buffers (2 or 3)
// FNET_WG_ABLATE (diagnostic builds only, tools/conv_bench.py variants 11-13; results are wrong, only the time means something):
// 1 = no MFMA, 2 = no DMA after a block's first chunks, 3 = no DMA wait / stage barrier
#ifndef FNET_WG_ABLATE
#define FNET_WG_ABLATE 0
#endif
static_assert(WG_STAGES >= 2 && WG_STAGES <= 4, "two to four stage buffers");
// 128-pixel chunks run two stage buffers by default (85 KB, the LDS footprint of three 64-pixel stages: a convolution block of lane 0 still
// fits beside the weight-gradient block on a CU; with three -- 129 KB -- the step was 0.3 ms slower although the kernel alone was faster)
#ifndef FNET_WG_STAGES128
#define FNET_WG_STAGES128 2
#endif
__host__ __device__ constexpr int wg_stages(int px) { return px == 64 ? WG_STAGES : FNET_WG_STAGES128; }

// LDS-DMA of 16 B per lane (64 lanes -> 1 KB landing linearly at `lds`) issued from inline assembly, so that hipcc does not know of it:
// ds_read_b64_tr_b16 has no memory operand the waitcnt pass could prove disjoint from the DMA's destination, and with the builtin form
// hipcc put `s_waitcnt vmcnt(0)` in front of the first fragment read after every DMA issue -- each chunk waited for the NEXT chunk's DMA
// to land before it computed (no overlap inside a block; found in round 2 from the ISA).  The kernel orders DMA and reads itself:
// wait_vmcnt<N>() + s_barrier before a stage buffer is read.  (hipcc's own vmcnt bookkeeping stays safe: operations it does not
// know of only make its waits stricter.)
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 make_rsrc_words(const void* p, int bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;                                            // wave-uniform by construction; readfirstlane keeps it in SGPRs for the asm operand
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffull));
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffull));     // stride 0
    r[2] = __builtin_amdgcn_readfirstlane(bytes);      // num_records (bytes: raw buffer)
    r[3] = 0x00020000;
    return r;
}
// (`lds` = the destination's LDS byte address as an integer: a generic pointer cast back to the LDS address space here made hipcc emit a
// null test against src_shared_base that failed its own machine verifier in one instantiation)
__device__ __forceinline__ void dma16_untracked(i32x4 rsrc, unsigned lds, int voff) {
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)lds);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory");
}

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt = simm16[15:14]:[3:0], expcnt [6:4] and lgkmcnt [11:8] left at their maxima)
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
}

// one stage of DMA for chunk c into LDS buffer `buf` (see k_wgrad_dma)
template <int NCO, int NCI, int PPW, bool S2, int WG_PX>
__device__ __forceinline__ void wg_issue(unsigned smem_lds, const bf16_t* __restrict__ src, const bf16_t* __restrict__ dout, int c, int buf,
                                         int wave, int H, int W, int ldc_in, int ldc_do, int log2cw, int cpi, int ncx,
                                         const int (&p_off)[PPW], const int (&p_lds)[PPW], const int (&p_cyx)[PPW]) {
    constexpr int SROWS = wg_srows(WG_PX, S2);
    constexpr int DB = NCO * WG_PX * PANEL_ROW;
    constexpr int SB = NCI * SROWS * PANEL_ROW;
    constexpr int STAGE = DB + SB;
    constexpr int NPD = NCO * (WG_PX / 16);
    constexpr int NP = NPD + NCI * (SROWS / 16);
    const int Hp = H + 2, Wp = W + 2;
    const int CW = 1 << log2cw, CH = WG_PX >> log2cw;
    const int n = c / cpi, rem = c - n * cpi;
    const int cyi = rem / ncx, cxi = rem - cyi * ncx;
    const int y0 = 1 + cyi * CH, x0 = 1 + cxi * CW;   // padded coordinates of the chunk origin
    const size_t img = (size_t)n * Hp * Wp;
    // S2: the source is the full-resolution P-tensor ((2H+2) x (2W+2)) read as its space-to-depth view (see k_conv3x3_dma, MODE 1)
    const int Hps = S2 ? 2 * H + 2 : Hp, Wps = S2 ? 2 * W + 2 : Wp;
    const size_t simg = (size_t)n * Hps * Wps;
    const i32x4 rsd = make_rsrc_words(dout + img * ldc_do, Hp * Wp * ldc_do * 2);
    const i32x4 rss = make_rsrc_words(src + simg * ldc_in, Hps * Wps * ldc_in * 2);
    const int dbase = (y0 * Wp + x0) * ldc_do * 2;
    const int sbase = S2 ? (2 * (y0 - 1) * Wps + 2 * (x0 - 1)) * ldc_in * 2 : ((y0 - 1) * Wp + x0 - 1) * ldc_in * 2;
    const unsigned base = smem_lds + (unsigned)(buf * STAGE);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pid = wave + 4 * i;
        if (pid < NPD) {
            const int cy = p_cyx[i] & 0xff, cx = p_cyx[i] >> 8;
            const bool ok = (y0 + cy <= H) && (x0 + cx <= W);
            const int off = ok ? p_off[i] + dbase : 0x7ffffff0;          // out of range -> the DMA writes zeros
            dma16_untracked(rsd, base + p_lds[i], off);
        } else if (pid < NP) {
            dma16_untracked(rss, base + p_lds[i], p_off[i] + sbase);
        }
    }
}

template <int NCO, int NCI, bool S2, int WG_PX>
__global__ __launch_bounds__(256, 2) void k_wgrad_dma(const bf16_t* __restrict__ src, const bf16_t* __restrict__ dout,
                                                      float* __restrict__ slab, int B, int H, int W, int ldc_in, int coff_in,
                                                      int ldc_do, int coff_do, int log2cw, int chunks_per_block, int nchunks,
                                                      int nparts, int n_ci_tiles, int Cs, int xcd_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KS = 4 / (NCO * NCI);       // waves that share one (co,ci) panel pair and split the k-steps
    constexpr int NSTG = wg_stages(WG_PX);
    constexpr int SROWS = wg_srows(WG_PX, S2);
    constexpr int DB = NCO * WG_PX * PANEL_ROW;      // dout bytes per stage
    constexpr int SB = NCI * SROWS * PANEL_ROW;      // source bytes per stage
    constexpr int STAGE = DB + SB;
    constexpr int NPD = NCO * (WG_PX / 16);          // DMA pieces (16 rows x 64 B) per stage: dout
    constexpr int NPSP = SROWS / 16;              // ... per source panel
    constexpr int NP = NPD + NCI * NPSP;
    constexpr int PPW = (NP + 3) / 4;                // pieces per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cop = wave % NCO, cip = (wave / NCO) % NCI, ksl = wave / (NCO * NCI);
    const int co_tile = blockIdx.y, ci_tile = blockIdx.z;
    const int co_base = co_tile * (32 * NCO), ci_base = ci_tile * (32 * NCI);
    constexpr int NTAPS = S2 ? 4 : 9, TW = S2 ? 2 : 3;
    const int Wp = W + 2, Wps = 2 * W + 2;
    const int CW = 1 << log2cw, CH = WG_PX >> log2cw, SW = CW + (S2 ? 1 : 2);
    const int ncx = (W + CW - 1) >> log2cw, ncy = (H + CH - 1) / CH;
    const int cpi = ncx * ncy;                       // chunks per image

    // this wave's DMA pieces: lane byte offsets relative to the chunk origin, LDS destinations, dout (cy,cx) for masking
    int p_off[PPW], p_lds[PPW], p_cyx[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pid = wave + 4 * i;
        const int row16 = lane >> 2, piece16 = (lane & 3) * 8;
        if (pid < NPD) {
            const int panel = pid / (WG_PX / 16), j = pid % (WG_PX / 16);
            const int row = 16 * j + row16;
            const int cy = row >> log2cw, cx = row & (CW - 1);
            p_off[i] = ((cy * Wp + cx) * ldc_do + coff_do + co_base + panel * 32 + piece16) * 2;
            p_lds[i] = panel * (WG_PX * PANEL_ROW) + j * 1024;
            p_cyx[i] = cy | (cx << 8);
        } else {
            const int ps = pid - NPD;
            const int panel = ps / NPSP, j = ps % NPSP;
            const int row = 16 * j + row16;
            const int iy = row / SW, ix = row - iy * SW;
            if (S2) {
                // 32-channel panel of the 4C axis = channels c0 .. c0+31 of phase (py, px): Cs is a multiple of 32
                const int cb = ci_base + panel * 32, ph = cb / Cs, c0 = cb - ph * Cs;
                p_off[i] = (((2 * iy + (ph >> 1)) * Wps + 2 * ix + (ph & 1)) * ldc_in + coff_in + c0 + piece16) * 2;
            } else {
                p_off[i] = ((iy * Wp + ix) * ldc_in + coff_in + ci_base + panel * 32 + piece16) * 2;
            }
            p_lds[i] = DB + panel * (SROWS * PANEL_ROW) + j * 1024;
            p_cyx[i] = 0;
        }
    }


    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // chunk ids of this block: c0 + i * cs, i < n_mine.  Contiguous ranges, or (xcd_map; gridDim.x a multiple of 8) the blocks of one XCD
    // -- blockIdx.x % 8 -- interleave over one eighth of the chunks, the same eighth of the batch that XCD's tiles of the data-gradient
    // convolution running beside this launch cover, and walk it front to back as that convolution does
    int c0 = blockIdx.x * chunks_per_block, cs = 1, n_mine = min(nchunks, c0 + chunks_per_block) - c0;
    if (xcd_map) {
        const int X = blockIdx.x & 7, j = blockIdx.x >> 3, SX = gridDim.x >> 3, R = (nchunks + 7) >> 3;
        const int lo = X * R, hi = min(nchunks, lo + R);
        c0 = lo + j; cs = SX;
        n_mine = c0 < hi ? (hi - c0 + SX - 1) / SX : 0;
    }
    if (n_mine < 0) n_mine = 0;
    const int c_begin = 0, c_end = n_mine;
    // NSTG stage buffers, the DMA runs NSTG - 1 chunks ahead: with one block per CU a single stage in flight (27 KB per CU, 7 MB
    // chip-wide) left the kernel latency-bound (solo: 512 blocks 19 % faster than 256).  A chunk has landed when this wave's DMA count is
    // down to the pieces of the chunks issued after it (vmcnt counts in order; the loop issues no other vector memory operation) and
    // every wave has passed the barrier; the buffer refilled in iteration c was last read in iteration c - 1, before that barrier.
    const int npw = (NP - 1 - wave) / 4 + 1;             // pieces this wave issues per chunk: PPW or PPW - 1
    const unsigned smem_lds = (unsigned)(size_t)(lds_void_t*)smem;      // LDS byte address of the dynamic array
#pragma unroll
    for (int k = 0; k < NSTG - 1; ++k)
        if (c_begin + k < c_end) wg_issue<NCO, NCI, PPW, S2, WG_PX>(smem_lds, src, dout, c0 + (c_begin + k) * cs, k, wave, H, W, ldc_in, ldc_do, log2cw, cpi, ncx, p_off, p_lds, p_cyx);
    int cur = 0;
    for (int c = c_begin; c < c_end; ++c) {
        const int ahead = min(c_end - 1 - c, NSTG - 2);   // chunks issued after chunk c and still allowed in flight
#if FNET_WG_ABLATE != 3
        if (ahead <= 0) wait_vmcnt<0>();
        else if (ahead == 1) { if (npw == PPW) wait_vmcnt<PPW>(); else wait_vmcnt<PPW - 1>(); }
        else { if (npw == PPW) wait_vmcnt<2 * PPW>(); else wait_vmcnt<2 * (PPW - 1)>(); }
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        {
            int nb = cur + NSTG - 1;
            nb = nb >= NSTG ? nb - NSTG : nb;
#if FNET_WG_ABLATE == 2
            if (false)
#else
            if (c + NSTG - 1 < c_end)
#endif
                wg_issue<NCO, NCI, PPW, S2, WG_PX>(smem_lds, src, dout, c0 + (c + NSTG - 1) * cs, nb, wave, H, W, ldc_in, ldc_do, log2cw, cpi, ncx, p_off, p_lds, p_cyx);
        }
        const unsigned char* pa = smem + cur * STAGE + cop * (WG_PX * PANEL_ROW);
        const unsigned char* pb = smem + cur * STAGE + DB + cip * (SROWS * PANEL_ROW);
#pragma unroll
        for (int kk = 0; kk < (WG_PX / 16) / KS; ++kk) {
            const int ks = ksl + kk * KS;
            const int row0 = 16 * ks;
            const int cy = row0 >> log2cw, cx0 = row0 & (CW - 1);
            const bf16x8 a = tr_frag(pa, row0, lane);
#pragma unroll
            for (int ty = 0; ty < TW; ++ty)
#pragma unroll
                for (int tx = 0; tx < TW; ++tx) {
                    const bf16x8 b = tr_frag(pb, (cy + ty) * SW + cx0 + tx, lane);
#if FNET_WG_ABLATE == 1
                    asm volatile("" ::"v"(a), "v"(b));
#else
                    acc[ty * TW + tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ty * TW + tx], 0, 0, 0);
#endif
                }
        }
        cur = cur + 1 == NSTG ? 0 : cur + 1;
    }
    // KS > 1: the KS waves that split the k-steps of one panel pair add their accumulators up through LDS (lane-linear 36-KB images, one
    // per pair, over the stage buffers) before anything goes to the slab: KS times fewer partials to write and to reduce
    if (KS > 1) {
        float* xch = reinterpret_cast<float*>(smem) + (size_t)(cop + NCO * cip) * WG_PART;
#pragma unroll 1
        for (int j = 1; j < KS; ++j) {
            __syncthreads();                  // stage buffers (j = 1) / the previous round's image are no longer read
            if (ksl == j) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) xch[(t * 16 + i) * 64 + lane] = acc[t][i];
            }
            __syncthreads();
            if (ksl == 0) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += xch[(t * 16 + i) * 64 + lane];
            }
        }
        if (ksl != 0) return;
    }
    // D[row = co][col = ci]: partial[t][co][ci], lanes run along ci (two 128-B segments per store instruction)
    const int r = lane & 31, h = lane >> 5;
    const size_t pair = (size_t)(co_tile * NCO + cop) * (n_ci_tiles * NCI) + (ci_tile * NCI + cip);
    float* part = slab + (pair * nparts + (size_t)blockIdx.x) * WG_PART;      // (the 2x2 form uses the first 4 of the 9 tap images)
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = (i & 3) + 8 * (i >> 2) + 4 * h;
            part[t * 1024 + co * 32 + r] = acc[t][i];
        }
}

// dWk[t][co][ci] += sum over partials of the (co/32, ci/32) pair.  grid = (elements / 256, partial groups); one float atomic
// per (element, group) -- a few MB per launch.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ dWk, int Cout, int Cin, int Cin_w,
                                                      int nparts, int parts_per_group, int n_ci_panels, int ntaps) {
    const int e = blockIdx.x * kThreads + threadIdx.x;
    if (e >= ntaps * Cout * Cin) return;
    const int ci = e % Cin, rest = e / Cin, co = rest % Cout, t = rest / Cout;
    const size_t pair = (size_t)(co >> 5) * n_ci_panels + (ci >> 5);
    const int s0 = blockIdx.y * parts_per_group, s1 = min(nparts, s0 + parts_per_group);
    const float* p = slab + (pair * nparts + s0) * WG_PART + t * 1024 + (co & 31) * 32 + (ci & 31);
    // a thread's partials are 36 KB apart: the kernel is a chain of load round trips, so eight are kept in flight per trip
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int s = s0;
    for (; s + 8 <= s1; s += 8, p += 8 * WG_PART) {
        const float v0 = p[0], v1 = p[WG_PART], v2 = p[2 * WG_PART], v3 = p[3 * WG_PART];
        const float v4 = p[4 * WG_PART], v5 = p[5 * WG_PART], v6 = p[6 * WG_PART], v7 = p[7 * WG_PART];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3; a4 += v4; a5 += v5; a6 += v6; a7 += v7;
    }
    for (; s + 4 <= s1; s += 4, p += 4 * WG_PART) {
        a0 += p[0]; a1 += p[WG_PART]; a2 += p[2 * WG_PART]; a3 += p[3 * WG_PART];
    }
    for (; s < s1; ++s, p += WG_PART) a0 += *p;
    atomicAdd(dWk + ((size_t)t * Cout + co) * Cin_w + ci, ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)));
}

// per kernel instantiation and device: the largest dynamic-LDS size already granted (the attribute call costs microseconds of host time and
// a step has ~360 convolution / weight-gradient launches)
template <typename K>
int set_lds_limit(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return FNET_OK;
    // granted sizes per (kernel, device): every instantiation of a kernel template has the same pointer TYPE, so the key is the pointer
    static std::mutex mu;
    static std::unordered_map<const void*, std::array<size_t, 16>> granted;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    std::lock_guard<std::mutex> lock(mu);
    std::array<size_t, 16>& g = granted[key];          // value-initialised to zeros on first use
    if (bytes <= g[dev]) return FNET_OK;
    hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    g[dev] = bytes;
    return FNET_OK;
}
// compute units of the current device, asked once per device (hipGetDeviceProperties fills a multi-KB struct on every call)
inline int device_cus() {
    static int cached[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cached[dev];
}

}  // namespace

// CONV3X3 (impl = MFMA)  i: 9 mode (0 = 3x3; 1 = stride-2 forward, 2 = stride-2 data gradient: see the kernel), modes 1 / 2: 10 = h (rows of
// the half-resolution image; Npix and Wp describe that half-resolution padded space), 11 = C (channels per phase: Cin = 4C in mode 1 and
// ldc_in / coff_in address the full-resolution source; Cout = 4C in mode 2 and ldc_out / coff_out address the full-resolution
// destination).  p[4] = cycle-stamp scratch of the diagnostic build, else unused.
int fnet_conv3x3_mfma(const FnetOp& op, hipStream_t st) {
    const int Npix = op.i[0], Wp = op.i[1], Cin = op.i[2], Cout = op.i[5], mode = op.i[9];
    FNET_CHECK_ARG(op.dtype == FNET_BF16 && mode >= 0 && mode <= 2);
    FNET_CHECK_ARG(Npix > 0 && Wp >= 3 && Cin % 8 == 0 && Cin > 0 && Cout % 8 == 0 && Cout > 0);
    FNET_CHECK_ARG(op.i[3] % 8 == 0 && op.i[4] % 8 == 0 && op.i[6] % 8 == 0 && op.i[7] % 8 == 0);
    FNET_CHECK_ARG(op.p[0] && op.p[1] && op.p[3] && (op.p[0] % 16) == 0 && (op.p[1] % 16) == 0 && (op.p[3] % 16) == 0);
    ConvS2 g2 = {0, 0, 1, 0.f, 0.f, 0.f};
    if (mode) {
        const int hh = op.i[10], Cs = op.i[11], wh = Wp - 2;
        FNET_CHECK_ARG(hh >= 1 && wh >= 1 && Cs >= 16 && Cs % 16 == 0 && Npix % ((hh + 2) * Wp) == 0 && Npix < (1 << 24));
        const long long full = (long long)(Npix / ((hh + 2) * Wp)) * (2 * hh + 2) * (2 * wh + 2);   // pixels of the full-resolution P-tensor
        if (mode == 1) {
            FNET_CHECK_ARG(Cin == 4 * Cs && op.i[4] + Cs <= op.i[3] && op.i[7] + Cout <= op.i[6] && full * op.i[3] * 2 < (1ll << 31));
        } else {
            FNET_CHECK_ARG(Cout == 4 * Cs && op.i[4] + Cin <= op.i[3] && op.i[7] + Cs <= op.i[6] && full * op.i[6] < (1ll << 31));
        }
        g2 = {hh, Cs, (hh + 2) * Wp, 1.f / (float)((hh + 2) * Wp), 1.f / (float)Wp, 1.f / (float)Cs};
    } else {
        FNET_CHECK_ARG(op.i[4] + Cin <= op.i[3] && op.i[7] + Cout <= op.i[6]);
    }
    float* stamps = FNET_CONV_STAMPS ? (float*)op.p[4] : nullptr;
    const int cus = device_cus();
    const int ntaps = mode ? 4 : 9;
    const int halo = mode ? Wp + 1 : 2 * Wp + 2;   // pixels of the input run beyond the tile
    // Tile shape.  Small: 256 px x {32,64} couts, 4 waves, two blocks per CU.  Big: 512 px x {64,128} couts, 8 waves, one block
    // per CU: 1.6-2.3x fewer staged bytes per FLOP, taken when the layer has enough big tiles to keep every CU busy for at
    // least ~3 rounds (a partial last round of big tiles costs more than it saves on the small layers).
    // FNET_CONV_TILE=small|big forces one shape (tuning / tools/conv_bench.py).
    static const int forced = [] { const char* e = getenv("FNET_CONV_TILE"); return !e ? 0 : (e[0] == 's' ? 1 : (e[0] == 'b' ? 2 : 0)); }();
    bool big = false;
    if (Cout > 32) {
        const int tn_b = Cout > 64 ? 128 : 64;
        const long long items_b = (long long)((Npix + 511) / 512) * ((Cout + tn_b - 1) / tn_b);
        big = items_b >= 3ll * cus;
        if (forced) big = forced == 2;
        if (mode == 1 && tn_b == 128) big = false;     // (that variant would spill: 8 gather offsets on top of 256 accumulator registers)
        // wide images: the big tile's two stage buffers (input run of 512 + halo pixels) must fit the CU's 160 KB of LDS
        const size_t rin_b = (size_t)(512 + halo + 31) / 32 * 32;
        size_t buf_b = rin_b * 32 + (size_t)ntaps * tn_b * 32;
        const size_t out_b = (size_t)512 * 64 * 2 + 2048;
        if (buf_b < out_b) buf_b = out_b;
        if (2 * buf_b > 160 * 1024) big = false;
    }
    const int tm = big ? 512 : TM;
    const int TN = big ? (Cout > 64 ? 128 : 64) : (Cout > 32 ? 64 : 32);
    const int rinp = (tm + halo + 31) / 32 * 32;
    // one stage buffer: weights + input run, and big enough for one output slice
    size_t bufsz = (size_t)rinp * 32 + (size_t)ntaps * TN * 32;
    const int tne = TN > 64 ? 64 : TN;
    const size_t need = (size_t)tm * tne * 2;
    if (bufsz < need) bufsz = need;
    const size_t lds = 2 * bufsz;
    FNET_CHECK_ARG(lds <= 160 * 1024);
    // the DMA window offsets are 32-bit
    FNET_CHECK_ARG((long long)(rinp + 2) * op.i[3] * 2 < (1ll << 30));
    // mode 1 keeps a lane's gather offsets of the pieces its wave issues in S2_MAXP registers
    FNET_CHECK_ARG(mode != 1 || (rinp / 32 + ntaps * TN / 32 + (big ? 8 : 4) - 1) / (big ? 8 : 4) <= S2_MAXP);
    const int n_mtiles = (Npix + tm - 1) / tm, n_ntiles = (Cout + TN - 1) / TN;
    const int n_items = 8 * ((n_mtiles + 7) / 8) * n_ntiles;
    // persistent grid: one (big) or two (small; LDS allows 2 at <= 80 KB each) blocks per CU, a multiple of 8 for the XCD mapping
    // (the 256 x 32 tile -- 116-133 registers, <= 53 KB -- runs three blocks per CU: its layers have one or two 16-channel stages per tile,
    // 18 MFMAs per wave and stage, far less than a DMA round trip, so what hides the latency is blocks in flight)
    static const int bpc32 = [] { const char* e = getenv("FNET_CONV_BPC32"); return e ? atoi(e) : 3; }();
    int bpc = (!big && lds <= 80 * 1024) ? 2 : 1;
    if (!big && TN == 32 && bpc32 >= 3 && 3 * lds <= 160 * 1024) bpc = 3;
    int nblk = bpc * cus;
    nblk = (nblk + 7) / 8 * 8;
    if (nblk > n_items) nblk = n_items;
    const dim3 grid(nblk);
    const bf16_t* src = (const bf16_t*)op.p[0];
    const bf16_t* Wk = (const bf16_t*)op.p[1];
    const float* bias = (const float*)op.p[2];
    bf16_t* dst = (bf16_t*)op.p[3];
#define FNET_LAUNCH_CONV(TN_, NW_, MODE_)                                                                                            \
    {                                                                                                                                \
        int rc = set_lds_limit(k_conv3x3_dma<TN_, NW_, MODE_>, lds);                                                                 \
        if (rc) return rc;                                                                                                           \
        hipLaunchKernelGGL((k_conv3x3_dma<TN_, NW_, MODE_>), grid, dim3(NW_ * 64), lds, st, src, Wk, bias, dst, Npix, Wp, Cin,       \
                           op.i[3], op.i[4], Cout, op.i[6], op.i[7], n_mtiles, n_ntiles, rinp, (int)bufsz, stamps, g2);              \
    }
#define FNET_LAUNCH_CONV_M(TN_, NW_)                                                                                                 \
    {                                                                                                                                \
        if (mode == 0) FNET_LAUNCH_CONV(TN_, NW_, 0)                                                                                 \
        else if (mode == 1) FNET_LAUNCH_CONV(TN_, NW_, 1)                                                                            \
        else FNET_LAUNCH_CONV(TN_, NW_, 2)                                                                                           \
    }
    if (big && TN == 128) {
        if (mode == 2) FNET_LAUNCH_CONV(128, 8, 2) else FNET_LAUNCH_CONV(128, 8, 0)
    }
    else if (big) FNET_LAUNCH_CONV_M(64, 8)
    else if (TN == 64) FNET_LAUNCH_CONV_M(64, 4)
    else FNET_LAUNCH_CONV_M(32, 4)
#undef FNET_LAUNCH_CONV_M
#undef FNET_LAUNCH_CONV
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// WGRAD (impl = MFMA)  i: 10 B, 11 H, 12 slab capacity in partials (36 KB each), 13 mode (0 = 3x3; 1 = the 2x2 space-to-depth form of a
// stride-2 convolution: H and Wp describe the half-resolution dout, src is the full-resolution input of C = i[14] channels, Cin = 4C,
// dWk = [4][Cout][Cin_w]).  p: 3 = partial slab.
int fnet_wgrad_mfma(const FnetOp& op, hipStream_t st) {
    const int Npix = op.i[0], Wp = op.i[1], Cin = op.i[2], Cout = op.i[5], Cin_w = op.i[9];
    const int B = op.i[10], H = op.i[11], W = Wp - 2, s2 = op.i[13], Cs = op.i[14];
    FNET_CHECK_ARG(op.dtype == FNET_BF16 && op.p[3] && (s2 == 0 || s2 == 1));
    FNET_CHECK_ARG(Cin % 8 == 0 && Cout % 8 == 0 && op.i[3] % 8 == 0 && op.i[4] % 8 == 0 && op.i[6] % 8 == 0 && op.i[7] % 8 == 0);
    const int nco = Cout > 32 ? 2 : 1, nci = Cin > 32 ? 2 : 1;
    const int ntaps = s2 ? 4 : 9;
    FNET_CHECK_ARG(B > 0 && H >= 1 && W >= 1 && (long long)B * (H + 2) * Wp == Npix);
    FNET_CHECK_ARG(op.i[7] + Cout <= op.i[6] && (long long)(H + 2) * Wp * op.i[6] * 2 < (1ll << 31));
    if (s2) {
        FNET_CHECK_ARG(Cs >= 32 && Cs % 32 == 0 && Cin == 4 * Cs && op.i[4] + Cs <= op.i[3]);
        FNET_CHECK_ARG((long long)(2 * H + 2) * (2 * W + 2) * op.i[3] * 2 < (1ll << 31));
    } else {
        FNET_CHECK_ARG(op.i[4] + Cin <= op.i[3] && (long long)(H + 2) * Wp * op.i[3] * 2 < (1ll << 31));
    }
    FNET_CHECK_ARG((op.p[0] % 16) == 0 && (op.p[1] % 16) == 0 && (op.p[3] % 16) == 0);
    int log2cw = W > 16 ? 5 : 4;
    int CW = 1 << log2cw;
    const int co_tiles = (Cout + 32 * nco - 1) / (32 * nco), ci_tiles = (Cin + 32 * nci - 1) / (32 * nci);
    const int tiles = co_tiles * ci_tiles;
    // one block per CU over the whole launch: the launch shares the chip with lane 0 (r2 sweep: 256 -> 38.10 ms, 512 -> 38.4 ms per K4 step)
    static const int target_blocks = [] { const char* e = getenv("FNET_WG_BLOCKS"); return e ? atoi(e) : 256; }();
    // 128-pixel chunks (20 % fewer staged bytes per pixel) where every block still gets >= 3 of them; FNET_WG_PX=64|128 forces one size
    // (a knob, so that one test process can run both)
    int px = 64;
    {
        const long long nch128 = (long long)B * ((W + 15) / 16) * ((H + 7) / 8);          // 8 x 16 chunks, see below
        const int want = (target_blocks + tiles - 1) / tiles;
        if (nch128 >= 3ll * want) px = 128;
        const int forced_px = fnet_knob(FNET_KNOB_WG_PX);
        if (forced_px) px = forced_px == 128 ? 128 : 64;
        // 128-pixel chunks are 8 rows x 16 columns at any image width: 180 source pixels per panel (1.41x) against 204 (1.59x) for 4 x 32
        // -- a pixel's 32 channels are one 64-B DMA segment whatever the chunk shape
        if (px == 128) { log2cw = 4; CW = 16; }
    }
    const int CH = px / CW;
    const int nch = B * ((W + CW - 1) / CW) * ((H + CH - 1) / CH);
    int splits = (target_blocks + tiles - 1) / tiles;
    if (splits > nch) splits = nch;
    int cpb = (nch + splits - 1) / splits;
    splits = (nch + cpb - 1) / cpb;
    const int nparts = splits;               // one partial per block and panel pair (the ks waves of a pair combine in LDS)
    const long long need = (long long)co_tiles * nco * ci_tiles * nci * nparts;
    FNET_CHECK_ARG(need <= op.i[12]);
    const bf16_t* src = (const bf16_t*)op.p[0];
    const bf16_t* dout = (const bf16_t*)op.p[1];
    float* slab = (float*)op.p[3];
    const dim3 grid(splits, co_tiles, ci_tiles);
    // (PMC: 136 -> 121 MB fetched per launch -- the data gradient's dout tiles are found in the XCD's L2; step 34.24 -> 34.18 ms)
    const int xcd_map = (fnet_knob(FNET_KNOB_WG_XCD) && splits % 8 == 0 && splits >= 8) ? 1 : 0;
#define WGD_LAUNCH3(A, B_, S2_, PX_)                                                                                                  \
    {                                                                                                                                \
        size_t lds = wg_stages(PX_) * ((size_t)A * PX_ * PANEL_ROW + (size_t)B_ * wg_srows(PX_, S2_) * PANEL_ROW);                         \
        if (A * B_ < 4 && lds < (size_t)A * B_ * WG_PART * 4) lds = (size_t)A * B_ * WG_PART * 4;                                     \
        int rc = set_lds_limit(k_wgrad_dma<A, B_, S2_, PX_>, lds);                                                                    \
        if (rc) return rc;                                                                                                           \
        hipLaunchKernelGGL((k_wgrad_dma<A, B_, S2_, PX_>), grid, dim3(kThreads), lds, st, src, dout, slab, B, H, W, op.i[3], op.i[4], \
                           op.i[6], op.i[7], log2cw, cpb, nch, nparts, ci_tiles, Cs, xcd_map);                                       \
    }
#define WGD_LAUNCH2(A, B_, S2_)                                                                                                       \
    {                                                                                                                                \
        if (px == 128) WGD_LAUNCH3(A, B_, S2_, 128) else WGD_LAUNCH3(A, B_, S2_, 64)                                                  \
    }
#define WGD_LAUNCH(A, B_)                                                                                                            \
    {                                                                                                                                \
        if (s2) WGD_LAUNCH2(A, B_, true) else WGD_LAUNCH2(A, B_, false)                                                              \
    }
    if (nco == 2 && nci == 2) WGD_LAUNCH(2, 2)
    else if (nco == 2 && nci == 1) WGD_LAUNCH(2, 1)
    else if (nco == 1 && nci == 2) WGD_LAUNCH(1, 2)
    else WGD_LAUNCH(1, 1)
#undef WGD_LAUNCH
#undef WGD_LAUNCH2
#undef WGD_LAUNCH3
    const int elems = ntaps * Cout * Cin;
    const int eb = (elems + kThreads - 1) / kThreads;
    static const int red_blocks = [] { const char* e = getenv("FNET_WG_RED_BLOCKS"); return e ? atoi(e) : 512; }();
    int groups = (red_blocks + eb - 1) / eb;
    if (groups > nparts) groups = nparts;
    const int ppg = (nparts + groups - 1) / groups;
    groups = (nparts + ppg - 1) / ppg;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(eb, groups), dim3(kThreads), 0, st, (const float*)slab, (float*)op.p[2], Cout, Cin, Cin_w,
                       nparts, ppg, ci_tiles * nci, ntaps);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}
