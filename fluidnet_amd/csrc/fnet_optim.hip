// Weight re-layout (OIHW fp32 master <-> [tap][Cout][Cin] kernel layout), global grad-norm, fused clip + Adam.
// Replaces clip_grad_norm_ + torch.optim.Adam.step (ml/training/trainer.py:52,192-195).
#include "fnet_common.h"

namespace {

constexpr int kThreads = 256;

// Wf[t][co][ci] = W[co][ci][t]            (ci < Cin and co < Cout, else 0)   forward conv, rows = Cout_pad, K dim = Cin_pad
// Wd[t][ci][co] = W[co][ci][8-t]          (same)                              dgrad as a conv: rows = Cout_d (padded Cin), K dim = Cout_pad
// Channel counts that are not a multiple of 8 (UNet_small-medium's 20) run zero-padded: the padded rows / columns are zeros here, the
// matching norm and bias parameters read as zeros from the flat buffer's alignment gaps, so padded channels carry exact zeros.
template <typename T>
__device__ __forceinline__ void prep_elem(int idx, const float* W, T* Wf, T* Wd, int Cout, int Cin, int Cin_pad, int Cout_d, int blocked,
                                          int Cout_pad, int Kd /* K dim of Wd: >= Cout_pad (the MFMA kernel walks K in 16s) */) {
    // logical layouts Wf[t][co][ci], Wd[t][ci][co] (flipped taps); "blocked" (MFMA convolution) stores the K axis in
    // 16-channel blocks outermost, [K/16][t][row][16], so that one LDS-DMA stage of the conv is a contiguous slab
    const int nf = 9 * Cout_pad * Cin_pad;
    const int nd = Wd ? 9 * Cout_d * Kd : 0;
    if (idx < nf) {
        const int ci = idx % Cin_pad, co = (idx / Cin_pad) % Cout_pad, t = idx / (Cin_pad * Cout_pad);
        const size_t o = (blocked & 1) ? ((((size_t)(ci >> 4) * 9 + t) * Cout_pad + co) * 16 + (ci & 15)) : (size_t)idx;
        st1<T>(Wf + o, (ci < Cin && co < Cout) ? W[((size_t)co * Cin + ci) * 9 + t] : 0.f);
    } else if (idx < nf + nd) {
        const int j = idx - nf;
        const int co = j % Kd, ci = (j / Kd) % Cout_d, t = j / (Kd * Cout_d);
        const size_t o = (blocked & 2) ? ((((size_t)(co >> 4) * 9 + t) * Cout_d + ci) * 16 + (co & 15)) : (size_t)j;
        st1<T>(Wd + o, (ci < Cin && co < Cout) ? W[((size_t)co * Cin + ci) * 9 + (8 - t)] : 0.f);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_prep_weights(const float* W, T* Wf, T* Wd, int Cout, int Cin, int Cin_pad, int Cout_d, int blocked,
                                                      int Cout_pad, int Kd) {
    prep_elem<T>(blockIdx.x * kThreads + threadIdx.x, W, Wf, Wd, Cout, Cin, Cin_pad, Cout_d, blocked, Cout_pad, Kd);
}

// nn.ConvTranspose2d(C, C, kernel_size=2, stride=2) (unet.py:150; weight [ci][co][a][b]) as a convolution with 4*Cout output channels
// (phase q = 2a + b outermost) whose only non-zero tap is the centre one; DEPTH2SPACE then interleaves the phases:
// Wf[4][q*Cout + co][ci] = W[ci][co][a][b], Wd[4][ci][q*Cout + co] = the same (the centre tap is its own flip).
template <typename T>
__device__ __forceinline__ void prep_elem_t2(int idx, const float* W, T* Wf, T* Wd, int Cout, int Cin, int Kf, int Cout_d, int blocked, int Kd) {
    const int R = 4 * Cout;
    const int nf = 9 * R * Kf;
    const int nd = Wd ? 9 * Cout_d * Kd : 0;
    if (idx < nf) {
        const int ci = idx % Kf, row = (idx / Kf) % R, t = idx / (Kf * R);
        const int q = row / Cout, co = row - q * Cout;
        const size_t o = (blocked & 1) ? ((((size_t)(ci >> 4) * 9 + t) * R + row) * 16 + (ci & 15)) : (size_t)idx;
        st1<T>(Wf + o, (t == 4 && ci < Cin) ? W[((size_t)ci * Cout + co) * 4 + q] : 0.f);
    } else if (idx < nf + nd) {
        const int j = idx - nf;
        const int k = j % Kd, ci = (j / Kd) % Cout_d, t = j / (Kd * Cout_d);
        const int q = k / Cout, co = k - q * Cout;
        const size_t o = (blocked & 2) ? ((((size_t)(k >> 4) * 9 + t) * Cout_d + ci) * 16 + (k & 15)) : (size_t)j;
        st1<T>(Wd + o, (t == 4 && k < R && ci < Cin) ? W[((size_t)ci * Cout + co) * 4 + q] : 0.f);
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_prep_weights_t2(const float* W, T* Wf, T* Wd, int Cout, int Cin, int Kf, int Cout_d, int blocked, int Kd) {
    prep_elem_t2<T>(blockIdx.x * kThreads + threadIdx.x, W, Wf, Wd, Cout, Cin, Kf, Cout_d, blocked, Kd);
}
// dW[ci][co][a][b] (+)= dWk[4][q*Cout + co][ci]
__device__ __forceinline__ void unprep_elem_t2(int idx, const float* dWk, float* dW, int Cout, int Cin, int Cin_w, int accumulate) {
    if (idx >= Cin * Cout * 4) return;
    const int q = idx & 3, co = (idx >> 2) % Cout, ci = idx / (4 * Cout);
    const float v = dWk[((size_t)4 * (4 * Cout) + q * Cout + co) * Cin_w + ci];
    dW[idx] = accumulate ? dW[idx] + v : v;
}
__global__ __launch_bounds__(256) void k_unprep_wgrad_t2(const float* dWk, float* dW, int Cout, int Cin, int Cin_w, int accumulate) {
    unprep_elem_t2(blockIdx.x * kThreads + threadIdx.x, dWk, dW, Cout, Cin, Cin_w, accumulate);
}

// Conv2d(C, Cout, 3, stride 2, padding 1) as the 2x2-tap convolution over the space-to-depth view of its input (k_conv3x3_dma, MODE 1 / 2):
// K index k = (py, px, c) = ph * C + c of the 4C axis, tap t' = (dy, dx) carries W[co][c][ky = 2dy + py][kx = 2dx + px] (0 if ky or kx = 3).
//   Wf[4][Cout_pad][4C], K-blocked [4C/16][4][Cout_pad][16];  Wd[(ty,tx)][k][co] = Wf[(1-ty, 1-tx)][co][k], K-blocked [Kd/16][4][4C][16]
template <typename T>
__device__ __forceinline__ void prep_elem_s2(int idx, const float* W, T* Wf, T* Wd, int Cout, int C, int Cout_pad, int Kd) {
    const int K = 4 * C;
    const int nf = 4 * Cout_pad * K;
    const int nd = Wd ? 4 * K * Kd : 0;
    int t, co, k;
    size_t o;
    T* out;
    if (idx < nf) {
        k = idx % K; co = (idx / K) % Cout_pad; t = idx / (K * Cout_pad);
        o = (((size_t)(k >> 4) * 4 + t) * Cout_pad + co) * 16 + (k & 15);
        out = Wf;
    } else if (idx < nf + nd) {
        const int j = idx - nf;
        co = j % Kd; k = (j / Kd) % K;
        const int td = j / (Kd * K);
        o = (((size_t)(co >> 4) * 4 + td) * K + k) * 16 + (co & 15);
        t = 3 - td;                        // (1 - ty, 1 - tx)
        out = Wd;
    } else {
        return;
    }
    const int ph = k / C, c = k - ph * C;
    const int ky = 2 * (t >> 1) + (ph >> 1), kx = 2 * (t & 1) + (ph & 1);
    st1<T>(out + o, (co < Cout && ky <= 2 && kx <= 2) ? W[((size_t)co * C + c) * 9 + ky * 3 + kx] : 0.f);
}
template <typename T>
__global__ __launch_bounds__(256) void k_prep_weights_s2(const float* W, T* Wf, T* Wd, int Cout, int C, int Cout_pad, int Kd) {
    prep_elem_s2<T>(blockIdx.x * kThreads + threadIdx.x, W, Wf, Wd, Cout, C, Cout_pad, Kd);
}

// Several PREP_WEIGHTS records in one launch (the plan executor hands consecutive records over together: a training step re-lays
// all ~30 weight tensors out before its first convolution, 4-13 us per launch with the chip idle in between).  Block b belongs to the
// descriptor j with blk0[j] <= b < blk0[j + 1]; each element is produced by the same code as in the single-record kernels.
constexpr int PREP_MAXB = 32;
struct PrepDesc {
    const float* W;
    void* Wf;
    void* Wd;
    int Cout, Cin, a2, Cout_d, blocked, Cout_pad, Kd, kind;   // a2 = Cin_pad (kind 0) / Kf (kind 1); kind 2 uses Cout, Cin (= C), Cout_pad, Kd
};
struct PrepBatch {
    PrepDesc d[PREP_MAXB];
    int blk0[PREP_MAXB + 1];
    int n;
};
template <typename T>
__global__ __launch_bounds__(256) void k_prep_weights_multi(PrepBatch pb) {
    int j = 0;
    while (j + 1 < pb.n && (int)blockIdx.x >= pb.blk0[j + 1]) ++j;      // wave-uniform scan over <= 32 entries
    const PrepDesc& d = pb.d[j];
    const int idx = ((int)blockIdx.x - pb.blk0[j]) * kThreads + threadIdx.x;
    if (d.kind == 1) prep_elem_t2<T>(idx, d.W, (T*)d.Wf, (T*)d.Wd, d.Cout, d.Cin, d.a2, d.Cout_d, d.blocked, d.Kd);
    else if (d.kind == 2) prep_elem_s2<T>(idx, d.W, (T*)d.Wf, (T*)d.Wd, d.Cout, d.Cin, d.Cout_pad, d.Kd);
    else prep_elem<T>(idx, d.W, (T*)d.Wf, (T*)d.Wd, d.Cout, d.Cin, d.a2, d.Cout_d, d.blocked, d.Cout_pad, d.Kd);
}
// dW[co][c][ky][kx] (+)= dWk[(ky>>1, kx>>1)][co][((ky&1), (kx&1), c)]
__device__ __forceinline__ void unprep_elem_s2(int idx, const float* dWk, float* dW, int Cout, int C, int Cin_w, int accumulate, int Cout_pad) {
    if (idx >= Cout * C * 9) return;
    const int tt = idx % 9, c = (idx / 9) % C, co = idx / (9 * C);
    const int ky = tt / 3, kx = tt - ky * 3;
    const int t = (ky >> 1) * 2 + (kx >> 1), ph = (ky & 1) * 2 + (kx & 1);
    const float v = dWk[((size_t)t * Cout_pad + co) * Cin_w + ph * C + c];
    dW[idx] = accumulate ? dW[idx] + v : v;
}
__global__ __launch_bounds__(256) void k_unprep_wgrad_s2(const float* dWk, float* dW, int Cout, int C, int Cin_w, int accumulate, int Cout_pad) {
    unprep_elem_s2(blockIdx.x * kThreads + threadIdx.x, dWk, dW, Cout, C, Cin_w, accumulate, Cout_pad);
}

// dW[co][ci][t] (+)= dWk[t][co][ci]   (dWk rows = Cout_pad, row pitch Cin_w; padded rows / columns are dropped)
__device__ __forceinline__ void unprep_elem(int idx, const float* dWk, float* dW, int Cout, int Cin, int Cin_w, int accumulate, int Cout_pad) {
    if (idx >= Cout * Cin * 9) return;
    const int t = idx % 9, ci = (idx / 9) % Cin, co = idx / (9 * Cin);
    const float v = dWk[((size_t)t * Cout_pad + co) * Cin_w + ci];
    dW[idx] = accumulate ? dW[idx] + v : v;
}
__global__ __launch_bounds__(256) void k_unprep_wgrad(const float* dWk, float* dW, int Cout, int Cin, int Cin_w, int accumulate, int Cout_pad) {
    unprep_elem(blockIdx.x * kThreads + threadIdx.x, dWk, dW, Cout, Cin, Cin_w, accumulate, Cout_pad);
}

// Several UNPREP_WGRAD records in one launch (as k_prep_weights_multi: the ~30 folds at the end of a backward pass were 3-4 us kernels
// with 15-75 us of idle queue between them in the kernel trace); the same element code as the single-record kernels.
struct UnprepDesc {
    const float* dWk;
    float* dW;
    int Cout, Cin, Cin_w, accumulate, Cout_pad, kind;
};
struct UnprepBatch {
    UnprepDesc d[PREP_MAXB];
    int blk0[PREP_MAXB + 1];
    int n;
};
__global__ __launch_bounds__(256) void k_unprep_wgrad_multi(UnprepBatch ub) {
    int j = 0;
    while (j + 1 < ub.n && (int)blockIdx.x >= ub.blk0[j + 1]) ++j;      // wave-uniform scan over <= 32 entries
    const UnprepDesc& d = ub.d[j];
    const int idx = ((int)blockIdx.x - ub.blk0[j]) * kThreads + threadIdx.x;
    if (d.kind == 1) unprep_elem_t2(idx, d.dWk, d.dW, d.Cout, d.Cin, d.Cin_w, d.accumulate);
    else if (d.kind == 2) unprep_elem_s2(idx, d.dWk, d.dW, d.Cout, d.Cin, d.Cin_w, d.accumulate, d.Cout_pad);
    else unprep_elem(idx, d.dWk, d.dW, d.Cout, d.Cin, d.Cin_w, d.accumulate, d.Cout_pad);
}

__global__ __launch_bounds__(256) void k_grad_norm(const float* g, double* out, long long n) {
    __shared__ double red[4];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const double v = (double)g[i];
        s += v * v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (||g|| + 1e-6));  torch.optim.Adam (no amsgrad, wd = 0):
// m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void k_adam(float* p, const float* g, float* m, float* v, const double* normsq, long long n, float lr,
                                              float b1, float b2, float eps, float bc1, float bc2_sqrt, float max_norm, float gscale) {
    float coef = gscale;
    if (max_norm > 0.f && normsq) {
        const float total = (float)sqrt(*normsq) * gscale;
        coef *= fminf(1.f, max_norm / (total + 1e-6f));
    }
    const float step = lr / bc1;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

}  // namespace

// PREP_WEIGHTS  i: 0 Cout,1 Cin,2 Cin_pad (K dim of Wf),3 Cout_d,4 blocked bits,5 Cout_pad (rows of Wf; 0 = Cout),6 Kd (K dim of Wd; 0 = Cout_pad),
//                  7 kind (0 conv3x3, 1 transposed 2x2 stride 2: see k_prep_weights_t2, 2 conv3x3 stride 2 in its 2x2 space-to-depth form:
//                    see k_prep_weights_s2; Cin = C, Cin_pad = Cout_d = 4C, both layouts K-blocked)
//               p: 0 W fp32 OIHW,1 Wf [9][Cout_pad][Cin_pad],2 Wd [9][Cout_d][Kd]|0
// validates one record and fills its descriptor and its element count (what the single-record launch and the batch both use)
static int prep_describe(const FnetOp& op, PrepDesc* d, int* total) {
    const int Cout = op.i[0], Cin = op.i[1], Cin_pad = op.i[2], Cout_d = op.i[3], Cout_pad = op.i[5] ? op.i[5] : Cout;
    FNET_CHECK_ARG(op.dtype == FNET_BF16 || op.dtype == FNET_F32);
    *d = PrepDesc{(const float*)op.p[0], (void*)op.p[1], (void*)op.p[2], Cout, Cin, Cin_pad, Cout_d, op.i[4], Cout_pad, 0, op.i[7]};
    if (op.i[7] == 1) {
        // transposed 2x2 stride-2 convolution (i[7] = 1): W fp32 [Cin][Cout][2][2] -> Wf [9][4*Cout][Cin_pad], Wd [9][Cout_d][Kd], Kd >= 4*Cout
        const int Kd = op.i[6] ? op.i[6] : 4 * Cout;
        FNET_CHECK_ARG(Cout > 0 && Cin > 0 && Cin_pad >= Cin && op.p[0] && op.p[1] && (op.p[2] == 0 || (Cout_d >= Cin && Kd >= 4 * Cout)));
        FNET_CHECK_ARG(op.i[4] >= 0 && op.i[4] <= 3 && (!(op.i[4] & 1) || Cin_pad % 16 == 0) && (!(op.i[4] & 2) || Kd % 16 == 0));
        d->Kd = Kd;
        *total = 9 * 4 * Cout * Cin_pad + (op.p[2] ? 9 * Cout_d * Kd : 0);
        return FNET_OK;
    }
    if (op.i[7] == 2) {
        const int Kd = op.i[6] ? op.i[6] : Cout_pad;
        FNET_CHECK_ARG(Cout > 0 && Cin >= 16 && Cin % 16 == 0 && Cin_pad == 4 * Cin && Cout_d == 4 * Cin && Cout_pad >= Cout && Kd >= Cout_pad);
        FNET_CHECK_ARG(op.i[4] == 3 && Kd % 16 == 0 && op.p[0] && op.p[1]);
        d->Kd = Kd;
        *total = 4 * Cout_pad * 4 * Cin + (op.p[2] ? 4 * 4 * Cin * Kd : 0);
        return FNET_OK;
    }
    FNET_CHECK_ARG(op.i[7] == 0);
    FNET_CHECK_ARG(Cout > 0 && Cin > 0 && Cin_pad >= Cin && Cout_pad >= Cout && op.p[0] && op.p[1] && (op.p[2] == 0 || Cout_d >= Cin));
    const int Kd = op.i[6] ? op.i[6] : Cout_pad;
    FNET_CHECK_ARG(op.i[4] >= 0 && op.i[4] <= 3 && (!(op.i[4] & 1) || Cin_pad % 16 == 0) && (!(op.i[4] & 2) || Kd % 16 == 0) && Kd >= Cout_pad);
    d->Kd = Kd;
    *total = 9 * Cout_pad * Cin_pad + (op.p[2] ? 9 * Cout_d * Kd : 0);
    return FNET_OK;
}

int fnet_op_prep_weights(const FnetOp& op, hipStream_t st) {
    PrepDesc d;
    int total = 0;
    const int rc = prep_describe(op, &d, &total);
    if (rc != FNET_OK) return rc;
    const dim3 grid((total + kThreads - 1) / kThreads);
    if (d.kind == 1) {
        DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_prep_weights_t2<T>, grid, dim3(kThreads), 0, st, d.W, (T*)d.Wf, (T*)d.Wd, d.Cout, d.Cin, d.a2,
                                                d.Cout_d, d.blocked, d.Kd));
    } else if (d.kind == 2) {
        DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_prep_weights_s2<T>, grid, dim3(kThreads), 0, st, d.W, (T*)d.Wf, (T*)d.Wd, d.Cout, d.Cin, d.Cout_pad,
                                                d.Kd));
    } else {
        DISPATCH_T(op.dtype, hipLaunchKernelGGL(k_prep_weights<T>, grid, dim3(kThreads), 0, st, d.W, (T*)d.Wf, (T*)d.Wd, d.Cout, d.Cin, d.a2,
                                                d.Cout_d, d.blocked, d.Cout_pad, d.Kd));
    }
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// n consecutive PREP_WEIGHTS records of one dtype in ceil(n / 32) launches.  Every record is validated like a single one; at the first
// bad record the records before it are still launched (as fnet_plan_run would have) and *n_done says how many were.
int fnet_op_prep_weights_batch(const FnetOp* ops, int n, hipStream_t st, int* n_done) {
    *n_done = 0;
    while (*n_done < n) {
        PrepBatch pb;
        pb.n = 0;
        pb.blk0[0] = 0;
        int rc = FNET_OK;
        while (pb.n < PREP_MAXB && *n_done + pb.n < n) {
            int total = 0;
            rc = prep_describe(ops[*n_done + pb.n], &pb.d[pb.n], &total);
            if (rc != FNET_OK) break;
            pb.blk0[pb.n + 1] = pb.blk0[pb.n] + (total + kThreads - 1) / kThreads;
            ++pb.n;
        }
        if (pb.n > 0) {
            for (int j = pb.n + 1; j <= PREP_MAXB; ++j) pb.blk0[j] = pb.blk0[pb.n];
            DISPATCH_T(ops[*n_done].dtype, hipLaunchKernelGGL(k_prep_weights_multi<T>, dim3(pb.blk0[pb.n]), dim3(kThreads), 0, st, pb));
            FNET_LAUNCH_CHECK();
            *n_done += pb.n;
        }
        if (rc != FNET_OK) return rc;
    }
    return FNET_OK;
}

// UNPREP_WGRAD  i: 0 Cout,1 Cin,2 Cin_w,3 accumulate,4 Cout_pad (0 = Cout),5 kind (0 conv3x3, 1 transposed 2x2, 2 conv3x3 stride 2 in 2x2
//                  form: dWk [4][Cout_pad][Cin_w >= 4 Cin]); p: 0 dWk fp32 [9][Cout_pad][Cin_w],1 dW fp32 OIHW
// validates one record and fills its descriptor and element count (what the single-record launch and the batch both use)
static int unprep_describe(const FnetOp& op, UnprepDesc* d, int* total) {
    const int Cout = op.i[0], Cin = op.i[1], Cin_w = op.i[2], Cout_pad = op.i[4] ? op.i[4] : Cout;
    FNET_CHECK_ARG(Cout > 0 && Cin > 0 && Cin_w >= Cin && Cout_pad >= Cout && op.p[0] && op.p[1]);
    FNET_CHECK_ARG(op.i[5] >= 0 && op.i[5] <= 2 && (op.i[5] != 2 || Cin_w >= 4 * Cin));
    *d = UnprepDesc{(const float*)op.p[0], (float*)op.p[1], Cout, Cin, Cin_w, op.i[3], Cout_pad, op.i[5]};
    *total = op.i[5] == 1 ? Cin * Cout * 4 : 9 * Cout * Cin;      // transposed 2x2: dWk fp32 [9][4*Cout][Cin_w] -> dW [Cin][Cout][2][2]
    return FNET_OK;
}

int fnet_op_unprep_wgrad(const FnetOp& op, hipStream_t st) {
    UnprepDesc d;
    int total = 0;
    const int rc = unprep_describe(op, &d, &total);
    if (rc != FNET_OK) return rc;
    const dim3 grid((total + kThreads - 1) / kThreads);
    if (d.kind == 1) hipLaunchKernelGGL(k_unprep_wgrad_t2, grid, dim3(kThreads), 0, st, d.dWk, d.dW, d.Cout, d.Cin, d.Cin_w, d.accumulate);
    else if (d.kind == 2) hipLaunchKernelGGL(k_unprep_wgrad_s2, grid, dim3(kThreads), 0, st, d.dWk, d.dW, d.Cout, d.Cin, d.Cin_w, d.accumulate, d.Cout_pad);
    else hipLaunchKernelGGL(k_unprep_wgrad, grid, dim3(kThreads), 0, st, d.dWk, d.dW, d.Cout, d.Cin, d.Cin_w, d.accumulate, d.Cout_pad);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// n consecutive UNPREP_WGRAD records in ceil(n / 32) launches (same contract as fnet_op_prep_weights_batch)
int fnet_op_unprep_wgrad_batch(const FnetOp* ops, int n, hipStream_t st, int* n_done) {
    *n_done = 0;
    while (*n_done < n) {
        UnprepBatch ub;
        ub.n = 0;
        ub.blk0[0] = 0;
        int rc = FNET_OK;
        while (ub.n < PREP_MAXB && *n_done + ub.n < n) {
            int total = 0;
            rc = unprep_describe(ops[*n_done + ub.n], &ub.d[ub.n], &total);
            if (rc != FNET_OK) break;
            ub.blk0[ub.n + 1] = ub.blk0[ub.n] + (total + kThreads - 1) / kThreads;
            ++ub.n;
        }
        if (ub.n > 0) {
            for (int j = ub.n + 1; j <= PREP_MAXB; ++j) ub.blk0[j] = ub.blk0[ub.n];
            hipLaunchKernelGGL(k_unprep_wgrad_multi, dim3(ub.blk0[ub.n]), dim3(kThreads), 0, st, ub);
            FNET_LAUNCH_CHECK();
            *n_done += ub.n;
        }
        if (rc != FNET_OK) return rc;
    }
    return FNET_OK;
}

// GRAD_NORM  i: 0 n & 0x7fffffff,1 n >> 31; p: 0 grads fp32,1 sum of squares double* (+=; zero it first)
int fnet_op_grad_norm(const FnetOp& op, hipStream_t st) {
    const long long n = (long long)op.i[0] + ((long long)op.i[1] << 31);
    FNET_CHECK_ARG(n > 0 && op.p[0] && op.p[1]);
    long long blocks = (n + kThreads * 8 - 1) / (kThreads * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_grad_norm, dim3((unsigned)blocks), dim3(kThreads), 0, st, (const float*)op.p[0], (double*)op.p[1], n);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}

// ADAM  i: 0 n & 0x7fffffff,1 n >> 31; f: 0 lr,1 beta1,2 beta2,3 eps,4 bc1 = 1-beta1^t,5 sqrt(1-beta2^t),6 max_norm (<=0: no clip),7 grad scale
//       p: 0 params,1 grads,2 exp_avg,3 exp_avg_sq,4 sum of squares double*|0
int fnet_op_adam(const FnetOp& op, hipStream_t st) {
    const long long n = (long long)op.i[0] + ((long long)op.i[1] << 31);
    FNET_CHECK_ARG(n > 0 && op.p[0] && op.p[1] && op.p[2] && op.p[3]);
    FNET_CHECK_ARG(op.f[4] > 0.f && op.f[5] > 0.f && (op.f[6] <= 0.f || op.p[4] != 0));
    long long blocks = (n + kThreads * 4 - 1) / (kThreads * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(kThreads), 0, st, (float*)op.p[0], (const float*)op.p[1], (float*)op.p[2],
                       (float*)op.p[3], (const double*)op.p[4], n, op.f[0], op.f[1], op.f[2], op.f[3], op.f[4], op.f[5], op.f[6],
                       op.f[7]);
    FNET_LAUNCH_CHECK();
    return FNET_OK;
}
