/*
 * fluidnet_hip.h -- C ABI of libfluidnet_hip.so, the MI355X (gfx950) compute library behind
 * the FluidNet hot path (UNet forward/backward, physics loss, K-step rollout loss, clip+Adam,
 * autoregressive rollout).
 *
 * The reference (ohmatheus/FluidNet) has no FFI for this path: its boundary is the Python module
 * API of ml/ (SURVEY.md section 8b).  This header is what a binding for that API would call; every
 * entry point cites the reference interface it replaces (paths relative to the reference root).
 * All pointers are DEVICE pointers unless named host_*; no torch types appear here.  Every launch
 * goes to the hipStream_t passed in (as void*); nothing here allocates, frees or synchronises
 * unless the comment says so.  Return value: 0 = ok, negative = FNET_E_* (see fnet_strerror).
 *
 * Data layout ("P-tensor"): activations live in HBM as channels-last, spatially PADDED arrays
 *     [B][H+2][W+2][ldc]   element type fp32 or bf16,
 * one ring of padding pixels filled according to the conv padding mode (zeros / replicate /
 * reflect / circular), so that every 3x3 convolution is a "valid" convolution over the
 * flattened pixel axis f = (n*(H+2)+y)*(W+2)+x with nine constant offsets.  A view selects
 * channels [coff, coff+C) of a buffer with ldc channels (this is how skip-connection concat
 * costs nothing).  Field planes at the API surface (inputs, targets, predictions) are plain
 * fp32 NCHW like the reference's tensors.
 */
#ifndef FLUIDNET_HIP_H
#define FLUIDNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FNET_ABI_VERSION 2        /* 2: FnetOp slot meanings of round 2 (CONV3X3 mode, lanes in slot 29), fnet_set_knob */

/* error codes */
#define FNET_OK 0
#define FNET_E_BADARG (-1)   /* shape / alignment / enum violation (host-side check, nothing launched) */
#define FNET_E_HIP (-2)      /* a HIP runtime call failed; fnet_last_hip_error() has the code */
#define FNET_E_UNSUPPORTED (-3)
#define FNET_E_TOO_LARGE (-4)  /* the input exceeds the stated maximum of an allocation-free entry point (nothing launched) */

/* element type of P-tensors */
#define FNET_F32 0
#define FNET_BF16 1

/* padding mode (ml/models/unet.py:13 PaddingType; ml/training/physics_loss.py:32-33) */
#define FNET_PAD_ZEROS 0
#define FNET_PAD_REPLICATE 1
#define FNET_PAD_REFLECT 2
#define FNET_PAD_CIRCULAR 3

/* activation (ml/models/unet.py:31-39) */
#define FNET_ACT_RELU 0
#define FNET_ACT_LEAKY 1
#define FNET_ACT_GELU 2
#define FNET_ACT_SILU 3
#define FNET_ACT_IDENTITY 4

/* output activation (ml/models/unet.py:296-301) */
#define FNET_OUT_LINEAR_CLAMP 0
#define FNET_OUT_SIGMOID_TANH 1

/* stencil (ml/training/physics_loss.py:16) */
#define FNET_STENCIL_FORWARD 0
#define FNET_STENCIL_CENTRAL 1

/* conv implementation selector */
#define FNET_CONV_REF 0   /* plain fp32-accumulate FMA kernels, any dtype: the debugging / exact path */
#define FNET_CONV_MFMA 1  /* bf16 MFMA implicit GEMM (32x32x16), LDS-staged halo rows */

/* ---- op records ------------------------------------------------------------------------
 * A "plan" is an array of FnetOp executed in order on one stream by fnet_plan_run().  The host
 * side (fluidnet_amd/engine/plan.py) builds one plan per (UNetConfig, B, H, W, K) and replays
 * it every step; slot meaning per opcode is documented in include/fluidnet_ops.h.
 */
typedef struct FnetOp {
    int32_t op;       /* FNET_OP_* */
    int32_t dtype;    /* FNET_F32 / FNET_BF16 for the P-tensors of this op */
    int32_t i[30];    /* integer arguments */
    float f[8];       /* float arguments */
    uint64_t p[10];   /* device pointers (0 = absent) */
} FnetOp;

enum FnetOpCode {
    FNET_OP_NOP = 0,
    FNET_OP_PACK_INPUT = 1,
    FNET_OP_CONV3X3 = 2,
    FNET_OP_WGRAD = 3,
    FNET_OP_IN_STATS = 4,
    FNET_OP_APPLY = 5,
    FNET_OP_UPSAMPLE2X = 6,
    FNET_OP_SUBSAMPLE_BIAS = 7,
    FNET_OP_PAD_RING = 8,
    FNET_OP_HEAD_FWD = 9,
    FNET_OP_LOSS_REDUCE = 10,
    FNET_OP_LOSS_FINALIZE = 11,
    FNET_OP_LOSS_GRAD = 12,
    FNET_OP_HEAD_BWD = 13,
    FNET_OP_INACT_BWD_REDUCE = 14,
    FNET_OP_INACT_BWD_APPLY = 15,
    FNET_OP_AFFINE_GRAD = 16,
    FNET_OP_FOLD_SUM = 17,
    FNET_OP_BIAS_GRAD = 18,
    FNET_OP_ZERO_INSERT = 19,
    FNET_OP_UPSAMPLE2X_BWD = 20,
    FNET_OP_UNPACK_GRAD = 21,
    FNET_OP_ZERO = 22,
    FNET_OP_PREP_WEIGHTS = 23,
    FNET_OP_GRAD_NORM = 24,
    FNET_OP_ADAM = 25,
    FNET_OP_COPY_PLANES = 26,
    FNET_OP_UNPREP_WGRAD = 27,
    FNET_OP_STENCIL = 28,
    FNET_OP_AFFINE_GRAD_MULTI = 29,
    FNET_OP_METRICS = 30,
    FNET_OP_GATHER_BATCH = 31,
    FNET_OP_POOL2X = 32,          /* 2x2 average / max pooling of the pooled `down` variants (unet.py:93-94) */
    FNET_OP_POOL2X_BWD = 33,
    FNET_OP_NORM_COMBINE = 34,    /* BatchNorm2d / GroupNorm statistics pooled from the per-(n,c) ones (unet.py:20-28) */
    FNET_OP_NORM_COMBINE_BWD = 35,
    FNET_OP_DEPTH2SPACE = 36,     /* phase interleave + bias of the transposed-conv upsample (unet.py:150) */
    FNET_OP_SPACE2DEPTH = 37,
    FNET_OP_EVENT_RECORD = 38,    /* lanes: record event i[0] on the op's lane (see FNET_LANE_SLOT) */
    FNET_OP_EVENT_WAIT = 39,      /* lanes: the op's lane waits for the last record of event i[0] */
    FNET_OP_COUNT_
};

/* Lanes.  Every op runs on lane op.i[FNET_LANE_SLOT]: lane 0 is the stream passed to fnet_plan_run, lanes 1..FNET_MAX_LANES-1 are
 * side streams the library creates once per device (hipStreamNonBlocking).  EVENT_RECORD / EVENT_WAIT (event ids 0 .. 2^20-1, a
 * per-device pool shared by all plans: an id is recorded before it is waited for inside one fnet_plan_run) order lanes against
 * each other.  The plan builder joins every side lane back into lane 0 before the plan ends.  Used to run the weight-gradient
 * kernels (MFMA-bound) beside the normalisation / activation adjoint kernels (HBM- and VALU-bound) of the backward pass. */
#define FNET_LANE_SLOT 29
#define FNET_MAX_LANES 4

/* Library identity / diagnostics. */
int fnet_abi_version(void);
const char* fnet_strerror(int code);
int fnet_last_hip_error(void);
/* Device properties the host uses to size grids: [0]=CU count, [1]=max LDS per block, [2]=warp size, [3]=gfx major*100+minor. */
int fnet_device_info(int32_t out[4]);

/*
 * Execute a plan: ops[0..n) in order on `stream`.  Replaces the chain of ATen calls under
 * UNet.forward (ml/models/unet.py:281-303), PhysicsAwareLoss.forward
 * (ml/training/physics_loss.py:210-278), Trainer._compute_rollout_loss
 * (ml/training/trainer.py:311-385), loss.backward()/clip_grad_norm_/Adam.step
 * (ml/training/trainer.py:190-195) and the rollout loop body (ml/scripts/simple_infer.py:226-260).
 * Every op validates its arguments on the host before launching; on the first failure nothing
 * further is launched and *failed_index (if non-null) receives the op index.
 * The executor may issue adjacent records as one launch where that is the same computation: a run of PREP_WEIGHTS records (one launch
 * per 32), IN_STATS immediately followed by the APPLY of the same InstanceNorm unit on an image of at most 64 x 64, and
 * INACT_BWD_REDUCE immediately followed by the INACT_BWD_APPLY of the same single-source unit on at most 32 x 32 (ml/models/unet.py:65-69
 * and its autograd adjoint in one kernel each; statistics summed in a fixed order, outputs within one bf16 ulp of the separate
 * launches); HEAD_FWD immediately followed by the LOSS_REDUCE of its prediction, and LOSS_GRAD immediately followed by the HEAD_BWD that
 * consumes its dpred (ml/models/unet.py:294-303 + ml/training/physics_loss.py:210-278 in one pass each: z, pred, dpred and the activation
 * gradient bit-identical to the separate launches, the fp64 loss sums and the head weight / bias gradients to summation order); a run of
 * UNPREP_WGRAD records of one lane (one launch per 32).  Records that do not qualify, or that would fail their own checks, take the ordinary
 * launchers; results and error reporting are otherwise those of record-by-record execution.
 */
int fnet_plan_run(const FnetOp* ops, int32_t n_ops, void* stream, int32_t* failed_index);
/* The hipStream_t behind side lane `lane` (1 .. FNET_MAX_LANES-1) of the current device, created on first use and owned by the
 * library.  No reference counterpart (the reference is single-stream): the data-parallel host uses it to order a gradient
 * bucket's RCCL all-reduce behind the lane that produced the bucket's last piece (fluidnet_amd/training/dp.py). */
int fnet_lane_stream(int32_t lane, void** stream_out);
/* Development / test control with no reference counterpart: overrides one of the launch heuristics that the library otherwise reads
 * from the environment once per process ("sa_minitems", "sa_cs", "sb_maxhw", "wg_px", "fuse_head", "wg_xcd" = FNET_SA_MINITEMS, FNET_SA_CS,
 * FNET_SB_MAXHW, FNET_WG_PX, FNET_FUSE_HEAD, FNET_WG_XCD; INTEGRATION.md "Development knobs").  value < 0 restores the environment's value or the default.  FNET_E_BADARG for an
 * unknown name.  Results never depend on a knob beyond summation order; production code does not call this. */
int fnet_set_knob(const char* name, int32_t value);

/*
 * hipGraph form of fnet_plan_run for launch-bound loops (the 600-frame rollout of
 * ml/scripts/simple_infer.py:226-260; BASELINE config 5).  fnet_graph_create captures the plan on
 * a private capture stream (this call allocates the graph objects and synchronises nothing else);
 * fnet_graph_launch replays it `repeat` times on `stream`; fnet_graph_destroy frees it.
 */
int fnet_graph_create(const FnetOp* ops, int32_t n_ops, void** graph_out);
int fnet_graph_launch(void* graph, void* stream, int32_t repeat);
int fnet_graph_destroy(void* graph);

/*
 * Single-op conveniences used by the host mirror of the reference's free functions and by the
 * parity tests (each is fnet_plan_run on a one-element plan):
 *   fnet_spatial_gradients  <- compute_spatial_gradients  (ml/training/physics_loss.py:19-50)
 *   fnet_divergence         <- compute_divergence         (ml/training/physics_loss.py:53-91)
 * field planes are fp32 [B][H][W] contiguous.
 */
int fnet_spatial_gradients(const float* field, float* grad_x, float* grad_y, int32_t B, int32_t H, int32_t W,
                           float dx, float dy, int32_t padding_mode, int32_t stencil_mode, void* stream);
int fnet_divergence(const float* velx, const float* vely, float* div, int32_t B, int32_t H, int32_t W, float dx,
                    float dy, int32_t padding_mode, int32_t stencil_mode, void* stream);

/*
 * Allocation-free physics-loss terms.  NO counterpart in the reference: it exists for callers that cannot call hipMalloc -- code
 * under hipGraph capture, the viewer's per-frame loop, harnesses that forbid allocation.  Computes the forward terms of
 * PhysicsAwareLoss (ml/training/physics_loss.py:210-278) exactly as fnet_plan_run does for the op sequence ZERO(sums) + LOSS_REDUCE +
 * LOSS_FINALIZE, but the eight fp64 sums those kernels need live in a 256-byte static __device__ array of the library (one copy per
 * GPU, held until the process ends) instead of a caller buffer; no hipMalloc / hipFree / allocating routine is called.
 *   pred [B,3,H,W]; target / density_in / emitter / collider: planes with the given batch strides (floats), as LOSS_REDUCE;
 *   flags bit0 divergence, bit1 gradient, bit2 emitter term; terms_out: DEVICE float[8] = mse, divergence, gradient, emitter, total,
 *   fluid-cell count (written on `stream`).
 * Limit: B*H*W <= 2^30 cells, else FNET_E_TOO_LARGE.  All calls on one GPU share the array (cleared at the start of every call):
 * calls that overlap on one GPU -- two streams or two threads -- race on it; serialise them.
 */
int fnet_physics_loss_terms_noalloc(const float* pred, const float* target, const float* density_in, const float* emitter,
                                    const float* collider, int32_t B, int32_t H, int32_t W, int32_t target_bstride, int32_t density_bstride,
                                    int32_t emitter_bstride, int32_t collider_bstride, int32_t padding_mode, int32_t stencil_mode, int32_t flags,
                                    float grid_spacing, float w_mse, float w_div, float w_grad, float w_emit, float* terms_out, void* stream);

/*
 * HIP-event timing of a stream region, on the stream the kernels were launched on (bench.py
 * uses this for roofline.achieved; torch.cuda.Event only sees torch's current stream).
 * fnet_timer_create allocates two events; start/stop record them; elapsed_ms synchronises on the
 * stop event and returns milliseconds.
 */
int fnet_timer_create(void** timer_out);
int fnet_timer_start(void* timer, void* stream);
int fnet_timer_stop(void* timer, void* stream);
int fnet_timer_elapsed_ms(void* timer, float* ms_out);
int fnet_timer_destroy(void* timer);

/*
 * Native inference rollout: the per-step work of the reference's C++ viewer
 * (engine/src/Simulation.cpp:236-367: pack six planes, Ort::Session::Run, copy the output into
 * the back buffer) as one device-resident loop.  Model family: the four shipped architectures' block
 * types (instance / batch / group norm, stride or pooled downsample, nearest or bilinear upsample);
 * weights = the float tensors of a reference checkpoint in state-dict order (ml/models/unet.py:204-279;
 * BatchNorm2d: weight, bias, running_mean, running_var -- not its int64 counter), concatenated as fp32;
 * bf16 MFMA arithmetic.
 *   fnet_unet_param_count   number of floats fnet_rollout_create expects for a configuration
 *   fnet_rollout_create     allocates every device buffer (hipMalloc), uploads and re-lays the weights,
 *                           captures one frame in a hipGraph; synchronises
 *   fnet_rollout_set_weights / set_masks / set_state   host -> device uploads (null = zeros); synchronise
 *   fnet_rollout_add_impulse  velx += ix, vely += iy from a host [B][2][H][W] array (Simulation.cpp:263-271) on `stream`; synchronises
 *                           `stream` before returning, so the host array may be reused or freed at once
 *   fnet_rollout_step       `frames` graph replays on `stream`: state_prev <- density, state <- UNet(state, prev, masks)
 *   fnet_rollout_state_ptr  device pointers of the state [B][3][H][W] and the previous density [B][H][W]
 *   fnet_rollout_read_state device -> host copy of the state; synchronises `stream`
 */
typedef struct FnetUNetConfig {
    int32_t in_channels;        /* 6 */
    int32_t out_channels;       /* 3 */
    int32_t base_channels;      /* multiple of 4, <= 64 (not a multiple of 8: those tensors run zero-padded) */
    int32_t depth;
    int32_t act;                /* FNET_ACT_* */
    int32_t padding_mode;       /* FNET_PAD_* */
    int32_t upsample;           /* 0 nearest, 1 bilinear */
    int32_t use_residual;       /* 0 / 1 */
    int32_t bottleneck_blocks;
    int32_t output_activation;  /* FNET_OUT_* */
    int32_t norm;               /* 0 instance, 1 batch (eval mode: running statistics), 2 group */
    int32_t downsample;         /* 0 stride-2 conv, 1 avgpool + conv, 2 maxpool + conv */
    int32_t group_norm_groups;  /* norm == 2: the reference's `groups` (largest divisor of the channel count <= it is used) */
} FnetUNetConfig;
int fnet_unet_param_count(const FnetUNetConfig* cfg, int64_t* n_floats);
int fnet_rollout_create(const FnetUNetConfig* cfg, const float* host_params, int64_t n_params, int32_t B, int32_t H, int32_t W,
                        void** handle_out);
int fnet_rollout_set_weights(void* handle, const float* host_params, int64_t n_params);
int fnet_rollout_set_masks(void* handle, const float* host_emitter, const float* host_collider);
int fnet_rollout_set_state(void* handle, const float* host_state, const float* host_prev_density);
int fnet_rollout_add_impulse(void* handle, const float* host_impulse_xy, void* stream);
int fnet_rollout_step(void* handle, int32_t frames, void* stream);
int fnet_rollout_state_ptr(void* handle, float** device_state, float** device_prev_density);
int fnet_rollout_read_state(void* handle, float* host_state, void* stream);
int fnet_rollout_destroy(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDNET_HIP_H */
