// Plan executor, hipGraph capture/replay, HIP-event timers: the native runtime under the Python host mirror.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "fnet_common.h"

static int dispatch_op(const FnetOp& op, hipStream_t st) {
    switch (op.op) {
        case FNET_OP_NOP: return FNET_OK;
        case FNET_OP_PACK_INPUT: return fnet_op_pack_input(op, st);
        case FNET_OP_CONV3X3: return fnet_op_conv3x3(op, st);
        case FNET_OP_WGRAD: return fnet_op_wgrad(op, st);
        case FNET_OP_IN_STATS: return fnet_op_in_stats(op, st);
        case FNET_OP_APPLY: return fnet_op_apply(op, st);
        case FNET_OP_UPSAMPLE2X: return fnet_op_upsample2x(op, st);
        case FNET_OP_SUBSAMPLE_BIAS: return fnet_op_subsample_bias(op, st);
        case FNET_OP_PAD_RING: return fnet_op_pad_ring(op, st);
        case FNET_OP_HEAD_FWD: return fnet_op_head_fwd(op, st);
        case FNET_OP_LOSS_REDUCE: return fnet_op_loss_reduce(op, st);
        case FNET_OP_LOSS_FINALIZE: return fnet_op_loss_finalize(op, st);
        case FNET_OP_LOSS_GRAD: return fnet_op_loss_grad(op, st);
        case FNET_OP_HEAD_BWD: return fnet_op_head_bwd(op, st);
        case FNET_OP_INACT_BWD_REDUCE: return fnet_op_inact_bwd_reduce(op, st);
        case FNET_OP_INACT_BWD_APPLY: return fnet_op_inact_bwd_apply(op, st);
        case FNET_OP_AFFINE_GRAD: return fnet_op_affine_grad(op, st);
        case FNET_OP_AFFINE_GRAD_MULTI: return fnet_op_affine_grad_multi(op, st);
        case FNET_OP_METRICS: return fnet_op_metrics(op, st);
        case FNET_OP_GATHER_BATCH: return fnet_op_gather_batch(op, st);
        case FNET_OP_POOL2X: return fnet_op_pool2x(op, st);
        case FNET_OP_POOL2X_BWD: return fnet_op_pool2x_bwd(op, st);
        case FNET_OP_NORM_COMBINE: return fnet_op_norm_combine(op, st);
        case FNET_OP_NORM_COMBINE_BWD: return fnet_op_norm_combine_bwd(op, st);
        case FNET_OP_DEPTH2SPACE: return fnet_op_depth2space(op, st);
        case FNET_OP_SPACE2DEPTH: return fnet_op_space2depth(op, st);
        case FNET_OP_FOLD_SUM: return fnet_op_fold_sum(op, st);
        case FNET_OP_BIAS_GRAD: return fnet_op_bias_grad(op, st);
        case FNET_OP_ZERO_INSERT: return fnet_op_zero_insert(op, st);
        case FNET_OP_UPSAMPLE2X_BWD: return fnet_op_upsample2x_bwd(op, st);
        case FNET_OP_UNPACK_GRAD: return fnet_op_unpack_grad(op, st);
        case FNET_OP_ZERO: return fnet_op_zero(op, st);
        case FNET_OP_PREP_WEIGHTS: return fnet_op_prep_weights(op, st);
        case FNET_OP_GRAD_NORM: return fnet_op_grad_norm(op, st);
        case FNET_OP_ADAM: return fnet_op_adam(op, st);
        case FNET_OP_COPY_PLANES: return fnet_op_copy_planes(op, st);
        case FNET_OP_UNPREP_WGRAD: return fnet_op_unprep_wgrad(op, st);
        case FNET_OP_STENCIL: return fnet_op_stencil(op, st);
        default: return FNET_E_BADARG;
    }
}

// ---- lanes: a plan may spread its ops over the caller's stream (lane 0) and up to FNET_MAX_LANES-1 library-owned side streams.
// An op's lane is op.i[FNET_LANE_SLOT]; EVENT_RECORD / EVENT_WAIT order lanes against each other (hipEventRecord on the op's lane,
// hipStreamWaitEvent of the op's lane on that event).  The plan builder is responsible for joining every side lane back into
// lane 0 before the plan ends (and before anything reuses a buffer a side lane still reads).  Streams and events are created
// lazily, once per device and process, and never destroyed (a few hundred bytes each).  The pools are guarded by one mutex, so two host
// threads may call fnet_plan_run concurrently without corrupting them -- but event ids are per device, not per plan: two plans WITH LANES
// running at the same time on one device share the events of equal id and may wait on each other's records.  Run plans that use lanes
// from one thread per device (plans without lanes, e.g. the rollout frame, are independent).
#define FNET_MAX_DEVICES 16
static std::mutex g_pool_mu;
static hipStream_t g_lane_stream[FNET_MAX_DEVICES][FNET_MAX_LANES];
static hipEvent_t* g_events[FNET_MAX_DEVICES];
static int g_n_events[FNET_MAX_DEVICES];

static int lane_stream(int lane, hipStream_t caller, hipStream_t* out) {
    if (lane == 0) { *out = caller; return FNET_OK; }
    if (lane < 0 || lane >= FNET_MAX_LANES) return FNET_E_BADARG;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= FNET_MAX_DEVICES) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (!g_lane_stream[dev][lane]) {
        e = hipStreamCreateWithFlags(&g_lane_stream[dev][lane], hipStreamNonBlocking);
        if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    }
    *out = g_lane_stream[dev][lane];
    return FNET_OK;
}

static int plan_event(int id, hipEvent_t* out) {
    if (id < 0 || id >= (1 << 20)) return FNET_E_BADARG;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= FNET_MAX_DEVICES) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (id >= g_n_events[dev]) {
        int n = g_n_events[dev] ? g_n_events[dev] : 256;
        while (n <= id) n *= 2;
        hipEvent_t* p = (hipEvent_t*)realloc(g_events[dev], sizeof(hipEvent_t) * n);
        if (!p) { g_fnet_last_hip_error = (int)hipErrorOutOfMemory; return FNET_E_HIP; }
        for (int k = g_n_events[dev]; k < n; ++k) p[k] = nullptr;
        g_events[dev] = p;
        g_n_events[dev] = n;
    }
    if (!g_events[dev][id]) {
        e = hipEventCreateWithFlags(&g_events[dev][id], hipEventDisableTiming);
        if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    }
    *out = g_events[dev][id];
    return FNET_OK;
}

// ---- tuning knobs (fnet_common.h): name, environment variable, default
static const struct { const char* name; const char* env; int dflt; } g_knob_def[FNET_KNOB_COUNT] = {
    {"sa_minitems", "FNET_SA_MINITEMS", 96},   // fewest (image, 16-channel slice) blocks for which a norm / activation pair is fused
    {"sa_cs", "FNET_SA_CS", 0},                // forced channel-slice width of the fused launches (0: widest with a block per CU)
    {"sb_maxhw", "FNET_SB_MAXHW", 32},         // largest image side whose adjoint pair is fused
    {"wg_px", "FNET_WG_PX", 0},                // forced weight-gradient chunk size, 64 | 128 (0: by chunk count)
    {"fuse_head", "FNET_FUSE_HEAD", 1},        // HEAD_FWD + LOSS_REDUCE and LOSS_GRAD + HEAD_BWD pairs as one launch each
    {"wg_xcd", "FNET_WG_XCD", 1},              // weight-gradient chunks dealt to blocks per XCD (the eighth of the batch the dgrad conv beside it walks)
};
static int g_knob[FNET_KNOB_COUNT];
static std::once_flag g_knob_once;
static int knob_from_env(int id) {
    const char* e = getenv(g_knob_def[id].env);
    return e ? atoi(e) : g_knob_def[id].dflt;
}
int fnet_knob(int id) {
    std::call_once(g_knob_once, [] { for (int k = 0; k < FNET_KNOB_COUNT; ++k) g_knob[k] = knob_from_env(k); });
    return (id >= 0 && id < FNET_KNOB_COUNT) ? g_knob[id] : 0;
}

struct FnetGraph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};
struct FnetTimer {
    hipEvent_t a, b;
};

#define HIP_TRY(x)                                \
    do {                                          \
        hipError_t e__ = (x);                     \
        if (e__ != hipSuccess) {                  \
            g_fnet_last_hip_error = (int)e__;     \
            return FNET_E_HIP;                    \
        }                                         \
    } while (0)

extern "C" {

int fnet_abi_version(void) { return FNET_ABI_VERSION; }

const char* fnet_strerror(int code) {
    switch (code) {
        case FNET_OK: return "ok";
        case FNET_E_BADARG: return "bad argument (shape, alignment or enum) -- nothing was launched for that op";
        case FNET_E_HIP: return "HIP runtime error (see fnet_last_hip_error)";
        case FNET_E_UNSUPPORTED: return "unsupported configuration";
        case FNET_E_TOO_LARGE: return "input larger than the stated maximum of this allocation-free entry point -- nothing was launched";
        default: return "unknown error";
    }
}

int fnet_last_hip_error(void) { return g_fnet_last_hip_error; }

int fnet_device_info(int32_t out[4]) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    out[0] = prop.multiProcessorCount;
    out[1] = (int32_t)prop.sharedMemPerBlock;
    out[2] = prop.warpSize;
    out[3] = prop.major * 100 + prop.minor;
    return FNET_OK;
}

int fnet_plan_run(const FnetOp* ops, int32_t n_ops, void* stream, int32_t* failed_index) {
    if (!ops || n_ops < 0) return FNET_E_BADARG;
    hipStream_t caller = (hipStream_t)stream;
    static const bool prep_batching = [] { const char* e = getenv("FNET_PREP_BATCH"); return !e || e[0] != '0'; }();   // A/B switch
    for (int32_t k = 0; k < n_ops; ++k) {
        hipStream_t st = caller;
        int rc = lane_stream(ops[k].i[FNET_LANE_SLOT], caller, &st);
        if (rc == FNET_OK) {
            if (ops[k].op == FNET_OP_EVENT_RECORD || ops[k].op == FNET_OP_EVENT_WAIT) {
                hipEvent_t ev = nullptr;
                rc = plan_event(ops[k].i[0], &ev);
                if (rc == FNET_OK) {
                    const hipError_t e = ops[k].op == FNET_OP_EVENT_RECORD ? hipEventRecord(ev, st) : hipStreamWaitEvent(st, ev, 0);
                    if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; rc = FNET_E_HIP; }
                }
            } else if (ops[k].op == FNET_OP_IN_STATS && k + 1 < n_ops && ops[k + 1].op == FNET_OP_APPLY &&
                       fnet_can_fuse_stats_apply(ops[k], ops[k + 1])) {
                rc = fnet_op_stats_apply(ops[k], ops[k + 1], st);      // one launch for the pair (small images)
                if (rc == FNET_OK) ++k;
            } else if (ops[k].op == FNET_OP_INACT_BWD_REDUCE && k + 1 < n_ops && ops[k + 1].op == FNET_OP_INACT_BWD_APPLY &&
                       fnet_can_fuse_inact_bwd(ops[k], ops[k + 1])) {
                rc = fnet_op_inact_bwd_fused(ops[k], ops[k + 1], st);  // both adjoint passes of a single-source unit in one launch (small images)
                if (rc == FNET_OK) ++k;
            } else if (ops[k].op == FNET_OP_HEAD_FWD && k + 1 < n_ops && ops[k + 1].op == FNET_OP_LOSS_REDUCE &&
                       fnet_can_fuse_head_loss_fwd(ops[k], ops[k + 1])) {
                rc = fnet_op_head_loss_fwd(ops[k], ops[k + 1], st);    // head + clamp + loss sums in one pass
                if (rc == FNET_OK) ++k;
            } else if (ops[k].op == FNET_OP_LOSS_GRAD && k + 1 < n_ops && ops[k + 1].op == FNET_OP_HEAD_BWD &&
                       fnet_can_fuse_loss_head_bwd(ops[k], ops[k + 1])) {
                rc = fnet_op_loss_head_bwd(ops[k], ops[k + 1], st);    // loss gradient formed inside the head adjoint
                if (rc == FNET_OK) ++k;
            } else if (ops[k].op == FNET_OP_PREP_WEIGHTS && k + 1 < n_ops && ops[k + 1].op == FNET_OP_PREP_WEIGHTS && prep_batching) {
                // a run of weight re-layouts (the start of every training step) goes out as one launch per 32 records
                int32_t m = 1;
                while (k + m < n_ops && ops[k + m].op == FNET_OP_PREP_WEIGHTS && ops[k + m].dtype == ops[k].dtype &&
                       ops[k + m].i[FNET_LANE_SLOT] == ops[k].i[FNET_LANE_SLOT])
                    ++m;
                int done = 0;
                rc = fnet_op_prep_weights_batch(ops + k, m, st, &done);
                k += rc == FNET_OK ? m - 1 : done;            // on failure k is the offending record
            } else if (ops[k].op == FNET_OP_UNPREP_WGRAD && k + 1 < n_ops && ops[k + 1].op == FNET_OP_UNPREP_WGRAD && prep_batching) {
                // ... and so does a run of weight-gradient folds of one lane (the end of a backward pass, a data-parallel bucket)
                int32_t m = 1;
                while (k + m < n_ops && ops[k + m].op == FNET_OP_UNPREP_WGRAD && ops[k + m].i[FNET_LANE_SLOT] == ops[k].i[FNET_LANE_SLOT]) ++m;
                int done = 0;
                rc = fnet_op_unprep_wgrad_batch(ops + k, m, st, &done);
                k += rc == FNET_OK ? m - 1 : done;
            } else {
                rc = dispatch_op(ops[k], st);
            }
        }
        if (rc != FNET_OK) {
            if (failed_index) *failed_index = k;
            return rc;
        }
    }
    if (failed_index) *failed_index = -1;
    return FNET_OK;
}

int fnet_set_knob(const char* name, int32_t value) {
    if (!name) return FNET_E_BADARG;
    fnet_knob(0);                                        // make sure the environment was read first
    for (int k = 0; k < FNET_KNOB_COUNT; ++k)
        if (!strcmp(name, g_knob_def[k].name)) {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            g_knob[k] = value < 0 ? knob_from_env(k) : value;
            return FNET_OK;
        }
    return FNET_E_BADARG;
}

int fnet_lane_stream(int32_t lane, void** stream_out) {
    if (!stream_out || lane < 1 || lane >= FNET_MAX_LANES) return FNET_E_BADARG;
    hipStream_t st = nullptr;
    const int rc = lane_stream(lane, nullptr, &st);
    if (rc != FNET_OK) return rc;
    *stream_out = (void*)st;
    return FNET_OK;
}

int fnet_graph_create(const FnetOp* ops, int32_t n_ops, void** graph_out) {
    if (!ops || n_ops <= 0 || !graph_out) return FNET_E_BADARG;
    hipStream_t cap;
    HIP_TRY(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; hipStreamDestroy(cap); return FNET_E_HIP; }
    int32_t bad = -1;
    const int rc = fnet_plan_run(ops, n_ops, (void*)cap, &bad);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(cap, &g);
    hipStreamDestroy(cap);
    if (rc != FNET_OK) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) { g_fnet_last_hip_error = (int)e; return FNET_E_HIP; }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; hipGraphDestroy(g); return FNET_E_HIP; }
    FnetGraph* fg = new FnetGraph{g, ex};
    *graph_out = (void*)fg;
    return FNET_OK;
}

int fnet_graph_launch(void* graph, void* stream, int32_t repeat) {
    if (!graph || repeat < 1) return FNET_E_BADARG;
    FnetGraph* fg = (FnetGraph*)graph;
    for (int32_t r = 0; r < repeat; ++r) HIP_TRY(hipGraphLaunch(fg->exec, (hipStream_t)stream));
    return FNET_OK;
}

int fnet_graph_destroy(void* graph) {
    if (!graph) return FNET_E_BADARG;
    FnetGraph* fg = (FnetGraph*)graph;
    hipGraphExecDestroy(fg->exec);
    hipGraphDestroy(fg->graph);
    delete fg;
    return FNET_OK;
}

int fnet_timer_create(void** timer_out) {
    if (!timer_out) return FNET_E_BADARG;
    FnetTimer* t = new FnetTimer;
    hipError_t e = hipEventCreate(&t->a);
    if (e == hipSuccess) e = hipEventCreate(&t->b);
    if (e != hipSuccess) { g_fnet_last_hip_error = (int)e; delete t; return FNET_E_HIP; }
    *timer_out = (void*)t;
    return FNET_OK;
}
int fnet_timer_start(void* timer, void* stream) {
    if (!timer) return FNET_E_BADARG;
    HIP_TRY(hipEventRecord(((FnetTimer*)timer)->a, (hipStream_t)stream));
    return FNET_OK;
}
int fnet_timer_stop(void* timer, void* stream) {
    if (!timer) return FNET_E_BADARG;
    HIP_TRY(hipEventRecord(((FnetTimer*)timer)->b, (hipStream_t)stream));
    return FNET_OK;
}
int fnet_timer_elapsed_ms(void* timer, float* ms_out) {
    if (!timer || !ms_out) return FNET_E_BADARG;
    FnetTimer* t = (FnetTimer*)timer;
    HIP_TRY(hipEventSynchronize(t->b));
    HIP_TRY(hipEventElapsedTime(ms_out, t->a, t->b));
    return FNET_OK;
}
int fnet_timer_destroy(void* timer) {
    if (!timer) return FNET_E_BADARG;
    FnetTimer* t = (FnetTimer*)timer;
    hipEventDestroy(t->a);
    hipEventDestroy(t->b);
    delete t;
    return FNET_OK;
}

}  // extern "C"
