"""ctypes binding of the C ABI in include/fluidnet_hip.h.  No fallback: if the library is missing this raises."""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FNET_LIB overrides the library path (diagnostic / ablation builds made by tools/conv_bench.py); the default is the in-tree build
LIB_PATH = os.environ.get("FNET_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libfluidnet_hip.so")

# mirrors `struct FnetOp` (include/fluidnet_hip.h)
OP_DTYPE = np.dtype([("op", "<i4"), ("dtype", "<i4"), ("i", "<i4", (30,)), ("f", "<f4", (8,)), ("p", "<u8", (10,))], align=True)
assert OP_DTYPE.itemsize == 240, OP_DTYPE.itemsize

F32, BF16 = 0, 1
PAD = {"zeros": 0, "replicate": 1, "reflect": 2, "circular": 3}
ACT = {"relu": 0, "leaky_relu": 1, "gelu": 2, "silu": 3, "identity": 4}
OUTACT = {"linear_clamp": 0, "sigmoid_tanh": 1}
STENCIL = {"forward": 0, "central": 1}
CONV_REF, CONV_MFMA = 0, 1

OPC = dict(NOP=0, PACK_INPUT=1, CONV3X3=2, WGRAD=3, IN_STATS=4, APPLY=5, UPSAMPLE2X=6, SUBSAMPLE_BIAS=7, PAD_RING=8,
           HEAD_FWD=9, LOSS_REDUCE=10, LOSS_FINALIZE=11, LOSS_GRAD=12, HEAD_BWD=13, INACT_BWD_REDUCE=14,
           INACT_BWD_APPLY=15, AFFINE_GRAD=16, FOLD_SUM=17, BIAS_GRAD=18, ZERO_INSERT=19, UPSAMPLE2X_BWD=20,
           UNPACK_GRAD=21, ZERO=22, PREP_WEIGHTS=23, GRAD_NORM=24, ADAM=25, COPY_PLANES=26, UNPREP_WGRAD=27, STENCIL=28,
           AFFINE_GRAD_MULTI=29, METRICS=30, GATHER_BATCH=31, POOL2X=32, POOL2X_BWD=33, NORM_COMBINE=34, NORM_COMBINE_BWD=35, DEPTH2SPACE=36, SPACE2DEPTH=37,
           EVENT_RECORD=38, EVENT_WAIT=39)
ABI_VERSION = 2       # include/fluidnet_hip.h FNET_ABI_VERSION: the FnetOp slot meanings this module writes
LANE_SLOT = 29        # FnetOp.i[29]: the lane (stream) an op runs on; include/fluidnet_hip.h FNET_LANE_SLOT
OPNAME = {v: k for k, v in OPC.items()}


class FluidNetHipError(RuntimeError):
    pass


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load libfluidnet_hip.so (built by `python -m fluidnet_amd.build`).  Fails loudly when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FluidNetHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m fluidnet_amd.build` "
            "(hipcc, gfx950). fluidnet_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    fp = ctypes.c_void_p
    L.fnet_abi_version.restype = ctypes.c_int
    L.fnet_strerror.restype = ctypes.c_char_p
    L.fnet_strerror.argtypes = [ctypes.c_int]
    L.fnet_last_hip_error.restype = ctypes.c_int
    L.fnet_device_info.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    L.fnet_plan_run.restype = ctypes.c_int
    L.fnet_plan_run.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
    L.fnet_graph_create.restype = ctypes.c_int
    L.fnet_graph_create.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    L.fnet_graph_launch.restype = ctypes.c_int
    L.fnet_graph_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    L.fnet_graph_destroy.restype = ctypes.c_int
    L.fnet_graph_destroy.argtypes = [ctypes.c_void_p]
    L.fnet_physics_loss_terms_noalloc.restype = ctypes.c_int
    L.fnet_physics_loss_terms_noalloc.argtypes = [fp] * 5 + [ctypes.c_int32] * 10 + [ctypes.c_float] * 5 + [fp, ctypes.c_void_p]
    L.fnet_lane_stream.restype = ctypes.c_int
    L.fnet_lane_stream.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    L.fnet_spatial_gradients.restype = ctypes.c_int
    L.fnet_spatial_gradients.argtypes = [fp, fp, fp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                         ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    L.fnet_divergence.restype = ctypes.c_int
    L.fnet_divergence.argtypes = [fp, fp, fp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                  ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    for name in ("fnet_timer_create",):
        getattr(L, name).restype = ctypes.c_int
        getattr(L, name).argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    for name in ("fnet_timer_start", "fnet_timer_stop"):
        getattr(L, name).restype = ctypes.c_int
        getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.fnet_timer_elapsed_ms.restype = ctypes.c_int
    L.fnet_timer_elapsed_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    L.fnet_timer_destroy.restype = ctypes.c_int
    L.fnet_timer_destroy.argtypes = [ctypes.c_void_p]
    i32, i64, vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p
    fpp = ctypes.POINTER(ctypes.c_void_p)
    sigs = {"fnet_unet_param_count": [vp, ctypes.POINTER(i64)],
            "fnet_rollout_create": [vp, vp, i64, i32, i32, i32, fpp],
            "fnet_rollout_set_weights": [vp, vp, i64], "fnet_rollout_set_masks": [vp, vp, vp], "fnet_rollout_set_state": [vp, vp, vp],
            "fnet_rollout_add_impulse": [vp, vp, vp], "fnet_rollout_step": [vp, i32, vp], "fnet_rollout_state_ptr": [vp, fpp, fpp],
            "fnet_rollout_read_state": [vp, vp, vp], "fnet_rollout_destroy": [vp]}
    for name, args in sigs.items():
        getattr(L, name).restype = ctypes.c_int
        getattr(L, name).argtypes = args
    L.fnet_set_knob.restype = ctypes.c_int
    L.fnet_set_knob.argtypes = [ctypes.c_char_p, ctypes.c_int32]
    if L.fnet_abi_version() != ABI_VERSION:
        raise FluidNetHipError(f"{LIB_PATH} has ABI version {L.fnet_abi_version()}, this package needs {ABI_VERSION} (FnetOp slot meanings "
                               "differ): rebuild with `python -m fluidnet_amd.build --force`")
    _lib = L
    return L


EXPORTS = ["fnet_abi_version", "fnet_strerror", "fnet_last_hip_error", "fnet_device_info", "fnet_plan_run", "fnet_lane_stream", "fnet_set_knob", "fnet_physics_loss_terms_noalloc", "fnet_graph_create",
           "fnet_graph_launch", "fnet_graph_destroy", "fnet_spatial_gradients", "fnet_divergence", "fnet_timer_create",
           "fnet_timer_start", "fnet_timer_stop", "fnet_timer_elapsed_ms", "fnet_timer_destroy", "fnet_unet_param_count",
           "fnet_rollout_create", "fnet_rollout_set_weights", "fnet_rollout_set_masks", "fnet_rollout_set_state", "fnet_rollout_add_impulse",
           "fnet_rollout_step", "fnet_rollout_state_ptr", "fnet_rollout_read_state", "fnet_rollout_destroy"]


def check(rc: int, what: str = "", ops: Optional[np.ndarray] = None, failed: int = -1) -> None:
    if rc == 0:
        return
    L = lib()
    msg = L.fnet_strerror(rc).decode()
    extra = ""
    if ops is not None and 0 <= failed < len(ops):
        o = ops[failed]
        extra = f" at op #{failed} ({OPNAME.get(int(o['op']), '?')}) i={list(o['i'][:18])} f={list(o['f'])}"
    if rc == -2:
        extra += f" hipError={L.fnet_last_hip_error()}"
    raise FluidNetHipError(f"{what}: {msg}{extra}")


def current_stream_ptr() -> int:
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


def run_ops(ops: np.ndarray, stream: Optional[int] = None) -> None:
    """fnet_plan_run on torch's current stream (or the given hipStream_t)."""
    assert ops.dtype == OP_DTYPE and ops.flags["C_CONTIGUOUS"]
    st = current_stream_ptr() if stream is None else stream
    failed = ctypes.c_int32(-1)
    rc = lib().fnet_plan_run(ops.ctypes.data_as(ctypes.c_void_p), len(ops), ctypes.c_void_p(st), ctypes.byref(failed))
    check(rc, "fnet_plan_run", ops, failed.value)


def set_knob(name: str, value: int) -> None:
    """fnet_set_knob: override a launch heuristic (tests / A-B runs); value < 0 restores the environment's value or the default."""
    check(lib().fnet_set_knob(name.encode(), int(value)), f"fnet_set_knob({name})")


def lane_stream(lane: int) -> int:
    """hipStream_t of side lane `lane` (1 .. FNET_MAX_LANES-1) on the current device, created on first use."""
    h = ctypes.c_void_p()
    check(lib().fnet_lane_stream(lane, ctypes.byref(h)), "fnet_lane_stream")
    return int(h.value)


class Graph:
    """hipGraph capture of a plan (fnet_graph_create / launch / destroy)."""

    def __init__(self, ops: np.ndarray):
        self._ops = ops
        h = ctypes.c_void_p()
        check(lib().fnet_graph_create(ops.ctypes.data_as(ctypes.c_void_p), len(ops), ctypes.byref(h)), "fnet_graph_create")
        self._h = h

    def launch(self, repeat: int = 1, stream: Optional[int] = None) -> None:
        st = current_stream_ptr() if stream is None else stream
        check(lib().fnet_graph_launch(self._h, ctypes.c_void_p(st), repeat), "fnet_graph_launch")

    def __del__(self):
        try:
            if self._h:
                lib().fnet_graph_destroy(self._h)
        except Exception:
            pass


class Timer:
    """HIP events recorded on the stream the kernels run on."""

    def __init__(self):
        h = ctypes.c_void_p()
        check(lib().fnet_timer_create(ctypes.byref(h)), "fnet_timer_create")
        self._h = h

    def start(self, stream: Optional[int] = None):
        check(lib().fnet_timer_start(self._h, ctypes.c_void_p(current_stream_ptr() if stream is None else stream)), "timer_start")

    def stop(self, stream: Optional[int] = None):
        check(lib().fnet_timer_stop(self._h, ctypes.c_void_p(current_stream_ptr() if stream is None else stream)), "timer_stop")

    def elapsed_ms(self) -> float:
        ms = ctypes.c_float()
        check(lib().fnet_timer_elapsed_ms(self._h, ctypes.byref(ms)), "timer_elapsed")
        return float(ms.value)

    def __del__(self):
        try:
            lib().fnet_timer_destroy(self._h)
        except Exception:
            pass


def device_info() -> dict:
    out = (ctypes.c_int32 * 4)()
    check(lib().fnet_device_info(out), "fnet_device_info")
    return {"cus": out[0], "lds_per_block": out[1], "wave": out[2], "gfx": out[3]}


_SEED_SLOT = {OPC["APPLY"]: 13, OPC["INACT_BWD_REDUCE"]: 17, OPC["INACT_BWD_APPLY"]: 12}


def patch_seed(ops: np.ndarray, seed: int) -> None:
    """Set the Dropout2d seed of every op that evaluates the channel-mask hash (forward and backward must agree)."""
    seed &= 0x7FFFFFFF
    for code, slot in _SEED_SLOT.items():
        m = ops["op"] == code
        if m.any():
            ops["i"][m, slot] = seed
