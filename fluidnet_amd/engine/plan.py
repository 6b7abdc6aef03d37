"""Plan builder: turns (UNetConfig, B, H, W, K) into flat op lists for fnet_plan_run.

The builder is tensor-agnostic: it emits OpRec records that reference Buf handles.  `bind()` allocates the
buffers on a device and lowers the records to the FnetOp array of include/fluidnet_hip.h.  (tests/sim_ops.py
interprets the very same records on CPU tensors, which is how the graph wiring is checked without a GPU.)

Forward graph = ml/models/unet.py:281-303; backward graph = its hand-derived adjoint;
rollout loss = ml/training/trainer.py:311-385; optimiser = trainer.py:190-195.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import runtime as rt
from .runtime import OPC


# ------------------------------------------------------------------------------------------------ handles
@dataclass(eq=False)
class Buf:
    name: str
    shape: Tuple[int, ...]
    kind: str                      # 'T' (plan element type), 'f32', 'f64'
    external: bool = False         # bound by the caller (params, inputs, outputs)
    tensor: object = None
    rc: int = 0                    # refcount for pooled temporaries
    parent: object = None          # (Buf, element offset): this buffer is a slice of another one (allocate() makes it a view)

    @property
    def numel(self) -> int:
        return int(np.prod(self.shape))


@dataclass(eq=False)
class Ref:
    buf: Buf
    off: int = 0                   # element offset


@dataclass(eq=False)
class PV:
    """View of channels [coff, coff+C) of a P-tensor buffer [B, H+2, W+2, ldc]."""
    buf: Buf
    C: int
    coff: int = 0

    @property
    def ldc(self) -> int:
        return self.buf.shape[3]

    @property
    def H(self) -> int:
        return self.buf.shape[1] - 2

    @property
    def W(self) -> int:
        return self.buf.shape[2] - 2

    @property
    def ref(self) -> Ref:
        return Ref(self.buf, 0)


@dataclass(eq=False)
class GradSrc:
    """A gradient contribution: a P view, to be read plainly (fold=0) or with the padding ring folded in (fold=1)."""
    pv: PV
    fold: int


@dataclass(eq=False)
class OpRec:
    code: int
    i: List[int] = field(default_factory=list)
    f: List[float] = field(default_factory=list)
    p: List[Optional[Ref]] = field(default_factory=list)
    dtype: int = 0
    tag: str = ""
    flops: float = 0.0             # algorithmic (useful) FLOPs of this launch, for roofline accounting
    lane: int = 0                  # 0 = the caller's stream, 1.. = a side stream of the library (fluidnet_hip.h "Lanes")


@dataclass(eq=False)
class Unit:
    """conv3x3 -> InstanceNorm -> activation -> Dropout2d [-> + residual]  (half of a ConvBlock, unet.py:65-69)."""
    conv: str
    norm: str
    xin: PV
    raw: Buf
    st: Buf
    out: PV
    res: Optional[PV]
    drop_site: int
    drop_p: float


@dataclass(eq=False)
class PlainConv:
    """conv3x3 + bias with no norm/activation: the stem (stride 1) and the stride-2 `down` convs."""
    conv: str
    xin: PV
    out: PV            # padded output at the conv's output resolution
    stride: int


@dataclass(eq=False)
class Tape:
    app: int
    in_p: Buf = None
    stem: PlainConv = None
    downs: list = field(default_factory=list)     # per level: dict(block=[u1,u2], res=[u1,u2]|None, down=PlainConv, cat=Buf)
    mids: list = field(default_factory=list)      # per block: [u1,u2]
    ups: list = field(default_factory=list)       # per level (forward order): dict(block, res, cat, xlow=PV, C)
    head_in: PV = None
    z: Buf = None
    pred: Buf = None


def unet_param_shapes(cfg) -> "OrderedDict[str, Tuple[int, ...]]":
    """State-dict order/shapes of the trainable tensors (ml/models/unet.py:204-279) for the supported family."""
    out: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def block(prefix, cin, cout):
        for idx, (ci, co) in enumerate(((cin, cout), (cout, cout)), start=1):
            out[f"{prefix}.c{idx}.weight"] = (co, ci, 3, 3)
            if cfg.norm == "none":
                out[f"{prefix}.c{idx}.bias"] = (co,)
            else:
                out[f"{prefix}.n{idx}.weight"] = (co,)
                out[f"{prefix}.n{idx}.bias"] = (co,)

    out["stem.weight"] = (cfg.base_channels, cfg.in_channels, 3, 3)
    out["stem.bias"] = (cfg.base_channels,)
    ch = cfg.base_channels
    skips = []
    for L in range(cfg.depth):
        oc = ch * 2
        block(f"downs.{L}.block", ch, oc)
        if cfg.use_residual:
            block(f"downs.{L}.res.b", oc, oc)
        dk = f"downs.{L}.down" if cfg.downsample == "stride" else f"downs.{L}.down.1"
        out[f"{dk}.weight"] = (oc, oc, 3, 3)
        out[f"{dk}.bias"] = (oc,)
        skips.append(oc)
        ch = oc
    for j in range(max(0, cfg.bottleneck_blocks)):
        block(f"mid.{j}.b", ch, ch)
    for i, L in enumerate(reversed(range(cfg.depth))):
        sc = skips[L]
        oc = sc // 2
        if cfg.upsample == "transpose":
            out[f"ups.{i}.up.weight"] = (ch, ch, 2, 2)
            out[f"ups.{i}.up.bias"] = (ch,)
        block(f"ups.{i}.block", ch + sc, oc)
        if cfg.use_residual:
            block(f"ups.{i}.res.b", oc, oc)
        ch = oc
    out["head.weight"] = (cfg.out_channels, ch, 1, 1)
    out["head.bias"] = (cfg.out_channels,)
    return out


def cpad(c: int) -> int:
    """Channel count a tensor of `c` channels is laid out with: the next multiple of 8 (UNet_small-medium's 20 -> 24).  Padded
    channels hold exact zeros: their conv rows / columns are zero (PREP_WEIGHTS) and their norm / bias parameters are the zero
    alignment gaps of param_layout."""
    return (c + 7) // 8 * 8


def kpad(c: int) -> int:
    """K extent of an MFMA convolution's weights over `c` input channels: the kernel walks K in 16-channel stages and, when c is
    only a multiple of 8, lets the last stage run into the next pixel against zero weights."""
    return (c + 15) // 16 * 16


def param_layout(cfg) -> "Tuple[OrderedDict[str, Tuple[int, ...]], OrderedDict[str, int], int]":
    """(shapes, offsets in floats, total floats) of the flat fp32 parameter / gradient / Adam buffers: tensors in state-dict order,
    each 32-B aligned, per-channel vectors given cpad(C) floats so that kernels may read a padded layer's gamma / beta / bias as zeros."""
    shapes = unet_param_shapes(cfg)
    offs: "OrderedDict[str, int]" = OrderedDict()
    off = 0
    for k, s in shapes.items():
        offs[k] = off
        n = cpad(s[0]) if len(s) == 1 else int(np.prod(s))
        off = (off + n + 7) // 8 * 8
    return shapes, offs, off


def group_size(groups: int, ch: int) -> int:
    """Channels per group of nn.GroupNorm as the reference builds it (unet.py:24-28): the largest g <= groups dividing ch."""
    g = min(groups, ch)
    while g > 1 and ch % g:
        g -= 1
    return ch // g


def norm_state_layout(cfg) -> "Tuple[OrderedDict[str, int], int]":
    """Offsets (floats) of BatchNorm2d's running_mean / running_var of every norm layer inside the flat norm-state buffer, in
    state-dict order, each 32-B aligned; empty unless norm == 'batch'."""
    offs: "OrderedDict[str, int]" = OrderedDict()
    off = 0
    if cfg.norm == "batch":
        for k, s in unet_param_shapes(cfg).items():
            if len(s) == 1 and ".n" in k and k.endswith(".weight"):
                base = k[: -len(".weight")]
                for leaf in ("running_mean", "running_var"):
                    offs[f"{base}.{leaf}"] = off
                    off = (off + cpad(s[0]) + 7) // 8 * 8
    return offs, off


WG_PART = 9 * 32 * 32          # floats per wave partial of the weight-gradient kernel (fnet_conv_mfma.hip)


def wgrad_slab_parts(cfg) -> int:
    """Upper bound of (co panels x ci panels x pixel splits x k-split waves) over the layers of `cfg`, mirroring the launch
    geometry of fnet_wgrad_mfma: ~512 blocks per launch, 4 waves each.  (The k-split waves of a panel pair now combine in LDS and write one
    partial per block and pair, so the last factor makes this a 1-4x over-estimate: kept as a safe upper bound.)"""
    worst = 0
    for k, s in unet_param_shapes(cfg).items():
        if len(s) == 4 and s[2] in (2, 3):
            # 3x3: [co][ci]; transposed 2x2 ([ci][co][2][2]): a centre-tap convolution into 4*co channels
            co, ci = (cpad(s[0]), max(16, cpad(s[1]))) if s[2] == 3 else (4 * s[1], max(16, s[0]))
            nco, nci = (2 if co > 32 else 1), (2 if ci > 32 else 1)
            cot, cit = -(-co // (32 * nco)), -(-ci // (32 * nci))
            splits = -(-512 // (cot * cit))
            worst = max(worst, cot * nco * cit * nci * splits * (4 // (nco * nci)))
            if s[2] == 3 and k.endswith(".down.weight") and cfg.downsample == "stride":
                # the 2x2 space-to-depth form of the stride-2 convolution: K axis 4*ci
                cit = -(-4 * ci // 64)
                splits = -(-512 // (cot * cit))
                worst = max(worst, cot * nco * cit * 2 * splits)
    return worst


def check_supported(cfg, H: int, W: int) -> None:
    """The HIP engine covers the UNet_medium family; anything else fails loudly (no fallback)."""
    bad = []
    if cfg.norm not in ("instance", "batch", "group", "none"):
        bad.append(f"norm={cfg.norm!r}")
    if cfg.downsample not in ("stride", "avgpool", "maxpool"):
        bad.append(f"downsample={cfg.downsample!r}")
    if cfg.upsample not in ("bilinear", "nearest", "transpose"):
        bad.append(f"upsample={cfg.upsample!r}")
    if not (1 <= cfg.in_channels <= 16):
        bad.append(f"in_channels={cfg.in_channels} (1..16; the K-step training and rollout plans need the reference's 6 fields)")
    if not (1 <= cfg.out_channels <= 4):
        bad.append(f"out_channels={cfg.out_channels}")
    if cfg.base_channels % 4 != 0:
        bad.append(f"base_channels={cfg.base_channels} (multiple of 4: every concatenated tensor needs a multiple of 8 channels)")
    if cfg.base_channels > 64:
        bad.append(f"base_channels={cfg.base_channels} (head kernels cover <= 64)")
    if cfg.depth < 1:
        raise ValueError("depth must be >= 1")
    div = 2 ** cfg.depth
    if H % div or W % div or H // div < 2 or W // div < 2:
        bad.append(f"grid {H}x{W} must be divisible by 2^depth with >= 2 cells left")
    if cfg.padding_mode not in rt.PAD or cfg.act not in rt.ACT or cfg.output_activation not in rt.OUTACT:
        bad.append("unknown padding_mode/act/output_activation")
    if bad:
        raise NotImplementedError("fluidnet_amd HIP engine does not support: " + "; ".join(bad))


class PlanBuilder:
    def __init__(self, cfg, B: int, H: int, W: int, dtype: str = "bf16", conv_impl: str = "auto", training: bool = True,
                 seed: int = 0):
        check_supported(cfg, H, W)
        assert dtype in ("fp32", "bf16")
        self.cfg, self.B, self.H, self.W = cfg, B, H, W
        self.dtype = rt.BF16 if dtype == "bf16" else rt.F32
        self.dtype_name = dtype
        self.conv_impl = conv_impl
        self.training = training
        self.seed = seed & 0x7FFFFFFF
        # Lanes: the weight-gradient launches of a backward pass go to side stream `wgrad_lane` (0 = keep everything on the caller's
        # stream).  Nothing on the data-gradient chain depends on them, and they are MFMA-bound where the normalisation / activation
        # adjoint kernels that follow on lane 0 are HBM- and VALU-bound, so the two overlap.  Buffers a side-lane launch reads are
        # held back from the temporary pool until the lanes are joined (emit_side_join, once per backward pass).
        self.wgrad_lane = int(os.environ.get("FNET_WGRAD_LANE", "1")) if (training and dtype == "bf16") else 0
        self._n_events = 0
        self._side_reads: set = set()
        self._side_held: List[Buf] = []
        self._side_dirty = False
        self.pm = rt.PAD[cfg.padding_mode]
        self.act = rt.ACT[cfg.act]
        self.bufs: List[Buf] = []
        self.s_slab: Optional[Buf] = None
        self._affine_pending: list = []
        self.pool: Dict[Tuple, List[Buf]] = {}
        self.shapes, self.offsets, off = param_layout(cfg)
        self.n_params_padded = off
        self.params = self.new_buf("params_flat", (off,), "f32", external=True)
        self.grads = self.new_buf("grads_flat", (off,), "f32", external=True)
        # per-block (sum, sumsq) partials of the one-pass InstanceNorm statistics: [B][chunks <= ceil(2048/B)][C][2]
        cmax = cfg.base_channels * 2 ** cfg.depth
        n_sc = B * ((2048 + B - 1) // B) * cmax * 2
        self.stats_scratch = self.new_buf("stats_scratch", (n_sc,), "f32")
        # per-wave partial results of the weight-gradient kernel (36 KB each); see wgrad_slab_parts()
        self.wgrad_slab = self.new_buf("wgrad_slab", (wgrad_slab_parts(cfg) * WG_PART,), "f32")
        # BatchNorm2d / GroupNorm ride on the InstanceNorm kernels: NORM_COMBINE pools the per-(n,c) statistics
        self.norm_kind = {"instance": 0, "batch": 1 if training else 3, "group": 2, "none": 0}[cfg.norm]
        # norm == "none" (conv + bias -> activation, unet.py:55-69) runs through the same kernels as an identity normalisation:
        # statistics (0, 1), gamma = 1, beta = the conv bias; its adjoint centres nothing (zero sums, the weighted-S form)
        self.no_norm = cfg.norm == "none"
        if self.no_norm:
            cm = cpad(cfg.base_channels * 2 ** cfg.depth)
            self.ones = self.new_buf("const/ones", (cm,), "f32")
            self.ident_stats = self.new_buf("const/identity_stats", (B * cm, 2), "f32")
            self.zero_sums = self.new_buf("const/zero_sums", (B * cm, 2), "f32")
        self.nstate_offsets, n_ns = norm_state_layout(cfg)
        # running statistics: the caller may bind its own tensor (UNet's buffers) before allocate(); else see init_norm_state()
        self.nstate = self.new_buf("norm_state", (max(n_ns, 8),), "f32")
        self.convs: "OrderedDict[str, dict]" = OrderedDict()
        for k, s in self.shapes.items():
            if len(s) == 4 and s[2] == 3:
                name = k[: -len(".weight")]
                co, ci = s[0], s[1]
                cip = 16 if name == "stem" else cpad(ci)  # the 6-channel stem input is padded to one MFMA k-step
                cop = cpad(co)
                kf = kpad(cip) if self.impl_for(cip, cop) == rt.CONV_MFMA else cip
                kd = kpad(cop) if self.impl_for(cop, cip) == rt.CONV_MFMA else cop
                if self._s2_native(name, ci, co):
                    # Conv2d(C, C, 3, stride 2) on the MFMA path: the 2x2-tap convolution over the space-to-depth view of its input
                    # (fnet_conv_mfma.hip, MODE 1 / 2): K axis 4C, four taps
                    self.convs[name] = dict(
                        Cout=co, Cin=ci, Cin_pad=cip, Cout_d=cip, Cout_pad=cop, Kf=4 * cip, Kd=kd, s2=1,
                        Wf=self.new_buf(f"Wf/{name}", (4, cop, 4 * cip), "T"),
                        Wd=self.new_buf(f"Wd/{name}", (4, 4 * cip, kd), "T"),
                        dWk=self.new_buf(f"dWk/{name}", (4, cop, 4 * cip), "f32"))
                    continue
                self.convs[name] = dict(
                    Cout=co, Cin=ci, Cin_pad=cip, Cout_d=cip, Cout_pad=cop, Kf=kf, Kd=kd,
                    Wf=self.new_buf(f"Wf/{name}", (9, cop, kf), "T"),
                    Wd=self.new_buf(f"Wd/{name}", (9, cip, kd), "T"),
                    dWk=self.new_buf(f"dWk/{name}", (9, cop, cip), "f32"))
            elif len(s) == 4 and s[2] == 2:
                # nn.ConvTranspose2d(C, C, 2, stride=2) [ci][co][2][2]: a centre-tap convolution C -> 4C (phase-major), then DEPTH2SPACE
                name = k[: -len(".weight")]
                ci, co = s[0], s[1]
                assert ci % 8 == 0 and co % 8 == 0
                rows = 4 * co
                kf = kpad(ci) if self.impl_for(ci, rows) == rt.CONV_MFMA else ci
                kd = kpad(rows) if self.impl_for(rows, ci) == rt.CONV_MFMA else rows
                self.convs[name] = dict(
                    Cout=co, Cin=ci, Cin_pad=ci, Cout_d=ci, Cout_pad=rows, Kf=kf, Kd=kd, tr=1,
                    Wf=self.new_buf(f"Wf/{name}", (9, rows, kf), "T"),
                    Wd=self.new_buf(f"Wd/{name}", (9, ci, kd), "T"),
                    dWk=self.new_buf(f"dWk/{name}", (9, rows, ci), "f32"))
        self._pack_dwk()

    def _s2_native(self, name: str, ci: int, co: int) -> bool:
        """Does the stride-2 convolution `name` (downs.L.down, unet.py:90-91) run in its 2x2 space-to-depth form?  bf16 MFMA plans, channel
        counts the kernels' phase arithmetic covers, index ranges the kernels' 24-bit pixel / 31-bit byte offsets cover."""
        if self.cfg.downsample != "stride" or not name.endswith(".down") or not name.startswith("downs.") or os.environ.get("FNET_S2", "1") != "1":
            return False
        if ci != co or ci % 32 or self.conv_impl == "ref" or self.dtype != rt.BF16:
            return False
        L = int(name.split(".")[1])
        h, w = self.H >> (L + 1), self.W >> (L + 1)
        return self.B * (h + 2) * (w + 2) < (1 << 24) and self.B * (2 * h + 2) * (2 * w + 2) * 2 * ci * 2 < (1 << 31) and w + 2 <= 500

    def _pack_dwk(self) -> None:
        """All weight-gradient accumulators live in one slab (each 32-B aligned): one ZERO per step instead of one per conv."""
        off = 0
        for c in self.convs.values():
            c["dWk"].parent = off
            off += (c["dWk"].numel + 7) // 8 * 8
        self.dwk_slab = self.new_buf("dWk_slab", (off,), "f32")
        for c in self.convs.values():
            c["dWk"].parent = (self.dwk_slab, c["dWk"].parent)

    # -------------------------------------------------------------------------------------------- buffers
    def new_buf(self, name, shape, kind, external=False) -> Buf:
        b = Buf(name, tuple(int(s) for s in shape), kind, external)
        self.bufs.append(b)
        return b

    def pbuf(self, name, H, W, C) -> Buf:
        return self.new_buf(name, (self.B, H + 2, W + 2, C), "T")

    def tmp(self, H, W, C) -> Buf:
        key = (H, W, C)
        lst = self.pool.setdefault(key, [])
        b = lst.pop() if lst else self.pbuf(f"tmp/{H}x{W}x{C}/{sum(1 for x in self.bufs if x.name.startswith('tmp/'))}", H, W, C)
        b.rc = 1
        return b

    def retain(self, b: Buf) -> None:
        if b.name.startswith("tmp/"):
            b.rc += 1

    def release(self, b: Buf) -> None:
        if not b.name.startswith("tmp/"):
            return
        b.rc -= 1
        assert b.rc >= 0, b.name
        if b.rc == 0:
            if id(b) in self._side_reads:       # a side lane may still be reading it: back to the pool at the next join
                self._side_held.append(b)
            else:
                self.pool[(b.shape[1] - 2, b.shape[2] - 2, b.shape[3])].append(b)

    def new_event(self) -> int:
        self._n_events += 1
        return self._n_events - 1

    def emit_lane_edge(self, ops, src_lane: int, dst_lane: int) -> None:
        """Everything issued so far on src_lane happens before whatever dst_lane is given next."""
        ev = self.new_event()
        self.op(ops, "EVENT_RECORD", [ev], lane=src_lane)
        self.op(ops, "EVENT_WAIT", [ev], lane=dst_lane)

    def emit_side_join(self, ops) -> None:
        """Lane 0 waits for the side lane; the temporaries it was reading return to the pool."""
        if self._side_dirty:
            self.emit_lane_edge(ops, self.wgrad_lane, 0)
            self._side_dirty = False
        for b in self._side_held:
            self._side_reads.discard(id(b))
            self.pool[(b.shape[1] - 2, b.shape[2] - 2, b.shape[3])].append(b)
        self._side_held = []

    def s_alloc(self, n: int) -> Ref:
        """n floats of the S slab (per-(n,c) backward sums of the InstanceNorm adjoint)."""
        if self.s_slab is None:
            self.s_slab = self.new_buf(f"S_slab/{len(self.bufs)}", (0,), "f32")
        off = self.s_slab.shape[0]
        self.s_slab.shape = (off + (n + 63) // 64 * 64,)
        return Ref(self.s_slab, off)

    def init_norm_state(self) -> None:
        """Fresh BatchNorm2d running statistics (mean 0, variance 1) in a norm-state buffer this plan allocated itself."""
        self.nstate.tensor.zero_()
        for k, o in self.nstate_offsets.items():
            if k.endswith("running_var"):
                c = self.shapes[k[: -len("running_var")] + "weight"][0]
                self.nstate.tensor[o:o + c] = 1.0

    def nref(self, name: str) -> Ref:
        return Ref(self.nstate, self.nstate_offsets[name])

    def e_norm_combine(self, ops, norm: str, st: Buf, C: int, HW: int) -> None:
        k = self.norm_kind
        if not k:
            return
        run = k in (1, 3)
        creal = self.shapes[norm + ".weight"][0]          # groups are formed over the real channels; zero padding keeps its own statistics
        self.op(ops, "NORM_COMBINE", [self.B, C, k, group_size(self.cfg.group_norm_groups, creal) if k == 2 else 1, HW, creal], f=[1e-5, 0.1],
                p=[Ref(st), self.nref(norm + ".running_mean") if run else None, self.nref(norm + ".running_var") if run else None],
                tag=norm)

    def gamma_ref(self, u: "Unit") -> Ref:
        return Ref(self.ones) if self.no_norm else self.pref(u.norm + ".weight")

    def beta_ref(self, u: "Unit") -> Ref:
        return self.pref(u.conv + ".bias") if self.no_norm else self.pref(u.norm + ".bias")

    def pref(self, name: str) -> Ref:
        return Ref(self.params, self.offsets[name])

    def gref(self, name: str) -> Ref:
        return Ref(self.grads, self.offsets[name])

    def impl_for(self, cin: int, cout: int) -> int:
        if self.conv_impl == "ref" or self.dtype != rt.BF16:
            return rt.CONV_REF
        ok = cin % 8 == 0 and cout % 8 == 0
        if self.conv_impl == "mfma" and not ok:
            raise NotImplementedError(f"MFMA conv needs channel counts that are multiples of 8 (got {cin}->{cout})")
        return rt.CONV_MFMA if ok else rt.CONV_REF

    # -------------------------------------------------------------------------------------------- op emitters
    def op(self, ops, code, i, f=(), p=(), tag="", lane: int = 0) -> None:
        ops.append(OpRec(OPC[code], [int(v) for v in i], [float(v) for v in f], list(p), self.dtype, tag, lane=lane))

    def e_pack(self, ops, planes, dst: Buf):
        """planes: six (Ref, batch stride) field planes (the rollout's channels), or ("nchw", Ref, n): one contiguous [B,n,H,W] tensor
        (the nn.Module path, any in_channels <= 16)"""
        H, W = dst.shape[1] - 2, dst.shape[2] - 2
        if planes and planes[0] == "nchw":
            _, ref, n = planes
            assert n == self.cfg.in_channels
            self.op(ops, "PACK_INPUT", [self.B, H, W, dst.shape[3], self.pm, n * H * W] + [0] * 9 + [n], p=[ref] + [None] * 5 + [Ref(dst)], tag=dst.name)
            return
        if len(planes) != 6 or self.cfg.in_channels != 6:
            raise NotImplementedError("the K-step training and rollout plans feed the six fields of the reference's input (in_channels = 6)")
        self.op(ops, "PACK_INPUT", [self.B, H, W, dst.shape[3], self.pm] + [s for _, s in planes],
                p=[r for r, _ in planes] + [Ref(dst)], tag=dst.name)

    def conv_flops(self, name: str, H: int, W: int) -> float:
        """2*MAC of the real convolution `name` evaluated on an HxW input grid (real channels; stride-2 convs count
        their H/2 x W/2 outputs) -- SURVEY.md section 8d's convention."""
        c = self.convs.get(name.split("/")[0])
        if c is None:                       # not a layer of the network (unit tests drive e_conv directly)
            return 0.0
        if c.get("tr"):                     # transposed 2x2: every input pixel feeds 4 output pixels, one tap each
            return 2.0 * self.B * H * W * 4 * c["Cin"] * c["Cout"]
        fl = 2.0 * self.B * H * W * 9 * c["Cin"] * c["Cout"]
        return fl / 4 if name.split("/")[0].endswith(".down") else fl

    def e_conv(self, ops, src: PV, wbuf: Buf, bias: Optional[Ref], dst: PV, cin: int, cout: int, tag=""):
        npix = self.B * (src.H + 2) * (src.W + 2)
        assert (dst.H, dst.W) == (src.H, src.W) and src.C == cin and dst.C == cout, (tag, src.C, cin, dst.C, cout)
        self.op(ops, "CONV3X3", [npix, src.W + 2, cin, src.ldc, src.coff, cout, dst.ldc, dst.coff, self.impl_for(cin, cout), 0],
                p=[src.ref, Ref(wbuf), bias, dst.ref, None], tag=tag)
        ops[-1].flops = self.conv_flops(tag, src.H, src.W)

    def e_conv_s2(self, ops, src: PV, wbuf: Buf, bias: Optional[Ref], dst: PV, mode: int, tag=""):
        """Stride-2 convolution in its 2x2 space-to-depth form.  mode 1 (forward): src = full-resolution input (C channels), dst = half
        resolution; mode 2 (data gradient): src = half-resolution gradient, dst = full-resolution folded gradient (C channels)."""
        full, half = (src, dst) if mode == 1 else (dst, src)
        assert (full.H, full.W) == (2 * half.H, 2 * half.W), tag
        C = full.C
        npix = self.B * (half.H + 2) * (half.W + 2)
        cin, cout = (4 * C, dst.C) if mode == 1 else (src.C, 4 * C)
        self.op(ops, "CONV3X3", [npix, half.W + 2, cin, src.ldc, src.coff, cout, dst.ldc, dst.coff, rt.CONV_MFMA, mode, half.H, C],
                p=[src.ref, Ref(wbuf), bias, dst.ref, None], tag=tag)
        ops[-1].flops = self.conv_flops(tag, full.H, full.W)

    def e_wgrad(self, ops, src: PV, dout: PV, conv: str):
        c = self.convs[conv]
        s2 = c.get("s2", 0)
        npix = self.B * (dout.H + 2) * (dout.W + 2)
        assert src.C == c["Cin_pad"] and dout.C == c["Cout_pad"], (conv, src.C, dout.C)
        assert (src.H, src.W) == ((2 * dout.H, 2 * dout.W) if s2 else (dout.H, dout.W)), conv
        lane = self.wgrad_lane
        if lane:
            # all weight gradients share one lane: the partial slab and the dWk accumulators are touched in stream order there
            self.emit_lane_edge(ops, 0, lane)
            self._side_dirty = True
            for b in (src.buf, dout.buf):
                if b.name.startswith("tmp/"):
                    self._side_reads.add(id(b))
        kc = 4 * src.C if s2 else src.C         # the GEMM's K-side channel count
        self.op(ops, "WGRAD", [npix, dout.W + 2, kc, src.ldc, src.coff, dout.C, dout.ldc, dout.coff,
                               self.impl_for(src.C, dout.C), 4 * c["Cin_pad"] if s2 else c["Cin_pad"], self.B, dout.H,
                               self.wgrad_slab.numel // WG_PART, s2, src.C if s2 else 0],
                p=[src.ref, dout.ref, Ref(c["dWk"]), Ref(self.wgrad_slab)], tag=conv, lane=lane)
        ops[-1].flops = self.conv_flops(conv, src.H, src.W)

    def e_stats(self, ops, raw: PV, st: Buf):
        sc = self.stats_scratch
        self.op(ops, "IN_STATS", [self.B, raw.H, raw.W, raw.C, raw.ldc, raw.coff, sc.numel], f=[1e-5],
                p=[raw.ref, Ref(st), Ref(sc)])

    def e_apply(self, ops, u: Unit):
        raw = PV(u.raw, u.out.C)
        res = u.res
        self.op(ops, "APPLY", [self.B, raw.H, raw.W, raw.C, raw.ldc, raw.coff, res.ldc if res else 8, res.coff if res else 0,
                               u.out.ldc, u.out.coff, self.act, self.pm, u.drop_site, self.seed],
                f=[u.drop_p],
                p=[raw.ref, Ref(u.st), self.gamma_ref(u), self.beta_ref(u), res.ref if res else None, u.out.ref],
                tag=u.conv)

    # -------------------------------------------------------------------------------------------- forward
    def _unit(self, ops, tape_name: str, conv: str, norm: str, xin: PV, cout: int, out: Optional[PV], res: Optional[PV],
              drop_site: int, drop_p: float) -> Unit:
        H, W = xin.H, xin.W
        c = self.convs[conv]
        raw = self.pbuf(f"{tape_name}/{conv}/raw", H, W, cout)
        st = self.ident_stats if self.no_norm else self.new_buf(f"{tape_name}/{conv}/stats", (self.B, cout, 2), "f32")
        if out is None:
            out = PV(self.pbuf(f"{tape_name}/{conv}/act", H, W, cout), cout)
        u = Unit(conv, norm, xin, raw, st, out, res, drop_site, drop_p)
        self.e_conv(ops, xin, c["Wf"], None, PV(raw, cout), c["Cin_pad"], cout, tag=conv)
        if self.norm_kind != 3 and not self.no_norm:   # eval-mode BatchNorm2d normalises with the running statistics alone
            self.e_stats(ops, PV(raw, cout), st)
        self.e_norm_combine(ops, norm, st, cout, H * W)
        self.e_apply(ops, u)
        return u

    def _conv_block(self, ops, tn, key, xin: PV, cout, out: Optional[PV], residual: bool, site: int) -> List[Unit]:
        p = self.cfg.dropout if self.training else 0.0
        u1 = self._unit(ops, tn, f"{key}.c1", f"{key}.n1", xin, cout, None, None, site, p)
        u2 = self._unit(ops, tn, f"{key}.c2", f"{key}.n2", u1.out, cout, out, xin if residual else None, 0, 0.0)
        return [u1, u2]

    def emit_forward(self, ops, app: int, planes: Sequence[Tuple[Ref, int]], pred: Optional[Buf] = None) -> Tape:
        cfg, B, H, W = self.cfg, self.B, self.H, self.W
        tn = f"tape{app}"
        t = Tape(app)
        site = app * 64
        t.in_p = self.pbuf(f"{tn}/in_p", H, W, 16)
        self.e_pack(ops, planes, t.in_p)
        base = cpad(cfg.base_channels)
        s_p = self.pbuf(f"{tn}/stem", H, W, base)
        t.stem = PlainConv("stem", PV(t.in_p, 16), PV(s_p, base), 1)
        self.e_conv(ops, t.stem.xin, self.convs["stem"]["Wf"], self.pref("stem.bias"), t.stem.out, 16, base, tag="stem")
        self.op(ops, "PAD_RING", [B, H, W, base, base, 0, self.pm], p=[Ref(s_p)])
        x = t.stem.out
        h, w, ch = H, W, cfg.base_channels
        for L in range(cfg.depth):
            C = ch * 2                          # a multiple of 8 (base % 4 == 0): only base-channel tensors are ever padded
            cat = self.pbuf(f"{tn}/cat{L}", h, w, 2 * C)
            skip = PV(cat, C, C)
            d = dict(cat=cat, C=C)
            if cfg.use_residual:
                d["block"] = self._conv_block(ops, tn, f"downs.{L}.block", x, C, None, False, site); site += 1
                d["res"] = self._conv_block(ops, tn, f"downs.{L}.res.b", d["block"][1].out, C, skip, True, site); site += 1
            else:
                d["block"] = self._conv_block(ops, tn, f"downs.{L}.block", x, C, skip, False, site); site += 1
                d["res"] = None
            dn = self.pbuf(f"{tn}/dn{L}", h // 2, w // 2, C)
            if cfg.downsample == "stride" and self.convs[f"downs.{L}.down"].get("s2"):
                name = f"downs.{L}.down"
                self.e_conv_s2(ops, skip, self.convs[name]["Wf"], self.pref(name + ".bias"), PV(dn, C), 1, tag=name)
                self.op(ops, "PAD_RING", [B, h // 2, w // 2, C, C, 0, self.pm], p=[Ref(dn)])
                d["down"] = PlainConv(name, skip, PV(dn, C), 2)
            elif cfg.downsample == "stride":
                full = self.tmp(h, w, C)
                name = f"downs.{L}.down"
                self.e_conv(ops, skip, self.convs[name]["Wf"], None, PV(full, C), C, C, tag=name)
                self.op(ops, "SUBSAMPLE_BIAS", [B, h, w, C, C, 0, C, 0, self.pm], p=[Ref(full), self.pref(name + ".bias"), Ref(dn)], tag=name)
                self.release(full)
                d["down"] = PlainConv(name, skip, PV(dn, C), 2)
            else:
                # Sequential(pool, conv3x3 + bias) (unet.py:93-94): the pooled tensor is kept, the conv's wgrad reads it
                name = f"downs.{L}.down.1"
                pooled = self.pbuf(f"{tn}/pool{L}", h // 2, w // 2, C)
                self.op(ops, "POOL2X", [B, h // 2, w // 2, C, skip.ldc, skip.coff, C, 0, 1 if cfg.downsample == "maxpool" else 0, self.pm],
                        p=[skip.ref, Ref(pooled)], tag=f"downs.{L}.down.0")
                self.e_conv(ops, PV(pooled, C), self.convs[name]["Wf"], self.pref(name + ".bias"), PV(dn, C), C, C, tag=name)
                self.op(ops, "PAD_RING", [B, h // 2, w // 2, C, C, 0, self.pm], p=[Ref(dn)])
                d["down"] = PlainConv(name, PV(pooled, C), PV(dn, C), 1)
                d["pool_in"] = skip
            t.downs.append(d)
            x = d["down"].out
            h, w, ch = h // 2, w // 2, C
        for j in range(max(0, cfg.bottleneck_blocks)):
            t.mids.append(self._conv_block(ops, tn, f"mid.{j}.b", x, ch, None, True, site)); site += 1
            x = t.mids[-1][1].out
        for i, L in enumerate(reversed(range(cfg.depth))):
            d = t.downs[L]
            C = d["C"]
            assert ch == C
            h, w = h * 2, w * 2
            if cfg.upsample == "transpose":
                name = f"ups.{i}.up"
                t4 = self.tmp(h // 2, w // 2, 4 * C)
                self.e_conv(ops, x, self.convs[name]["Wf"], None, PV(t4, 4 * C), C, 4 * C, tag=name)
                self.op(ops, "DEPTH2SPACE", [B, h // 2, w // 2, C, 4 * C, 0, 2 * C, 0, self.pm],
                        p=[Ref(t4), self.pref(name + ".bias"), Ref(d["cat"])], tag=name)
                self.release(t4)
            else:
                self.op(ops, "UPSAMPLE2X", [B, h // 2, w // 2, C, x.ldc, x.coff, 2 * C, 0, 1 if cfg.upsample == "bilinear" else 0, self.pm],
                        p=[x.ref, Ref(d["cat"])], tag=f"ups.{i}.up")
            u = dict(cat=d["cat"], C=C, xlow=x)
            u["block"] = self._conv_block(ops, tn, f"ups.{i}.block", PV(d["cat"], 2 * C, 0), cpad(C // 2), None, False, site); site += 1
            x = u["block"][1].out
            if cfg.use_residual:
                u["res"] = self._conv_block(ops, tn, f"ups.{i}.res.b", x, cpad(C // 2), None, True, site); site += 1
                x = u["res"][1].out
            else:
                u["res"] = None
            t.ups.append(u)
            ch = C // 2
        t.head_in = x
        co = cfg.out_channels
        t.z = self.new_buf(f"{tn}/z", (B, co, H, W), "f32")
        t.pred = pred if pred is not None else self.new_buf(f"{tn}/pred", (B, co, H, W), "f32")
        self.op(ops, "HEAD_FWD", [B, H, W, ch, x.ldc, x.coff, co, rt.OUTACT[cfg.output_activation]],
                p=[x.ref, self.pref("head.weight"), self.pref("head.bias"), Ref(t.z), Ref(t.pred)], tag="head")
        return t

    # -------------------------------------------------------------------------------------------- backward
    def _unit_bwd(self, ops, u: Unit, srcs: List[GradSrc], need_dx: bool = True, need_G: bool = False) -> Tuple[Optional[GradSrc], Optional[Buf]]:
        """Backward of one unit.  Consumes (releases) the temporaries behind `srcs`.  Returns (gradient wrt the unit's
        input as a folded source or None, the materialised total gradient G wrt the unit's output or None (caller releases)).
        G is materialised when the caller needs it (need_G: it is also the gradient of a residual branch) or when there are two
        gradient sources to add."""
        assert 1 <= len(srcs) <= 2
        B, H, W, C = self.B, u.out.H, u.out.W, u.out.C
        S = self.s_alloc(B * C * 2)         # float2[B][C] in the S slab (zeroed once per backward pass: emit_backward)
        g1 = srcs[0]
        g2 = srcs[1] if len(srcs) > 1 else None
        # One gradient source: pass 1 writes no G (a third of its traffic); pass 2 reads the source itself and folds its ring the
        # same way.  Two sources: their folded sum is materialised once as G.
        if os.environ.get("FNET_KEEP_G", "0") == "1":     # A/B switch: always materialise G (the two-source form everywhere)
            need_G = True
        G = self.tmp(H, W, C) if (g2 is not None or need_G) else None
        self.op(ops, "INACT_BWD_REDUCE",
                [B, H, W, C, g1.pv.ldc, g1.pv.coff, g2.pv.ldc if g2 else 8, g2.pv.coff if g2 else 0, C, 0, C, 0,
                 g1.fold, g2.fold if g2 else 0, self.act, self.pm, u.drop_site, self.seed],
                f=[u.drop_p],
                p=[g1.pv.ref, g2.pv.ref if g2 else None, Ref(u.raw), Ref(u.st), self.gamma_ref(u),
                   self.beta_ref(u), Ref(G) if G is not None else None, S], tag=u.conv)
        draw = self.tmp(H, W, C)
        if G is not None:
            gp, gl, gc, gf = Ref(G), C, 0, 0
        else:
            gp, gl, gc, gf = g1.pv.ref, g1.pv.ldc, g1.pv.coff, g1.fold
        Sa = Ref(self.zero_sums) if self.no_norm else S
        if self.norm_kind:
            # BatchNorm2d / GroupNorm: centre with the gamma-weighted mean of S over the normalisation group (S stays for dgamma/dbeta)
            Sa = self.s_alloc(B * C * 2)
            creal = self.shapes[u.norm + ".weight"][0]
            self.op(ops, "NORM_COMBINE_BWD", [B, C, self.norm_kind, group_size(self.cfg.group_norm_groups, creal) if self.norm_kind == 2 else 1, creal],
                    p=[S, self.pref(u.norm + ".weight"), Sa], tag=u.norm)
        self.op(ops, "INACT_BWD_APPLY", [B, H, W, C, gl, gc, C, 0, C, 0, self.act, u.drop_site, self.seed, gf, self.pm,
                                         1 if (self.norm_kind or self.no_norm) else 0], f=[u.drop_p],
                p=[gp, Ref(u.raw), Ref(u.st), self.gamma_ref(u), self.beta_ref(u), Sa, Ref(draw)],
                tag=u.conv)
        for s in srcs:                      # only now: in the single-source form pass 2 still reads srcs[0]
            self.release(s.pv.buf)
        # dgamma/dbeta: a separate tiny kernel (adding them from every block of the reduce kernel, hundreds of adders per
        # address, cost more), batched over the layers of the backward pass by emit_backward (flush_affine)
        if self.no_norm:                   # only the bias has a gradient: sum of dz (dgamma slot -1 = skipped)
            self._affine_pending.append((S, C, None, self.gref(u.conv + ".bias"), u.conv))
        else:
            self._affine_pending.append((S, C, self.gref(u.norm + ".weight"), self.gref(u.norm + ".bias"), u.norm))
        self.e_wgrad(ops, u.xin, PV(draw, C), u.conv)
        dx = None
        if need_dx:
            cin = u.xin.C
            dpad = self.tmp(H, W, cin)
            self.e_conv(ops, PV(draw, C), self.convs[u.conv]["Wd"], None, PV(dpad, cin), C, cin, tag=u.conv + "/dgrad")
            dx = GradSrc(PV(dpad, cin), 1)
        self.release(draw)
        return dx, G

    def _block_bwd(self, ops, units: List[Unit], srcs: List[GradSrc], residual: bool, need_dx: bool = True) -> List[GradSrc]:
        """Backward of a ConvBlock (residual=False) or ResBlock y = x + B(x) (residual=True); returns sources for x."""
        u1, u2 = units
        # y = x + B(x): the gradient of y also reaches x directly.  With ONE incoming source that gradient is the source itself (read
        # again, with its fold flag, by whoever consumes x's gradient): nothing is copied.  Two sources are summed once (G).
        alias = None
        if residual and len(srcs) == 1 and os.environ.get("FNET_KEEP_G", "0") != "1":
            alias = GradSrc(srcs[0].pv, srcs[0].fold)
            self.retain(alias.pv.buf)
        g_a1, Gy = self._unit_bwd(ops, u2, srcs, need_G=residual and alias is None)
        if not residual and Gy is not None:
            self.release(Gy)
        g_x, G1 = self._unit_bwd(ops, u1, [g_a1], need_dx=need_dx)
        if G1 is not None:
            self.release(G1)
        out = []
        if g_x is not None:
            out.append(g_x)
        if residual:
            out.append(alias if alias is not None else GradSrc(PV(Gy, u2.out.C), 0))
        return out

    def _plain_conv_bwd(self, ops, pc: PlainConv, srcs: List[GradSrc], need_dx: bool) -> Optional[GradSrc]:
        """Backward of conv+bias (stem: stride 1; down: stride 2 via zero-insert).  Consumes srcs."""
        B = self.B
        C = pc.out.C
        H, W = pc.out.H, pc.out.W
        d = self.tmp(H, W, C)
        g1 = srcs[0]
        g2 = srcs[1] if len(srcs) > 1 else None
        self.op(ops, "FOLD_SUM", [B, H, W, C, g1.pv.ldc, g1.pv.coff, g2.pv.ldc if g2 else 8, g2.pv.coff if g2 else 0, C, 0,
                                  g1.fold, g2.fold if g2 else 0, self.pm],
                p=[g1.pv.ref, g2.pv.ref if g2 else None, Ref(d)], tag=pc.conv)
        for s in srcs:
            self.release(s.pv.buf)
        # (r2: on the weight-gradient lane this launch and the loss sums cost 0.13 ms per step more than they hide; they stay on lane 0)
        self.op(ops, "BIAS_GRAD", [B, H, W, C, C, 0], p=[Ref(d), self.gref(pc.conv + ".bias")], tag=pc.conv)
        s2 = self.convs[pc.conv].get("s2", 0)
        if pc.stride == 2 and not s2:
            full = self.tmp(2 * H, 2 * W, C)
            self.op(ops, "ZERO_INSERT", [B, H, W, C, C, 0, C, 0], p=[Ref(d), Ref(full)], tag=pc.conv)
            self.release(d)
            d = full
        self.e_wgrad(ops, pc.xin, PV(d, C), pc.conv)
        dx = None
        if need_dx:
            cin = pc.xin.C
            dpad = self.tmp(pc.xin.H, pc.xin.W, cin)
            if s2:
                self.e_conv_s2(ops, PV(d, C), self.convs[pc.conv]["Wd"], None, PV(dpad, cin), 2, tag=pc.conv + "/dgrad")
            else:
                self.e_conv(ops, PV(d, C), self.convs[pc.conv]["Wd"], None, PV(dpad, cin), C, cin, tag=pc.conv + "/dgrad")
            dx = GradSrc(PV(dpad, cin), 1)
        self.release(d)
        return dx

    def flush_affine(self, ops) -> None:
        """AFFINE_GRAD of every unit visited since the last flush, 6 per launch (their S sums all live in the current slab)."""
        pend, self._affine_pending = self._affine_pending, []
        for k in range(0, len(pend), 6):
            grp = pend[k:k + 6]
            iv = [self.B, len(grp)]
            for (S, C, gw, gb, _) in grp:
                assert S.buf is self.s_slab and (gw is None or gw.buf is self.grads) and gb.buf is self.grads
                iv += [S.off, C, gw.off if gw is not None else -1, gb.off]
            self.op(ops, "AFFINE_GRAD_MULTI", iv, p=[Ref(self.s_slab), Ref(self.grads)], tag="+".join(g[4] for g in grp))

    def emit_backward(self, ops, t: Tape, dpred: Buf, dx_out: Optional[Buf] = None, dx_channels: int = 6, bucket_cb=None,
                      pre_head=None) -> None:
        """Adjoint of emit_forward.  Accumulates parameter gradients (dWk scratch for 3x3 weights, grads_flat for the
        rest); writes d(input) [B,dx_channels,H,W] fp32 into dx_out when given.
        bucket_cb(ops, prefixes): called when every layer whose name starts with one of `prefixes` has emitted its last gradient
        op of THIS pass, decoder first (head + ups.*, mid.*, the deepest encoder level); the remaining layers (outer encoder levels,
        stem) finish with the pass itself.  The data-parallel step starts a bucket's all-reduce there (training/dp.py).
        pre_head(ops): emits the producer of dpred immediately before HEAD_BWD."""
        cfg, B, H, W = self.cfg, self.B, self.H, self.W
        zero_at = len(ops)
        self.op(ops, "NOP", [], tag="S_slab")     # placeholder: becomes the ZERO of this pass's S slab once its size is known
        self.s_slab = None                   # a fresh slab per backward pass: buffers already allocated never change size
        s_begin = 0
        x = t.head_in
        G = self.tmp(H, W, x.C)
        if pre_head is not None:
            pre_head(ops)                    # the op that writes dpred (LOSS_GRAD), directly in front of its consumer
        self.op(ops, "HEAD_BWD", [B, H, W, self.shapes["head.weight"][1], x.ldc, x.coff, x.C, 0, cfg.out_channels,
                                  rt.OUTACT[cfg.output_activation], x.C],
                p=[Ref(dpred), Ref(t.z), Ref(t.pred), x.ref, self.pref("head.weight"), Ref(G), self.gref("head.weight"),
                   self.gref("head.bias")], tag="head")
        srcs = [GradSrc(PV(G, x.C), 0)]
        dcat: Dict[int, Buf] = {}
        for i in reversed(range(cfg.depth)):
            u = t.ups[i]
            L = cfg.depth - 1 - i
            C = u["C"]
            if u["res"] is not None:
                srcs = self._block_bwd(ops, u["res"], srcs, True)
            srcs = self._block_bwd(ops, u["block"], srcs, False)
            (gcat,) = srcs
            dcat[L] = gcat.pv.buf                       # [.., 2C]: [0:C] -> upsample adjoint now, [C:2C] -> skip, later
            self.retain(dcat[L])
            h, w = u["xlow"].H, u["xlow"].W
            if cfg.upsample == "transpose":
                name = f"ups.{i}.up"
                d4 = self.tmp(h, w, 4 * C)
                self.op(ops, "SPACE2DEPTH", [B, h, w, C, 2 * C, 0, 4 * C, 0, 1, self.pm], p=[Ref(dcat[L]), Ref(d4)], tag=name)
                self.release(dcat[L])
                for q in range(4):              # the bias reaches every output pixel: all four phases
                    self.op(ops, "BIAS_GRAD", [B, h, w, C, 4 * C, q * C], p=[Ref(d4), self.gref(name + ".bias")], tag=name)
                self.e_wgrad(ops, u["xlow"], PV(d4, 4 * C), name)
                glow = self.tmp(h, w, C)
                self.e_conv(ops, PV(d4, 4 * C), self.convs[name]["Wd"], None, PV(glow, C), 4 * C, C, tag=name + "/dgrad")
                self.release(d4)
            else:
                glow = self.tmp(h, w, C)
                self.op(ops, "UPSAMPLE2X_BWD", [B, h, w, C, 2 * C, 0, C, 0, 1, 1 if cfg.upsample == "bilinear" else 0, self.pm],
                        p=[Ref(dcat[L]), Ref(glow)], tag=f"ups.{i}.up")
                self.release(dcat[L])
            srcs = [GradSrc(PV(glow, C), 0)]
        if bucket_cb is not None:
            bucket_cb(ops, ("head.", "ups."))
        for j in reversed(range(len(t.mids))):
            srcs = self._block_bwd(ops, t.mids[j], srcs, True)
        if bucket_cb is not None and t.mids:
            bucket_cb(ops, ("mid.",))
        for L in reversed(range(cfg.depth)):
            d = t.downs[L]
            C = d["C"]
            g = self._plain_conv_bwd(ops, d["down"], srcs, True)
            if "pool_in" in d:
                xs = d["pool_in"]
                mx = cfg.downsample == "maxpool"
                gfull = self.tmp(xs.H, xs.W, C)
                self.op(ops, "POOL2X_BWD", [B, xs.H // 2, xs.W // 2, C, g.pv.ldc, g.pv.coff, xs.ldc, xs.coff, C, 0, g.fold, 1 if mx else 0, self.pm],
                        p=[g.pv.ref, xs.ref if mx else None, Ref(gfull)], tag=f"downs.{L}.down.0")
                self.release(g.pv.buf)
                g = GradSrc(PV(gfull, C), 0)
            srcs = [g, GradSrc(PV(dcat[L], C, C), 1)]
            if d["res"] is not None:
                srcs = self._block_bwd(ops, d["res"], srcs, True)
            srcs = self._block_bwd(ops, d["block"], srcs, False)
            if bucket_cb is not None and L == cfg.depth - 1 and cfg.depth >= 2:
                bucket_cb(ops, (f"downs.{L}.",))
        need_dx = dx_out is not None
        g = self._plain_conv_bwd(ops, t.stem, srcs, need_dx)
        if need_dx:
            self.op(ops, "UNPACK_GRAD", [B, H, W, 16, dx_channels, self.pm], p=[g.pv.ref, Ref(dx_out)], tag="dx")
            self.release(g.pv.buf)
        self.flush_affine(ops)
        self.emit_side_join(ops)
        if self.s_slab is not None and self.s_slab.shape[0] > s_begin:
            nbytes = (self.s_slab.shape[0] - s_begin) * 4
            ops[zero_at] = OpRec(OPC["ZERO"], [nbytes & 0x7FFFFFFF, nbytes >> 31], [], [Ref(self.s_slab, s_begin)], self.dtype, "S_slab")

    # -------------------------------------------------------------------------------------------- weights / optimiser
    def emit_prep_weights(self, ops) -> None:
        for name, c in self.convs.items():
            if c.get("s2"):
                self.op(ops, "PREP_WEIGHTS", [c["Cout"], c["Cin"], 4 * c["Cin"], 4 * c["Cin"], 3, c["Cout_pad"], c["Kd"], 2],
                        p=[self.pref(name + ".weight"), Ref(c["Wf"]), Ref(c["Wd"])], tag=name)
                continue
            blocked = (1 if self.impl_for(c["Cin_pad"], c["Cout_pad"]) == rt.CONV_MFMA else 0) | \
                      (2 if self.impl_for(c["Cout_pad"], c["Cout_d"]) == rt.CONV_MFMA else 0)
            self.op(ops, "PREP_WEIGHTS", [c["Cout"], c["Cin"], c["Kf"], c["Cout_d"], blocked, c["Cout_pad"], c["Kd"], c.get("tr", 0)],
                    p=[self.pref(name + ".weight"), Ref(c["Wf"]), Ref(c["Wd"])], tag=name)

    def emit_zero(self, ops, buf: Buf) -> None:
        esz = {"f32": 4, "f64": 8}.get(buf.kind, 2 if self.dtype == rt.BF16 else 4)
        nbytes = buf.numel * esz
        self.op(ops, "ZERO", [nbytes & 0x7FFFFFFF, nbytes >> 31], p=[Ref(buf)], tag=buf.name)

    def emit_zero_grads(self, ops) -> None:
        self.emit_zero(ops, self.grads)
        self.emit_zero(ops, self.dwk_slab)

    def emit_unprep(self, ops, prefixes=None, skip=(), lane: int = 0, accumulate: bool = False) -> List[str]:
        """dWk (tap-major accumulators) -> OIHW gradients in grads_flat, for the convolutions whose name starts with one of `prefixes`
        (default: all) and is not in `skip`; returns the names emitted."""
        done = []
        for name, c in self.convs.items():
            if name in skip or (prefixes is not None and not name.startswith(tuple(prefixes))):
                continue
            self.op(ops, "UNPREP_WGRAD", [c["Cout"], c["Cin"], 4 * c["Cin_pad"] if c.get("s2") else c["Cin_pad"], 1 if accumulate else 0, c["Cout_pad"],
                                          2 if c.get("s2") else c.get("tr", 0)],
                    p=[Ref(c["dWk"]), self.gref(name + ".weight")], tag=name, lane=lane)
            done.append(name)
        return done

    def param_range(self, prefixes) -> Tuple[int, int]:
        """[lo, hi) in floats of the flat parameter / gradient buffer covered by the tensors whose name starts with one of `prefixes`
        (they are contiguous: state-dict order is stem, downs.*, mid.*, ups.*, head)."""
        names = list(self.offsets)
        idx = [i for i, k in enumerate(names) if k.startswith(tuple(prefixes))]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), prefixes
        hi = self.offsets[names[idx[-1] + 1]] if idx[-1] + 1 < len(names) else self.n_params_padded
        return self.offsets[names[idx[0]]], hi

    # -------------------------------------------------------------------------------------------- loss
    def loss_ops_args(self, lc, tgt: Tuple[Ref, int], in0: Tuple[Ref, int], emit: Tuple[Ref, int], coll: Tuple[Ref, int]):
        flags = (1 if (lc.enable_divergence and lc.divergence_weight > 0) else 0) | \
                (2 if (lc.enable_gradient and lc.gradient_weight > 0) else 0) | \
                (4 if (lc.enable_emitter and lc.emitter_weight > 0) else 0)
        i = [self.B, self.H, self.W, tgt[1], in0[1], emit[1], coll[1], rt.PAD[lc.padding_mode], rt.STENCIL[lc.stencil_mode], flags]
        return i, flags

    def emit_loss_fwd(self, ops, lc, pred: Buf, tgt, in0, emit, coll, sums: Buf, terms: Buf, accum: Optional[Buf], wk: float,
                      gcnt: Optional[Ref] = None, world: int = 1) -> None:
        i, flags = self.loss_ops_args(lc, tgt, in0, emit, coll)
        self.emit_zero(ops, sums)
        self.op(ops, "LOSS_REDUCE", i, f=[lc.grid_spacing], p=[Ref(pred), tgt[0], in0[0], emit[0], coll[0], Ref(sums)], tag="loss")
        self.op(ops, "LOSS_FINALIZE", [self.B, self.H, self.W, flags, world],
                f=[lc.mse_weight, lc.divergence_weight, lc.gradient_weight, lc.emitter_weight, wk],
                p=[Ref(sums), gcnt, Ref(terms), Ref(accum) if accum is not None else None], tag="loss")

    def emit_loss_grad(self, ops, lc, pred: Buf, tgt, in0, emit, coll, sums: Buf, dnext: Optional[Tuple[Ref, int]],
                       gscale: Optional[Ref], dpred: Buf, wk: float, gcnt: Optional[Ref] = None, world: int = 1) -> None:
        i, flags = self.loss_ops_args(lc, tgt, in0, emit, coll)
        self.op(ops, "LOSS_GRAD", i + [world, dnext[1] if dnext else 0],
                f=[lc.grid_spacing, lc.mse_weight, lc.divergence_weight, lc.gradient_weight, lc.emitter_weight, wk],
                p=[Ref(pred), tgt[0], in0[0], emit[0], coll[0], Ref(sums), gcnt, dnext[0] if dnext else None, gscale, Ref(dpred)],
                tag="loss")


# ------------------------------------------------------------------------------------------------ binding
def _torch_dtype(kind: str, plan_dtype: int):
    import torch
    if kind == "T":
        return torch.bfloat16 if plan_dtype == rt.BF16 else torch.float32
    return {"f32": torch.float32, "f64": torch.float64}[kind]


def allocate(pb: PlanBuilder, device) -> int:
    """Allocate every non-external buffer of the builder on `device` (zero-filled).  Returns bytes allocated."""
    import torch
    total = 0
    for b in pb.bufs:
        if b.external or b.tensor is not None or b.parent is not None:
            continue
        b.tensor = torch.zeros(b.shape, dtype=_torch_dtype(b.kind, pb.dtype), device=device)
        total += b.tensor.numel() * b.tensor.element_size()
        if b is pb.nstate:
            pb.init_norm_state()
    for b in pb.bufs:                      # slices of other buffers (after their parents exist)
        if b.parent is not None and b.tensor is None:
            par, off = b.parent
            b.tensor = par.tensor[off:off + b.numel].view(b.shape)
        if pb.no_norm and b is pb.ones:
            b.tensor.fill_(1.0)
        if pb.no_norm and b is pb.ident_stats:
            b.tensor[:, 1] = 1.0
    return total


def lower(pb: PlanBuilder, recs: List[OpRec]) -> np.ndarray:
    """OpRec list -> FnetOp array with absolute device pointers (all buffers must be bound)."""
    arr = np.zeros(len(recs), dtype=rt.OP_DTYPE)
    for k, r in enumerate(recs):
        a = arr[k]
        a["op"] = r.code
        a["dtype"] = r.dtype
        assert len(r.i) <= rt.LANE_SLOT and len(r.f) <= 8 and len(r.p) <= 10, r
        a["i"][: len(r.i)] = r.i
        a["i"][rt.LANE_SLOT] = r.lane
        a["f"][: len(r.f)] = r.f
        for j, ref in enumerate(r.p):
            if ref is None:
                continue
            t = ref.buf.tensor
            if t is None:
                raise RuntimeError(f"buffer {ref.buf.name} is not bound")
            a["p"][j] = t.data_ptr() + ref.off * t.element_size()
    return arr
