// Native autoregressive rollout: what the reference's C++ viewer does per simulation step
// with Ort::Session::Run (engine/src/Simulation.cpp:236-367) -- pack [density, velx + impulse, vely + impulse, previous
// density, emitter, collider], run the UNet, hand the output over as the next state -- as a self-contained C ABI on top of
// the op plans: fnet_rollout_create builds the forward plan of one frame from a UNet configuration and a flat weight
// array (the state-dict order of ml/models/unet.py:204-279, i.e. the tensors of a reference checkpoint concatenated),
// allocates its device buffers and captures the frame in a hipGraph; no Python, no torch.  The emission mirrors
// fluidnet_amd/engine/plan.py:emit_forward (inference mode) record for record, so both hosts produce bit-identical states
// (tests/test_gpu_native_rollout.py).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "fnet_common.h"

namespace {

struct PV {            // channel view of a P-tensor
    bf16_t* buf;
    int H, W, ldc, coff, C;
    bf16_t* ptr() const { return buf; }
};

struct ConvEnt { int cout, cin, cin_pad; long long w_off; bf16_t* Wf; int cout_pad = 0, kf = 0; };
inline int cpad(int c) { return (c + 7) / 8 * 8; }      // plan.py:cpad -- channel counts run zero-padded to a multiple of 8
inline int kpad(int c) { return (c + 15) / 16 * 16; }   // plan.py:kpad -- K extent of the MFMA kernel's weights

struct Rollout {
    FnetUNetConfig cfg;
    int B, H, W;
    std::vector<void*> allocs;
    std::map<std::string, long long> off;          // parameter name -> offset (floats) in the padded device buffer
    std::vector<std::pair<long long, long long>> spans;   // (padded offset, length) of every tensor in state-dict order
    std::map<std::string, ConvEnt> convs;
    std::vector<FnetOp> prep, frame;
    float* params = nullptr;                        // device, every tensor 32-B aligned
    long long n_params_ref = 0, n_params_padded = 0;
    float *cur = nullptr, *prev = nullptr, *emit = nullptr, *coll = nullptr, *pred = nullptr, *z = nullptr, *impulse = nullptr;
    float* stats_scratch = nullptr;
    long long scratch_floats = 0;
    void* graph = nullptr;
    bool fail = false;

    ~Rollout() {
        if (graph) fnet_graph_destroy(graph);
        for (void* p : allocs) (void)hipFree(p);
    }
    void* alloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess || hipMemset(p, 0, bytes ? bytes : 16) != hipSuccess) { fail = true; return nullptr; }
        allocs.push_back(p);
        return p;
    }
    PV pbuf(int h, int w, int C) { return PV{(bf16_t*)alloc((size_t)B * (h + 2) * (w + 2) * C * sizeof(bf16_t)), h, w, C, 0, C}; }
    float* pp(const std::string& name) { return params + off.at(name); }
    int impl_for(int cin, int cout) const { return (cin % 8 == 0 && cout % 8 == 0) ? FNET_CONV_MFMA : FNET_CONV_REF; }
    // the stride-2 convolution of level L in its 2x2 space-to-depth form: the same rule as engine/plan.py _s2_native
    bool s2_native(int L, int C) const {
        static const bool on = [] { const char* e = getenv("FNET_S2"); return !e || (e[0] == '1' && !e[1]); }();
        if (!on || cfg.downsample != 0 || C % 32) return false;
        const long long h = H >> (L + 1), w = W >> (L + 1);
        return (long long)B * (h + 2) * (w + 2) < (1ll << 24) && (long long)B * (2 * h + 2) * (2 * w + 2) * 2 * C * 2 < (1ll << 31) && w + 2 <= 500;
    }
};

FnetOp mkop(int code) {
    FnetOp o;
    memset(&o, 0, sizeof(o));
    o.op = code;
    o.dtype = FNET_BF16;
    return o;
}

// ---- parameter table: the float tensors of the state dict, in its order (no transposed conv; BatchNorm2d contributes weight, bias,
// running_mean, running_var -- its int64 num_batches_tracked is not part of the array)
void add_param(Rollout& r, const std::string& name, long long n) {
    r.off[name] = r.n_params_padded;
    r.spans.push_back({r.n_params_padded, n});
    r.n_params_ref += n;
    r.n_params_padded += (n + 7) / 8 * 8;
}
void add_block(Rollout& r, const std::string& pre, int cin, int cout) {
    const int ci[2] = {cin, cout};
    for (int k = 0; k < 2; ++k) {
        const std::string c = pre + ".c" + std::to_string(k + 1), n = pre + ".n" + std::to_string(k + 1);
        add_param(r, c + ".weight", (long long)cout * ci[k] * 9);
        add_param(r, n + ".weight", cout);
        add_param(r, n + ".bias", cout);
        if (r.cfg.norm == 1) {
            add_param(r, n + ".running_mean", cout);
            add_param(r, n + ".running_var", cout);
        }
        r.convs[c] = ConvEnt{cout, ci[k], cpad(ci[k]), r.off[c + ".weight"], nullptr};
    }
}
void build_params(Rollout& r) {
    const FnetUNetConfig& c = r.cfg;
    add_param(r, "stem.weight", (long long)c.base_channels * c.in_channels * 9);
    add_param(r, "stem.bias", c.base_channels);
    r.convs["stem"] = ConvEnt{c.base_channels, c.in_channels, 16, r.off["stem.weight"], nullptr};
    int ch = c.base_channels;
    std::vector<int> skips;
    for (int L = 0; L < c.depth; ++L) {
        const int oc = ch * 2;
        const std::string d = "downs." + std::to_string(L);
        add_block(r, d + ".block", ch, oc);
        if (c.use_residual) add_block(r, d + ".res.b", oc, oc);
        const std::string dk = d + (c.downsample == 0 ? ".down" : ".down.1");     // Sequential(pool, conv) names its conv "1"
        add_param(r, dk + ".weight", (long long)oc * oc * 9);
        add_param(r, dk + ".bias", oc);
        r.convs[d + ".down"] = ConvEnt{oc, oc, oc, r.off[dk + ".weight"], nullptr};
        skips.push_back(oc);
        ch = oc;
    }
    for (int j = 0; j < c.bottleneck_blocks; ++j) add_block(r, "mid." + std::to_string(j) + ".b", ch, ch);
    for (int i = 0; i < c.depth; ++i) {
        const int sc = skips[c.depth - 1 - i], oc = sc / 2;
        const std::string u = "ups." + std::to_string(i);
        add_block(r, u + ".block", ch + sc, oc);
        if (c.use_residual) add_block(r, u + ".res.b", oc, oc);
        ch = oc;
    }
    add_param(r, "head.weight", (long long)c.out_channels * ch);
    add_param(r, "head.bias", c.out_channels);
}

// ---- op emitters (slots: include/fluidnet_ops.h)
void e_conv(Rollout& r, const PV& src, const ConvEnt& ce, const float* bias, const PV& dst, int cin, int cout) {
    FnetOp o = mkop(FNET_OP_CONV3X3);
    const int v[10] = {r.B * (src.H + 2) * (src.W + 2), src.W + 2, cin, src.ldc, src.coff, cout, dst.ldc, dst.coff, r.impl_for(cin, cout), 0};
    memcpy(o.i, v, sizeof(v));
    o.p[0] = (uint64_t)src.buf; o.p[1] = (uint64_t)ce.Wf; o.p[2] = (uint64_t)bias; o.p[3] = (uint64_t)dst.buf;
    r.frame.push_back(o);
}
// Conv2d(C, C, 3, stride 2) forward as the 2x2-tap convolution over the space-to-depth view of src (fnet_conv_mfma.hip, MODE 1)
void e_conv_s2(Rollout& r, const PV& src, const ConvEnt& ce, const float* bias, const PV& dst) {
    FnetOp o = mkop(FNET_OP_CONV3X3);
    const int v[12] = {r.B * (dst.H + 2) * (dst.W + 2), dst.W + 2, 4 * src.C, src.ldc, src.coff, dst.C, dst.ldc, dst.coff, FNET_CONV_MFMA, 1, dst.H, src.C};
    memcpy(o.i, v, sizeof(v));
    o.p[0] = (uint64_t)src.buf; o.p[1] = (uint64_t)ce.Wf; o.p[2] = (uint64_t)bias; o.p[3] = (uint64_t)dst.buf;
    r.frame.push_back(o);
}
void e_pad_ring(Rollout& r, const PV& x) {
    FnetOp o = mkop(FNET_OP_PAD_RING);
    const int v[7] = {r.B, x.H, x.W, x.C, x.ldc, x.coff, r.cfg.padding_mode};
    memcpy(o.i, v, sizeof(v));
    o.p[0] = (uint64_t)x.buf;
    r.frame.push_back(o);
}
int group_size(int groups, int ch) {          // channels per group of nn.GroupNorm as unet.py:24-28 builds it
    int g = std::min(groups, ch);
    while (g > 1 && ch % g) --g;
    return ch / g;
}
PV unit(Rollout& r, const std::string& conv, const std::string& norm, const PV& xin, int cout, const PV* out, const PV* res) {
    const ConvEnt& ce = r.convs.at(conv);
    PV raw = r.pbuf(xin.H, xin.W, cout);
    float* st = (float*)r.alloc((size_t)r.B * cout * 2 * sizeof(float));
    PV o = out ? *out : r.pbuf(xin.H, xin.W, cout);
    e_conv(r, xin, ce, nullptr, raw, ce.cin_pad, cout);
    if (r.cfg.norm != 1) {                      // eval-mode BatchNorm2d normalises with the running statistics alone
        FnetOp s = mkop(FNET_OP_IN_STATS);
        const int v[7] = {r.B, raw.H, raw.W, raw.C, raw.ldc, raw.coff, (int)r.scratch_floats};
        memcpy(s.i, v, sizeof(v));
        s.f[0] = 1e-5f;
        s.p[0] = (uint64_t)raw.buf; s.p[1] = (uint64_t)st; s.p[2] = (uint64_t)r.stats_scratch;
        r.frame.push_back(s);
    }
    if (r.cfg.norm != 0) {
        FnetOp s = mkop(FNET_OP_NORM_COMBINE);
        const int batch = r.cfg.norm == 1;
        // groups are formed over the layer's REAL channels (ce.cout); `cout` may be the zero-padded count (base_channels % 8 != 0)
        const int v[6] = {r.B, cout, batch ? 3 : 2, batch ? 1 : group_size(r.cfg.group_norm_groups, ce.cout), raw.H * raw.W, ce.cout};
        memcpy(s.i, v, sizeof(v));
        s.f[0] = 1e-5f; s.f[1] = 0.1f;
        s.p[0] = (uint64_t)st;
        if (batch) { s.p[1] = (uint64_t)r.pp(norm + ".running_mean"); s.p[2] = (uint64_t)r.pp(norm + ".running_var"); }
        r.frame.push_back(s);
    }
    {
        FnetOp a = mkop(FNET_OP_APPLY);
        const int v[14] = {r.B, raw.H, raw.W, raw.C, raw.ldc, raw.coff, res ? res->ldc : 8, res ? res->coff : 0, o.ldc, o.coff,
                           r.cfg.act, r.cfg.padding_mode, 0, 0};
        memcpy(a.i, v, sizeof(v));
        a.f[0] = 0.f;                               // inference: no dropout
        a.p[0] = (uint64_t)raw.buf; a.p[1] = (uint64_t)st; a.p[2] = (uint64_t)r.pp(norm + ".weight"); a.p[3] = (uint64_t)r.pp(norm + ".bias");
        a.p[4] = res ? (uint64_t)res->buf : 0; a.p[5] = (uint64_t)o.buf;
        r.frame.push_back(a);
    }
    return o;
}
PV conv_block(Rollout& r, const std::string& key, const PV& xin, int cout, const PV* out, bool residual) {
    PV u1 = unit(r, key + ".c1", key + ".n1", xin, cout, nullptr, nullptr);
    return unit(r, key + ".c2", key + ".n2", u1, cout, out, residual ? &xin : nullptr);
}

void build_frame(Rollout& r) {
    const FnetUNetConfig& c = r.cfg;
    const int B = r.B, H = r.H, W = r.W, HW = H * W, pm = c.padding_mode;
    // weights: fp32 OIHW -> bf16 forward layout, once per weight upload
    for (auto& kv : r.convs) {
        ConvEnt& ce = kv.second;
        ce.cout_pad = cpad(ce.cout);
        const std::string& name = kv.first;
        if (name.size() > 5 && name.compare(name.size() - 5, 5, ".down") == 0 && r.s2_native(atoi(name.c_str() + 6), ce.cout)) {
            ce.kf = 4 * ce.cin;
            ce.Wf = (bf16_t*)r.alloc((size_t)4 * ce.cout_pad * ce.kf * sizeof(bf16_t));
            FnetOp o = mkop(FNET_OP_PREP_WEIGHTS);
            const int v[8] = {ce.cout, ce.cin, 4 * ce.cin, 4 * ce.cin, 3, ce.cout_pad, (ce.cout_pad + 15) / 16 * 16, 2};
            memcpy(o.i, v, sizeof(v));
            o.p[0] = (uint64_t)(r.params + ce.w_off); o.p[1] = (uint64_t)ce.Wf;
            r.prep.push_back(o);
            continue;
        }
        const bool mfma = r.impl_for(ce.cin_pad, ce.cout_pad) == FNET_CONV_MFMA;
        ce.kf = mfma ? kpad(ce.cin_pad) : ce.cin_pad;
        ce.Wf = (bf16_t*)r.alloc((size_t)9 * ce.cout_pad * ce.kf * sizeof(bf16_t));
        FnetOp o = mkop(FNET_OP_PREP_WEIGHTS);
        const int v[7] = {ce.cout, ce.cin, ce.kf, 0, mfma ? 1 : 0, ce.cout_pad, 0};
        memcpy(o.i, v, sizeof(v));
        o.p[0] = (uint64_t)(r.params + ce.w_off); o.p[1] = (uint64_t)ce.Wf;
        r.prep.push_back(o);
    }
    // pack: [density, velx, vely] of the current state, previous density, emitter, collider
    PV in_p = r.pbuf(H, W, 16);
    {
        FnetOp o = mkop(FNET_OP_PACK_INPUT);
        const int v[11] = {B, H, W, 16, pm, 3 * HW, 3 * HW, 3 * HW, HW, HW, HW};
        memcpy(o.i, v, sizeof(v));
        for (int k = 0; k < 3; ++k) o.p[k] = (uint64_t)(r.cur + (size_t)k * HW);
        o.p[3] = (uint64_t)r.prev; o.p[4] = (uint64_t)r.emit; o.p[5] = (uint64_t)r.coll; o.p[6] = (uint64_t)in_p.buf;
        r.frame.push_back(o);
    }
    const int base = cpad(c.base_channels);
    PV x = r.pbuf(H, W, base);
    e_conv(r, in_p, r.convs.at("stem"), r.pp("stem.bias"), x, 16, base);
    {
        FnetOp o = mkop(FNET_OP_PAD_RING);
        const int v[7] = {B, H, W, base, base, 0, pm};
        memcpy(o.i, v, sizeof(v));
        o.p[0] = (uint64_t)x.buf;
        r.frame.push_back(o);
    }
    int h = H, w = W, ch = c.base_channels;
    std::vector<PV> cats;
    std::vector<int> Cs;
    for (int L = 0; L < c.depth; ++L) {
        const int C = ch * 2;
        PV cat = r.pbuf(h, w, 2 * C);
        PV skip = PV{cat.buf, h, w, 2 * C, C, C};
        const std::string d = "downs." + std::to_string(L);
        if (c.use_residual) {
            PV b = conv_block(r, d + ".block", x, C, nullptr, false);
            conv_block(r, d + ".res.b", b, C, &skip, true);
        } else {
            conv_block(r, d + ".block", x, C, &skip, false);
        }
        PV dn = r.pbuf(h / 2, w / 2, C);
        if (c.downsample == 0 && r.s2_native(L, C)) {
            e_conv_s2(r, skip, r.convs.at(d + ".down"), r.pp(d + ".down.bias"), dn);
            e_pad_ring(r, dn);
        } else if (c.downsample == 0) {
            PV full = r.pbuf(h, w, C);
            e_conv(r, skip, r.convs.at(d + ".down"), nullptr, full, C, C);
            FnetOp o = mkop(FNET_OP_SUBSAMPLE_BIAS);
            const int v[9] = {B, h, w, C, C, 0, C, 0, pm};
            memcpy(o.i, v, sizeof(v));
            o.p[0] = (uint64_t)full.buf; o.p[1] = (uint64_t)r.pp(d + ".down.bias"); o.p[2] = (uint64_t)dn.buf;
            r.frame.push_back(o);
        } else {
            PV pooled = r.pbuf(h / 2, w / 2, C);
            FnetOp o = mkop(FNET_OP_POOL2X);
            const int v[10] = {B, h / 2, w / 2, C, skip.ldc, skip.coff, C, 0, c.downsample == 2 ? 1 : 0, pm};
            memcpy(o.i, v, sizeof(v));
            o.p[0] = (uint64_t)skip.buf; o.p[1] = (uint64_t)pooled.buf;
            r.frame.push_back(o);
            e_conv(r, pooled, r.convs.at(d + ".down"), r.pp(d + ".down.1.bias"), dn, C, C);
            FnetOp q = mkop(FNET_OP_PAD_RING);
            const int u[7] = {B, h / 2, w / 2, C, C, 0, pm};
            memcpy(q.i, u, sizeof(u));
            q.p[0] = (uint64_t)dn.buf;
            r.frame.push_back(q);
        }
        cats.push_back(cat);
        Cs.push_back(C);
        x = dn;
        h /= 2; w /= 2; ch = C;
    }
    for (int j = 0; j < c.bottleneck_blocks; ++j) x = conv_block(r, "mid." + std::to_string(j) + ".b", x, ch, nullptr, true);
    for (int i = 0; i < c.depth; ++i) {
        const int L = c.depth - 1 - i, C = Cs[L];
        h *= 2; w *= 2;
        FnetOp o = mkop(FNET_OP_UPSAMPLE2X);
        const int v[10] = {B, h / 2, w / 2, C, x.ldc, x.coff, 2 * C, 0, c.upsample, pm};
        memcpy(o.i, v, sizeof(v));
        o.p[0] = (uint64_t)x.buf; o.p[1] = (uint64_t)cats[L].buf;
        r.frame.push_back(o);
        const std::string u = "ups." + std::to_string(i);
        PV catv = PV{cats[L].buf, h, w, 2 * C, 0, 2 * C};
        x = conv_block(r, u + ".block", catv, cpad(C / 2), nullptr, false);
        if (c.use_residual) x = conv_block(r, u + ".res.b", x, cpad(C / 2), nullptr, true);
        ch = C / 2;
    }
    {
        FnetOp o = mkop(FNET_OP_HEAD_FWD);
        const int v[8] = {B, H, W, ch, x.ldc, x.coff, c.out_channels, c.output_activation};
        memcpy(o.i, v, sizeof(v));
        o.p[0] = (uint64_t)x.buf; o.p[1] = (uint64_t)r.pp("head.weight"); o.p[2] = (uint64_t)r.pp("head.bias");
        o.p[3] = (uint64_t)r.z; o.p[4] = (uint64_t)r.pred;
        r.frame.push_back(o);
    }
    // state hand-off (Simulation.cpp double buffer; simple_infer.py:259-260): prev <- current density, current <- output
    {
        FnetOp o = mkop(FNET_OP_COPY_PLANES);
        const int v[4] = {B, HW, 3 * HW, HW};
        memcpy(o.i, v, sizeof(v));
        o.p[0] = (uint64_t)r.cur; o.p[1] = (uint64_t)r.prev;
        r.frame.push_back(o);
        FnetOp q = mkop(FNET_OP_COPY_PLANES);
        const int u[4] = {B, 3 * HW, 3 * HW, 3 * HW};
        memcpy(q.i, u, sizeof(u));
        q.p[0] = (uint64_t)r.pred; q.p[1] = (uint64_t)r.cur;
        r.frame.push_back(q);
    }
}

int cfg_ok(const FnetUNetConfig& c, int B, int H, int W) {
    if (c.in_channels != 6 || c.out_channels != 3) return 0;
    if (c.base_channels < 8 || c.base_channels > 64 || c.base_channels % 4) return 0;
    if (c.depth < 1 || c.depth > 6 || c.bottleneck_blocks < 0 || c.bottleneck_blocks > 8) return 0;
    if (c.act < 0 || c.act > 4 || c.padding_mode < 0 || c.padding_mode > 3 || c.upsample < 0 || c.upsample > 1) return 0;
    if (c.output_activation < 0 || c.output_activation > 1 || (c.use_residual != 0 && c.use_residual != 1)) return 0;
    if (c.norm < 0 || c.norm > 2 || c.downsample < 0 || c.downsample > 2 || (c.norm == 2 && c.group_norm_groups < 1)) return 0;
    const int div = 1 << c.depth;
    if (B < 1 || B > 4096 || H % div || W % div || H / div < 2 || W / div < 2) return 0;
    return 1;
}

__global__ __launch_bounds__(256) void k_add_impulse(float* state, const float* imp, int B, int HW) {
    // velx += impulse_x, vely += impulse_y (Simulation.cpp:263-271); state [B,3,H,W], imp [B,2,H,W]
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * 2 * HW) return;
    const int n = i / (2 * HW), r = i - n * 2 * HW;
    state[(size_t)n * 3 * HW + HW + r] += imp[i];
}

}  // namespace

extern "C" {

int fnet_unet_param_count(const FnetUNetConfig* cfg, int64_t* n_floats) {
    if (!cfg || !n_floats || !cfg_ok(*cfg, 1, 1 << cfg->depth << 1, 1 << cfg->depth << 1)) return FNET_E_BADARG;
    Rollout r;
    r.cfg = *cfg;
    build_params(r);
    *n_floats = r.n_params_ref;
    return FNET_OK;
}

int fnet_rollout_create(const FnetUNetConfig* cfg, const float* host_params, int64_t n_params, int32_t B, int32_t H, int32_t W, void** handle) {
    if (!cfg || !host_params || !handle || !cfg_ok(*cfg, B, H, W)) return FNET_E_BADARG;
    Rollout* r = new Rollout();
    r->cfg = *cfg; r->B = B; r->H = H; r->W = W;
    build_params(*r);
    if (n_params != r->n_params_ref) { delete r; return FNET_E_BADARG; }
    const size_t HW = (size_t)H * W;
    r->params = (float*)r->alloc((size_t)r->n_params_padded * sizeof(float));
    r->cur = (float*)r->alloc(B * 3 * HW * 4); r->prev = (float*)r->alloc(B * HW * 4);
    r->emit = (float*)r->alloc(B * HW * 4); r->coll = (float*)r->alloc(B * HW * 4);
    r->pred = (float*)r->alloc(B * 3 * HW * 4); r->z = (float*)r->alloc(B * 3 * HW * 4);
    r->impulse = (float*)r->alloc(B * 2 * HW * 4);
    int cmax = cfg->base_channels << cfg->depth;
    r->scratch_floats = (long long)B * ((2048 + B - 1) / B) * cmax * 2;
    r->stats_scratch = (float*)r->alloc((size_t)r->scratch_floats * 4);
    if (!r->fail) build_frame(*r);
    if (r->fail) { delete r; return FNET_E_HIP; }
    int rc = fnet_rollout_set_weights(r, host_params, n_params);
    if (rc == FNET_OK) rc = fnet_graph_create(r->frame.data(), (int32_t)r->frame.size(), &r->graph);
    if (rc != FNET_OK) { delete r; return rc; }
    *handle = r;
    return FNET_OK;
}

int fnet_rollout_set_weights(void* handle, const float* host_params, int64_t n_params) {
    Rollout* r = (Rollout*)handle;
    if (!r || !host_params || n_params != r->n_params_ref) return FNET_E_BADARG;
    // reference order (tensors back to back) -> device layout (every tensor 32-B aligned), tensor by tensor
    std::vector<float> padded((size_t)r->n_params_padded, 0.f);
    long long src = 0;
    for (auto& sp : r->spans) {
        memcpy(padded.data() + sp.first, host_params + src, (size_t)sp.second * sizeof(float));
        src += sp.second;
    }
    if (hipMemcpy(r->params, padded.data(), padded.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return FNET_E_HIP;
    int32_t failed = -1;
    int rc = fnet_plan_run(r->prep.data(), (int32_t)r->prep.size(), nullptr, &failed);
    if (rc != FNET_OK) return rc;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return FNET_E_HIP;
    return FNET_OK;
}

int fnet_rollout_set_masks(void* handle, const float* host_emitter, const float* host_collider) {
    Rollout* r = (Rollout*)handle;
    if (!r) return FNET_E_BADARG;
    const size_t bytes = (size_t)r->B * r->H * r->W * 4;
    hipError_t e = host_emitter ? hipMemcpy(r->emit, host_emitter, bytes, hipMemcpyHostToDevice) : hipMemset(r->emit, 0, bytes);
    if (e == hipSuccess) e = host_collider ? hipMemcpy(r->coll, host_collider, bytes, hipMemcpyHostToDevice) : hipMemset(r->coll, 0, bytes);
    return e == hipSuccess ? FNET_OK : FNET_E_HIP;
}

int fnet_rollout_set_state(void* handle, const float* host_state, const float* host_prev_density) {
    Rollout* r = (Rollout*)handle;
    if (!r) return FNET_E_BADARG;
    const size_t hw = (size_t)r->B * r->H * r->W * 4;
    hipError_t e = host_state ? hipMemcpy(r->cur, host_state, 3 * hw, hipMemcpyHostToDevice) : hipMemset(r->cur, 0, 3 * hw);
    if (e == hipSuccess) e = host_prev_density ? hipMemcpy(r->prev, host_prev_density, hw, hipMemcpyHostToDevice) : hipMemset(r->prev, 0, hw);
    return e == hipSuccess ? FNET_OK : FNET_E_HIP;
}

int fnet_rollout_add_impulse(void* handle, const float* host_impulse_xy, void* stream) {
    Rollout* r = (Rollout*)handle;
    if (!r || !host_impulse_xy) return FNET_E_BADARG;
    const int HW = r->H * r->W, n = r->B * 2 * HW;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(r->impulse, host_impulse_xy, (size_t)n * 4, hipMemcpyHostToDevice, st) != hipSuccess) return FNET_E_HIP;
    hipLaunchKernelGGL(k_add_impulse, dim3((n + 255) / 256), dim3(256), 0, st, r->cur, r->impulse, r->B, HW);
    if (hipGetLastError() != hipSuccess) return FNET_E_HIP;
    // the host array may be pageable and may go away as soon as this returns (the viewer's input thread): wait for the upload
    return hipStreamSynchronize(st) == hipSuccess ? FNET_OK : FNET_E_HIP;
}

int fnet_rollout_step(void* handle, int32_t frames, void* stream) {
    Rollout* r = (Rollout*)handle;
    if (!r || frames < 1) return FNET_E_BADARG;
    return fnet_graph_launch(r->graph, stream, frames);
}

int fnet_rollout_state_ptr(void* handle, float** device_state, float** device_prev_density) {
    Rollout* r = (Rollout*)handle;
    if (!r) return FNET_E_BADARG;
    if (device_state) *device_state = r->cur;
    if (device_prev_density) *device_prev_density = r->prev;
    return FNET_OK;
}

int fnet_rollout_read_state(void* handle, float* host_state, void* stream) {
    Rollout* r = (Rollout*)handle;
    if (!r || !host_state) return FNET_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(host_state, r->cur, (size_t)r->B * 3 * r->H * r->W * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return FNET_E_HIP;
    return hipStreamSynchronize(st) == hipSuccess ? FNET_OK : FNET_E_HIP;
}

int fnet_rollout_destroy(void* handle) {
    delete (Rollout*)handle;
    return FNET_OK;
}

}  // extern "C"
