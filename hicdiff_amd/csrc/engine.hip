// HiCDiff engine for gfx950: network plans (UNet / hicedrn, unconditional, conditional, SR3), weight
// packing, activation pool, and the C ABI of include/hicdiff_hip.h.
#include "hd_common.h"
#include "../../include/hicdiff_hip.h"
#include "../../include/hicdiff_hip_debug.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

static thread_local std::string g_err;
void hd_set_error(const std::string& msg) { g_err = msg; }

#define HD_TRY(expr)                  \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != 0) return rc_;     \
    } while (0)

// ---- activation pool: first-fit free list over one device block; deterministic for a given call
// sequence, so a captured graph always sees the same addresses.  `dry` simulates the same sequence
// on an unbounded arena to size the block (hd_workspace_bytes / hd_reserve).
struct Pool {
    char* base = nullptr;
    size_t cap = 0;
    bool dry = false;
    size_t high = 0;
    std::vector<std::pair<size_t, size_t>> free_;   // (offset, size), sorted by offset
    std::unordered_map<size_t, size_t> live;        // offset -> size
    void reset(bool dry_) {
        dry = dry_; high = 0; free_.clear(); live.clear();
        free_.push_back({0, dry ? (size_t)1 << 60 : cap});
    }
    bool alloc(size_t bytes, float** out) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        for (size_t i = 0; i < free_.size(); ++i) {
            if (free_[i].second >= bytes) {
                size_t off = free_[i].first;
                free_[i].first += bytes; free_[i].second -= bytes;
                if (free_[i].second == 0) free_.erase(free_.begin() + i);
                live[off] = bytes;
                high = std::max(high, off + bytes);
                *out = dry ? reinterpret_cast<float*>((uintptr_t)256 + off) : reinterpret_cast<float*>(base + off);
                return true;
            }
        }
        return false;
    }
    void release(float* p) {
        if (!p) return;
        size_t off = dry ? (size_t)((uintptr_t)p - 256) : (size_t)((char*)p - base);
        auto it = live.find(off);
        if (it == live.end()) return;
        size_t sz = it->second;
        live.erase(it);
        auto pos = std::lower_bound(free_.begin(), free_.end(), std::make_pair(off, (size_t)0));
        pos = free_.insert(pos, {off, sz});
        size_t i = pos - free_.begin();
        if (i + 1 < free_.size() && free_[i].first + free_[i].second == free_[i + 1].first) {
            free_[i].second += free_[i + 1].second; free_.erase(free_.begin() + i + 1);
        }
        if (i > 0 && free_[i - 1].first + free_[i - 1].second == free_[i].first) {
            free_[i - 1].second += free_[i].second; free_.erase(free_.begin() + i);
        }
    }
};

struct ResW {
    ConvW c1, c2, res;
    float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
    bool has_res = false, has_norm = true;
    int film_off = 0, cin = 0, cout = 0;
    std::string name;
};
struct AttnW {
    float* norm_g = nullptr; float* out_g = nullptr;
    ConvW qkv, out;
    ConvW qonly;                       // fused path: q = to_qkv[0:128] only
    unsigned short* wkv = nullptr;     // fused path: split k/v weight image with the LayerNorm gain folded in
    unsigned short* wq = nullptr;      // chained q path (dim 64): split q weight image with the LayerNorm gain folded in
    bool fused = false;
    int dim = 0; bool linear = true;
    std::string name;
};
struct StageW { ResW r1, r2; AttnW attn; ConvW resample; bool last = false; int cin = 0, cout = 0; };

struct hd_ctx {
    hd_arch_desc arch{};
    int device = 0;
    bool loaded = false;
    std::string err;
    std::vector<void*> owned;      // every weight allocation
    // shared
    float *first_w = nullptr, *first_b = nullptr;   // init_conv / head, torch layout
    int first_ks = 7, first_cout = 64, cin0 = 1;
    int time_in = 64, time_dim = 256;
    float *w1t = nullptr, *b1 = nullptr, *w3t = nullptr, *b3 = nullptr;
    float *film_wt = nullptr, *film_b = nullptr; int film_n = 0;
    // unet
    std::vector<StageW> downs, ups;
    ResW mid1, mid2, final_res; AttnW mid_attn;
    float *final_w = nullptr, *final_b = nullptr;
    // hicedrn
    std::vector<ResW> body; ConvW body_tail, tail;
    // workspace
    Pool pool;
    float* eps_buf = nullptr; size_t eps_cap = 0;   // eps of the fused step calls
    int resB = 0, resS = 0;
    int precision = HD_PREC_BF16X3;   // arithmetic of the wide convolutions (hd_set_precision)
    int f16w2 = 0;                    // with BF16X3: the 3x3 convolutions take two (1) / one (2) fp16 products per multiply (HD_PRECISION_F16W2 / _F16W1, or one step's hd_ddpm_coef.arith)
    int ck = 16;                      // K slice of the split-bf16 weights: 32 when every channel count allows it
    // hipGraph replay of the fused sampler steps (device-generated noise only): one graph per
    // (kind, B, S, tensor addresses); the step's scalars are written to the lane's `sp_dev` by a 1-thread kernel.
    // A LANE is a stream of the context with its own graph cache and step-parameter block.  Lane 0 replays whole-batch steps out of
    // `pool`.  Large steps (lanes_for) are cut into two half batches, one per lane, each with a workspace of its own: tiles are
    // independent, so the halves may run side by side -- inside an hd_chain_begin / hd_chain_end bracket they are only
    // ordered against the caller's stream at the two ends and drift apart, which lets one half's HBM-bound launches (1x1, attention,
    // element-wise) run under the other half's MFMA-bound 3x3 convolutions.
    struct StepGraph { int kind, B, S, precision; const void *x, *aux, *x0; int seen; hipGraphExec_t exec; };
    struct Lane {
        hipStream_t st = nullptr;
        hipEvent_t ev_out = nullptr;
        StepParams* sp_dev = nullptr;
        std::vector<StepGraph> graphs;
        Pool pool;                 // half-batch workspace (lanes_for(B, S) == 2 only)
        bool dirty = false;        // holds work the caller's stream has not been ordered after yet (chain bracket)
        void drop_graphs() { for (auto& g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec); graphs.clear(); }
    };
    static constexpr int MAX_LANES = 4;
    Lane lane[MAX_LANES];
    hipEvent_t ev_in = nullptr;
    int use_graphs = -1;           // 1 / 0: always / never replay the fused steps from a hipGraph; -1 (default): by the amount of work, see run_step
    int chains = -1;               // n >= 1: always cut a replayed step into n sub-batch lanes (1: never); -1 (default): by the amount of work, see lanes_for
    int lane_delay_us = 0;         // experiment knob (HICDIFF_LANE_DELAY_US): lane 1 starts a chain this much later than lane 0
    unsigned* f16_absmax = nullptr;   // device slots of the loader's fp16 range guard (one per 3x3 layer)
    bool in_chain = false;         // between hd_chain_begin and hd_chain_end
    bool forked = false;           // in_chain: the lanes already wait behind the caller's stream
    // test-only capture of intermediates (hicdiff_hip_debug.h)
    bool capture = false;
    std::unordered_map<std::string, Act> captured;
};

static int fail(hd_ctx* c, int code, const std::string& msg) {
    g_err = msg;
    if (c) c->err = msg;
    return code;
}
static int keep_err(hd_ctx* c, int rc) { if (rc != 0 && c) c->err = g_err; return rc; }

// ---- weights -----------------------------------------------------------------------------------
struct Loader {
    hd_ctx* c; hipStream_t st;
    std::unordered_map<std::string, const hd_named_tensor*> map;
    // Range guard of the fp16 weight images (the schedule's two- and one-product arithmetic): hi = fp16(w), lo = fp16(w - hi) is exact to ~2^-25
    // ABSOLUTE (fp16 subnormals), so a layer whose largest weight is below 2^-8 would carry less than the 2^-17 relative accuracy the error budget
    //  assumes, and one above 2^15 would overflow hi.  Such a layer keeps three bf16 products at every step.  The packed
    // weights are what is measured (after weight standardisation where the layer has it), one atomicMax slot per layer, read back once per load.
    std::vector<ConvW*> f16_layers;
    static constexpr int kF16Slots = 1024;
    int f16_slot(ConvW* w, unsigned** slot) {
        if (!c->f16_absmax) {
            if (hipMalloc((void**)&c->f16_absmax, kF16Slots * sizeof(unsigned)) != hipSuccess) return fail(c, HD_EHIP, "hipMalloc(weights) failed");
            c->owned.push_back(c->f16_absmax);
        }
        if (f16_layers.empty() && hipMemsetAsync(c->f16_absmax, 0, kF16Slots * sizeof(unsigned), st) != hipSuccess) return fail(c, HD_EHIP, "weight range memset failed");
        if ((int)f16_layers.size() >= kF16Slots) { *slot = nullptr; return 0; }       // (no network here has that many layers: unguarded beyond)
        *slot = c->f16_absmax + f16_layers.size();
        f16_layers.push_back(w);
        return 0;
    }
    int finish_f16_range() {
        if (f16_layers.empty()) return 0;
        std::vector<unsigned> bits(f16_layers.size());
        if (hipMemcpyAsync(bits.data(), c->f16_absmax, bits.size() * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(c, HD_EHIP, "weight range readback failed");
        bool changed = false;
        for (size_t i = 0; i < bits.size(); ++i) {
            const float m = __builtin_bit_cast(float, bits[i]);
            const int ok = m >= 0.00390625f && m <= 32768.f ? 1 : 0;              // (NaN / inf weights fail both)
            changed |= ok != f16_layers[i]->f16_range_ok;
            f16_layers[i]->f16_range_ok = ok;
        }
        if (changed) {                                                              // a captured step holds the kernel choice
            if (hipDeviceSynchronize() != hipSuccess) return fail(c, HD_EHIP, "device synchronize failed");
            for (auto& ln : c->lane) ln.drop_graphs();
        }
        return 0;
    }
    const hd_named_tensor* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = map.find(name);
        if (it == map.end()) { fail(c, HD_ENOWEIGHT, "missing state-dict entry '" + name + "'"); return nullptr; }
        const hd_named_tensor* t = it->second;
        bool ok = t->ndim == (int)shape.size();
        int i = 0;
        for (int64_t s : shape) { if (ok && t->shape[i] != s) ok = false; ++i; }
        if (!ok) { fail(c, HD_ENOWEIGHT, "state-dict entry '" + name + "' has the wrong shape"); return nullptr; }
        return t;
    }
    float* dev(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)) != hipSuccess) { fail(c, HD_EHIP, "hipMalloc(weights) failed"); return nullptr; }
        c->owned.push_back(p);
        return (float*)p;
    }
    // copy a small 1-D style parameter
    int vec(const std::string& name, std::initializer_list<int64_t> shape, float** dst) {
        const hd_named_tensor* t = get(name, shape);
        if (!t) return HD_ENOWEIGHT;
        size_t n = 1; for (int64_t s : shape) n *= (size_t)s;
        if (!*dst) { *dst = dev(n); if (!*dst) return HD_EHIP; }
        if (hipMemcpyAsync(*dst, t->data, n * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(c, HD_EHIP, "weight copy failed");
        return 0;
    }
    int conv(const std::string& name, int cout, int cin, int k, bool standardize, bool unshuffle, bool bias, ConvW* w) {
        const hd_named_tensor* t = unshuffle ? get(name + ".weight", {cout, (int64_t)cin * 4, 1, 1}) : get(name + ".weight", {cout, cin, k, k});
        if (!t) return HD_ENOWEIGHT;
        const int KH = unshuffle ? 2 : k;
        w->KH = w->KW = KH; w->Cin = cin; w->Cout = cout; w->CoutPad = (cout + 63) / 64 * 64;
        if (!w->w) { w->w = dev((size_t)KH * KH * cin * w->CoutPad); if (!w->w) return HD_EHIP; }
        HD_TRY(launch_pack_conv((const float*)t->data, w->w, cout, cin, KH, KH, w->CoutPad, standardize ? 1 : 0, unshuffle ? 1 : 0, st));
        w->ck = (cin % c->ck == 0) ? c->ck : 16;
        if (!w->wsplit) {
            void* p = nullptr;
            if (hipMalloc(&p, (size_t)KH * KH * cin * w->CoutPad * 2 * sizeof(unsigned short)) != hipSuccess) return fail(c, HD_EHIP, "hipMalloc(weights) failed");
            c->owned.push_back(p);
            w->wsplit = (unsigned short*)p;
        }
        // the split image is packed in 16-channel k-steps whatever the kernel's activation slice (w->ck) is: conv_bf16x3_kernel.h
        HD_TRY(launch_split_conv(w->w, w->wsplit, KH * KH, cin, w->CoutPad, 16, st));
        // 3x3 filters also as fp16 hi | lo: the two-product arithmetic of the early timestep band (hd_ddpm_coef.arith; conv_bf16x3_kernel.h AR = 2)
        if (KH == 3 && !unshuffle && cin % 16 == 0) {
            if (!w->wsplit16) {
                void* p = nullptr;
                if (hipMalloc(&p, (size_t)9 * cin * w->CoutPad * 2 * sizeof(unsigned short)) != hipSuccess) return fail(c, HD_EHIP, "hipMalloc(weights) failed");
                c->owned.push_back(p);
                w->wsplit16 = (unsigned short*)p;
            }
            unsigned* slot = nullptr;
            HD_TRY(f16_slot(w, &slot));
            HD_TRY(launch_split_conv(w->w, w->wsplit16, 9, cin, w->CoutPad, 16, st, 1, slot));
        }
        // Winograd image of the 3x3 filters (conv_winograd.hip): only while that opt-in path is switched on (HICDIFF_WINOGRAD=1 / hd_debug_winograd(1)
        // before the weights are loaded) -- the image is 16/9 of the filter bytes twice over and one more pack launch per layer; without it
        // conv_uses_winograd() is false and the layer takes the implicit-GEMM kernel
        if (conv_winograd_enabled() && KH == 3 && !unshuffle && cin % 16 == 0 && cout % 4 == 0) {
            if (!w->wino) {
                void* p = nullptr;
                if (hipMalloc(&p, conv_winograd_weight_bytes(cin, w->CoutPad)) != hipSuccess) return fail(c, HD_EHIP, "hipMalloc(weights) failed");
                c->owned.push_back(p);
                w->wino = (unsigned short*)p;
            }
            HD_TRY(launch_pack_winograd(w->w, w->wino, cin, w->CoutPad, st));
        }
        if (bias) HD_TRY(vec(name + ".bias", {cout}, &w->bias));
        return 0;
    }
};

static int load_film(Loader& L, const std::string& name, int nout, int* off_io, int* my_off) {
    hd_ctx* c = L.c;
    const bool sr3 = c->arch.sr3;
    const std::string key = name + (sr3 ? ".noise_func.noise_func.0" : ".mlp.1");
    const hd_named_tensor* w = L.get(key + ".weight", {nout, c->time_dim});
    const hd_named_tensor* b = L.get(key + ".bias", {nout});
    if (!w || !b) return HD_ENOWEIGHT;
    *my_off = *off_io;
    HD_TRY(launch_transpose((const float*)w->data, c->film_wt, nout, c->time_dim, c->film_n, *off_io, L.st));
    if (hipMemcpyAsync(c->film_b + *off_io, b->data, (size_t)nout * sizeof(float), hipMemcpyDeviceToDevice, L.st) != hipSuccess)
        return fail(c, HD_EHIP, "film bias copy failed");
    *off_io += nout;
    return 0;
}

static int load_unet_res(Loader& L, const std::string& p, int cin, int cout, ResW* r, int* film_off) {
    const bool sr3 = L.c->arch.sr3;
    r->name = p; r->cin = cin; r->cout = cout; r->has_res = cin != cout; r->has_norm = true;
    HD_TRY(load_film(L, p, sr3 ? cout : 2 * cout, film_off, &r->film_off));
    HD_TRY(L.conv(p + ".block1.proj", cout, cin, 3, true, false, true, &r->c1));
    HD_TRY(L.vec(p + ".block1.norm.weight", {cout}, &r->g1));
    HD_TRY(L.vec(p + ".block1.norm.bias", {cout}, &r->b1));
    HD_TRY(L.conv(p + ".block2.proj", cout, cout, 3, true, false, true, &r->c2));
    HD_TRY(L.vec(p + ".block2.norm.weight", {cout}, &r->g2));
    HD_TRY(L.vec(p + ".block2.norm.bias", {cout}, &r->b2));
    if (r->has_res) HD_TRY(L.conv(p + ".res_conv", cout, cin, 1, false, false, true, &r->res));
    return 0;
}

static int load_attn(Loader& L, const std::string& p, int dim, bool linear, AttnW* a) {
    a->name = p; a->dim = dim; a->linear = linear;
    HD_TRY(L.vec(p + ".fn.norm.g", {1, dim, 1, 1}, &a->norm_g));
    HD_TRY(L.conv(p + ".fn.fn.to_qkv", 384, dim, 1, false, false, false, &a->qkv));
    if (linear && (dim == 64 || dim == 128 || dim == 256)) {
        // fused key/value side (linattn_fused.hip): q-only projection + packed k/v weights
        const hd_named_tensor* t = L.get(p + ".fn.fn.to_qkv.weight", {384, dim, 1, 1});
        if (!t) return HD_ENOWEIGHT;
        hd_ctx* c = L.c;
        ConvW& q = a->qonly;
        q.KH = q.KW = 1; q.Cin = dim; q.Cout = 128; q.CoutPad = 128; q.ck = (dim % c->ck == 0) ? c->ck : 16;
        if (!q.w) { q.w = L.dev((size_t)dim * 128); if (!q.w) return HD_EHIP; }
        HD_TRY(launch_pack_conv((const float*)t->data, q.w, 128, dim, 1, 1, 128, 0, 0, L.st));
        if (!q.wsplit) { q.wsplit = (unsigned short*)L.dev((size_t)dim * 128); if (!q.wsplit) return HD_EHIP; }   // 2 shorts per weight
        HD_TRY(launch_split_conv(q.w, q.wsplit, 1, dim, 128, 16, L.st));
        if (!a->wkv) { a->wkv = (unsigned short*)L.dev((size_t)256 * dim); if (!a->wkv) return HD_EHIP; }
        HD_TRY(launch_pack_kv((const float*)t->data, a->norm_g, dim, a->wkv, L.st));
        if (dim == 64) {
            if (!a->wq) { a->wq = (unsigned short*)L.dev((size_t)128 * dim); if (!a->wq) return HD_EHIP; }   // 128*dim*2 shorts
            HD_TRY(launch_pack_q((const float*)t->data, a->norm_g, dim, a->wq, L.st));
        }
        a->fused = true;
    }
    if (linear) {
        HD_TRY(L.conv(p + ".fn.fn.to_out.0", dim, 128, 1, false, false, true, &a->out));
        HD_TRY(L.vec(p + ".fn.fn.to_out.1.g", {1, dim, 1, 1}, &a->out_g));
    } else {
        HD_TRY(L.conv(p + ".fn.fn.to_out", dim, 128, 1, false, false, true, &a->out));
    }
    return 0;
}

static int unet_film_total(const hd_arch_desc& a) {
    const int per = a.sr3 ? 1 : 2;
    int total = 0, d_prev = a.dim;
    std::vector<int> dims{a.dim};
    for (int i = 0; i < a.n_mults; ++i) dims.push_back(a.dim * a.mults[i]);
    for (int i = 0; i < a.n_mults; ++i) total += 2 * per * dims[i];          // downs: blocks are dim_in -> dim_in
    total += 2 * per * dims.back();                                           // mid blocks
    for (int i = a.n_mults - 1; i >= 0; --i) total += 2 * per * dims[i + 1];  // ups: blocks produce dim_out
    total += per * a.dim;                                                     // final_res_block
    (void)d_prev;
    return total;
}

static int load_common_time(Loader& L, const std::string& first) {
    hd_ctx* c = L.c;
    const hd_named_tensor* w1 = L.get("time_mlp.1.weight", {c->time_dim, c->time_in});
    const hd_named_tensor* w3 = L.get("time_mlp.3.weight", {c->time_dim, c->time_dim});
    if (!w1 || !w3) return HD_ENOWEIGHT;
    if (!c->w1t) { c->w1t = L.dev((size_t)c->time_in * c->time_dim); c->w3t = L.dev((size_t)c->time_dim * c->time_dim); }
    if (!c->w1t || !c->w3t) return HD_EHIP;
    HD_TRY(launch_transpose((const float*)w1->data, c->w1t, c->time_dim, c->time_in, c->time_dim, 0, L.st));
    HD_TRY(launch_transpose((const float*)w3->data, c->w3t, c->time_dim, c->time_dim, c->time_dim, 0, L.st));
    HD_TRY(L.vec("time_mlp.1.bias", {c->time_dim}, &c->b1));
    HD_TRY(L.vec("time_mlp.3.bias", {c->time_dim}, &c->b3));
    HD_TRY(L.vec(first + ".weight", {c->first_cout, c->cin0, c->first_ks, c->first_ks}, &c->first_w));
    HD_TRY(L.vec(first + ".bias", {c->first_cout}, &c->first_b));
    if (!c->film_wt) { c->film_wt = L.dev((size_t)c->time_dim * c->film_n); c->film_b = L.dev(c->film_n); }
    if (!c->film_wt || !c->film_b) return HD_EHIP;
    return 0;
}

static int load_unet(Loader& L) {
    hd_ctx* c = L.c;
    const hd_arch_desc& a = c->arch;
    HD_TRY(load_common_time(L, "init_conv"));
    std::vector<int> dims{a.dim};
    for (int i = 0; i < a.n_mults; ++i) dims.push_back(a.dim * a.mults[i]);
    const int n = a.n_mults;
    c->downs.resize(n); c->ups.resize(n);
    int off = 0;
    for (int i = 0; i < n; ++i) {
        StageW& s = c->downs[i];
        const int di = dims[i], dout = dims[i + 1];
        const std::string p = "downs." + std::to_string(i);
        s.cin = di; s.cout = dout; s.last = i >= n - 1;
        HD_TRY(load_unet_res(L, p + ".0", di, di, &s.r1, &off));
        HD_TRY(load_unet_res(L, p + ".1", di, di, &s.r2, &off));
        HD_TRY(load_attn(L, p + ".2", di, true, &s.attn));
        if (s.last) HD_TRY(L.conv(p + ".3", dout, di, 3, false, false, true, &s.resample));
        else HD_TRY(L.conv(p + ".3.1", dout, di, 1, false, true, true, &s.resample));
    }
    const int mid = dims.back();
    HD_TRY(load_unet_res(L, "mid_block1", mid, mid, &c->mid1, &off));
    HD_TRY(load_attn(L, "mid_attn", mid, false, &c->mid_attn));
    HD_TRY(load_unet_res(L, "mid_block2", mid, mid, &c->mid2, &off));
    for (int i = 0; i < n; ++i) {
        StageW& s = c->ups[i];
        const int di = dims[n - 1 - i], dout = dims[n - i];   // reversed(in_out)[i] = (dim_in, dim_out)
        const std::string p = "ups." + std::to_string(i);
        s.cin = di; s.cout = dout; s.last = i == n - 1;
        HD_TRY(load_unet_res(L, p + ".0", dout + di, dout, &s.r1, &off));
        HD_TRY(load_unet_res(L, p + ".1", dout + di, dout, &s.r2, &off));
        HD_TRY(load_attn(L, p + ".2", dout, true, &s.attn));
        if (s.last) HD_TRY(L.conv(p + ".3", di, dout, 3, false, false, true, &s.resample));
        else HD_TRY(L.conv(p + ".3.1", di, dout, 3, false, false, true, &s.resample));
    }
    HD_TRY(load_unet_res(L, "final_res_block", a.dim * 2, a.dim, &c->final_res, &off));
    HD_TRY(L.vec("final_conv.weight", {1, a.dim, 1, 1}, &c->final_w));
    HD_TRY(L.vec("final_conv.bias", {1}, &c->final_b));
    if (off != c->film_n) return fail(c, HD_EINVAL, "internal: FiLM column count mismatch");
    return 0;
}

static int load_hicedrn(Loader& L) {
    hd_ctx* c = L.c;
    const hd_arch_desc& a = c->arch;
    HD_TRY(load_common_time(L, "head"));
    const int F = a.dim;
    c->body.resize(a.number_resnet);
    int off = 0;
    for (int i = 0; i < a.number_resnet; ++i) {
        ResW& r = c->body[i];
        const std::string p = "body." + std::to_string(i);
        r.name = p; r.cin = r.cout = F; r.has_norm = false;
        HD_TRY(load_film(L, p, a.sr3 ? F : 2 * F, &off, &r.film_off));
        HD_TRY(L.conv(p + ".conv.proj", F, F, 3, false, false, true, &r.c1));
    }
    HD_TRY(L.conv("body_tail", F, F, 3, false, false, true, &c->body_tail));
    HD_TRY(L.conv("tail", 1, F, 3, false, false, true, &c->tail));
    return 0;
}

// ---- forward plans -----------------------------------------------------------------------------
struct Run {
    hd_ctx* c; hipStream_t st; bool dry;
    Pool* pool;
    int B, S;
    // time embedding rows: Bt == B (per-sample t) or 1 (every tile at the same step)
    int Bt; const float* film; int film_bs;
    int alloc(size_t n, float** p) {
        if (!pool->alloc(n * sizeof(float), p)) return fail(c, HD_ENOMEM, "workspace too small: call hd_reserve(ctx, B, S) with the batch and tile size first");
        return 0;
    }
    int act(int H, int W, int C, Act* a) { a->B = B; a->H = H; a->W = W; a->C = C; return alloc(a->numel(), &a->p); }
    void free(float* p) { pool->release(p); }
    void free(Act& a) { pool->release(a.p); a.p = nullptr; }
};

static int probe(Run& r, const std::string& label, const Act& a) {
    hd_ctx* c = r.c;
    if (r.dry || !c->capture) return 0;
    Act copy = a;
    void* p = nullptr;
    if (hipMalloc(&p, a.numel() * sizeof(float)) != hipSuccess) return fail(c, HD_EHIP, "hipMalloc(capture) failed");
    if (hipMemcpyAsync(p, a.p, a.numel() * sizeof(float), hipMemcpyDeviceToDevice, r.st) != hipSuccess) return fail(c, HD_EHIP, "capture copy failed");
    copy.p = (float*)p;
    auto it = c->captured.find(label);
    if (it != c->captured.end()) { (void)hipFree(it->second.p); }
    c->captured[label] = copy;
    return 0;
}

// Activation slice of a launch.  3x3 layers with at most 64 output channels take 16-channel slices on maps of >= 4096 pixels when their
// loader is the plain one: their 256 x 64 tile then needs ~40 KB of LDS and (registers capped at 168, conv_bf16x3_kernel.h) a third
// workgroup fits a CU -- 271 -> 258 us on the dominant unet64 kernel in round 2; the transforming loaders need more registers than three
// workgroups per CU leave, and smaller maps keep 32 (unet40: 3.07 vs 3.09 ms).  A rule by the layer, its loader and the map size only.
// HICDIFF_CK16_NARROW=0 turns it off.
static void pick_slices(ConvArgs& a) {
    static const bool ck16n = !(getenv("HICDIFF_CK16_NARROW") && atoi(getenv("HICDIFF_CK16_NARROW")) == 0);
    if (ck16n && a.cw.KH == 3 && a.cw.KW == 3 && a.cw.CoutPad == 64 && a.cw.ck == 32 && a.H * a.W >= 4096 && a.in_mode == IN_NONE) a.cw.ck = 16;
}

// The arithmetic of one 3x3 layer under the context's level: 1 / 2 = two / one fp16 products everywhere, 3 = two products on the maps of at
// most (S/4)^2 pixels only (the "low" layer class of the error-budget study: the layers whose error the LATE half of a chain tolerates).
static int layer_f16(const Run& r, const ConvArgs& a) {
    const int lvl = r.c->f16w2;
    if (lvl != 3) return lvl;
    return (long long)a.H * a.W * 16 <= (long long)r.S * r.S ? 1 : 0;
}

static int run_conv(Run& r, ConvArgs& a) {
    pick_slices(a);
    a.precision = r.c->precision;
    a.f16w2 = layer_f16(r, a);
    // the dry run sizes the workspace for either arithmetic (hd_set_precision may switch later): plan the split as the fast path would
    ConvArgs probe = a; probe.precision = HD_PREC_BF16X3;
    const int ks = conv_splitk(probe);
    float* ws = nullptr;
    if (ks > 1) HD_TRY(r.alloc((size_t)ks * a.B * a.H * a.W * a.cw.Cout, &ws));
    a.splitk_ws = ws;
    const int rc = r.dry ? 0 : launch_conv(a, r.st, nullptr);
    if (ws) r.free(ws);          // stream order: whatever reuses the block runs after the reduce kernel
    return rc;
}

// GroupNorm'd conv of the UNet: conv (+ fused per-channel partial sums when the tile geometry allows)
// followed by gn_finalize -> per-(sample, channel) affine (A, Bv[, E]).
static int conv_gn(Run& r, ConvArgs& a, int C, const float* gamma, const float* beta, int film_mode, int film_off,
                   float** A, float** Bv, float** E) {
    const int HW = a.H * a.W;
    pick_slices(a);
    a.precision = r.c->precision;
    a.f16w2 = layer_f16(r, a);
    int slots = conv_gn_slots(a);
    const bool fused = slots > 0;
    if (!fused) slots = (HW + 255) / 256;
    float* part = nullptr;
    HD_TRY(r.alloc((size_t)a.B * slots * C * 2, &part));
    a.gn_part = fused ? part : nullptr;
    HD_TRY(r.alloc((size_t)a.B * C, A));
    HD_TRY(r.alloc((size_t)a.B * C, Bv));
    *E = nullptr;
    if (film_mode == 2) HD_TRY(r.alloc((size_t)a.B * C, E));
    // where one workgroup of the convolution (or of its split-K reduce) sees a whole sample, it writes the affine itself: no gn_finalize launch
    a.gn_fin.gamma = gamma; a.gn_fin.beta = beta; a.gn_fin.film = r.film; a.gn_fin.film_bs = r.film_bs; a.gn_fin.film_off = film_off;
    a.gn_fin.film_mode = film_mode; a.gn_fin.groups = r.c->arch.groups; a.gn_fin.A = *A; a.gn_fin.Bv = *Bv; a.gn_fin.E = *E;
    HD_TRY(run_conv(r, a));
    if (!fused && !r.dry) { int s2 = 0; HD_TRY(launch_gn_partial(a.out, a.B, HW, C, part, &s2, r.st)); }
    if (!r.dry && !(fused && conv_gn_direct(a)))
        HD_TRY(launch_gn_finalize(part, slots, a.B, HW, C, r.c->arch.groups, gamma, beta, r.film, r.film_bs, film_off, film_mode, *A,
                                  *Bv, *E, r.st));
    r.free(part);
    return 0;
}

// ResnetBlock (src/hicdiff.py:185-197; SR3: src/hicdiff_sr3.py:246-251).  in1 != null: channel concat.
// stats (optional): [pixels][2] buffer for the channel-LayerNorm statistics of the block's output, for the attention block
// that consumes it; *stats_done is set when the block's tail produced them (C = 64 / 128 [/ 256 with identity shortcut]).
static int unet_resblock(Run& r, const ResW& w, const Act& in0, const Act* in1, Act* out, float* stats = nullptr, bool* stats_done = nullptr) {
    const int H = in0.H, W = in0.W, C = w.cout;
    const bool sr3 = r.c->arch.sr3;
    Act h1, h2;
    HD_TRY(r.act(H, W, C, &h1));
    ConvArgs a;
    a.in0 = in0.p; a.C0 = in0.C; a.in1 = in1 ? in1->p : nullptr; a.C1 = in1 ? in1->C : 0;
    a.B = r.B; a.H = H; a.W = W; a.IH = H; a.IW = W; a.stride = 1; a.pad = 1; a.cw = w.c1; a.out = h1.p;
    float *A1, *B1, *E1;
    HD_TRY(conv_gn(r, a, C, w.g1, w.b1, sr3 ? 2 : 1, w.film_off, &A1, &B1, &E1));
    HD_TRY(r.act(H, W, C, &h2));
    ConvArgs b;
    b.in0 = h1.p; b.C0 = C; b.B = r.B; b.H = H; b.W = W; b.IH = H; b.IW = W; b.stride = 1; b.pad = 1; b.cw = w.c2; b.out = h2.p;
    b.in_mode = IN_AFFINE_SILU; b.inA = A1; b.inB = B1; b.inE = E1; b.in_bstride = C;
    float *A2, *B2, *E2;
    HD_TRY(conv_gn(r, b, C, w.g2, w.b2, 0, 0, &A2, &B2, &E2));
    if (r.c->capture && !r.dry) {      // test-only: the block's internals (per-sample affines as (B, 1, 1, C) maps)
        auto vec = [&](float* p) { Act v{}; v.B = r.B; v.H = 1; v.W = 1; v.C = C; v.p = p; return v; };
        HD_TRY(probe(r, w.name + ".h1", h1)); HD_TRY(probe(r, w.name + ".A1", vec(A1))); HD_TRY(probe(r, w.name + ".B1", vec(B1)));
        HD_TRY(probe(r, w.name + ".h2", h2)); HD_TRY(probe(r, w.name + ".A2", vec(A2))); HD_TRY(probe(r, w.name + ".B2", vec(B2)));
    }
    r.free(h1); r.free(A1); r.free(B1); r.free(E1);
    HD_TRY(r.act(H, W, C, out));
    if (w.has_res) {
        ConvArgs s;
        s.in0 = in0.p; s.C0 = in0.C; s.in1 = in1 ? in1->p : nullptr; s.C1 = in1 ? in1->C : 0;
        s.B = r.B; s.H = H; s.W = W; s.IH = H; s.IW = W; s.stride = 1; s.pad = 0; s.cw = w.res; s.out = out->p;
        s.ep = EP_RES_AFFINE_SILU; s.res = h2.p; s.resA = A2; s.resB = B2; s.res_bstride = C;
        if (stats && !r.dry && r.c->precision == HD_PREC_BF16X3 && (C == 64 || C == 128)) {   // tile width == C there
            s.ep |= EP_LN_STATS; s.ln_stats_out = stats;
            if (stats_done) *stats_done = true;
        }
        HD_TRY(run_conv(r, s));
    } else if (!r.dry) {
        // (the exact-fp32 arithmetic keeps its separately validated ln_stats pass: DDIM amplifies 1e-7 reorderings to 1e-3)
        const int rc = launch_affine_silu_add(h2.p, A2, B2, in0.p, out->p, r.B, H * W, C, r.st, r.c->precision == HD_PREC_BF16X3 ? stats : nullptr);
        if (rc < 0) return keep_err(r.c, rc);
        if (rc == 1 && stats_done) *stats_done = true;
    }
    r.free(h2); r.free(A2); r.free(B2);
    if (r.c->capture && !r.dry) HD_TRY(probe(r, w.name + ".out", *out));
    return 0;
}

// Residual(PreNorm(LinearAttention)) / Residual(PreNorm(Attention)), src/hicdiff.py:199-251.
// stats_in: the PreNorm statistics of x when the producing block already wrote them (ownership passes to this call).
static int unet_attention(Run& r, const AttnW& w, const Act& x, Act* out, float* stats_in = nullptr, bool stats_ready = false) {
    const int H = x.H, W = x.W, C = x.C, HW = H * W, heads = 4;
    const size_t P = x.pixels();
    float* stats = stats_in;
    if (!stats) HD_TRY(r.alloc(P * 2, &stats));
    if (!r.dry && !stats_ready) HD_TRY(launch_ln_stats(x.p, P, C, stats, r.st));
    const bool fused = w.linear && w.fused && r.c->precision == HD_PREC_BF16X3;
    // chained q side (linattn_q_fused.hip): 64-channel maps of >= 256 pixels -- q, the attention output and the
    // pre-LayerNorm tensor never exist; the kernel reads x and the statistics and writes the block's output
    const bool qchain = fused && C == 64 && HW >= 256 && w.wq && w.out.CoutPad == 64;
    // (the dry run takes exactly the allocations of the real one -- same sizes, same order -- for the arithmetic mode and time-row form it is
    // given; dry_bytes sizes the workspace over every combination: a first-fit pool only replays a plan it was sized with)
    Act qkv{};
    if (!qchain) {
        HD_TRY(r.act(H, W, fused ? 128 : 384, &qkv));
        ConvArgs q;
        q.in0 = x.p; q.C0 = C; q.B = r.B; q.H = H; q.W = W; q.IH = H; q.IW = W; q.stride = 1; q.pad = 0; q.cw = fused ? w.qonly : w.qkv; q.out = qkv.p;
        q.in_mode = IN_LAYERNORM; q.ln_stats = stats; q.ln_g = w.norm_g;
        if (r.c->capture) { Act sa{}; sa.B = r.B; sa.H = H; sa.W = W; sa.C = 2; sa.p = stats; HD_TRY(probe(r, w.name + ".ln_stats", sa)); }
        HD_TRY(run_conv(r, q));
        HD_TRY(probe(r, w.name + ".q", qkv));
        r.free(stats); stats = nullptr;
    }
    // q side, fused form (split-bf16 arithmetic, feature maps of >= 256 pixels so a conv tile never mixes samples):
    // the context is folded into to_out's weight per sample and the q-softmax runs in to_out's loader, so neither the
    // attention output nor (for C = 64 / 128, where a tile holds whole channel rows) the pre-LayerNorm tensor exists.
    const bool qfuse = fused && HW >= 256 && w.out.ck == 32;
    const bool lnfuse = qfuse && (C == 64 || C == 128);
    Act att{};
    if (!qfuse && !qchain) HD_TRY(r.act(H, W, 128, &att));
    unsigned short* wfold = nullptr;
    const size_t wfold_bytes = (size_t)heads * w.out.CoutPad * 64 * sizeof(unsigned short);   // per sample
    if (fused) {
        const int nsplit = linattn_kv_nsplit(HW);
        const size_t slots = (size_t)r.B * heads * nsplit;
        float *ctx, *scr;
        HD_TRY(r.alloc((size_t)r.B * heads * 32 * 32, &ctx));
        HD_TRY(r.alloc(std::max(slots * (32 + 32 + 32 * 32), linattn_scratch_floats(r.B, HW, heads)), &scr));
        if (qfuse || qchain) { float* wf; HD_TRY(r.alloc((size_t)r.B * wfold_bytes / sizeof(float), &wf)); wfold = (unsigned short*)wf; }
        if (!r.dry) {
            const int nparts = linattn_kv_nparts(HW, C);                  // the workspace keeps its per-chunk size; the merging kernel fills a prefix
            const size_t pslots = (size_t)r.B * heads * nparts;
            float* pmax = scr; float* psum = pmax + pslots * 32; float* pctx = psum + pslots * 32;
            HD_TRY(launch_linattn_kv_fused(x.p, w.wkv, r.B, HW, C, pmax, psum, pctx, r.st));
            HD_TRY(launch_linattn_combine(pmax, psum, pctx, r.B, heads, nparts, HW, ctx, r.st));
            if (r.c->capture) { Act ca{}; ca.B = r.B; ca.H = heads; ca.W = 32; ca.C = 32; ca.p = ctx; HD_TRY(probe(r, w.name + ".ctx", ca)); }
            if (qfuse || qchain) HD_TRY(launch_linattn_fold_out(w.out.w, ctx, r.B, w.out.CoutPad, wfold, r.st, qchain ? 1 : 0));
            else { HD_TRY(launch_linattn_apply(qkv.p, 128, ctx, r.B, HW, heads, att.p, r.st)); HD_TRY(probe(r, w.name + ".att", att)); }
        }
        r.free(ctx); r.free(scr);
    } else if (w.linear) {
        float* ctx; HD_TRY(r.alloc((size_t)r.B * heads * 32 * 32, &ctx));
        float* scr; HD_TRY(r.alloc(std::max(linattn_scratch_floats(r.B, HW, heads), (size_t)r.B * heads * linattn_kv_nsplit(HW) * (32 + 32 + 32 * 32)), &scr));
        if (!r.dry) {
            HD_TRY(launch_linattn_context(qkv.p, r.B, HW, heads, scr, ctx, r.st));
            HD_TRY(launch_linattn_apply(qkv.p, 384, ctx, r.B, HW, heads, att.p, r.st));
        }
        r.free(ctx); r.free(scr);
    } else if (!r.dry) {
        HD_TRY(launch_attn_full(qkv.p, r.B, HW, heads, att.p, r.st));
    }
    if (qchain) {
        HD_TRY(r.act(H, W, C, out));
        if (!r.dry) HD_TRY(launch_linattn_q_fused(x.p, stats, w.wq, wfold, w.out.bias, w.out_g, out->p, r.B, HW, C, r.st));
        r.free(stats);
        if (wfold) r.free((float*)wfold);
        if (att.p) r.free(att);
        return 0;
    }
    if (!qfuse) r.free(qkv);
    HD_TRY(r.act(H, W, C, out));
    ConvArgs o;
    o.in0 = att.p; o.C0 = 128; o.B = r.B; o.H = H; o.W = W; o.IH = H; o.IW = W; o.stride = 1; o.pad = 0; o.cw = w.out;
    if (qfuse) {
        o.in0 = qkv.p; o.in_mode = IN_SOFTMAX32; o.cw.wsplit = wfold; o.w_bstride = wfold_bytes;
    }
    if (w.linear) {
        Act y{};
        if (!lnfuse) HD_TRY(r.act(H, W, C, &y));
        if (lnfuse) {
            o.out = out->p; o.ep = EP_LN_RES; o.ep_ln_g = w.out_g; o.res = x.p;
            HD_TRY(run_conv(r, o));
        } else {
            o.out = y.p;
            HD_TRY(run_conv(r, o));
            HD_TRY(probe(r, w.name + ".y", y));
            if (!r.dry) HD_TRY(launch_ln_residual(y.p, w.out_g, x.p, out->p, P, C, r.st));
        }
        if (y.p) r.free(y);
    } else {
        o.out = out->p; o.ep = EP_RES; o.alpha = 1.f; o.res = x.p;
        HD_TRY(run_conv(r, o));
    }
    if (qfuse) r.free(qkv);
    if (wfold) r.free((float*)wfold);
    if (att.p) r.free(att);
    return 0;
}

static int time_and_film(Run& r, const void* t, int t_kind, float tval, bool uniform, const StepParams* sp, float** film_out) {
    hd_ctx* c = r.c;
    r.Bt = uniform ? 1 : r.B;
    float *temb, *tact, *film;
    HD_TRY(r.alloc((size_t)r.Bt * c->time_dim, &temb));
    HD_TRY(r.alloc((size_t)r.Bt * c->time_dim, &tact));
    HD_TRY(r.alloc((size_t)r.Bt * c->film_n, &film));
    if (!r.dry) {
        if (uniform && c->precision == HD_PREC_BF16X3 && c->time_dim % 256 == 0 && c->time_dim <= 1024 && c->time_in <= 256 && c->time_in % 16 == 0 && c->film_n % 4 == 0) {   // one time row: MLP + every FiLM projection in one launch (the exact-fp32 mode keeps the two kernels whose summation order its parity margins were measured with: DDIM amplifies 1e-7 to 1e-3)
            HD_TRY(launch_time_film(tval, sp, c->arch.sr3, c->time_in, c->time_dim, c->w1t, c->b1, c->w3t, c->b3, c->film_wt, c->film_b, c->film_n, film, r.st));
        } else {
            HD_TRY(launch_time_mlp(uniform ? nullptr : t, t_kind, tval, sp, c->arch.sr3, r.Bt, c->time_in, c->time_dim, c->w1t, c->b1, c->w3t,
                                   c->b3, temb, tact, r.st));
            HD_TRY(launch_film(tact, r.Bt, c->time_dim, c->film_wt, c->film_b, c->film_n, film, r.st));
        }
    }
    r.free(temb); r.free(tact);
    r.film = film; r.film_bs = uniform ? 0 : c->film_n;
    *film_out = film;
    return 0;
}

static int unet_forward(Run& r, const float* x, const float* cond, float* eps) {
    hd_ctx* c = r.c;
    const int S = r.S, n = c->arch.n_mults;
    Act h0; HD_TRY(r.act(S, S, c->arch.dim, &h0));
    if (!r.dry) HD_TRY(launch_conv_small_cin(x, cond, c->first_w, c->first_b, h0.p, r.B, S, 7, c->cin0, c->arch.dim, r.st, c->precision == HD_PREC_BF16X3));
    HD_TRY(probe(r, "init_conv", h0));
    std::vector<Act> skips;
    Act cur = h0;
    for (int i = 0; i < n; ++i) {
        const StageW& s = c->downs[i];
        Act a1, a2, a3, d;
        HD_TRY(unet_resblock(r, s.r1, cur, nullptr, &a1));
        if (cur.p != h0.p) r.free(cur);
        skips.push_back(a1);
        HD_TRY(probe(r, "downs." + std::to_string(i) + ".0", a1));
        float* st; bool st_done = false;
        HD_TRY(r.alloc(a1.pixels() * 2, &st));
        HD_TRY(unet_resblock(r, s.r2, a1, nullptr, &a2, st, &st_done));
        HD_TRY(unet_attention(r, s.attn, a2, &a3, st, st_done));
        HD_TRY(probe(r, "downs." + std::to_string(i) + ".2", a3));
        r.free(a2);
        skips.push_back(a3);
        ConvArgs k;
        k.in0 = a3.p; k.C0 = a3.C; k.B = r.B; k.IH = a3.H; k.IW = a3.W; k.cw = s.resample;
        if (s.last) { k.H = a3.H; k.W = a3.W; k.stride = 1; k.pad = 1; }
        else { k.H = a3.H / 2; k.W = a3.W / 2; k.stride = 2; k.pad = 0; }
        HD_TRY(r.act(k.H, k.W, s.cout, &d));
        k.out = d.p;
        HD_TRY(run_conv(r, k));
        HD_TRY(probe(r, "downs." + std::to_string(i), d));
        cur = d;
    }
    {
        Act m1, m2, m3;
        HD_TRY(unet_resblock(r, c->mid1, cur, nullptr, &m1)); r.free(cur);
        HD_TRY(unet_attention(r, c->mid_attn, m1, &m2)); r.free(m1);
        HD_TRY(probe(r, "mid_attn", m2));
        HD_TRY(unet_resblock(r, c->mid2, m2, nullptr, &m3)); r.free(m2);
        HD_TRY(probe(r, "mid", m3));
        cur = m3;
    }
    for (int i = 0; i < n; ++i) {
        const StageW& s = c->ups[i];
        Act a1, a2, a3, u;
        Act sk = skips.back(); skips.pop_back();
        HD_TRY(unet_resblock(r, s.r1, cur, &sk, &a1)); r.free(cur); r.free(sk);
        sk = skips.back(); skips.pop_back();
        float* st; bool st_done = false;
        HD_TRY(r.alloc(a1.pixels() * 2, &st));
        HD_TRY(unet_resblock(r, s.r2, a1, &sk, &a2, st, &st_done)); r.free(a1); r.free(sk);
        HD_TRY(unet_attention(r, s.attn, a2, &a3, st, st_done)); r.free(a2);
        ConvArgs k;
        k.in0 = a3.p; k.C0 = a3.C; k.B = r.B; k.IH = a3.H; k.IW = a3.W; k.cw = s.resample; k.stride = 1; k.pad = 1;
        if (s.last) { k.H = a3.H; k.W = a3.W; } else { k.H = a3.H * 2; k.W = a3.W * 2; k.upsample = 1; }
        HD_TRY(r.act(k.H, k.W, s.cin, &u));
        k.out = u.p;
        HD_TRY(run_conv(r, k));
        r.free(a3);
        HD_TRY(probe(r, "ups." + std::to_string(i), u));
        cur = u;
    }
    Act f;
    HD_TRY(unet_resblock(r, c->final_res, cur, &h0, &f));
    r.free(cur); r.free(h0);
    HD_TRY(probe(r, "final_res", f));
    if (!r.dry) HD_TRY(launch_rowdot(f.p, c->final_w, c->final_b, eps, f.pixels(), f.C, r.st));
    r.free(f);
    return 0;
}

static int hicedrn_forward(Run& r, const float* x, const float* cond, float* eps) {
    hd_ctx* c = r.c;
    const int S = r.S, F = c->arch.dim;
    const bool sr3 = c->arch.sr3;
    Act head; HD_TRY(r.act(S, S, F, &head));
    if (!r.dry) HD_TRY(launch_conv_small_cin(x, cond, c->first_w, c->first_b, head.p, r.B, S, 3, c->cin0, F, r.st, c->precision == HD_PREC_BF16X3));
    HD_TRY(probe(r, "head", head));
    Act cur = head;
    for (size_t i = 0; i < c->body.size(); ++i) {
        const ResW& w = c->body[i];
        Act h, nx;
        HD_TRY(r.act(S, S, F, &h));
        ConvArgs a;
        a.in0 = cur.p; a.C0 = F; a.B = r.B; a.H = S; a.W = S; a.IH = S; a.IW = S; a.stride = 1; a.pad = 1; a.cw = w.c1; a.out = h.p;
        a.ep_bstride = r.film_bs;
        if (sr3) { a.ep = EP_ADD_SILU; a.epShift = r.film + w.film_off; }
        else { a.ep = EP_FILM_SILU; a.epScale = r.film + w.film_off; a.epShift = r.film + w.film_off + F; }
        HD_TRY(run_conv(r, a));
        HD_TRY(r.act(S, S, F, &nx));
        ConvArgs b;
        b.in0 = h.p; b.C0 = F; b.B = r.B; b.H = S; b.W = S; b.IH = S; b.IW = S; b.stride = 1; b.pad = 1; b.cw = w.c1; b.out = nx.p;
        b.ep = EP_RES; b.alpha = 0.1f; b.res = cur.p;
        HD_TRY(run_conv(r, b));
        r.free(h);
        if (cur.p != head.p) r.free(cur);
        cur = nx;
        if (i == 0 || i + 1 == c->body.size()) HD_TRY(probe(r, "body." + std::to_string(i), nx));
    }
    Act y; HD_TRY(r.act(S, S, F, &y));
    ConvArgs t;
    t.in0 = cur.p; t.C0 = F; t.B = r.B; t.H = S; t.W = S; t.IH = S; t.IW = S; t.stride = 1; t.pad = 1; t.cw = c->body_tail; t.out = y.p;
    t.ep = EP_RES; t.alpha = 1.f; t.res = head.p;
    HD_TRY(run_conv(r, t));
    if (cur.p != head.p) r.free(cur);
    r.free(head);
    ConvArgs o;
    o.in0 = y.p; o.C0 = F; o.B = r.B; o.H = S; o.W = S; o.IH = S; o.IW = S; o.stride = 1; o.pad = 1; o.cw = c->tail; o.out = eps;
    HD_TRY(run_conv(r, o));
    r.free(y);
    return 0;
}

static int forward(hd_ctx* c, const float* x, const void* t, int t_kind, float tval, bool uniform, const float* cond, float* eps,
                   int B, int S, hipStream_t st, bool dry, const StepParams* sp = nullptr, Pool* pool = nullptr) {
    if (B < 1 || S < 8) return fail(c, HD_EINVAL, "bad batch or tile size");
    if (c->arch.kind == HD_ARCH_UNET) {
        int div = 1; for (int i = 0; i + 1 < c->arch.n_mults; ++i) div *= 2;
        if (S % div) return fail(c, HD_EINVAL, "tile size must be divisible by 2^(len(dim_mults)-1)");
    }
    if ((S * S) % 4) return fail(c, HD_EINVAL, "S*S must be a multiple of 4");
    if (!dry && !c->loaded) return fail(c, HD_ESTATE, "weights not loaded: call hd_load_weights first");
    if (!dry && (c->arch.self_condition != 0) != (cond != nullptr)) return fail(c, HD_EINVAL, "cond must be given iff self_condition");
    if (!pool) pool = &c->pool;
    Run r{c, st, dry, pool, B, S, B, nullptr, 0};
    pool->reset(dry);
    float* film = nullptr;
    HD_TRY(time_and_film(r, t, t_kind, tval, uniform, sp, &film));
    int rc = c->arch.kind == HD_ARCH_UNET ? unet_forward(r, x, cond, eps) : hicedrn_forward(r, x, cond, eps);
    r.free(film);
    return rc;
}

static const char* const kCoefSizeMsg =
    "coefficient struct: struct_bytes is not a size this library knows (set it to sizeof of your struct; bindings written before ABI revision 3 "
    "lack the field -- see include/hicdiff_hip.h)";

bool hd_raise_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return true;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done.insert({dev, kernel});
    return true;
}

// ---- C ABI -------------------------------------------------------------------------------------
extern "C" {

const char* hd_version(void) { return "hicdiff_hip 0.3 (gfx950; split-bf16 x3 / exact fp32 MFMA)"; }
int hd_abi_version(void) { return HD_ABI_VERSION; }

const char* hd_last_error(const hd_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int hd_create(hd_ctx** out, int device, const hd_arch_desc* a) {
    if (!out || !a) return HD_EINVAL;
    *out = nullptr;
    if (a->channels != 1) { g_err = "only channels == 1 is supported"; return HD_EINVAL; }
    if (a->kind == HD_ARCH_UNET) {
        if (a->dim < 16 || a->dim % 16 || a->n_mults < 1 || a->n_mults > 8 || a->groups < 1 || a->dim % a->groups) {
            g_err = "Unet: dim must be a multiple of 16 (and of groups), 1 <= len(dim_mults) <= 8"; return HD_EINVAL;
        }
        for (int i = 0; i < a->n_mults; ++i) if (a->mults[i] < 1) { g_err = "Unet: dim_mults must be positive"; return HD_EINVAL; }
    } else if (a->kind == HD_ARCH_HICEDRN) {
        if (a->dim < 16 || a->dim % 16 || a->number_resnet < 1) { g_err = "hicedrn: n_feat must be a multiple of 16"; return HD_EINVAL; }
    } else { g_err = "unknown architecture kind"; return HD_EINVAL; }
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return HD_EHIP; }
    hd_ctx* c = new hd_ctx();
    c->arch = *a; c->device = device;
    c->cin0 = a->self_condition ? 2 : 1;
    c->time_in = a->dim; c->time_dim = a->dim * 4;
    c->ck = (a->dim % 32 == 0) ? 32 : 16;
    if (const char* k = getenv("HICDIFF_CK")) { if (atoi(k) == 16) c->ck = 16; }   // tuning experiments
    if (const char* g = getenv("HICDIFF_GRAPHS")) c->use_graphs = atoi(g) != 0 ? 1 : 0;
    if (const char* g = getenv("HICDIFF_CHAINS")) c->chains = std::max(1, std::min(atoi(g), (int)hd_ctx::MAX_LANES));
    if (const char* g = getenv("HICDIFF_LANE_DELAY_US")) c->lane_delay_us = std::max(0, std::min(atoi(g), 100000));
    if (const char* e = getenv("HICDIFF_PRECISION")) c->precision = (std::string(e) == "f32") ? HD_PREC_F32 : HD_PREC_BF16X3;
    if (a->kind == HD_ARCH_UNET) { c->first_ks = 7; c->first_cout = a->dim; c->film_n = unet_film_total(*a); }
    else { c->first_ks = 3; c->first_cout = a->dim; c->film_n = a->number_resnet * (a->sr3 ? 1 : 2) * a->dim; }
    *out = c;
    return HD_OK;
}

void hd_destroy(hd_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (void* p : c->owned) (void)hipFree(p);
    if (c->pool.base) (void)hipFree(c->pool.base);
    if (c->eps_buf) (void)hipFree(c->eps_buf);
    for (auto& L : c->lane) {
        L.drop_graphs();
        if (L.pool.base) (void)hipFree(L.pool.base);
        if (L.sp_dev) (void)hipFree(L.sp_dev);
        if (L.ev_out) (void)hipEventDestroy(L.ev_out);
        if (L.st) (void)hipStreamDestroy(L.st);
    }
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    delete c;
}

int hd_load_weights(hd_ctx* c, const hd_named_tensor* tensors, int n, void* stream) {
    if (!c || !tensors) return HD_EINVAL;
    Loader L{c, (hipStream_t)stream, {}};
    for (int i = 0; i < n; ++i) L.map[tensors[i].name] = &tensors[i];
    int rc = c->arch.kind == HD_ARCH_UNET ? load_unet(L) : load_hicedrn(L);
    if (rc == 0) rc = L.finish_f16_range();
    if (rc == 0) c->loaded = true;
    // the allocation plan follows what was loaded (which attention blocks have fused weight images): a workspace reserved before the
    // first load is re-sized for the plan as it is now (grow-only; a no-op in the usual order, load then reserve)
    if (rc == 0 && c->resB > 0 && !c->in_chain) rc = hd_reserve(c, c->resB, c->resS);
    return keep_err(c, rc);
}

// Replay threshold: steps of at least this many pixels (B * S * S) are replayed from hipGraphs, and as TWO half-batch chains; smaller ones are
// launched eagerly as one batch.  Measured (profiles/r04_y_replay_threshold.txt): 64 tiles of 40 x 40 (102 k) and 32 of 64 x 64 (131 k) run
// 2 % faster eagerly, 96 / 128 / 160 tiles of 40 x 40 (154 k / 205 k / 256 k) run 8 / 7 / 13 % faster as two replayed chains -- at those sizes
// a kernel does not fill the chip and the two chains' launches really run side by side.  (Round 3's threshold, 256 k for one replayed chain, was
// the cross-over of eager launches against ONE graph.)
static const long long kReplayMinPixels = 150000;

// How a replayed step of B tiles of S x S is cut: two half-batch lanes (measured: profiles/r04_a_*, r04_y_*), one
// whole-batch lane below.  A rule by the amount of work only; sub-batches start at even tiles so that kernels that pair images keep their pairs.
static int lanes_for(const hd_ctx* c, int B, int S) {
    if (c->chains >= 0) return std::max(1, std::min(c->chains, B / 2));
    return B >= 4 && (long long)B * S * S >= kReplayMinPixels ? 2 : 1;
}
static int lane_start(int B, int n, int i) { return i >= n ? B : (int)(((long long)B * i / n + 1) & ~1LL); }      // first tile of lane i of n

static int dry_bytes(hd_ctx* c, int B, int S, size_t* out) {
    // A first-fit pool replays a plan only if it was sized with that plan: the allocations differ between the two arithmetic modes
    // (hd_set_precision may switch later) and between one time row (the sampler steps) and one per tile (hd_eps_forward).
    const int saved = c->precision;
    size_t high = 0;
    int rc = 0;
    for (int prec : {HD_PREC_BF16X3, HD_PREC_F32}) {
        for (bool uniform : {false, true}) {
            Pool scratch;
            c->precision = prec;
            rc = forward(c, nullptr, nullptr, HD_T_FLOAT32, 0.f, uniform, nullptr, nullptr, B, S, nullptr, true, nullptr, &scratch);
            if (rc != 0) break;
            high = std::max(high, scratch.high);
        }
        if (rc != 0) break;
    }
    c->precision = saved;
    if (rc != 0) return keep_err(c, rc);
    *out = high;
    return HD_OK;
}

int hd_workspace_bytes(const hd_ctx* cc, int B, int S, size_t* out) {
    hd_ctx* c = const_cast<hd_ctx*>(cc);
    if (!c || !out) return HD_EINVAL;
    return dry_bytes(c, B, S, out);
}

static int grow_block(hd_ctx* c, Pool& pool, size_t need, bool* synced) {
    if (need <= pool.cap) return HD_OK;
    if (!*synced) {
        if (hipDeviceSynchronize() != hipSuccess) return fail(c, HD_EHIP, "device synchronize failed");
        *synced = true;
        for (auto& L : c->lane) L.drop_graphs();     // captured addresses die with the old block
    }
    if (pool.base) (void)hipFree(pool.base);
    pool.base = nullptr; pool.cap = 0;
    void* p = nullptr;
    if (hipMalloc(&p, need) != hipSuccess) return fail(c, HD_EHIP, "hipMalloc(workspace) failed");
    pool.base = (char*)p; pool.cap = need;
    return HD_OK;
}

int hd_reserve(hd_ctx* c, int B, int S) {
    if (!c) return HD_EINVAL;
    if (c->in_chain) return fail(c, HD_ESTATE, "hd_reserve inside an hd_chain_begin / hd_chain_end bracket");
    size_t need = 0;
    HD_TRY(dry_bytes(c, B, S, &need));
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, HD_EHIP, "hipSetDevice failed");
    bool synced = false;
    HD_TRY(grow_block(c, c->pool, need, &synced));
    if (const int nl = lanes_for(c, B, S); nl > 1) {
        for (int i = 0; i < nl; ++i) {
            size_t ni = 0;
            HD_TRY(dry_bytes(c, lane_start(B, nl, i + 1) - lane_start(B, nl, i), S, &ni));
            HD_TRY(grow_block(c, c->lane[i].pool, ni, &synced));
        }
    }
    const size_t en = (size_t)B * S * S;
    if (en > c->eps_cap) {
        if (!synced) {
            if (hipDeviceSynchronize() != hipSuccess) return fail(c, HD_EHIP, "device synchronize failed");
            for (auto& L : c->lane) L.drop_graphs();
        }
        if (c->eps_buf) (void)hipFree(c->eps_buf);
        c->eps_buf = nullptr; c->eps_cap = 0;
        void* p = nullptr;
        if (hipMalloc(&p, en * sizeof(float)) != hipSuccess) return fail(c, HD_EHIP, "hipMalloc(eps) failed");
        c->eps_buf = (float*)p; c->eps_cap = en;
    }
    c->resB = std::max(c->resB, B); c->resS = std::max(c->resS, S);
    return HD_OK;
}

int hd_eps_forward(hd_ctx* c, const float* x, const void* t, int t_kind, const float* cond, float* eps, int B, int S, void* stream) {
    if (!c || !x || !t || !eps) return HD_EINVAL;
    return keep_err(c, forward(c, x, t, t_kind, 0.f, false, cond, eps, B, S, (hipStream_t)stream, false));
}

// One fused sampler step: eps-net + update.  kind 0: ancestral (aux = cond), 1: DDRM (aux = y).
static int step_body(hd_ctx* c, int kind, float* x, const float* aux, const float* noise, const StepParams& v, float* x0_out,
                     int B, int S, hipStream_t st, const StepParams* sp, Pool* pool = nullptr, float* eps = nullptr) {
    const float* cond = kind == 0 ? aux : nullptr;
    if (!eps) eps = c->eps_buf;
    HD_TRY(forward(c, x, nullptr, HD_T_FLOAT32, v.f[0], true, cond, eps, B, S, st, false, sp, pool));
    if (kind == 0)
        return launch_ddpm_update(x, eps, noise, v.f[1], v.f[2], v.f[3], v.f[4], v.f[5], x0_out, B, S, v.seed, v.tile_off, v.step, sp, st, v.f[6]);
    return launch_ddrm_update(x, eps, aux, noise, v.f[1], v.f[2], v.f[3], v.f[4], v.f[5], v.f[6], v.f[7], v.f[8], x0_out, B, S, v.seed,
                              v.tile_off, v.step, sp, st);
}

static int lanes_setup(hd_ctx* c) {
    if (c->ev_in) return HD_OK;
    for (auto& L : c->lane) {
        if (hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&L.ev_out, hipEventDisableTiming) != hipSuccess ||
            hipMalloc((void**)&L.sp_dev, sizeof(StepParams)) != hipSuccess)
            return fail(c, HD_EHIP, "graph stream setup failed");
    }
    if (hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess) { c->ev_in = nullptr; return fail(c, HD_EHIP, "graph stream setup failed"); }
    return HD_OK;
}

// Order the caller's stream after everything the lanes hold (the end of a chain bracket, or a step inside one that has to run on the
// caller's stream); the next replayed step forks again.
static int lanes_join(hd_ctx* c, hipStream_t user) {
    for (auto& L : c->lane) {
        if (!L.dirty) continue;
        if (hipEventRecord(L.ev_out, L.st) != hipSuccess || hipStreamWaitEvent(user, L.ev_out, 0) != hipSuccess) return fail(c, HD_EHIP, "stream hand-off failed");
        L.dirty = false;
    }
    c->forked = false;
    return HD_OK;
}

// One lane's share of a replayed step: its scalars, then the eager first call / the capture / the replay, all on the lane's stream.
static int lane_step(hd_ctx* c, hd_ctx::Lane& L, Pool* pool, int kind, float* x, const float* aux, const StepParams& v, float* x0_out, float* eps,
                     int B, int S) {
    hd_ctx::StepGraph* g = nullptr;
    const int arith = c->precision == HD_PREC_BF16X3 && c->f16w2 ? 1 + c->f16w2 : c->precision;     // a captured step holds the kernels of the arithmetic it was captured under
    for (auto& e : L.graphs) if (e.kind == kind && e.B == B && e.S == S && e.precision == arith && e.x == x && e.aux == aux && e.x0 == x0_out) { g = &e; break; }
    if (!g) {
        if (L.graphs.size() >= 16) {        // callers that pass fresh tensors every step must not grow the cache
            if (L.graphs.front().exec) (void)hipGraphExecDestroy(L.graphs.front().exec);
            L.graphs.erase(L.graphs.begin());
        }
        L.graphs.push_back({kind, B, S, arith, x, aux, x0_out, 0, nullptr});
        g = &L.graphs.back();
    }
    HD_TRY(launch_set_step_params(L.sp_dev, v, L.st));
    L.dirty = true;
    if (g->seen < 1) {                    // first call: eager (sets function attributes, warms caches)
        g->seen++;
        return step_body(c, kind, x, aux, nullptr, v, x0_out, B, S, L.st, L.sp_dev, pool, eps);
    }
    if (!g->exec) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(L.st, hipStreamCaptureModeThreadLocal) != hipSuccess) return fail(c, HD_EHIP, "hipStreamBeginCapture failed");
        const int rc = step_body(c, kind, x, aux, nullptr, v, x0_out, B, S, L.st, L.sp_dev, pool, eps);
        hipError_t e1 = hipStreamEndCapture(L.st, &graph);
        if (rc != 0 || e1 != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); return rc ? rc : fail(c, HD_EHIP, "hipStreamEndCapture failed"); }
        hipError_t e2 = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e2 != hipSuccess) { g->exec = nullptr; return fail(c, HD_EHIP, "hipGraphInstantiate failed"); }
    }
    if (hipGraphLaunch(g->exec, L.st) != hipSuccess) return fail(c, HD_EHIP, "hipGraphLaunch failed");
    return HD_OK;
}

static int run_step(hd_ctx* c, int kind, float* x, const float* aux, const float* noise, const StepParams& v, float* x0_out, int B,
                    int S, hipStream_t user) {
    if ((size_t)B * S * S > c->eps_cap) return fail(c, HD_ENOMEM, "workspace too small: call hd_reserve(ctx, B, S) first");
    // Replayed noise changes address every step and event profiling records per launch: both run eagerly.  So do small batches unless graphs are
    // forced on: below ~150 k pixels per step (kReplayMinPixels) the kernels are latency-bound and the eager launches of an idle host thread ran 6-10 % faster than
    // the replay (unet40: 4 tiles 2.23 vs 2.46 ms per step, 64 tiles 2.89 vs 3.08; profiles/r03_f_*), at full batches the two measure the same
    // (unet64, 256 tiles: 13.0-13.2 ms either way) and the replay keeps the step independent of what the host thread is doing.
    const bool graphs = c->use_graphs >= 0 ? c->use_graphs != 0 : (long long)B * S * S >= kReplayMinPixels;
    if (!graphs || noise != nullptr || hd_prof_is_on() || c->capture) {
        if (c->in_chain) HD_TRY(lanes_join(c, user));
        return step_body(c, kind, x, aux, noise, v, x0_out, B, S, user, nullptr);
    }
    HD_TRY(lanes_setup(c));
    int nl = lanes_for(c, B, S);
    for (int i = 0; i < nl && nl > 1; ++i)
        if (c->lane[i].pool.cap == 0) { nl = 1; break; }      // reserved before hd_set_chains(n): whole-batch lane
    // order after the caller's stream (every step; inside a chain bracket only at its first replayed step), run on the lanes, hand back
    if (!c->in_chain || !c->forked) {
        if (hipEventRecord(c->ev_in, user) != hipSuccess) return fail(c, HD_EHIP, "stream hand-off failed");
        for (auto& L : c->lane) if (hipStreamWaitEvent(L.st, c->ev_in, 0) != hipSuccess) return fail(c, HD_EHIP, "stream hand-off failed");
        if (c->in_chain) {
            c->forked = true;
            for (int i = 1; i < nl && c->lane_delay_us > 0; ++i) HD_TRY(launch_spin_us(c->lane_delay_us * i, c->lane[i].st));
        }
    }
    int rc = 0;
    if (nl == 1) {
        rc = lane_step(c, c->lane[0], &c->pool, kind, x, aux, v, x0_out, c->eps_buf, B, S);
    } else {
        for (int i = 0; i < nl && rc == 0; ++i) {
            const int b0 = lane_start(B, nl, i), bn = lane_start(B, nl, i + 1) - b0;
            const size_t off = (size_t)b0 * S * S;
            StepParams vi = v;
            vi.tile_off += (uint64_t)b0;
            rc = lane_step(c, c->lane[i], &c->lane[i].pool, kind, x + off, aux ? aux + off : nullptr, vi, x0_out ? x0_out + off : nullptr,
                           c->eps_buf + off, bn, S);
        }
    }
    if (!c->in_chain || rc != 0) { const int rj = lanes_join(c, user); if (rc == 0) rc = rj; }
    return rc;
}

int hd_ddpm_step(hd_ctx* c, float* x, const float* cond, const float* noise, const hd_ddpm_coef* kin, float* x0_out, int B, int S,
                 uint64_t seed, uint64_t tile_offset, uint32_t step, void* stream) {
    if (!c || !x || !kin) return HD_EINVAL;
    hd_ddpm_coef kk, *k = &kk;
    if (!hd_read_prefixed(kin, offsetof(hd_ddpm_coef, eps_coef), k)) return fail(c, HD_EINVAL, kCoefSizeMsg);   // (without eps_coef: an ancestral step)
    if (!c->loaded) return fail(c, HD_ESTATE, "weights not loaded: call hd_load_weights first");
    if ((c->arch.self_condition != 0) != (cond != nullptr)) return fail(c, HD_EINVAL, "cond must be given iff self_condition");
    StepParams v{};
    v.f[0] = k->time_value; v.f[1] = k->sqrt_recip_alphas_cumprod; v.f[2] = k->sqrt_recipm1_alphas_cumprod;
    v.f[3] = k->posterior_mean_coef1; v.f[4] = k->posterior_mean_coef2; v.f[5] = k->sigma; v.f[6] = k->eps_coef;
    v.step = step; v.seed = seed; v.tile_off = tile_offset;
    // this step's arithmetic (hd_ddpm_coef.arith): the host's precision schedule over the chain
    const int saved = c->f16w2;
    if (k->arith > HD_ARITH_F16W2_LOW) return fail(c, HD_EINVAL, "hd_ddpm_coef.arith: unknown value");
    if (k->arith != HD_ARITH_DEFAULT && c->precision == HD_PREC_BF16X3) c->f16w2 = (int)k->arith;
    const int rc = run_step(c, 0, x, cond, noise, v, x0_out, B, S, (hipStream_t)stream);
    c->f16w2 = saved;
    return keep_err(c, rc);
}

int hd_ddrm_step(hd_ctx* c, float* x, const float* y, const float* z, const hd_ddrm_coef* kin, float* x0_out, int B, int S,
                 uint64_t seed, uint64_t tile_offset, uint32_t step, void* stream) {
    if (!c || !x || !y || !kin) return HD_EINVAL;
    hd_ddrm_coef kk, *k = &kk;
    if (!hd_read_prefixed(kin, offsetof(hd_ddrm_coef, skip_network), k)) return fail(c, HD_EINVAL, kCoefSizeMsg);   // (without skip_network: revision 3)
    if (!c->loaded) return fail(c, HD_ESTATE, "weights not loaded: call hd_load_weights first");
    StepParams v{};
    v.f[0] = k->time_value; v.f[1] = k->sqrt_at; v.f[2] = k->sqrt_1m_at; v.f[3] = k->sqrt_at_next; v.f[4] = k->sigma_next;
    v.f[5] = k->sigma_0; v.f[6] = k->etaA; v.f[7] = k->etaB; v.f[8] = k->etaC;
    v.step = step; v.seed = seed; v.tile_off = tile_offset;
    if (k->skip_network) {
        // The identity degradation's third case with etaB == 1 (src/functions/denoising.py:99-100): the next state is built from y and fresh
        // noise alone.  The SAME update kernel runs on an all-zero eps -- its x0 term is multiplied by (1 - etaB) = 0 -- so the state equals
        // the full step's bit for bit; only the network's launches are gone.
        if (k->skip_network != 1 || !(k->etaB == 1.f) || !(k->sigma_next > k->sigma_0))
            return fail(c, HD_EINVAL, "hd_ddrm_coef.skip_network: this step's update reads the network (it needs etaB == 1 and sigma_next > sigma_0)");
        if ((size_t)B * S * S > c->eps_cap) return fail(c, HD_ENOMEM, "workspace too small: call hd_reserve(ctx, B, S) first");
        hipStream_t user = (hipStream_t)stream;
        if (c->in_chain) HD_TRY(keep_err(c, lanes_join(c, user)));
        if (hipMemsetAsync(c->eps_buf, 0, (size_t)B * S * S * sizeof(float), user) != hipSuccess) return fail(c, HD_EHIP, "hipMemsetAsync failed");
        return keep_err(c, launch_ddrm_update(x, c->eps_buf, y, z, v.f[1], v.f[2], v.f[3], v.f[4], v.f[5], v.f[6], v.f[7], v.f[8], x0_out, B, S, v.seed,
                                              v.tile_off, v.step, nullptr, user));
    }
    return keep_err(c, run_step(c, 1, x, y, z, v, x0_out, B, S, (hipStream_t)stream));
}

int hd_set_graphs(hd_ctx* c, int enable) {
    if (!c) return HD_EINVAL;
    c->use_graphs = enable != 0 ? 1 : 0;
    return HD_OK;
}

int hd_set_chains(hd_ctx* c, int n) {
    if (!c || n < 0 || n > hd_ctx::MAX_LANES) return HD_EINVAL;
    if (c->in_chain) return fail(c, HD_ESTATE, "hd_set_chains inside an hd_chain_begin / hd_chain_end bracket");
    c->chains = n == 0 ? -1 : n;
    if (c->resB > 0 && hd_reserve(c, c->resB, c->resS) != HD_OK) return HD_EHIP;      // the half-batch workspaces, if they are now needed
    return HD_OK;
}

int hd_chains_for(const hd_ctx* c, int B, int S) {
    if (!c || B < 1 || S < 1) return HD_EINVAL;
    const bool graphs = c->use_graphs >= 0 ? c->use_graphs != 0 : (long long)B * S * S >= kReplayMinPixels;
    return graphs ? lanes_for(c, B, S) : 1;
}

int hd_chain_begin(hd_ctx* c, void* stream) {
    if (!c) return HD_EINVAL;
    if (c->in_chain) return fail(c, HD_ESTATE, "hd_chain_begin: a bracket is already open on this context");
    (void)stream;
    c->in_chain = true; c->forked = false;
    return HD_OK;
}

int hd_chain_end(hd_ctx* c, void* stream) {
    if (!c) return HD_EINVAL;
    if (!c->in_chain) return fail(c, HD_ESTATE, "hd_chain_end without hd_chain_begin");
    c->in_chain = false;
    return keep_err(c, lanes_join(c, (hipStream_t)stream));
}

int hd_q_sample(hd_ctx* c, const float* x0, const float* noise, const float* a, const float* s, float* out, int B, int S, void* stream) {
    if (!c || !x0 || !noise || !a || !s || !out) return HD_EINVAL;
    return keep_err(c, launch_q_sample(x0, noise, a, s, out, B, S, (hipStream_t)stream));
}

int hd_loss_per_sample(hd_ctx* c, const float* pred, const float* target, int l2, float* out, int B, int S, void* stream) {
    if (!c || !pred || !target || !out) return HD_EINVAL;
    return keep_err(c, launch_loss(pred, target, l2, out, B, S, (hipStream_t)stream));
}

int hd_randn(hd_ctx* c, float* out, int B, int S, uint64_t seed, uint64_t tile_offset, uint32_t step, void* stream) {
    if (!c || !out) return HD_EINVAL;
    if ((S * S) % 4) return fail(c, HD_EINVAL, "S*S must be a multiple of 4");
    return keep_err(c, launch_randn(out, B, S, seed, tile_offset, step, (hipStream_t)stream));
}

int hd_tile_metrics(const float* pred, const float* target, int B, int S, int rescale, double* partial, double* sums, float* ssim_each,
                    void* stream) {
    if (!pred || !target || !partial || !sums || B < 1) return HD_EINVAL;
    return launch_tile_metrics(pred, target, B, S, rescale, partial, sums, ssim_each, (hipStream_t)stream) == 0 ? HD_OK : HD_EINVAL;
}

int hd_split_pieces(const float* mat, int n, const int* origins, int ntiles, int piece, float* tiles, void* stream) {
    if (n < 0 || ntiles < 0 || piece < 1) return HD_EINVAL;
    if (ntiles == 0) return HD_OK;
    if (!mat || !origins || !tiles || n < 1) return HD_EINVAL;
    return launch_split_pieces(mat, n, origins, ntiles, piece, tiles, (hipStream_t)stream) == 0 ? HD_OK : HD_EINVAL;
}

int hd_stitch_pieces(const float* tiles, const int* tile_of, int nb, int piece, int step, float* mat, int n, void* stream) {
    if (n < 0 || nb < 0 || piece < 1 || step < piece) return HD_EINVAL;
    if (n == 0) return HD_OK;
    if (!tiles || !tile_of || !mat || nb < 1) return HD_EINVAL;
    return launch_stitch_pieces(tiles, tile_of, nb, piece, step, mat, n, (hipStream_t)stream) == 0 ? HD_OK : HD_EINVAL;
}

int hd_profile_enable(int enable) { hd_prof_enable(enable != 0); return HD_OK; }

int hd_set_precision(hd_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 3 /* HD_PRECISION_F16W1 */) return HD_EINVAL;
    // a captured step holds the kernels of the arithmetic it was captured under: the graph cache is keyed by the mode (run_step), so a
    // sampler that switches back and forth (ddim_sample runs exact fp32) keeps both sets instead of re-capturing every time
    c->precision = mode >= 2 ? HD_PREC_BF16X3 : mode;
    c->f16w2 = mode >= 2 ? mode - 1 : 0;          // 1: two fp16 products in the 3x3 convolutions, 2: one
    return HD_OK;
}

int hd_profile_read(hd_profile_row* rows, int max_rows) {
    if (!rows || max_rows <= 0) return HD_EINVAL;
    if (max_rows > HD_PROFILE_MAX_ROWS) max_rows = HD_PROFILE_MAX_ROWS;
    const char* names[HD_PROFILE_MAX_ROWS]; double ms[HD_PROFILE_MAX_ROWS], fl[HD_PROFILE_MAX_ROWS], by[HD_PROFILE_MAX_ROWS]; long long n[HD_PROFILE_MAX_ROWS];
    const int count = hd_prof_collect(names, ms, fl, by, n, max_rows);
    for (int v = 0; v < count; ++v) { rows[v].kernel = names[v]; rows[v].launches = n[v]; rows[v].total_ms = ms[v]; rows[v].flops = fl[v]; rows[v].bytes = by[v]; }
    return count;
}

// ---- test-only entry points (include/hicdiff_hip_debug.h) ----
int hd_debug_linattn_out(const float* q, const float* ctx, const float* wout, const float* bias, const float* g, const float* res,
                         int B, int H, int W, int C, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!q || !ctx || !wout || !g || !res || !out || C % 64) return HD_EINVAL;
    ConvW cw; cw.KH = cw.KW = 1; cw.Cin = 128; cw.Cout = C; cw.CoutPad = C; cw.ck = 32; cw.bias = const_cast<float*>(bias);
    void *pw = nullptr, *pf = nullptr, *py = nullptr;
    const size_t fold_bytes = (size_t)4 * C * 64 * sizeof(unsigned short);
    if (hipMalloc(&pw, (size_t)128 * C * sizeof(float)) != hipSuccess || hipMalloc(&pf, (size_t)B * fold_bytes) != hipSuccess) return HD_EHIP;
    cw.w = (float*)pw;
    int rc = launch_pack_conv(wout, cw.w, C, 128, 1, 1, C, 0, 0, st);
    if (rc == 0) rc = launch_linattn_fold_out(cw.w, ctx, B, C, (unsigned short*)pf, st);
    ConvArgs a;
    a.in0 = q; a.C0 = 128; a.B = B; a.H = H; a.W = W; a.IH = H; a.IW = W; a.stride = 1; a.pad = 0; a.cw = cw;
    a.cw.wsplit = (unsigned short*)pf; a.w_bstride = fold_bytes; a.in_mode = IN_SOFTMAX32; a.precision = HD_PREC_BF16X3;
    const size_t P = (size_t)B * H * W;
    if (C == 64 || C == 128) {
        a.out = out; a.ep = EP_LN_RES; a.ep_ln_g = g; a.res = res;
        if (rc == 0) rc = launch_conv(a, st, nullptr);
    } else {
        if (hipMalloc(&py, P * C * sizeof(float)) != hipSuccess) rc = HD_EHIP;
        a.out = (float*)py;
        if (rc == 0) rc = launch_conv(a, st, nullptr);
        if (rc == 0) rc = launch_ln_residual((const float*)py, g, res, out, P, C, st);
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(pw); (void)hipFree(pf); if (py) (void)hipFree(py);
    return rc;
}

int hd_debug_linattn_q(const float* x, const float* norm_g, const float* wqkv, const float* ctx, const float* wout, const float* bias,
                       const float* gout, int B, int H, int W, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int C = 64;
    if (!x || !norm_g || !wqkv || !ctx || !wout || !gout || !out) return HD_EINVAL;
    const size_t P = (size_t)B * H * W;
    void *pw = nullptr, *pf = nullptr, *pq = nullptr, *ps = nullptr;
    if (hipMalloc(&pw, (size_t)128 * C * sizeof(float)) != hipSuccess || hipMalloc(&pf, (size_t)B * 4 * C * 64 * sizeof(unsigned short)) != hipSuccess ||
        hipMalloc(&pq, (size_t)128 * C * 2 * sizeof(unsigned short)) != hipSuccess || hipMalloc(&ps, P * 2 * sizeof(float)) != hipSuccess) return HD_EHIP;
    int rc = launch_pack_conv(wout, (float*)pw, C, 128, 1, 1, C, 0, 0, st);
    if (rc == 0) rc = launch_linattn_fold_out((const float*)pw, ctx, B, C, (unsigned short*)pf, st, 1);
    if (rc == 0) rc = launch_pack_q(wqkv, norm_g, C, (unsigned short*)pq, st);
    if (rc == 0) rc = launch_ln_stats(x, P, C, (float*)ps, st);
    if (rc == 0) rc = launch_linattn_q_fused(x, (const float*)ps, (const unsigned short*)pq, (const unsigned short*)pf, bias, gout, out, B, H * W, C, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(pw); (void)hipFree(pf); (void)hipFree(pq); (void)hipFree(ps);
    return rc;
}

int hd_debug_randn(float* out, int B, int S, uint64_t seed, uint64_t tile_offset, uint32_t step, uint32_t noise_stream, void* stream) {
    if (!out || B < 1 || (S * S) % 4 || noise_stream > 3) return HD_EINVAL;
    return launch_randn(out, B, S, seed, tile_offset, step, (hipStream_t)stream, noise_stream);
}

int hd_debug_winograd(int mode) {
    if (mode < -1 || mode > 1) return HD_EINVAL;
    conv_set_winograd(mode);
    return HD_OK;
}

int hd_debug_capture(hd_ctx* c, int enable) {
    if (!c) return HD_EINVAL;
    c->capture = enable != 0;
    if (!enable) { for (auto& kv : c->captured) (void)hipFree(kv.second.p); c->captured.clear(); }
    return HD_OK;
}

int hd_debug_read(hd_ctx* c, const char* label, float* dst, size_t n, int32_t dims[4]) {
    if (!c || !label || !dims) return HD_EINVAL;
    auto it = c->captured.find(label);
    if (it == c->captured.end()) return fail(c, HD_EINVAL, std::string("no capture named '") + label + "'");
    const Act& a = it->second;
    dims[0] = a.B; dims[1] = a.H; dims[2] = a.W; dims[3] = a.C;
    if (dst) {
        if (n < a.numel()) return fail(c, HD_EINVAL, "capture buffer too small");
        if (hipMemcpy(dst, a.p, a.numel() * sizeof(float), hipMemcpyDeviceToDevice) != hipSuccess) return fail(c, HD_EHIP, "capture read failed");
    }
    return HD_OK;
}

int hd_debug_conv(const float* in0, int C0, const float* in1, int C1, int B, int IH, int IW, const float* w, const float* bias,
                  int Cout, int K, int mode, const float* A, const float* Bv, const float* E, float* out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const bool up = mode & 1, ws = mode & 2, unsh = mode & 4, aff = mode & 8, ln = mode & 16;
    const int Cin = C0 + C1, KH = unsh ? 2 : K;
    ConvW cw; cw.KH = cw.KW = KH; cw.Cin = Cin; cw.Cout = Cout; cw.CoutPad = (Cout + 63) / 64 * 64;
    cw.ck = (mode & 64) && Cin % 32 == 0 && C0 % 32 == 0 ? 32 : 16;
    void* psplit = nullptr;
    void *pw = nullptr, *pst = nullptr;
    if (hipMalloc(&pw, (size_t)KH * KH * Cin * cw.CoutPad * sizeof(float)) != hipSuccess) return HD_EHIP;
    cw.w = (float*)pw; cw.bias = const_cast<float*>(bias);
    int rc = launch_pack_conv(w, cw.w, Cout, Cin, KH, KH, cw.CoutPad, ws, unsh, st);
    ConvArgs a;
    a.in0 = in0; a.C0 = C0; a.in1 = in1; a.C1 = C1; a.B = B; a.IH = IH; a.IW = IW; a.cw = cw; a.out = out;
    if (unsh) { a.H = IH / 2; a.W = IW / 2; a.stride = 2; a.pad = 0; }
    else if (up) { a.H = IH * 2; a.W = IW * 2; a.stride = 1; a.pad = K / 2; a.upsample = 1; }
    else { a.H = IH; a.W = IW; a.stride = 1; a.pad = K / 2; }
    if (aff) { a.in_mode = IN_AFFINE_SILU; a.inA = A; a.inB = Bv; a.inE = E; a.in_bstride = Cin; }
    if (ln && rc == 0) {
        if (mode & 128) {                      // precomputed statistics passed in Bv ([pixels][2])
            a.ln_stats = Bv;
        } else {
            if (hipMalloc(&pst, (size_t)B * IH * IW * 2 * sizeof(float)) != hipSuccess) { (void)hipFree(pw); return HD_EHIP; }
            rc = launch_ln_stats(in0, (size_t)B * IH * IW, C0, (float*)pst, st);
            a.ln_stats = (float*)pst;
        }
        a.in_mode = IN_LAYERNORM; a.ln_g = A;
    }
    if ((mode & 32) && rc == 0) {
        if (hipMalloc(&psplit, (size_t)KH * KH * Cin * cw.CoutPad * 2 * sizeof(unsigned short)) != hipSuccess) { (void)hipFree(pw); return HD_EHIP; }
        rc = launch_split_conv(cw.w, (unsigned short*)psplit, KH * KH, Cin, cw.CoutPad, 16, st);
        a.cw.wsplit = (unsigned short*)psplit; a.cw.ck = cw.ck; a.precision = HD_PREC_BF16X3;
        if (mode & 2048) a.plain_bf16 = 1;     // ONE bf16 product (the training step's optional arithmetic), where the kernel form exists
    }
    void* psplit16 = nullptr;
    if ((mode & 512) && (mode & 32) && rc == 0) {          // two fp16 products per multiply (3x3 only; launch_conv falls back to x3 where it does not apply)
        if (KH != 3 || unsh || Cin % 16 || hipMalloc(&psplit16, (size_t)9 * Cin * cw.CoutPad * 2 * sizeof(unsigned short)) != hipSuccess) rc = HD_EINVAL;
        if (rc == 0) rc = launch_split_conv(cw.w, (unsigned short*)psplit16, 9, Cin, cw.CoutPad, 16, st, 1);
        a.cw.wsplit16 = (unsigned short*)psplit16; a.f16w2 = (mode & 1024) ? 2 : 1;      // 1024: ONE fp16 product, xh wh
    }
    void* pwino = nullptr;
    if ((mode & 256) && rc == 0) {             // Winograd image + switch: launch_conv takes the F(2x2,3x3) kernel where the shape allows it
        conv_set_winograd(1);
        if (KH != 3 || unsh || Cin % 16 || hipMalloc(&pwino, conv_winograd_weight_bytes(Cin, cw.CoutPad)) != hipSuccess) rc = HD_EINVAL;
        if (rc == 0) rc = launch_pack_winograd(cw.w, (unsigned short*)pwino, Cin, cw.CoutPad, st);
        a.cw.wino = (unsigned short*)pwino;
        if (rc == 0 && !conv_uses_winograd(a)) { g_err = "hd_debug_conv: this shape does not take the Winograd kernel"; rc = HD_EINVAL; }
    }
    if (rc == 0) rc = launch_conv(a, st, nullptr);
    if (mode & 256) conv_set_winograd(-1);
    (void)hipStreamSynchronize(st);
    (void)hipFree(pw); if (pst) (void)hipFree(pst); if (psplit) (void)hipFree(psplit); if (pwino) (void)hipFree(pwino); if (psplit16) (void)hipFree(psplit16);
    return rc;
}

}  // extern "C"
