// fastgen_amd engine: network plan, weight packing, workspace arena, forward orchestration, sampler loop with hipGraph
// replay, and the C ABI declared in include/fastgen_amd.h.  Host-side C++ only; every kernel lives in conv/attn/misc.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <map>
#include <vector>

#include "../../include/fastgen_amd.h"
#include "conv.h"
#include "misc.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (hipError_t)(expr);                                                                    \
        if (e_ != hipSuccess) return fail(FG_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct Param {
    std::string name;
    int ndim = 0;
    int64_t shape[4] = {1, 1, 1, 1};
    int64_t numel = 0;
    const float* ptr = nullptr;  // borrowed device pointer (fp32, reference layout)
    float* grad = nullptr;       // borrowed fp32 gradient accumulator (fg_edm_bind_grad), nullptr = not requested
};

enum Kind { K_STEM, K_BLOCK, K_AUX_NORM, K_AUX_CONV };

struct Block {
    std::string key;
    Kind kind = K_BLOCK;
    int cin = 0, cout = 0, res_in = 0, res_out = 0;
    bool up = false, down = false, attn = false, has_skip = false, is_dec = false;
    int skip_c = 0;    // decoder: channels taken from the skip stack (concat), else 0
    int temb_off = 0;  // column offset of this block's affine() in the stacked embedding projection
    // parameter indices (-1 = absent)
    int norm0_w = -1, norm0_b = -1, conv0_w = -1, conv0_b = -1, aff_w = -1, aff_b = -1, norm1_w = -1, norm1_b = -1,
        conv1_w = -1, conv1_b = -1, skip_w = -1, skip_b = -1, norm2_w = -1, norm2_b = -1, qkv_w = -1, qkv_b = -1,
        proj_w = -1, proj_b = -1, w = -1, b = -1;
    // packed weights (owned device memory, compute dtype)
    void *p_conv0 = nullptr, *p_conv1 = nullptr, *p_skip = nullptr, *p_qkv = nullptr, *p_proj = nullptr;
    void *p_conv0_ws = nullptr, *p_conv1_ws = nullptr;  // conv_ws.hip layout, where that kernel applies
    int tap = -1;  // index among the encoder's `block3` outputs (feature taps), or -1
    float* qkv_bias = nullptr;  // [3C] permuted to q|k|v
    void* p_aux = nullptr;      // K_AUX_CONV: weights packed for the MFMA output head
    void* p_stem = nullptr;     // K_STEM: weights packed for the MFMA stem
};

struct Arena {
    char* base = nullptr;
    size_t off = 0, cap = 0;
    bool dry = false;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = dry ? nullptr : base + off;
        off += bytes;
        return p;
    }
    template <typename T>
    T* get(size_t n) {
        return reinterpret_cast<T*>(take(n * sizeof(T)));
    }
};

// An NHWC activation plus the GroupNorm partial statistics its producing conv left behind (st == nullptr: none, the
// consumer falls back to a full statistics pass over the tensor).
struct Act {
    void* p = nullptr;  // compute dtype (fp32 / bf16)
    float2* st = nullptr;
    int slots = 0;
};

struct Workspace {
    float *coef, *emb0, *emb1, *emb, *temb;
    float2 *ab0, *ab1, *ab2;
    float2 *mr0 = nullptr, *mr1 = nullptr, *mr2 = nullptr;  // {mean, rstd} of norm0 / norm1 / norm2, kept when a backward pass follows
    std::vector<Act> skip;  // encoder outputs
    Act xa, xb, h, xattn;
    void *sbuf, *aout, *pool;
    void *cvt1, *cvt2;  // fg_edm_run_block: caller's fp32 tensors converted to the activation dtype
    void *q, *k, *vt;
    // sampler state
    float *x, *x_pred, *eps;
    double* tl;      // [65]
    uint64_t* seed;  // [2]
};

// What a block's backward pass reads besides the block's inputs; kept per block by a forward that a backward follows.
struct BlockStash {
    void* h = nullptr;                      // conv0 output
    float2 *ab0 = nullptr, *ab1 = nullptr, *ab2 = nullptr, *mr0 = nullptr, *mr1 = nullptr, *mr2 = nullptr;
    void *xattn = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *aout = nullptr;  // attention blocks
};
struct TrainStash {
    std::vector<Act> dec_store;      // decoder block outputs (the encoder's live on the skip stack anyway)
    std::vector<BlockStash> blocks;  // indexed like fg_edm::blocks
    float2* aux_ab = nullptr;        // coefficients of aux_norm
    float2* aux_mr = nullptr;        // its {mean, rstd}
};

struct GraphKey {
    int B = 0, steps = 0, type = 0, loop = 0;
    uint64_t zero_mask = 0;
    const void *noise = nullptr, *labels = nullptr, *eps = nullptr, *out = nullptr, *ws = nullptr;
    bool device_rng = false;
    bool operator==(const GraphKey& o) const {
        return B == o.B && steps == o.steps && type == o.type && loop == o.loop && zero_mask == o.zero_mask && noise == o.noise &&
               labels == o.labels && eps == o.eps && out == o.out && ws == o.ws && device_rng == o.device_rng;
    }
};

}  // namespace

struct fg_edm {
    fg_edm_config cfg;
    int dtype = 0;
    int num_taps = 0;  // encoder `block3` outputs available as feature taps
    int emb_ch = 0, noise_ch = 0, cond_ch = 0;  // cond_ch = noise_ch * (1 + r_timestep), EDM/network.py:376
    std::vector<Param> params;
    std::vector<Block> enc, dec;  // dec includes aux_norm / aux_conv entries
    std::vector<Block*> blocks;   // UNetBlocks only, encoder then decoder order
    int temb_total = 0;
    bool packed = false;
    bool device_ready = false;
    // data-gradient weights (transposed, flipped, packed), built on first use per weight version (fg_edm_pack_weights)
    struct DgradW {
        void* packed = nullptr;
        uint64_t epoch = 0;
    };
    std::map<const float*, DgradW> dgrad_cache;
    uint64_t pack_epoch = 0;
    float* aff_wT = nullptr;  // [emb_ch][temb_total]: the stacked affine matrix transposed, for the batched embedding gradient
    uint64_t aff_wT_epoch = 0;
    // owned device memory
    float* freqs = nullptr;      // [noise_ch/2]
    float* aff_w = nullptr;      // [temb_total][emb_ch]
    float* aff_b = nullptr;      // [temb_total]
    std::vector<void*> owned;
    // graph cache
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    GraphKey graph_key;
    // Per-call scalars (timesteps, RNG seed) reach the device through a ring of pinned host slots, copied on the
    // caller's stream BEFORE the graph launch (not a graph node: a node would re-read host memory that the next call
    // may already have overwritten).  A slot is reused only after the event recorded behind its copy has completed.
    static constexpr int kSlots = 8;
    struct Slot {
        double tl[72];
        uint64_t seed[8];
    };
    Slot* slots = nullptr;  // pinned
    hipEvent_t slot_ev[kSlots] = {};
    bool slot_used[kSlots] = {};
    int slot_next = 0;
    // live timing of the dominant kernel (conv 3x3, no resample, 32x32 output): HIP events on the launch stream
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;  // start/stop pairs
    double prof_flops = 0.0;
    hipStream_t cap_stream = nullptr;  // capture-only stream (the legacy default stream cannot be captured)

    int find(const std::string& n) const {
        for (size_t i = 0; i < params.size(); ++i)
            if (params[i].name == n) return (int)i;
        return -1;
    }
    int add(const std::string& n, std::initializer_list<int64_t> shp) {
        Param p;
        p.name = n;
        p.ndim = (int)shp.size();
        p.numel = 1;
        int i = 0;
        for (int64_t s : shp) {
            p.shape[i++] = s;
            p.numel *= s;
        }
        params.push_back(p);
        return (int)params.size() - 1;
    }
    const float* P(int idx) const { return idx >= 0 ? params[idx].ptr : nullptr; }
    float* G(int idx) const { return idx >= 0 ? params[idx].grad : nullptr; }
};

namespace {

// Module order of SongUNet (standard encoder / decoder), reference EDM/network.py:417-486; state-dict names as there.
void build_layout(fg_edm* h) {
    const fg_edm_config& c = h->cfg;
    const int E = h->emb_ch, N = h->cond_ch;
    if (c.label_dim) {
        h->add("model.map_label.weight", {N, c.label_dim});
        h->add("model.map_label.bias", {N});
    }
    if (c.augment_dim) h->add("model.map_augment.weight", {N, c.augment_dim});
    h->add("model.map_layer0.weight", {E, N});
    h->add("model.map_layer0.bias", {E});
    h->add("model.map_layer1.weight", {E, E});
    h->add("model.map_layer1.bias", {E});

    auto is_attn_res = [&](int res) {
        for (int i = 0; i < c.num_attn_resolutions; ++i)
            if (c.attn_resolutions[i] == res) return true;
        return false;
    };
    auto make_block = [&](const std::string& key, int cin, int cout, int res_out, bool up, bool down, bool attn,
                          int skip_c, bool is_dec) {
        Block b;
        b.key = key;
        b.kind = K_BLOCK;
        b.cin = cin;
        b.cout = cout;
        b.res_out = res_out;
        b.res_in = down ? res_out * 2 : (up ? res_out / 2 : res_out);
        b.up = up;
        b.down = down;
        b.attn = attn;
        b.skip_c = skip_c;
        b.is_dec = is_dec;
        b.has_skip = (cin != cout) || up || down;
        b.norm0_w = h->add(key + ".norm0.weight", {cin});
        b.norm0_b = h->add(key + ".norm0.bias", {cin});
        b.conv0_w = h->add(key + ".conv0.weight", {cout, cin, 3, 3});
        b.conv0_b = h->add(key + ".conv0.bias", {cout});
        b.aff_w = h->add(key + ".affine.weight", {cout, E});
        b.aff_b = h->add(key + ".affine.bias", {cout});
        b.norm1_w = h->add(key + ".norm1.weight", {cout});
        b.norm1_b = h->add(key + ".norm1.bias", {cout});
        b.conv1_w = h->add(key + ".conv1.weight", {cout, cout, 3, 3});
        b.conv1_b = h->add(key + ".conv1.bias", {cout});
        if (b.has_skip) {
            b.skip_w = h->add(key + ".skip.weight", {cout, cin, 1, 1});
            b.skip_b = h->add(key + ".skip.bias", {cout});
        }
        if (attn) {
            b.norm2_w = h->add(key + ".norm2.weight", {cout});
            b.norm2_b = h->add(key + ".norm2.bias", {cout});
            b.qkv_w = h->add(key + ".qkv.weight", {3 * cout, cout, 1, 1});
            b.qkv_b = h->add(key + ".qkv.bias", {3 * cout});
            b.proj_w = h->add(key + ".proj.weight", {cout, cout, 1, 1});
            b.proj_b = h->add(key + ".proj.bias", {cout});
        }
        b.temb_off = h->temb_total;
        h->temb_total += cout;
        return b;
    };
    auto res_name = [](int r) { return std::to_string(r) + "x" + std::to_string(r); };

    std::vector<int> skips;
    int cout = c.img_channels;
    for (int level = 0; level < c.num_levels; ++level) {
        const int res = c.img_resolution >> level;
        if (level == 0) {
            Block b;
            b.key = "model.enc." + res_name(res) + "_conv";
            b.kind = K_STEM;
            b.cin = cout;
            b.cout = cout = c.model_channels;
            b.res_in = b.res_out = res;
            b.w = h->add(b.key + ".weight", {b.cout, b.cin, 3, 3});
            b.b = h->add(b.key + ".bias", {b.cout});
            h->enc.push_back(b);
        } else {
            h->enc.push_back(make_block("model.enc." + res_name(res) + "_down", cout, cout, res, false, true, false, 0, false));
        }
        skips.push_back(cout);
        for (int idx = 0; idx < c.num_blocks; ++idx) {
            const int cin = cout;
            cout = c.model_channels * c.channel_mult[level];
            h->enc.push_back(make_block("model.enc." + res_name(res) + "_block" + std::to_string(idx), cin, cout, res,
                                        false, false, is_attn_res(res), 0, false));
            // feature taps: every encoder entry whose name contains "block3" (EDM/network.py:535), numbered in order
            if (std::to_string(idx).rfind("3", 0) == 0) h->enc.back().tap = h->num_taps++;
            skips.push_back(cout);
        }
    }
    for (int level = c.num_levels - 1; level >= 0; --level) {
        const int res = c.img_resolution >> level;
        if (level == c.num_levels - 1) {
            h->dec.push_back(make_block("model.dec." + res_name(res) + "_in0", cout, cout, res, false, false, true, 0, true));
            h->dec.push_back(make_block("model.dec." + res_name(res) + "_in1", cout, cout, res, false, false, false, 0, true));
        } else {
            h->dec.push_back(make_block("model.dec." + res_name(res) + "_up", cout, cout, res, true, false, false, 0, true));
        }
        for (int idx = 0; idx <= c.num_blocks; ++idx) {
            const int sk = skips.back();
            skips.pop_back();
            const int cin = cout + sk;
            cout = c.model_channels * c.channel_mult[level];
            const bool attn = (idx == c.num_blocks) && is_attn_res(res);
            h->dec.push_back(make_block("model.dec." + res_name(res) + "_block" + std::to_string(idx), cin, cout, res,
                                        false, false, attn, sk, true));
        }
        if (level == 0) {
            Block n;
            n.key = "model.dec." + res_name(res) + "_aux_norm";
            n.kind = K_AUX_NORM;
            n.cin = n.cout = cout;
            n.res_in = n.res_out = res;
            n.w = h->add(n.key + ".weight", {cout});
            n.b = h->add(n.key + ".bias", {cout});
            h->dec.push_back(n);
            Block a;
            a.key = "model.dec." + res_name(res) + "_aux_conv";
            a.kind = K_AUX_CONV;
            a.cin = cout;
            a.cout = c.img_channels;
            a.res_in = a.res_out = res;
            a.w = h->add(a.key + ".weight", {a.cout, a.cin, 3, 3});
            a.b = h->add(a.key + ".bias", {a.cout});
            h->dec.push_back(a);
        }
    }
    h->add("model.logvar_linear.weight", {1, h->noise_ch});  // Linear(noise_channels, 1), EDM/network.py:487
    h->add("model.logvar_linear.bias", {1});
    for (auto& b : h->enc)
        if (b.kind == K_BLOCK) h->blocks.push_back(&b);
    for (auto& b : h->dec)
        if (b.kind == K_BLOCK) h->blocks.push_back(&b);
}

int check_supported(const fg_edm* h) {
    const fg_edm_config& c = h->cfg;
    const int kc = h->dtype ? 64 : 32;
    if (c.img_resolution != 32 && c.img_resolution != 16 && c.img_resolution != 8)
        return fail(FG_EINVAL, "img_resolution %d unsupported (8, 16, 32)", c.img_resolution);
    if ((c.img_resolution >> (c.num_levels - 1)) < 8)
        return fail(FG_EINVAL, "lowest resolution %d < 8 unsupported", c.img_resolution >> (c.num_levels - 1));
    if (c.img_channels > 4) return fail(FG_EINVAL, "img_channels %d > 4 unsupported", c.img_channels);
    for (const Block* b : h->blocks) {
        if (b->cout != 256) return fail(FG_EINVAL, "%s: out_channels %d unsupported (kernels are tiled for 256)", b->key.c_str(), b->cout);
        if (b->cin % kc || (b->cin - b->skip_c) % kc)
            return fail(FG_EINVAL, "%s: in_channels %d not a multiple of %d", b->key.c_str(), b->cin, kc);
        if (b->attn && b->res_out > 16) return fail(FG_EINVAL, "%s: attention at %dx%d unsupported", b->key.c_str(), b->res_out, b->res_out);
    }
    return FG_OK;
}

size_t plan_workspace(const fg_edm* h, int B, Arena& A, Workspace& w) {
    const fg_edm_config& c = h->cfg;
    const size_t tsz = h->dtype ? 2 : 4;
    const int R = c.img_resolution;
    size_t max_act = 0;  // largest [res,res,C] activation per image
    int max_c = 0, max_attn_hw = 0;
    for (const Block& b : h->enc) max_act = std::max(max_act, (size_t)b.res_out * b.res_out * b.cout);
    for (const Block* b : h->blocks) {
        max_act = std::max(max_act, (size_t)b->res_out * b->res_out * b->cout);
        max_c = std::max(max_c, std::max(b->cin, b->cout));
        if (b->attn) max_attn_hw = std::max(max_attn_hw, b->res_out * b->res_out);
    }
    w.coef = A.get<float>(5 * (size_t)B);
    w.emb0 = A.get<float>((size_t)B * h->cond_ch);
    w.emb1 = A.get<float>((size_t)B * h->emb_ch);
    w.emb = A.get<float>((size_t)B * h->emb_ch);
    w.temb = A.get<float>((size_t)B * h->temb_total);
    w.ab0 = A.get<float2>((size_t)B * max_c);
    w.ab1 = A.get<float2>((size_t)B * max_c);
    w.ab2 = A.get<float2>((size_t)B * max_c);
    const size_t st_elems = (size_t)B * 8 * 64;  // [B][<= 8 slots][256/4 quads]
    auto act = [&](size_t elems, size_t st_n) {
        Act a;
        a.p = A.take(elems * tsz);
        a.st = A.get<float2>(st_n);
        return a;
    };
    w.skip.clear();
    for (const Block& b : h->enc)  // the MFMA stem writes one statistics slot per 32 pixels: [B][hw/32][128/4]
        w.skip.push_back(act((size_t)B * b.res_out * b.res_out * b.cout,
                             (b.kind == K_STEM && b.p_stem) ? (size_t)B * b.res_out * b.res_out : st_elems));
    w.xa = act((size_t)B * max_act, st_elems);
    w.xb = act((size_t)B * max_act, st_elems);
    w.h = act((size_t)B * max_act, st_elems);
    w.sbuf = A.take((size_t)B * max_act * tsz);
    w.pool = A.take((size_t)B * max_act * tsz);
    w.cvt1 = A.take((size_t)B * max_act * 4 * tsz);  // run_block inputs: up to 512 channels at the input resolution
    w.cvt2 = A.take((size_t)B * max_act * 4 * tsz);
    w.xattn = act((size_t)B * max_attn_hw * 256, st_elems);
    w.aout = A.take((size_t)B * max_attn_hw * 256 * tsz);
    w.q = A.take((size_t)B * max_attn_hw * 256 * tsz);
    w.k = A.take((size_t)B * max_attn_hw * 256 * tsz);
    w.vt = A.take((size_t)B * max_attn_hw * 256 * tsz);
    const size_t img = (size_t)B * c.img_channels * R * R;
    w.x = A.get<float>(img);
    w.x_pred = A.get<float>(img);
    w.eps = A.get<float>(img);
    w.tl = A.get<double>(72);
    w.seed = A.get<uint64_t>(8);
    return (A.off + 255) & ~(size_t)255;
}

int dev_alloc(fg_edm* h, void** p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes));
    h->owned.push_back(*p);
    return FG_OK;
}

// launch_conv_fused + optional event pair when profiling the dominant kernel class
int conv_launch(fg_edm* h, int ks, int pro, int res, int outmode, const ConvArgs& a, hipStream_t s) {
    const bool timed = h->prof_on && ks == 3 && res == RES_NONE && a.W == 32 && outmode == OUT_NHWC;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (int)hipErrorUnknown;
        (void)hipEventRecord(e0, s);
    }
    const int rc = launch_conv_fused(h->dtype, ks, pro, res, outmode, a, s);
    if (timed) {
        (void)hipEventRecord(e1, s);
        h->prof_ev.push_back(e0);
        h->prof_ev.push_back(e1);
        h->prof_flops += 2.0 * a.B * a.H * a.W * (double)a.Cout * 9.0 * (a.C1 + a.C2);
    }
    return rc;
}

const float kSkipScale = (float)std::sqrt(0.5);  // block_kwargs.skip_scale, EDM/network.py:385
const float kBlockEps = 1e-6f;                   // block_kwargs.eps :386, aux_norm :483

// GroupNorm coefficients of the virtual concat [x1 | x2]: from the producers' partial statistics when every source
// has them, else one full pass over the tensor(s).
int norm_coeffs(int dtype, const Act& x1, int c1, const Act& x2, int c2, const float* gamma, const float* beta, float2* ab,
                int B, int hw, hipStream_t s, float2* mr = nullptr) {
    if (x1.st && (!c2 || x2.st))
        HIP_TRY(launch_gn_finalize(x1.st, c1, x1.slots, c2 ? x2.st : nullptr, c2, c2 ? x2.slots : 0, gamma, beta, kBlockEps, ab, B, hw, s, mr));
    else
        HIP_TRY(launch_gn_coeffs(dtype, x1.p, c1, c2 ? x2.p : nullptr, c2, gamma, beta, kBlockEps, ab, B, hw, s, mr));
    return FG_OK;
}

// Point the per-block scratch of the workspace at a block's stash (its forward then leaves everything the backward reads there).
void use_stash(Workspace& w, const BlockStash& st) {
    w.h.p = st.h;
    w.ab0 = st.ab0, w.ab1 = st.ab1, w.mr0 = st.mr0, w.mr1 = st.mr1;
    if (st.xattn) {
        w.xattn.p = st.xattn;
        w.ab2 = st.ab2, w.mr2 = st.mr2;
        w.q = st.q, w.k = st.k, w.vt = st.vt, w.aout = st.aout;
    }
}
int block_index(const fg_edm* h, const Block* b) {
    for (size_t i = 0; i < h->blocks.size(); ++i)
        if (h->blocks[i] == b) return (int)i;
    return -1;
}

// One UNetBlock (EDM/network.py:274-299) as 5-10 kernel launches.  `out.st` receives the block output's statistics.
int run_block(fg_edm* h, const Block& b, const Act& x1, int c1, const Act& x2, int c2, const float* temb, Act& out, int B,
              Workspace& w, hipStream_t s) {
    if (c1 + c2 != b.cin) return fail(FG_EINVAL, "%s: got %d+%d input channels, expected %d", b.key.c_str(), c1, c2, b.cin);
    const int hw_in = b.res_in * b.res_in, hw = b.res_out * b.res_out;
    const int res_mode = b.down ? RES_DOWN : (b.up ? RES_UP : RES_NONE);
    const int slots = conv_stat_slots(b.res_out);
    int rc;
    // h = conv0(silu(norm0(x))) + affine(emb)
    if ((rc = norm_coeffs(h->dtype, x1, c1, x2, c2, h->P(b.norm0_w), h->P(b.norm0_b), w.ab0, B, hw_in, s, w.mr0))) return rc;
    ConvArgs a{};
    a.src1 = x1.p; a.src2 = c2 ? x2.p : nullptr; a.C1 = c1; a.C2 = c2;
    a.Hs = a.Ws = b.res_in; a.H = a.W = b.res_out; a.B = B;
    a.ab = w.ab0; a.wpack = b.p_conv0; a.wpack_ws = b.p_conv0_ws; a.bias = h->P(b.conv0_b);
    a.temb = temb + b.temb_off; a.temb_stride = h->temb_total;
    a.resid = nullptr; a.scale = 1.0f; a.out = w.h.p; a.Cout = b.cout; a.stats = w.h.st;
    if (b.down && !c2) {
        // pooled silu(norm0(x)) written once by a small pass (4x less transform work than pooling inside the conv)
        HIP_TRY(launch_gn_silu_pool(h->dtype, x1.p, w.ab0, w.pool, B, b.res_out, b.res_out, c1, s));
        a.src1 = w.pool; a.Hs = a.Ws = b.res_out; a.ab = nullptr;
        HIP_TRY(conv_launch(h, 3, PRO_NONE, RES_NONE, OUT_NHWC, a, s));
    } else {
        HIP_TRY(conv_launch(h, 3, PRO_GN_SILU, res_mode, OUT_NHWC, a, s));
    }
    w.h.slots = slots;
    // skip path
    const void* resid = x1.p;
    if (b.has_skip) {
        ConvArgs k{};
        k.src1 = x1.p; k.src2 = c2 ? x2.p : nullptr; k.C1 = c1; k.C2 = c2;
        k.Hs = k.Ws = b.res_in; k.H = k.W = b.res_out; k.B = B;
        k.wpack = b.p_skip; k.bias = h->P(b.skip_b); k.scale = 1.0f; k.out = w.sbuf; k.Cout = b.cout;
        HIP_TRY(conv_launch(h, 1, PRO_NONE, res_mode, OUT_NHWC, k, s));
        resid = w.sbuf;
    }
    // x = (conv1(silu(norm1(h))) + skip) * sqrt(.5)
    HIP_TRY(launch_gn_finalize(w.h.st, b.cout, slots, nullptr, 0, 0, h->P(b.norm1_w), h->P(b.norm1_b), kBlockEps, w.ab1, B, hw, s, w.mr1));
    Act& x_mid = b.attn ? w.xattn : out;
    ConvArgs d{};
    d.src1 = w.h.p; d.C1 = b.cout; d.Hs = d.Ws = d.H = d.W = b.res_out; d.B = B;
    d.ab = w.ab1; d.wpack = b.p_conv1; d.wpack_ws = b.p_conv1_ws; d.bias = h->P(b.conv1_b);
    d.resid = resid; d.scale = kSkipScale; d.out = x_mid.p; d.Cout = b.cout; d.stats = x_mid.st;
    HIP_TRY(conv_launch(h, 3, PRO_GN_SILU, RES_NONE, OUT_NHWC, d, s));
    x_mid.slots = slots;
    if (b.attn) {
        HIP_TRY(launch_gn_finalize(x_mid.st, b.cout, slots, nullptr, 0, 0, h->P(b.norm2_w), h->P(b.norm2_b), kBlockEps, w.ab2, B, hw, s, w.mr2));
        ConvArgs q{};
        q.src1 = x_mid.p; q.C1 = b.cout; q.Hs = q.Ws = q.H = q.W = b.res_out; q.B = B;
        q.ab = w.ab2; q.wpack = b.p_qkv; q.bias = b.qkv_bias; q.scale = 1.0f; q.Cout = 3 * b.cout;
        q.q_out = w.q; q.k_out = w.k; q.vt_out = w.vt;
        HIP_TRY(conv_launch(h, 1, PRO_GN, RES_NONE, OUT_QKV, q, s));
        HIP_TRY(launch_attention(h->dtype, w.q, w.k, w.vt, w.aout, B, hw, s));
        ConvArgs p{};
        p.src1 = w.aout; p.C1 = b.cout; p.Hs = p.Ws = p.H = p.W = b.res_out; p.B = B;
        p.wpack = b.p_proj; p.bias = h->P(b.proj_b); p.resid = x_mid.p; p.scale = kSkipScale; p.out = out.p; p.Cout = b.cout;
        p.stats = out.st;
        HIP_TRY(conv_launch(h, 1, PRO_NONE, RES_NONE, OUT_NHWC, p, s));
        out.slots = slots;
    }
    return FG_OK;
}

int run_mapping(fg_edm* h, const float* labels, int B, Workspace& w, hipStream_t s) {
    const fg_edm_config& c = h->cfg;
    HIP_TRY(launch_mapping_in(w.coef + B, w.coef + 4 * (size_t)B, h->freqs, labels, c.label_dim,
                              h->P(h->find("model.map_label.weight")), h->P(h->find("model.map_label.bias")), w.emb0, B,
                              h->cond_ch, h->noise_ch, s));
    HIP_TRY(launch_linear(w.emb0, h->P(h->find("model.map_layer0.weight")), h->P(h->find("model.map_layer0.bias")), w.emb1,
                          B, h->cond_ch, h->emb_ch, 1, s));
    HIP_TRY(launch_linear(w.emb1, h->P(h->find("model.map_layer1.weight")), h->P(h->find("model.map_layer1.bias")), w.emb, B,
                          h->emb_ch, h->emb_ch, 1, s));
    HIP_TRY(launch_linear(w.emb, h->aff_w, h->aff_b, w.temb, B, h->emb_ch, h->temb_total, 0, s));
    return FG_OK;
}

// EDMPrecond.forward (eval, fwd_pred_type = net_pred_type): EDM/network.py:881-974 + SongUNet.forward :489-574.
// r (target time of r_timestep networks) is required iff cfg.r_timestep.
// feats (nullable): one entry per resolution level, the NCHW fp32 destination of that level's `block3` encoder output
// (SongUNet.forward feature taps, EDM/network.py:535-539) or nullptr; early: return after the encoder (:542-544).
int run_forward(fg_edm* h, const float* x_t, const double* t, int t_stride, const double* r, int r_stride,
                const float* labels, float* out, int B, Workspace& w, hipStream_t s, float* const* feats = nullptr,
                bool early = false, TrainStash* ts = nullptr) {
    const fg_edm_config& c = h->cfg;
    HIP_TRY(launch_precond_coef(t, t_stride, c.r_timestep ? r : nullptr, r_stride, c.sigma_data, c.sigma_shift, 1e-6,
                                c.drop_precond, w.coef, B, s));
    int rc = run_mapping(h, labels, B, w, s);
    if (rc) return rc;
    // encoder
    const Act none;
    const Act* x = nullptr;
    for (size_t i = 0; i < h->enc.size(); ++i) {
        const Block& b = h->enc[i];
        if (b.kind == K_STEM) {
            if (b.p_stem) {  // MFMA stem: leaves GroupNorm partial statistics, one slot per 32 pixels
                HIP_TRY(launch_stem(h->dtype, x_t, w.coef, b.p_stem, h->P(b.b), w.skip[i].p, w.skip[i].st, B, b.res_out, b.cin, s));
                w.skip[i].slots = b.res_out * b.res_out / 32;
            } else {
                HIP_TRY(launch_conv_in(h->dtype, x_t, w.coef, h->P(b.w), h->P(b.b), w.skip[i].p, B, b.res_out, b.cin, b.cout, s));
                w.skip[i].st = nullptr;  // no statistics: block0's norm0 takes the full-pass fallback
            }
        } else {
            if (ts) use_stash(w, ts->blocks[block_index(h, &b)]);
            rc = run_block(h, b, *x, b.cin, none, 0, w.temb, w.skip[i], B, w, s);
            if (rc) return rc;
            if (feats && b.tap >= 0 && feats[b.tap])
                HIP_TRY(launch_act_to_nchw(h->dtype, w.skip[i].p, feats[b.tap], B, b.cout, b.res_out * b.res_out, s));
        }
        x = &w.skip[i];
    }
    if (early) return FG_OK;
    // decoder (skip stack popped from the back; concat is virtual)
    int sp = (int)h->enc.size();
    Act* pong[2] = {&w.xa, &w.xb};
    int cur = 0;
    size_t di = 0;
    const float2* aux_ab = nullptr;
    for (const Block& b : h->dec) {
        if (b.kind == K_BLOCK) {
            const Act& x2 = b.skip_c ? w.skip[--sp] : none;
            // a backward pass follows: every block output is kept instead of ping-ponging two buffers, and the block's
            // intermediates land in its stash
            Act& dst = ts ? ts->dec_store[di++] : *pong[cur];
            if (ts) use_stash(w, ts->blocks[block_index(h, &b)]);
            rc = run_block(h, b, *x, b.cin - b.skip_c, x2, b.skip_c, w.temb, dst, B, w, s);
            if (rc) return rc;
            x = &dst;
            cur ^= 1;
        } else if (b.kind == K_AUX_NORM) {
            if (ts) w.ab0 = ts->aux_ab, w.mr0 = ts->aux_mr;  // not a block's stash
            if ((rc = norm_coeffs(h->dtype, *x, b.cin, none, 0, h->P(b.w), h->P(b.b), w.ab0, B, b.res_in * b.res_in, s, ts ? w.mr0 : nullptr))) return rc;
            aux_ab = w.ab0;
        } else if (b.kind == K_AUX_CONV) {
            if (b.p_aux)
                HIP_TRY(launch_aux_head(h->dtype, x->p, aux_ab, b.p_aux, h->P(b.b), x_t, w.coef, out, B, b.cin, b.cout, s));
            else
                HIP_TRY(launch_aux_out(h->dtype, x->p, aux_ab, h->P(b.w), h->P(b.b), x_t, w.coef, out, B, b.res_out, b.cin, b.cout, s));
        }
    }
    return FG_OK;
}

int enqueue_sampler(fg_edm* h, const float* noise, const float* labels, const double* t_list, int steps, int type,
                    int loop, const float* eps, float* out, int B, Workspace& w, hipStream_t s) {
    const fg_edm_config& c = h->cfg;
    const int sched = c.schedule;
    const int64_t total = (int64_t)B * c.img_channels * c.img_resolution * c.img_resolution;
    HIP_TRY(launch_latents(noise, 0.0, w.tl, 0, w.x, total, s));  // latents = noise * sigma(t_0), noise_schedule.py:72-88
    auto sde_noise = [&](int i, const float** e) -> int {
        if (eps) {
            *e = eps + (size_t)i * total;
        } else {
            HIP_TRY(launch_randn(w.eps, total, 0, (uint64_t)i, w.seed, s));
            *e = w.eps;
        }
        return FG_OK;
    };
    if (loop == FG_LOOP_MEANFLOW) {
        // MeanFlowModel._student_sample_loop (consistency_model/mean_flow.py:336-381): the network output is the
        // average velocity u(x, t, r); 'sde' jumps to r = 0 and re-noises, 'ode' integrates t_cur -> t_next.
        for (int i = 0; i < steps; ++i) {
            const double* r = (type == FG_SAMPLE_SDE) ? w.tl + steps : w.tl + i + 1;  // t_list[steps] == 0
            int rc = run_forward(h, w.x, w.tl + i, 0, r, 0, labels, w.x_pred, B, w, s);
            if (rc) return rc;
            const bool renoise = type == FG_SAMPLE_SDE && t_list[i + 1] > 0;
            float* dst = (i == steps - 1 && !renoise) ? out : w.x;
            HIP_TRY(launch_meanflow_update(w.x, w.x_pred, w.tl, i, type == FG_SAMPLE_SDE ? -1 : i + 1, dst, total, s));
            if (renoise) {
                const float* e = nullptr;
                if ((rc = sde_noise(i, &e))) return rc;
                HIP_TRY(launch_forward_process(w.x, e, 0.0, w.tl, i + 1, sched, w.x, total, s));
            }
        }
        return FG_OK;
    }
    // FastGenModel._student_sample_loop (methods/model.py:315-372): x0 prediction, then re-noise to t_next
    for (int i = 0; i < steps; ++i) {
        float* pred = (i == steps - 1) ? out : w.x_pred;
        int rc = run_forward(h, w.x, w.tl + i, 0, w.tl + steps, 0, labels, pred, B, w, s);
        if (rc) return rc;
        if (t_list[i + 1] > 0) {  // methods/model.py:356 — decided on the host, baked into the graph
            const float* e = nullptr;
            if (type == FG_SAMPLE_SDE) {
                if ((rc = sde_noise(i, &e))) return rc;
            } else {
                HIP_TRY(launch_x0_to_eps(w.x, pred, 0.0, w.tl, i, sched, 1e-6, w.eps, total, s));
                e = w.eps;
            }
            // t_list[-1] must be 0 (model.py:410), so the last step never re-noises
            HIP_TRY(launch_forward_process(pred, e, 0.0, w.tl, i + 1, sched, w.x, total, s));
        }
    }
    return FG_OK;
}

void drop_graph(fg_edm* h) {
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->graph_exec = nullptr;
    h->graph = nullptr;
}

int setup_ws(const fg_edm* h, int B, void* workspace, size_t bytes, Workspace& w) {
    if (B <= 0) return fail(FG_EINVAL, "batch must be positive");
    if (!workspace) return fail(FG_EINVAL, "workspace is null");
    if (((uintptr_t)workspace) & 255) return fail(FG_EINVAL, "workspace must be 256-byte aligned");
    Arena A;
    A.base = (char*)workspace;
    const size_t need = plan_workspace(h, B, A, w);
    if (need > bytes) return fail(FG_ENOMEM, "workspace too small: need %zu bytes for batch %d, got %zu", need, B, bytes);
    return FG_OK;
}

// Device-side state (packed-weight storage, stacked affine matrix, frequency table, pinned staging).  Deferred to the
// first fg_edm_pack_weights() so that fg_edm_create() is host-only and works without a GPU.
int ensure_device_state(fg_edm* h) {
    if (h->device_ready) return FG_OK;
    int rc;
    const size_t tsz = h->dtype ? 2 : 4;
    for (Block* b : h->blocks) {
        if ((rc = dev_alloc(h, &b->p_conv0, conv_pack_elems(b->cout, b->cin, 3) * tsz))) return rc;
        if ((rc = dev_alloc(h, &b->p_conv1, conv_pack_elems(b->cout, b->cout, 3) * tsz))) return rc;
        if (conv_ws_shape_ok(h->dtype, b->cout, b->cin, b->res_out) && !b->down &&
            (rc = dev_alloc(h, &b->p_conv0_ws, conv_pack_elems(b->cout, b->cin, 3) * tsz)))
            return rc;
        if (conv_ws_shape_ok(h->dtype, b->cout, b->cout, b->res_out) &&
            (rc = dev_alloc(h, &b->p_conv1_ws, conv_pack_elems(b->cout, b->cout, 3) * tsz)))
            return rc;
        if (b->has_skip && (rc = dev_alloc(h, &b->p_skip, conv_pack_elems(b->cout, b->cin, 1) * tsz))) return rc;
        if (b->attn) {
            if ((rc = dev_alloc(h, &b->p_qkv, conv_pack_elems(3 * b->cout, b->cout, 1) * tsz))) return rc;
            if ((rc = dev_alloc(h, &b->p_proj, conv_pack_elems(b->cout, b->cout, 1) * tsz))) return rc;
            if ((rc = dev_alloc(h, (void**)&b->qkv_bias, sizeof(float) * 3 * b->cout))) return rc;
        }
    }
    for (Block& b : h->enc)
        if (b.kind == K_STEM && stem_supported(b.res_out, b.cin, b.cout))
            if ((rc = dev_alloc(h, &b.p_stem, stem_pack_elems() * tsz))) return rc;
    for (Block& b : h->dec)
        if (b.kind == K_AUX_CONV && aux_head_supported(h->dtype, b.res_out, b.cin, b.cout))
            if ((rc = dev_alloc(h, &b.p_aux, aux_pack_elems(b.cin) * tsz))) return rc;
    if ((rc = dev_alloc(h, (void**)&h->aff_w, sizeof(float) * (size_t)h->temb_total * h->emb_ch))) return rc;
    if ((rc = dev_alloc(h, (void**)&h->aff_b, sizeof(float) * (size_t)h->temb_total))) return rc;
    // PositionalEmbedding(endpoint=True) frequencies in fp32, EDM/network.py:314-316
    const int half = h->noise_ch / 2;
    std::vector<float> fr(half);
    for (int j = 0; j < half; ++j) fr[j] = powf(1.0f / 10000.0f, (float)j / (float)(half - 1));
    if ((rc = dev_alloc(h, (void**)&h->freqs, sizeof(float) * half))) return rc;
    HIP_TRY(hipMemcpy(h->freqs, fr.data(), sizeof(float) * half, hipMemcpyHostToDevice));
    if (conv_prepare_all(h->dtype) != 0) return fail(FG_EHIP, "hipFuncSetAttribute(dynamic LDS) failed");
    if (launch_attention(h->dtype, nullptr, nullptr, nullptr, nullptr, 1, 256, nullptr) != 0) return fail(FG_EHIP, "attention prepare failed");
    HIP_TRY(hipHostMalloc((void**)&h->slots, sizeof(fg_edm::Slot) * fg_edm::kSlots));
    for (int i = 0; i < fg_edm::kSlots; ++i) HIP_TRY(hipEventCreateWithFlags(&h->slot_ev[i], hipEventDisableTiming));
    h->device_ready = true;
    return FG_OK;
}

}  // namespace

// ================================================ C ABI =====================================================
extern "C" {

const char* fg_last_error(void) { return g_err.c_str(); }
const char* fg_version(void) { return "fastgen_amd 0.1 gfx950"; }

int fg_edm_create(const fg_edm_config* cfg, fg_edm** out) {
    if (!cfg || !out) return fail(FG_EINVAL, "null argument");
    if (cfg->num_levels < 1 || cfg->num_levels > FG_MAX_LEVELS || cfg->num_attn_resolutions < 0 ||
        cfg->num_attn_resolutions > FG_MAX_LEVELS)
        return fail(FG_EINVAL, "bad num_levels / num_attn_resolutions");
    if (cfg->compute_dtype != FG_DTYPE_F32 && cfg->compute_dtype != FG_DTYPE_BF16) return fail(FG_EINVAL, "bad compute_dtype");
    if ((cfg->drop_precond & ~3) || (cfg->schedule != FG_SCHEDULE_EDM && cfg->schedule != FG_SCHEDULE_RF) ||
        (cfg->r_timestep & ~1))
        return fail(FG_EINVAL, "bad r_timestep / drop_precond / schedule");
    if (cfg->model_channels <= 0 || cfg->model_channels % 16 || cfg->channel_mult_noise < 1 || cfg->channel_mult_emb < 1)
        return fail(FG_EINVAL, "bad channel configuration");
    fg_edm* h = new fg_edm();
    h->cfg = *cfg;
    h->dtype = cfg->compute_dtype;
    h->emb_ch = cfg->model_channels * cfg->channel_mult_emb;
    h->noise_ch = cfg->model_channels * cfg->channel_mult_noise;
    h->cond_ch = h->noise_ch * (cfg->r_timestep ? 2 : 1);
    build_layout(h);
    int rc = check_supported(h);
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return FG_OK;
}

void fg_edm_destroy(fg_edm* h) {
    if (!h) return;
    drop_graph(h);
    for (void* p : h->owned) (void)hipFree(p);
    if (h->slots) (void)hipHostFree(h->slots);
    for (hipEvent_t e : h->slot_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    delete h;
}

int fg_edm_num_params(const fg_edm* h) { return h ? (int)h->params.size() : 0; }

int fg_edm_param_info(const fg_edm* h, int index, const char** name, int* ndim, int64_t shape[4]) {
    if (!h || index < 0 || index >= (int)h->params.size()) return fail(FG_EINVAL, "param index out of range");
    const Param& p = h->params[index];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = p.ndim;
    if (shape)
        for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
    return FG_OK;
}

int fg_edm_bind_param(fg_edm* h, const char* name, const float* device_ptr, int64_t numel) {
    if (!h || !name || !device_ptr) return fail(FG_EINVAL, "null argument");
    const int i = h->find(name);
    if (i < 0) return fail(FG_EINVAL, "unknown parameter '%s'", name);
    if (h->params[i].numel != numel)
        return fail(FG_EINVAL, "parameter '%s': expected %lld elements, got %lld", name, (long long)h->params[i].numel, (long long)numel);
    h->params[i].ptr = device_ptr;
    h->packed = false;
    return FG_OK;
}

int fg_edm_pack_weights(fg_edm* h, void* stream) {
    if (!h) return fail(FG_EINVAL, "null handle");
    hipStream_t s = (hipStream_t)stream;
    int rc0 = ensure_device_state(h);
    if (rc0) return rc0;
    for (const Param& p : h->params) {
        const bool unused = p.name == "model.map_augment.weight" || p.name.rfind("model.logvar_linear", 0) == 0;
        if (!p.ptr && !unused) return fail(FG_ENOTREADY, "parameter '%s' is not bound", p.name.c_str());
    }
    ++h->pack_epoch;
    for (Block* b : h->blocks) {
        HIP_TRY(launch_pack_conv_weights(h->dtype, h->P(b->conv0_w), b->p_conv0, b->cout, b->cin, 3, 0, s));
        HIP_TRY(launch_pack_conv_weights(h->dtype, h->P(b->conv1_w), b->p_conv1, b->cout, b->cout, 3, 0, s));
        if (b->p_conv0_ws) HIP_TRY(launch_pack_conv_weights_ws(h->P(b->conv0_w), b->p_conv0_ws, b->cout, b->cin, s));
        if (b->p_conv1_ws) HIP_TRY(launch_pack_conv_weights_ws(h->P(b->conv1_w), b->p_conv1_ws, b->cout, b->cout, s));
        if (b->has_skip) HIP_TRY(launch_pack_conv_weights(h->dtype, h->P(b->skip_w), b->p_skip, b->cout, b->cin, 1, 0, s));
        if (b->attn) {
            HIP_TRY(launch_pack_conv_weights(h->dtype, h->P(b->qkv_w), b->p_qkv, 3 * b->cout, b->cout, 1, 1, s));
            HIP_TRY(launch_pack_conv_weights(h->dtype, h->P(b->proj_w), b->p_proj, b->cout, b->cout, 1, 0, s));
            // bias o' = plane*C + c  <-  reference channel c*3 + plane: three strided 2-D copies
            for (int plane = 0; plane < 3; ++plane)
                HIP_TRY(hipMemcpy2DAsync(b->qkv_bias + plane * b->cout, sizeof(float), h->P(b->qkv_b) + plane,
                                         3 * sizeof(float), sizeof(float), b->cout, hipMemcpyDeviceToDevice, s));
        }
        HIP_TRY(hipMemcpyAsync(h->aff_w + (size_t)b->temb_off * h->emb_ch, h->P(b->aff_w),
                               sizeof(float) * (size_t)b->cout * h->emb_ch, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->aff_b + b->temb_off, h->P(b->aff_b), sizeof(float) * b->cout, hipMemcpyDeviceToDevice, s));
    }
    for (Block& b : h->enc)
        if (b.kind == K_STEM && b.p_stem) HIP_TRY(launch_pack_stem_weights(h->dtype, h->P(b.w), b.p_stem, b.cin, s));
    for (Block& b : h->dec)
        if (b.kind == K_AUX_CONV && b.p_aux) HIP_TRY(launch_pack_aux_weights(h->dtype, h->P(b.w), b.p_aux, b.cin, b.cout, s));
    drop_graph(h);
    h->packed = true;
    return FG_OK;
}

size_t fg_edm_workspace_bytes(const fg_edm* h, int batch) {
    if (!h || batch <= 0) return 0;
    Arena A;
    A.dry = true;
    Workspace w;
    return plan_workspace(h, batch, A, w);
}

int fg_edm_forward(fg_edm* h, const float* x_t, const double* t, const double* r, const float* class_labels, float* out,
                   float* emb_out, int batch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !x_t || !t || !out) return fail(FG_EINVAL, "null argument");
    if (r && !h->cfg.r_timestep) return fail(FG_EINVAL, "r_noise_labels provided, but r_timestep is not set");  // EDM/network.py:510
    if (!r && h->cfg.r_timestep) return fail(FG_EINVAL, "this network was built with r_timestep: r is required");
    if (!h->packed) return fail(FG_ENOTREADY, "weights are not packed (call fg_edm_pack_weights)");
    if (out == x_t) return fail(FG_EINVAL, "out must not alias x_t");
    Workspace w;
    int rc = setup_ws(h, batch, workspace, workspace_bytes, w);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = run_forward(h, x_t, t, 1, r, 1, class_labels, out, batch, w, s);
    if (rc) return rc;
    if (emb_out) HIP_TRY(hipMemcpyAsync(emb_out, w.emb, sizeof(float) * (size_t)batch * h->emb_ch, hipMemcpyDeviceToDevice, s));
    return FG_OK;
}

int fg_edm_num_feature_taps(const fg_edm* h) { return h ? h->num_taps : 0; }

int fg_edm_feature_info(const fg_edm* h, int index, const char** key, int* channels, int* resolution) {
    if (!h) return fail(FG_EINVAL, "null handle");
    for (const Block& b : h->enc)
        if (b.tap == index && index >= 0) {
            if (key) *key = b.key.c_str();
            if (channels) *channels = b.cout;
            if (resolution) *resolution = b.res_out;
            return FG_OK;
        }
    return fail(FG_EINVAL, "feature index out of range");
}

int fg_edm_forward_features(fg_edm* h, const float* x_t, const double* t, const double* r, const float* class_labels,
                            float* out, float* const* features, int batch, void* workspace, size_t workspace_bytes,
                            void* stream) {
    if (!h || !x_t || !t || !features) return fail(FG_EINVAL, "null argument");
    if (!h->packed) return fail(FG_ENOTREADY, "weights are not packed (call fg_edm_pack_weights)");
    if (out == x_t) return fail(FG_EINVAL, "out must not alias x_t");
    if (r && !h->cfg.r_timestep) return fail(FG_EINVAL, "r_noise_labels provided, but r_timestep is not set");
    if (!r && h->cfg.r_timestep) return fail(FG_EINVAL, "this network was built with r_timestep: r is required");
    for (const Block& b : h->enc)
        if (b.tap >= 0 && features[b.tap] && ((b.cout % 32) || ((b.res_out * b.res_out) % 32)))
            return fail(FG_EINVAL, "feature tap %d: channels and pixels must be multiples of 32", b.tap);
    Workspace w;
    int rc = setup_ws(h, batch, workspace, workspace_bytes, w);
    if (rc) return rc;
    return run_forward(h, x_t, t, 1, r, 1, class_labels, out, batch, w, (hipStream_t)stream, features, out == nullptr);
}

int fg_edm_t_list(int sample_steps, double* out_host) {
    if (sample_steps < 1 || sample_steps > 64 || !out_host) return fail(FG_EINVAL, "bad sample_steps");
    // EDMNoiseSchedule: sigma table (noise_schedule.py:752-756) and get_t_list (:940-973), all in fp64
    const int num_steps = 1000;
    const double min_t = 0.002, max_t = 80.0, rho = 7.0;
    const double a = std::pow(min_t, 1.0 / rho), b = std::pow(max_t, 1.0 / rho);
    const int lo = (int)(0.002 * num_steps), hi = (int)(0.998 * num_steps);
    for (int i = 0; i <= sample_steps; ++i) {
        // torch.linspace(hi, lo, n+1) in fp32 then .long(): symmetric evaluation as ATen does (first half from start,
        // second half from end)
        const int n = sample_steps + 1;
        const float step = ((float)lo - (float)hi) / (float)(n - 1);
        const float v = (i < n / 2) ? (float)hi + step * (float)i : (float)lo - step * (float)(n - 1 - i);
        const int idx = (int)v;
        // sigmas = flip((b + ramp*(a-b))^rho), ramp = linspace(0,1,1000) in fp64 (same symmetric evaluation)
        const int j = num_steps - 1 - idx;
        const double dstep = 1.0 / (double)(num_steps - 1);
        const double ramp = (j < num_steps / 2) ? dstep * j : 1.0 - dstep * (double)(num_steps - 1 - j);
        double sig = std::pow(b + ramp * (a - b), rho);
        if (i == sample_steps) sig = 0.0;
        out_host[i] = sig > max_t ? max_t : sig;
    }
    return FG_OK;
}

int fg_rf_t_list(int sample_steps, double* out_host) {
    if (sample_steps < 1 || sample_steps > 64 || !out_host) return fail(FG_EINVAL, "bad sample_steps");
    // BaseNoiseSchedule.get_t_list (noise_schedule.py:259-272): linspace(max_t, 0, n+1) in fp64, ATen's symmetric form
    const double max_t = 0.999;
    const int n = sample_steps + 1;
    const double step = (0.0 - max_t) / (double)(n - 1);
    for (int i = 0; i < n; ++i) {
        const double v = (i < n / 2) ? max_t + step * (double)i : 0.0 - step * (double)(n - 1 - i);
        out_host[i] = v > max_t ? max_t : v;
    }
    return FG_OK;
}

int fg_sampler_run(fg_edm* h, const float* noise, const float* class_labels, const double* t_list, int steps,
                   int sample_type, int loop_kind, const float* eps, uint64_t seed, float* out, int batch, void* workspace,
                   size_t workspace_bytes, int use_graph, void* stream) {
    if (!h || !noise || !t_list || !out) return fail(FG_EINVAL, "null argument");
    if (!h->packed) return fail(FG_ENOTREADY, "weights are not packed (call fg_edm_pack_weights)");
    if (steps < 1 || steps > 64) return fail(FG_EINVAL, "steps must be in [1, 64]");
    if (sample_type != FG_SAMPLE_SDE && sample_type != FG_SAMPLE_ODE) return fail(FG_EINVAL, "bad sample_type");
    if (t_list[steps] != 0.0) return fail(FG_EINVAL, "t_list[-1] must be zero");  // methods/model.py:410
    if (loop_kind != FG_LOOP_X0 && loop_kind != FG_LOOP_MEANFLOW) return fail(FG_EINVAL, "bad loop_kind");
    if ((loop_kind == FG_LOOP_MEANFLOW) != (h->cfg.r_timestep != 0))
        return fail(FG_EINVAL, "FG_LOOP_MEANFLOW needs an r_timestep network and FG_LOOP_X0 a network without one");
    const double t_lo = h->cfg.schedule == FG_SCHEDULE_RF ? 0.0 : 0.002, t_hi = h->cfg.schedule == FG_SCHEDULE_RF ? 0.999 : 80.0;
    for (int i = 0; i < steps; ++i)
        if (!(t_list[i] >= t_lo * (1 - 1e-12) && t_list[i] <= t_hi * (1 + 1e-12)))  // is_t_valid, noise_schedule.py:409-423
            return fail(FG_EINVAL, "t_list[%d] = %g outside [%g, %g]", i, t_list[i], t_lo, t_hi);
    Workspace w;
    int rc = setup_ws(h, batch, workspace, workspace_bytes, w);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    {   // upload this call's timesteps and seed (see fg_edm::Slot)
        const int si = h->slot_next;
        h->slot_next = (si + 1) % fg_edm::kSlots;
        if (h->slot_used[si]) HIP_TRY(hipEventSynchronize(h->slot_ev[si]));
        fg_edm::Slot& sl = h->slots[si];
        for (int i = 0; i <= steps; ++i) sl.tl[i] = t_list[i];
        sl.seed[0] = seed;
        sl.seed[1] = 0;
        HIP_TRY(hipMemcpyAsync(w.tl, sl.tl, sizeof(double) * (steps + 1), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(w.seed, sl.seed, sizeof(uint64_t) * 2, hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(h->slot_ev[si], s));
        h->slot_used[si] = true;
    }
    if (!use_graph || h->prof_on)
        return enqueue_sampler(h, noise, class_labels, t_list, steps, sample_type, loop_kind, eps, out, batch, w, s);

    GraphKey key;
    key.B = batch; key.steps = steps; key.type = sample_type; key.loop = loop_kind;
    for (int i = 1; i <= steps; ++i)
        if (t_list[i] > 0) key.zero_mask |= (1ull << i);
    key.noise = noise; key.labels = class_labels; key.eps = eps; key.out = out; key.ws = workspace;
    key.device_rng = (sample_type == FG_SAMPLE_SDE && !eps);
    if (!h->graph_exec || !(h->graph_key == key)) {
        drop_graph(h);
        if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
        hipStream_t cs = h->cap_stream;  // capture records, it does not execute: the graph is launched on `s` below
        HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        rc = enqueue_sampler(h, noise, class_labels, t_list, steps, sample_type, loop_kind, eps, out, batch, w, cs);
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(cs, &g);
        if (rc) {
            if (g) (void)hipGraphDestroy(g);
            return rc;
        }
        if (e != hipSuccess) return fail(FG_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        h->graph = g;
        HIP_TRY(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
        h->graph_key = key;
    }
    HIP_TRY(hipGraphLaunch(h->graph_exec, s));
    return FG_OK;
}

int fg_edm_profile_begin(fg_edm* h) {
    if (!h) return fail(FG_EINVAL, "null handle");
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    h->prof_ev.clear();
    h->prof_flops = 0.0;
    h->prof_on = true;
    return FG_OK;
}

int fg_edm_profile_end(fg_edm* h, int64_t* launches, double* total_ms, double* total_flops) {
    if (!h) return fail(FG_EINVAL, "null handle");
    h->prof_on = false;
    double ms = 0.0;
    for (size_t i = 0; i + 1 < h->prof_ev.size(); i += 2) {
        HIP_TRY(hipEventSynchronize(h->prof_ev[i + 1]));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, h->prof_ev[i], h->prof_ev[i + 1]));
        ms += t;
    }
    if (launches) *launches = (int64_t)h->prof_ev.size() / 2;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = h->prof_flops;
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    h->prof_ev.clear();
    return FG_OK;
}

// Debug/ablation micro-benchmark of one fused 3x3 conv (bf16 or fp32): allocates its own buffers, times `iters`
// launches with HIP events.  Not part of the product path; used by scripts/conv_ablate.py only.
int fg_debug_conv_bench(int dtype, int batch, int cin, int res, int ks, int with_resid, int dbg, int iters, float* ms_out) {
    const size_t npix = (size_t)batch * res * res;
    float *x = nullptr, *out = nullptr, *resid = nullptr, *bias = nullptr;
    float2* ab = nullptr;
    void* wp = nullptr;
    HIP_TRY(hipMalloc((void**)&x, npix * cin * 4));
    HIP_TRY(hipMalloc((void**)&out, npix * 256 * 4));
    HIP_TRY(hipMalloc((void**)&resid, npix * 256 * 4));
    HIP_TRY(hipMalloc((void**)&bias, 256 * 4));
    HIP_TRY(hipMalloc((void**)&ab, (size_t)batch * cin * 8));
    HIP_TRY(hipMalloc(&wp, (size_t)256 * cin * ks * ks * 4));
    // Pseudo-random finite operands (|v| in [2^-7, 2) as bf16 pairs, i.e. a valid fp32 too): constant fills run at a higher
    // clock than real data (DVFS) and ranked kernel variants differently from the end-to-end bench.
    {
        const size_t blk = 4u << 20;
        std::vector<uint16_t> hb(blk / 2);
        uint32_t st = 0x12345u;
        for (auto& v : hb) {
            st = st * 1664525u + 1013904223u;
            v = (uint16_t)(((st >> 16) & 0x8000u) | ((0x78u + ((st >> 12) & 7u)) << 7) | ((st >> 20) & 0x7fu));
        }
        auto fill = [&](void* dst, size_t bytes) -> int {
            for (size_t o = 0; o < bytes; o += blk)
                HIP_TRY(hipMemcpy((char*)dst + o, hb.data(), std::min(blk, bytes - o), hipMemcpyHostToDevice));
            return FG_OK;
        };
        int rc;
        if ((rc = fill(x, npix * cin * 4)) || (rc = fill(resid, npix * 256 * 4)) || (rc = fill(ab, (size_t)batch * cin * 8)) ||
            (rc = fill(wp, (size_t)256 * cin * ks * ks * (dtype ? 2 : 4))))
            return rc;
    }
    HIP_TRY(hipMemset(bias, 0, 256 * 4));
    if (conv_prepare_all(dtype) != 0) return fail(FG_EHIP, "prepare failed");
    ConvArgs a{};
    a.src1 = x; a.C1 = cin; a.Hs = a.Ws = a.H = a.W = res; a.B = batch;
    a.ab = ab; a.wpack = wp; a.wpack_ws = wp; a.bias = bias; a.resid = with_resid ? resid : nullptr; a.scale = 1.f; a.out = out; a.Cout = 256;
    a.dbg = dbg;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    auto run = [&]() { return dbg >= 64 ? launch_conv_ws_debug(a, dbg - 64, nullptr) : dbg < 0 ? launch_conv_fused(dtype, ks, ks == 3 ? PRO_GN_SILU : PRO_NONE, RES_NONE, OUT_NHWC, a, nullptr) : launch_conv_debug(dtype, a, nullptr); };
    for (int i = 0; i < 3; ++i) HIP_TRY(run());
    HIP_TRY(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) HIP_TRY(run());
    HIP_TRY(hipEventRecord(e1, nullptr));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(x); (void)hipFree(out); (void)hipFree(resid); (void)hipFree(bias); (void)hipFree(ab); (void)hipFree(wp);
    return FG_OK;
}

int fg_edm_num_blocks(const fg_edm* h) { return h ? (int)h->blocks.size() : 0; }

int fg_edm_block_info(const fg_edm* h, int index, const char** key, int* cin, int* cout, int* res_in, int* res_out,
                      int* has_attention) {
    if (!h || index < 0 || index >= (int)h->blocks.size()) return fail(FG_EINVAL, "block index out of range");
    const Block* b = h->blocks[index];
    if (key) *key = b->key.c_str();
    if (cin) *cin = b->cin;
    if (cout) *cout = b->cout;
    if (res_in) *res_in = b->res_in;
    if (res_out) *res_out = b->res_out;
    if (has_attention) *has_attention = b->attn ? 1 : 0;
    return FG_OK;
}

int fg_edm_run_block(fg_edm* h, int index, const float* x1, int c1, const float* x2, int c2, const float* emb, float* out,
                     int batch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !x1 || !emb || !out) return fail(FG_EINVAL, "null argument");
    if (!h->packed) return fail(FG_ENOTREADY, "weights are not packed (call fg_edm_pack_weights)");
    if (index < 0 || index >= (int)h->blocks.size()) return fail(FG_EINVAL, "block index out of range");
    Workspace w;
    int rc0 = setup_ws(h, batch, workspace, workspace_bytes, w);
    if (rc0) return rc0;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_linear(emb, h->aff_w, h->aff_b, w.temb, batch, h->emb_ch, h->temb_total, 0, s));
    // caller tensors are fp32 and carry no partial statistics: convert to the activation dtype; norm0 takes the
    // full-pass fallback
    const Block& b = *h->blocks[index];
    const size_t npix_in = (size_t)batch * b.res_in * b.res_in, npix_out = (size_t)batch * b.res_out * b.res_out;
    Act a1, a2;
    HIP_TRY(launch_to_act(h->dtype, x1, w.cvt1, (int64_t)npix_in * c1, s));
    a1.p = w.cvt1;
    if (c2) {
        HIP_TRY(launch_to_act(h->dtype, x2, w.cvt2, (int64_t)npix_in * c2, s));
        a2.p = w.cvt2;
    }
    int rc = run_block(h, b, a1, c1, a2, c2, w.temb, w.xa, batch, w, s);
    if (rc) return rc;
    HIP_TRY(launch_from_act(h->dtype, w.xa.p, out, (int64_t)npix_out * b.cout, s));
    return FG_OK;
}

// ---- training step, block level -------------------------------------------------------------------------
namespace {
struct BwdScratch {
    void *g1, *aop, *da, *dh0, *dskip, *dxin, *wpk, *wg;
    void *gmid = nullptr, *dq = nullptr, *dk = nullptr, *dvt = nullptr, *dqkv = nullptr, *att = nullptr;  // attention blocks only
    float2 *P, *S, *mr0, *mr1, *mr2;
    float *dtemb, *wt;
    size_t wg_bytes;
};
int pad256(int c) { return (c + 255) / 256 * 256; }
// split-K scratch the weight-gradient kernel needs for one block's convolutions (the split count depends on every dimension)
size_t block_wgrad_bytes(int B, int res, int cin, int cout, bool attn) {
    size_t m = 0;
    const int shapes[5][3] = {{cout, cout, 3}, {cin, cout, 3}, {cin, cout, 1}, {cout, 3 * cout, 1}, {cout, cout, 1}};
    for (int i = 0; i < (attn ? 5 : 3); ++i)
        if (conv_wgrad_supported(res, shapes[i][0], shapes[i][1], shapes[i][2]))
            m = std::max(m, conv_wgrad_workspace_bytes(B, res, shapes[i][0], shapes[i][1], shapes[i][2]));
    return m;
}
size_t plan_block_bwd(int B, int res_in, int res, int cin, int cout, Arena& A, BwdScratch& q, int attn_hw = 0, size_t wg_bytes = 0) {
    const size_t npix = (size_t)B * res * res, npix_in = (size_t)B * res_in * res_in;
    const int cp = pad256(cin), cm = cin > cout ? cin : cout;
    q.g1 = A.take(npix * cout * 2);
    q.aop = A.take(npix * cm * 2);
    q.da = A.take(npix * (cp > cout ? cp : cout) * 2);
    q.dh0 = A.take(npix * cout * 2);
    q.dskip = A.take(npix * cp * 2);
    q.dxin = A.take(npix_in * cin * 2);
    q.P = A.get<float2>((size_t)B * cm);
    q.S = A.get<float2>((size_t)B * 32);
    q.mr0 = A.get<float2>((size_t)B * 32);
    q.mr1 = A.get<float2>((size_t)B * 32);
    q.mr2 = A.get<float2>((size_t)B * 32);
    q.dtemb = A.get<float>((size_t)B * cout * 3);
    if (attn_hw) {
        q.gmid = A.take(npix * cout * 2);
        q.dq = A.take(npix * cout * 2);
        q.dk = A.take(npix * cout * 2);
        q.dvt = A.take(npix * cout * 2);
        q.dqkv = A.take(npix * cout * 3 * 2);
        q.att = A.take(attention_backward_scratch_bytes(B, attn_hw, cout));
    }
    const size_t welems = (size_t)(cp > cout ? cp : cout) * (cin > cout ? cin : cout) * 9;
    q.wt = A.get<float>(welems);
    q.wpk = A.take(welems * 2);
    q.wg_bytes = wg_bytes ? wg_bytes : block_wgrad_bytes(B, res, cin, cout, attn_hw != 0);
    q.wg = A.take(q.wg_bytes);
    return A.off;
}
// data gradient of a conv = the forward conv kernel on dY with transposed, flipped weights; output [npix][pad256(cin)]
int conv_dgrad(fg_edm* h, const float* w_oihw, int cout, int cin, int ks, const void* dy, void* out, int B, int res, BwdScratch& q,
               hipStream_t s, bool cache = true, float oscale = 1.0f) {
    const int cp = pad256(cin);
    void* wpk = q.wpk;
    if (cache) {  // parameters: packed once per weight version; scratch-built weights (the padded head) are not cached
        fg_edm::DgradW& e = h->dgrad_cache[w_oihw];
        if (!e.packed) {
            int rc = dev_alloc(h, &e.packed, (size_t)cp * cout * ks * ks * 2);
            if (rc) return rc;
        }
        if (e.epoch != h->pack_epoch) {
            HIP_TRY(launch_dgrad_weights(w_oihw, q.wt, cout, cin, cp, ks * ks, s));
            HIP_TRY(launch_pack_conv_weights(1, q.wt, e.packed, cp, cout, ks, 0, s));
            e.epoch = h->pack_epoch;
        }
        wpk = e.packed;
    } else {
        HIP_TRY(launch_dgrad_weights(w_oihw, q.wt, cout, cin, cp, ks * ks, s));
        HIP_TRY(launch_pack_conv_weights(1, q.wt, q.wpk, cp, cout, ks, 0, s));
    }
    ConvArgs a{};
    a.src1 = dy; a.C1 = cout; a.C2 = 0;
    a.Hs = a.Ws = a.H = a.W = res; a.B = B;
    a.wpack = wpk; a.scale = oscale; a.out = out; a.Cout = cp;
    HIP_TRY(launch_conv_fused(1, ks, PRO_NONE, RES_NONE, OUT_NHWC, a, s));
    return FG_OK;
}

// weight gradient with a host-side check of the split-K scratch (an undersized scratch would be an out-of-bounds write)
int wgrad_checked(const void* act, const void* dy, float* dw, int B, int res, int cin, int cout, int ks, int accumulate, void* wg,
                  size_t wg_bytes, hipStream_t s, float scale = 1.0f) {
    const size_t need = conv_wgrad_workspace_bytes(B, res, cin, cout, ks);
    if (need > wg_bytes) return fail(FG_ENOMEM, "weight-gradient scratch too small: %zu > %zu (res %d, %d -> %d, k%d)", need, wg_bytes, res, cin, cout, ks);
    HIP_TRY(launch_conv_wgrad(act, dy, dw, B, res, cin, cout, ks, accumulate, wg, s, scale));
    return FG_OK;
}

// Backward of one UNetBlock (bf16).  a1 / a2: the block's inputs (virtual concat) at res_in; gout: dL/d(block output), bf16
// [B, res_out^2, cout]; dxin: bf16 [B, res_in^2, cin] (overwritten); demb [B, emb_ch] and the bound parameter gradients are
// accumulated.  The block's forward is recomputed here (activation checkpointing at block granularity).
int block_backward(fg_edm* h, const Block& b, const Act& a1, int c1, const Act& a2, int c2, const float* emb, const float* temb,
                   const void* gout, void* dxin, float* demb, int B, Workspace& w, BwdScratch& q, hipStream_t s,
                   float* dtemb_all = nullptr, const BlockStash* stash = nullptr, void* dxin2 = nullptr, int dx_accumulate = 0) {
    // dxin2 != nullptr: the gradients of the two concat sources go to their own dense tensors (dxin: [.., c1], dxin2: [.., c2]);
    // dx_accumulate: added to what dxin (/ dxin2) hold
    const int res = b.res_out, res_in = b.res_in, hw = res * res, cin = b.cin, cout = b.cout, cp = pad256(cin);
    const int rm = b.down ? 1 : (b.up ? 2 : 0);
    if (!conv_wgrad_supported(res, cout, cout, 3) || !conv_wgrad_supported(res, cin, cout, 3))
        return fail(FG_EINVAL, "%s: shape not covered by the weight-gradient kernel", b.key.c_str());
    const size_t npix = (size_t)B * hw;
    int rc = FG_OK;
    if (stash) {
        // the forward that preceded this call left the block's intermediates in its stash: nothing to recompute
        use_stash(w, *stash);
        q.mr0 = stash->mr0, q.mr1 = stash->mr1, q.mr2 = stash->mr2;
    } else {
        w.mr0 = q.mr0;
        w.mr1 = q.mr1;
        w.mr2 = q.mr2;
        rc = run_block(h, b, a1, c1, a2, c2, temb, w.xa, B, w, s);
        w.mr0 = w.mr1 = w.mr2 = nullptr;
        if (rc) return rc;
    }
    // The gradient reaching a block's output is used as is; the residual scale sigma = sqrt(1/2) of
    // out = (branch + skip) * sigma travels as a factor into every consumer (no scaled copy, one bf16 rounding less).
    const float sg = kSkipScale;
    if (b.attn) {
        // out = (proj(attention(qkv(norm2(x_mid)))) + x_mid) * sigma, EDM/network.py:290-298
        if (!q.att) return fail(FG_EINVAL, "%s: scratch was planned without the attention part", b.key.c_str());
        const int C3 = 3 * cout;
        HIP_TRY(launch_colsum(gout, cout, cout, q.dtemb, B, hw, sg, s));
        if (h->G(b.proj_b)) HIP_TRY(launch_batchsum_add(q.dtemb, h->G(b.proj_b), B, cout, s));
        if (h->G(b.proj_w) && (rc = wgrad_checked(w.aout, gout, h->G(b.proj_w), B, res, cout, cout, 1, 1, q.wg, q.wg_bytes, s, sg))) return rc;
        if ((rc = conv_dgrad(h, h->P(b.proj_w), cout, cout, 1, gout, q.da, B, res, q, s, true, sg))) return rc;
        HIP_TRY(launch_attention_backward(w.q, w.k, w.vt, q.da, q.dq, q.dk, q.dvt, q.att, B, hw, cout, s));
        HIP_TRY(launch_qkv_interleave(q.dq, q.dk, q.dvt, q.dqkv, B, hw, cout, s));
        HIP_TRY(launch_colsum(q.dqkv, C3, C3, q.dtemb, B, hw, 1.0f, s));
        if (h->G(b.qkv_b)) HIP_TRY(launch_batchsum_add(q.dtemb, h->G(b.qkv_b), B, C3, s));
        if (h->G(b.qkv_w)) {
            HIP_TRY(launch_gn_act(1, w.xattn.p, cout, nullptr, 0, w.ab2, q.aop, B, res, 0, s));
            if ((rc = wgrad_checked(q.aop, q.dqkv, h->G(b.qkv_w), B, res, cout, C3, 1, 1, q.wg, q.wg_bytes, s))) return rc;
        }
        if ((rc = conv_dgrad(h, h->P(b.qkv_w), C3, cout, 1, q.dqkv, q.da, B, res, q, s))) return rc;
        // norm2 (no activation); the residual x_mid -> out contributes sigma * gout directly
        HIP_TRY(launch_gn_bwd(1, w.xattn.p, cout, nullptr, 0, q.da, pad256(cout), w.ab2, q.mr2, h->P(b.norm2_w), q.P, q.S,
                              h->G(b.norm2_w), h->G(b.norm2_b), gout, cout, sg, q.gmid, B, res, 0, s));
        gout = q.gmid;
    }
    // x_mid = (conv1(act1) + skip) * sigma: sigma * gout reaches conv1's output and the skip path alike
    HIP_TRY(launch_colsum(gout, cout, cout, q.dtemb, B, hw, sg, s));
    {
        float* d1 = h->G(b.conv1_b);
        float* d2 = b.has_skip ? h->G(b.skip_b) : nullptr;
        if (d1 || d2) HIP_TRY(launch_batchsum_add(q.dtemb, d1 ? d1 : d2, B, cout, s, d1 ? d2 : nullptr));
    }
    // conv1
    if (h->G(b.conv1_w)) {
        HIP_TRY(launch_gn_act(0, w.h.p, cout, nullptr, 0, w.ab1, q.aop, B, res, 0, s));
        if ((rc = wgrad_checked(q.aop, gout, h->G(b.conv1_w), B, res, cout, cout, 3, 1, q.wg, q.wg_bytes, s, sg))) return rc;
    }
    if ((rc = conv_dgrad(h, h->P(b.conv1_w), cout, cout, 3, gout, q.da, B, res, q, s, true, sg))) return rc;
    // norm1 + silu
    HIP_TRY(launch_gn_bwd(0, w.h.p, cout, nullptr, 0, q.da, pad256(cout), w.ab1, q.mr1, h->P(b.norm1_w), q.P, q.S, h->G(b.norm1_w),
                          h->G(b.norm1_b), nullptr, 0, 0.f, q.dh0, B, res, 0, s));
    // bias of conv0 and the embedding affine see the pixel sum of dh0
    if (dtemb_all) {
        // whole-network pass: parked in the stacked [B][temb_total] matrix; biases, affine weights and demb follow in one go
        HIP_TRY(launch_colsum(q.dh0, cout, cout, dtemb_all + b.temb_off, B, hw, 1.0f, s, h->temb_total));
    } else {
        HIP_TRY(launch_colsum(q.dh0, cout, cout, q.dtemb, B, hw, 1.0f, s));
        if (h->G(b.conv0_b)) HIP_TRY(launch_batchsum_add(q.dtemb, h->G(b.conv0_b), B, cout, s));
        if (h->G(b.aff_b)) HIP_TRY(launch_batchsum_add(q.dtemb, h->G(b.aff_b), B, cout, s));
        HIP_TRY(launch_affine_bwd(q.dtemb, emb, h->P(b.aff_w), h->G(b.aff_w), demb, B, cout, h->emb_ch, s));
    }
    // conv0: its operand is silu(norm0(x)), resampled to the output resolution
    if (h->G(b.conv0_w)) {
        HIP_TRY(launch_gn_act(0, a1.p, c1, a2.p, c2, w.ab0, q.aop, B, res, rm, s));
        if ((rc = wgrad_checked(q.aop, q.dh0, h->G(b.conv0_w), B, res, cin, cout, 3, 1, q.wg, q.wg_bytes, s))) return rc;
    }
    if ((rc = conv_dgrad(h, h->P(b.conv0_w), cout, cin, 3, q.dh0, q.da, B, res, q, s))) return rc;
    // skip path: its gradient joins dx_in inside the norm0 backward pass (both live at the output resolution)
    const void* add = gout;
    int ca = cout;
    float add_scale = sg;
    if (b.has_skip) {
        if (h->G(b.skip_w)) {
            HIP_TRY(launch_gn_act(2, a1.p, c1, a2.p, c2, nullptr, q.aop, B, res, rm, s));
            if ((rc = wgrad_checked(q.aop, gout, h->G(b.skip_w), B, res, cin, cout, 1, 1, q.wg, q.wg_bytes, s, sg))) return rc;
        }
        if ((rc = conv_dgrad(h, h->P(b.skip_w), cout, cin, 1, gout, q.dskip, B, res, q, s, true, sg))) return rc;
        add = q.dskip;
        ca = cp;
        add_scale = 1.0f;
    }
    HIP_TRY(launch_gn_bwd(0, a1.p, c1, a2.p, c2, q.da, cp, w.ab0, q.mr0, h->P(b.norm0_w), q.P, q.S, h->G(b.norm0_w), h->G(b.norm0_b),
                          add, ca, add_scale, dxin, B, res_in, rm, s, dxin2, dx_accumulate));
    return FG_OK;
}

// ---- whole network: forward keeping every block input, then the blocks in reverse ----------------------------------------------
struct NetBwd {
    TrainStash ts;
    std::vector<void*> genc;      // gradient w.r.t. each encoder output (bf16)
    void *ga, *gb;                // running gradient, ping-pong
    void *dfp, *op32;             // padded head gradient [npix][128], padded stem operand [npix][32]
    float *wtmp, *wpad, *vec;     // padded weight-gradient / weight scratch, small vector scratch
    float *demb, *pre, *d1, *d0;  // embedding MLP
    float *dtemb_all, *vec_all;   // [B][temb_total] pixel sums of every block's dh0, and their batch sum
    void* wgx;                    // split-K scratch of the head / stem weight gradients
    size_t wgx_bytes;
    BwdScratch q;
};
size_t plan_net_bwd(const fg_edm* h, int B, Arena& A, NetBwd& nb) {
    const size_t tsz = 2;
    size_t max_act = 0, max_in = 0;
    int max_res = 0, max_cin = 0, attn_hw = 0;
    size_t wg_max = 0;
    const size_t st_elems = (size_t)B * 8 * 64;
    nb.ts.dec_store.clear();
    nb.ts.blocks.clear();
    nb.genc.clear();
    nb.ts.aux_ab = A.get<float2>((size_t)B * 256);
    nb.ts.aux_mr = A.get<float2>((size_t)B * 32);
    for (const Block* b : h->blocks) {
        const size_t np = (size_t)B * b->res_out * b->res_out;
        BlockStash st;
        st.h = A.take(np * b->cout * tsz);
        st.ab0 = A.get<float2>((size_t)B * b->cin);
        st.ab1 = A.get<float2>((size_t)B * b->cout);
        st.mr0 = A.get<float2>((size_t)B * 32);
        st.mr1 = A.get<float2>((size_t)B * 32);
        if (b->attn) {
            st.ab2 = A.get<float2>((size_t)B * b->cout);
            st.mr2 = A.get<float2>((size_t)B * 32);
            st.xattn = A.take(np * b->cout * tsz);
            st.q = A.take(np * b->cout * tsz);
            st.k = A.take(np * b->cout * tsz);
            st.vt = A.take(np * b->cout * tsz);
            st.aout = A.take(np * b->cout * tsz);
        }
        nb.ts.blocks.push_back(st);
    }
    for (const Block& b : h->enc) {
        nb.genc.push_back(A.take((size_t)B * b.res_out * b.res_out * b.cout * tsz));
        max_act = std::max(max_act, (size_t)b.res_out * b.res_out * b.cout);
    }
    for (const Block& b : h->dec) {
        if (b.kind != K_BLOCK) continue;
        Act a;
        a.p = A.take((size_t)B * b.res_out * b.res_out * b.cout * tsz);
        a.st = A.get<float2>(st_elems);
        nb.ts.dec_store.push_back(a);
    }
    for (const Block* b : h->blocks) {
        max_act = std::max(max_act, (size_t)b->res_out * b->res_out * b->cout);
        max_in = std::max(max_in, (size_t)b->res_in * b->res_in * b->cin);
        max_res = std::max(max_res, std::max(b->res_in, b->res_out));
        max_cin = std::max(max_cin, b->cin);
        if (b->attn) attn_hw = std::max(attn_hw, b->res_out * b->res_out);
        wg_max = std::max(wg_max, block_wgrad_bytes(B, b->res_out, b->cin, b->cout, b->attn));
    }
    nb.ga = A.take((size_t)B * std::max(max_act, max_in) * tsz);
    nb.gb = A.take((size_t)B * std::max(max_act, max_in) * tsz);
    const size_t npix = (size_t)B * h->cfg.img_resolution * h->cfg.img_resolution;
    nb.dfp = A.take(npix * 128 * tsz);
    nb.op32 = A.take(npix * 32 * tsz);
    nb.wtmp = A.get<float>((size_t)128 * 256 * 9);
    nb.wpad = A.get<float>((size_t)256 * 256 * 9);  // head weights padded to 128 (backward) / 256 (forward-mode) output rows
    nb.vec = A.get<float>(1024);
    nb.demb = A.get<float>((size_t)B * h->emb_ch);
    nb.pre = A.get<float>((size_t)B * h->emb_ch);
    nb.d1 = A.get<float>((size_t)B * h->emb_ch);
    nb.d0 = A.get<float>((size_t)B * std::max(h->cond_ch, h->emb_ch));
    nb.dtemb_all = A.get<float>((size_t)B * h->temb_total);
    nb.vec_all = A.get<float>((size_t)h->temb_total);
    nb.wgx_bytes = std::max(conv_wgrad_workspace_bytes(B, h->cfg.img_resolution, 256, 128, 3),
                            conv_wgrad_workspace_bytes(B, h->cfg.img_resolution, 32, 128, 3));
    nb.wgx = A.take(nb.wgx_bytes);
    plan_block_bwd(B, max_res, max_res, max_cin, 256, A, nb.q, attn_hw, wg_max);
    return A.off;
}

// dout == nullptr: only the encoder is differentiated (the forward returned the feature taps early); dfeats[tap] (nullable, NCHW
// fp32) joins the gradient of that tap's encoder output; dx_t (nullable) receives the gradient of the network input.
int run_backward(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* dout, float* out,
                 int B, Workspace& w, NetBwd& nb, hipStream_t s, bool have_forward, const float* const* dfeats = nullptr,
                 float* dx_t = nullptr) {
    const fg_edm_config& c = h->cfg;
    const int res = c.img_resolution, hw = res * res;
    const size_t npix = (size_t)B * hw;
    BwdScratch& q = nb.q;
    int rc = FG_OK;
    const bool early = dout == nullptr;
    if (!have_forward && (rc = run_forward(h, x_t, t, 1, r, 1, labels, out, B, w, s, nullptr, early, &nb.ts))) return rc;
    const Block *aux_norm = nullptr, *aux_conv = nullptr;
    for (const Block& b : h->dec) {
        if (b.kind == K_AUX_NORM) aux_norm = &b;
        if (b.kind == K_AUX_CONV) aux_conv = &b;
    }
    if (!aux_norm || !aux_conv || aux_conv->cin != 256 || aux_conv->cout > 8 || h->enc[0].cout != 128 || h->enc[0].cin > 8)
        return fail(FG_EINVAL, "backward: head / stem shape not covered (needs model_channels 128, 256-channel head)");
    const Act none;
    if (early) {
        // nothing downstream of the encoder: the gradients start at the feature taps
        for (size_t i = 0; i < h->enc.size(); ++i)
            HIP_TRY(hipMemsetAsync(nb.genc[i], 0, (size_t)B * h->enc[i].res_out * h->enc[i].res_out * h->enc[i].cout * 2, s));
        HIP_TRY(hipMemsetAsync(nb.dtemb_all, 0, sizeof(float) * (size_t)B * h->temb_total, s));  // decoder blocks contribute nothing
    } else {
    // ---- output head: F = aux_conv(silu(aux_norm(y))), out = c_skip x + c_out F ------------------------------------------------
    const Act& y = nb.ts.dec_store.back();
    w.ab0 = nb.ts.aux_ab;  // aux_norm's coefficients and {mean, rstd} were left there by the kept forward
    q.mr0 = nb.ts.aux_mr;
    HIP_TRY(launch_head_grad(dout, w.coef + 3 * (size_t)B, nb.dfp, B, aux_conv->cout, 128, hw, s));
    if (h->G(aux_conv->b)) {
        HIP_TRY(launch_colsum(nb.dfp, 128, 8, q.dtemb, B, hw, 1.0f, s));
        HIP_TRY(hipMemsetAsync(nb.vec, 0, sizeof(float) * 8, s));
        HIP_TRY(launch_batchsum_add(q.dtemb, nb.vec, B, 8, s));
        HIP_TRY(launch_add_sub_tensor(nb.vec, 8, h->G(aux_conv->b), 1, aux_conv->cout, 1, s));
    }
    if (h->G(aux_conv->w)) {
        HIP_TRY(launch_gn_act(0, y.p, 256, nullptr, 0, w.ab0, q.aop, B, res, 0, s));
        if ((rc = wgrad_checked(q.aop, nb.dfp, nb.wtmp, B, res, 256, 128, 3, 0, nb.wgx, nb.wgx_bytes, s))) return rc;
        HIP_TRY(launch_add_sub_tensor(nb.wtmp, 256, h->G(aux_conv->w), aux_conv->cout, 256, 9, s));
    }
    HIP_TRY(launch_pad_rows(h->P(aux_conv->w), nb.wpad, aux_conv->cout, 128, 256 * 9, s));
    if ((rc = conv_dgrad(h, nb.wpad, 128, 256, 3, nb.dfp, q.da, B, res, q, s, false))) return rc;
    void* g_cur = nb.ga;
    void* g_alt = nb.gb;
    HIP_TRY(launch_gn_bwd(0, y.p, 256, nullptr, 0, q.da, 256, w.ab0, q.mr0, h->P(aux_norm->w), q.P, q.S, h->G(aux_norm->w),
                          h->G(aux_norm->b), nullptr, 0, 0.f, g_cur, B, res, 0, s));
    // ---- decoder blocks in reverse ---------------------------------------------------------------------------------------
    struct Rec {
        const Block* b;
        const Act* x;
        const Act* x2;
        int sk;
    };
    std::vector<Rec> recs;
    {
        int sp = (int)h->enc.size();
        const Act* x = &w.skip.back();
        size_t di = 0;
        for (const Block& b : h->dec) {
            if (b.kind != K_BLOCK) continue;
            Rec rcd{&b, x, nullptr, -1};
            if (b.skip_c) {
                rcd.sk = --sp;
                rcd.x2 = &w.skip[rcd.sk];
            }
            recs.push_back(rcd);
            x = &nb.ts.dec_store[di++];
        }
    }
    for (int i = (int)recs.size() - 1; i >= 0; --i) {
        const Block& b = *recs[i].b;
        const int c2 = b.skip_c, c1 = b.cin - c2;
        const size_t npin = (size_t)B * b.res_in * b.res_in;
        // the x part of the input gradient becomes the next (earlier) block's incoming gradient, the skip part that encoder output's
        if ((rc = block_backward(h, b, *recs[i].x, c1, c2 ? *recs[i].x2 : none, c2, w.emb, w.temb, g_cur, g_alt, nb.demb, B, w, q, s, nb.dtemb_all,
                                 &nb.ts.blocks[block_index(h, &b)], c2 ? nb.genc[recs[i].sk] : nullptr, 0)))
            return rc;
        (void)npin;
        std::swap(g_cur, g_alt);
    }
    // the decoder's first block read the last encoder output directly
    {
        const Block& last = h->enc.back();
        HIP_TRY(launch_add_bf16(nb.genc.back(), g_cur, (int64_t)B * last.res_out * last.res_out * last.cout, s));
    }
    }  // !early
    // gradients arriving at the feature taps (the encoder's `block3` outputs, EDM/network.py:535-539)
    if (dfeats)
        for (size_t i = 0; i < h->enc.size(); ++i) {
            const Block& b = h->enc[i];
            if (b.tap >= 0 && dfeats[b.tap]) HIP_TRY(launch_add_nchw_to_nhwc(dfeats[b.tap], nb.genc[i], B, b.cout, b.res_out * b.res_out, s));
        }
    // ---- encoder blocks in reverse -------------------------------------------------------------------------------------------
    for (int i = (int)h->enc.size() - 1; i >= 1; --i) {
        const Block& b = h->enc[i];
        // accumulated straight into the previous encoder output's gradient (which already holds the decoder's share)
        if ((rc = block_backward(h, b, w.skip[i - 1], b.cin, none, 0, w.emb, w.temb, nb.genc[i], nb.genc[i - 1], nb.demb, B, w, q, s, nb.dtemb_all,
                                 &nb.ts.blocks[block_index(h, &b)], nullptr, 1)))
            return rc;
    }
    // ---- stem: conv(c_in * x_t) ---------------------------------------------------------------------------------------------------
    {
        const Block& b = h->enc[0];
        if (h->G(b.b)) {
            HIP_TRY(launch_colsum(nb.genc[0], b.cout, b.cout, q.dtemb, B, hw, 1.0f, s));
            HIP_TRY(launch_batchsum_add(q.dtemb, h->G(b.b), B, b.cout, s));
        }
        if (h->G(b.w)) {
            HIP_TRY(launch_stem_operand(x_t, w.coef, nb.op32, B, b.cin, 32, hw, s));
            if ((rc = wgrad_checked(nb.op32, nb.genc[0], nb.wtmp, B, res, 32, b.cout, 3, 0, nb.wgx, nb.wgx_bytes, s))) return rc;
            HIP_TRY(launch_add_sub_tensor(nb.wtmp, 32, h->G(b.w), b.cout, b.cin, 9, s));
        }
        if (dx_t) {
            // d x_t = c_in * conv^T(g, W_stem)  (+ c_skip * dout through precond_output); drop_precond 'input' has c_in = 1
            if ((rc = conv_dgrad(h, h->P(b.w), b.cout, b.cin, 3, nb.genc[0], q.da, B, res, q, s))) return rc;
            HIP_TRY(launch_input_grad(q.da, 256, w.coef, w.coef + 2 * (size_t)B, dout, dx_t, B, b.cin, hw, s));
        }
    }
    // ---- all 33 embedding affines at once (stacked as in the forward): biases, weights, and demb = dtemb_all @ aff_w ---------------
    {
        const int E = h->emb_ch, TT = h->temb_total;
        HIP_TRY(hipMemsetAsync(nb.vec_all, 0, sizeof(float) * TT, s));
        HIP_TRY(launch_batchsum_add(nb.dtemb_all, nb.vec_all, B, TT, s));
        for (const Block* b : h->blocks) {
            if (h->G(b->conv0_b)) HIP_TRY(launch_add_sub_tensor(nb.vec_all + b->temb_off, b->cout, h->G(b->conv0_b), 1, b->cout, 1, s));
            if (h->G(b->aff_b)) HIP_TRY(launch_add_sub_tensor(nb.vec_all + b->temb_off, b->cout, h->G(b->aff_b), 1, b->cout, 1, s));
            if (h->G(b->aff_w))
                HIP_TRY(launch_linear_bwd(nb.dtemb_all + b->temb_off, w.emb, nullptr, h->G(b->aff_w), nullptr, nullptr, B, b->cout, E, 1.0f, s, TT));
        }
        if (!h->aff_wT) {
            int rc2 = dev_alloc(h, (void**)&h->aff_wT, sizeof(float) * (size_t)E * TT);
            if (rc2) return rc2;
        }
        if (h->aff_wT_epoch != h->pack_epoch) {
            HIP_TRY(launch_transpose_f32(h->aff_w, h->aff_wT, TT, E, s));
            h->aff_wT_epoch = h->pack_epoch;
        }
        HIP_TRY(launch_linear(nb.dtemb_all, h->aff_wT, nullptr, nb.demb, B, TT, E, 0, s));
    }
    // ---- embedding MLP: emb = silu(L1(silu(L0(emb0)))), emb0 = posemb + map_label(labels sqrt(L))  (:501-521) ------------------
    {
        const int E = h->emb_ch, N = h->cond_ch;
        const int w1 = h->find("model.map_layer1.weight"), b1 = h->find("model.map_layer1.bias");
        const int w0 = h->find("model.map_layer0.weight"), b0 = h->find("model.map_layer0.bias");
        HIP_TRY(launch_linear(w.emb1, h->P(w1), h->P(b1), nb.pre, B, E, E, 0, s));
        HIP_TRY(launch_silu_bwd(nb.demb, nb.pre, nb.d1, B * E, s));
        HIP_TRY(hipMemsetAsync(nb.d0, 0, sizeof(float) * (size_t)B * E, s));
        HIP_TRY(launch_linear_bwd(nb.d1, w.emb1, h->P(w1), h->G(w1), h->G(b1), nb.d0, B, E, E, 1.0f, s));
        HIP_TRY(launch_linear(w.emb0, h->P(w0), h->P(b0), nb.pre, B, N, E, 0, s));
        HIP_TRY(launch_silu_bwd(nb.d0, nb.pre, nb.d1, B * E, s));
        HIP_TRY(hipMemsetAsync(nb.d0, 0, sizeof(float) * (size_t)B * N, s));
        HIP_TRY(launch_linear_bwd(nb.d1, w.emb0, h->P(w0), h->G(w0), h->G(b0), nb.d0, B, E, N, 1.0f, s));
        if (c.label_dim > 0) {
            const int wl = h->find("model.map_label.weight"), bl = h->find("model.map_label.bias");
            HIP_TRY(launch_linear_bwd(nb.d0, labels, nullptr, labels ? h->G(wl) : nullptr, h->G(bl), nullptr, B, N, c.label_dim,
                                      std::sqrt((float)c.label_dim), s));
        }
    }
    return FG_OK;
}

// ---- forward mode: tangents of (x_t, t, r) pushed through the network next to the kept forward -----------------
// conv of a tangent: the forward kernel without prologue or bias on the forward's packed weights
int tangent_conv(fg_edm* h, int ks, int res_mode, const void* src, int cin, int res_in, int res_out, const void* wpack, int cout,
                 const float* temb, const void* resid, float scale, void* out, int B, hipStream_t s) {
    ConvArgs a{};
    a.src1 = src; a.C1 = cin; a.C2 = 0;
    a.Hs = a.Ws = res_in; a.H = a.W = res_out; a.B = B;
    a.wpack = wpack; a.bias = nullptr;
    a.temb = temb; a.temb_stride = h->temb_total;
    a.resid = resid; a.scale = scale; a.out = out; a.Cout = cout;
    HIP_TRY(launch_conv_fused(1, ks, PRO_NONE, res_mode, OUT_NHWC, a, s));
    return FG_OK;
}

// tangent of one UNetBlock: xd1 / xd2 are the tangents of its inputs, yd (bf16 [B, res_out^2, cout]) of its output
int block_jvp(fg_edm* h, const Block& b, const Act& a1, int c1, const void* xd1, const Act& a2, int c2, const void* xd2,
              const float* dtemb, void* yd, int B, Workspace& w, BwdScratch& q, const BlockStash& st, const float2* ident_ab,
              hipStream_t s) {
    const int res = b.res_out, res_in = b.res_in, hw = res * res, cin = b.cin, cout = b.cout;
    const int rm = b.down ? 1 : (b.up ? 2 : 0);
    const int res_mode = b.down ? RES_DOWN : (b.up ? RES_UP : RES_NONE);
    int rc;
    use_stash(w, st);
    // tangent of the (virtual) concat as one dense tensor
    const void* xd = xd1;
    if (c2) {
        HIP_TRY(launch_gn_act(2, xd1, c1, xd2, c2, nullptr, q.aop, B, res_in, 0, s));
        xd = q.aop;
    }
    // conv0(silu(norm0(x))) + affine(emb)
    HIP_TRY(launch_gn_jvp(0, a1.p, c1, c2 ? a2.p : nullptr, c2, xd, st.ab0, st.mr0, q.P, q.S, q.da, B, res_in, s));
    const void* ad0 = q.da;
    if (rm) {
        HIP_TRY(launch_gn_act(2, q.da, cin, nullptr, 0, nullptr, q.dskip, B, res, rm, s));
        ad0 = q.dskip;
    }
    if ((rc = tangent_conv(h, 3, RES_NONE, ad0, cin, res, res, b.p_conv0, cout, dtemb + b.temb_off, nullptr, 1.0f, q.dh0, B, s))) return rc;
    // silu(norm1(h0))
    HIP_TRY(launch_gn_jvp(0, st.h, cout, nullptr, 0, q.dh0, st.ab1, st.mr1, q.P, q.S, q.g1, B, res, s));
    // skip path
    const void* sd = xd;  // identity skip: same resolution and width
    if (b.has_skip) {
        if ((rc = tangent_conv(h, 1, res_mode, xd, cin, res_in, res, b.p_skip, cout, nullptr, nullptr, 1.0f, q.dxin, B, s))) return rc;
        sd = q.dxin;
    }
    void* mid = b.attn ? q.gmid : yd;
    if ((rc = tangent_conv(h, 3, RES_NONE, q.g1, cout, res, res, b.p_conv1, cout, nullptr, sd, kSkipScale, mid, B, s))) return rc;
    if (b.attn) {
        // (proj(attention(qkv(norm2(x_mid)))) + x_mid) * sigma
        HIP_TRY(launch_gn_jvp(1, st.xattn, cout, nullptr, 0, q.gmid, st.ab2, st.mr2, q.P, q.S, q.da, B, res, s));
        ConvArgs qa{};
        qa.src1 = q.da; qa.C1 = cout; qa.Hs = qa.Ws = qa.H = qa.W = res; qa.B = B;
        qa.ab = ident_ab; qa.wpack = b.p_qkv; qa.bias = nullptr; qa.scale = 1.0f; qa.Cout = 3 * cout;
        qa.q_out = q.dq; qa.k_out = q.dk; qa.vt_out = q.dvt;
        HIP_TRY(launch_conv_fused(1, 1, PRO_GN, RES_NONE, OUT_QKV, qa, s));
        HIP_TRY(launch_attention_jvp(st.q, st.k, st.vt, q.dq, q.dk, q.dvt, q.da, q.att, B, hw, cout, s));
        if ((rc = tangent_conv(h, 1, RES_NONE, q.da, cout, res, res, b.p_proj, cout, nullptr, q.gmid, kSkipScale, yd, B, s))) return rc;
    }
    return FG_OK;
}

int run_jvp(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* vx, const float* vt,
            const float* vr, float* out, float* jvp, int B, Workspace& w, NetBwd& nb, hipStream_t s) {
    const fg_edm_config& c = h->cfg;
    const int res = c.img_resolution, hw = res * res, C = c.img_channels;
    BwdScratch& q = nb.q;
    int rc = run_forward(h, x_t, t, 1, r, 1, labels, out, B, w, s, nullptr, false, &nb.ts);
    if (rc) return rc;
    const Block *aux_norm = nullptr, *aux_conv = nullptr;
    for (const Block& b : h->dec) {
        if (b.kind == K_AUX_NORM) aux_norm = &b;
        if (b.kind == K_AUX_CONV) aux_conv = &b;
    }
    if (!aux_norm || !aux_conv || aux_conv->cin != 256 || !h->enc[0].p_stem || !q.att)
        return fail(FG_EINVAL, "jvp: head / stem shape not covered");
    const int E = h->emb_ch, N = h->cond_ch, TT = h->temb_total;
    // scratch reuse (nothing of the backward runs here): coefficient tangents, embedding tangents, identity GroupNorm coefficients
    float* ct = nb.vec_all;                 // [8][B]  (temb_total floats >= 8 B for B <= 1056)
    if (8 * B > TT) return fail(FG_EINVAL, "jvp: batch too large for the coefficient scratch");
    float* e0 = nb.d0;                      // [B][N] tangent of emb0, later reused
    float* dtemb = nb.dtemb_all;            // [B][TT] tangent of the stacked affine outputs
    float2* ident = (float2*)nb.pre;        // [B][256] {1, 0}: `pre` has B * E floats = B * 256 float2
    float* ones = nb.vec;                   // [<= 1024]
    float* zeros = nb.vec + 512;
    if (B > 512) return fail(FG_EINVAL, "jvp: batch too large for the constant scratch");
    HIP_TRY(launch_fill_f32(ones, 1.0f, 512, s));
    HIP_TRY(launch_fill_f32(zeros, 0.0f, 512, s));
    HIP_TRY(launch_jvp_coef(t, c.r_timestep ? r : nullptr, vt, vr, c.sigma_data, c.sigma_shift, c.drop_precond, ct, B, s));
    // ---- embedding: emb = silu(L1(silu(L0(emb0)))), temb = A emb + b ------------------------------------------------------------
    {
        const int w1 = h->find("model.map_layer1.weight"), b1 = h->find("model.map_layer1.bias");
        const int w0 = h->find("model.map_layer0.weight"), b0 = h->find("model.map_layer0.bias");
        HIP_TRY(launch_jvp_embed(w.coef + B, w.coef + 4 * (size_t)B, ct + 2 * (size_t)B, ct + 3 * (size_t)B, h->freqs, e0, B, N, h->noise_ch, s));
        HIP_TRY(launch_linear(w.emb0, h->P(w0), h->P(b0), nb.d1, B, N, E, 0, s));        // pre-activation of layer 0
        HIP_TRY(launch_linear(e0, h->P(w0), nullptr, nb.demb, B, N, E, 0, s));             // W0 emb0_dot
        HIP_TRY(launch_silu_bwd(nb.demb, nb.d1, e0, B * E, s));                            // emb1_dot -> e0 ([B][E] fits: N <= E)
        HIP_TRY(launch_linear(w.emb1, h->P(w1), h->P(b1), nb.d1, B, E, E, 0, s));        // pre-activation of layer 1
        HIP_TRY(launch_linear(e0, h->P(w1), nullptr, nb.demb, B, E, E, 0, s));
        HIP_TRY(launch_silu_bwd(nb.demb, nb.d1, e0, B * E, s));                            // emb_dot
        HIP_TRY(launch_linear(e0, h->aff_w, nullptr, dtemb, B, E, TT, 0, s));
    }
    HIP_TRY(launch_fill_f2(ident, 1.0f, 0.0f, B * 256, s));
    // ---- stem: conv(c_in x)  ->  conv(c_in vx + dc_in x) --------------------------------------------------------------------
    float* xin = (float*)q.wg;  // fp32 [B, C, H, W]: the split-K scratch is idle here and far larger
    HIP_TRY(launch_jvp_input(vx, x_t, ct, ct + B, xin, B, C * hw, s));
    HIP_TRY(launch_stem(1, xin, ones, h->enc[0].p_stem, zeros, nb.genc[0], nullptr, B, res, h->enc[0].cin, s));
    const Act none;
    // ---- encoder -------------------------------------------------------------------------------------------------------------
    for (size_t i = 1; i < h->enc.size(); ++i) {
        const Block& b = h->enc[i];
        if ((rc = block_jvp(h, b, w.skip[i - 1], b.cin, nb.genc[i - 1], none, 0, nullptr, dtemb, nb.genc[i], B, w, q,
                            nb.ts.blocks[block_index(h, &b)], ident, s)))
            return rc;
    }
    // ---- decoder -------------------------------------------------------------------------------------------------------------
    const Act* x = &w.skip.back();
    const void* xd = nb.genc.back();
    void* pong[2] = {nb.ga, nb.gb};
    int cur = 0, sp = (int)h->enc.size();
    size_t di = 0;
    for (const Block& b : h->dec) {
        if (b.kind != K_BLOCK) continue;
        const int c2 = b.skip_c, c1 = b.cin - c2;
        const Act* x2 = &none;
        const void* xd2 = nullptr;
        if (c2) {
            --sp;
            x2 = &w.skip[sp];
            xd2 = nb.genc[sp];
        }
        if ((rc = block_jvp(h, b, *x, c1, xd, *x2, c2, xd2, dtemb, pong[cur], B, w, q, nb.ts.blocks[block_index(h, &b)], ident, s)))
            return rc;
        x = &nb.ts.dec_store[di++];
        xd = pong[cur];
        cur ^= 1;
    }
    // ---- head: F = aux_conv(silu(aux_norm(y))), out = c_skip x + c_out F ---------------------------------------------------------
    HIP_TRY(launch_gn_jvp(0, x->p, 256, nullptr, 0, xd, nb.ts.aux_ab, nb.ts.aux_mr, q.P, q.S, q.da, B, res, s));
    HIP_TRY(launch_pad_rows(h->P(aux_conv->w), nb.wpad, aux_conv->cout, 256, 256 * 9, s));   // 256 output rows, C real ones
    HIP_TRY(launch_pack_conv_weights(1, nb.wpad, q.wpk, 256, 256, 3, 0, s));
    if ((rc = tangent_conv(h, 3, RES_NONE, q.da, 256, res, res, q.wpk, 256, nullptr, nullptr, 1.0f, q.dh0, B, s))) return rc;
    HIP_TRY(launch_jvp_output(q.dh0, 256, out, x_t, vx, ct, jvp, B, C, hw, s));
    return FG_OK;
}
}  // namespace

int fg_edm_jvp(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* vx, const float* vt,
               const float* vr, float* out, float* jvp, int batch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !x_t || !t || !vx || !out || !jvp) return fail(FG_EINVAL, "null argument");
    if (!h->packed) return fail(FG_ENOTREADY, "weights are not packed (call fg_edm_pack_weights)");
    if (!h->dtype) return fail(FG_EINVAL, "the forward-mode pass runs in the bf16 compute mode only");
    if (h->cfg.r_timestep && !r) return fail(FG_EINVAL, "r is required by an r_timestep network");
    if (batch <= 0 || !workspace || (((uintptr_t)workspace) & 255)) return fail(FG_EINVAL, "bad batch / workspace");
    Arena A;
    A.base = (char*)workspace;
    Workspace w;
    plan_workspace(h, batch, A, w);
    NetBwd nb;
    const size_t need = plan_net_bwd(h, batch, A, nb);
    if (need > workspace_bytes) return fail(FG_ENOMEM, "workspace too small: need %zu bytes for batch %d, got %zu", need, batch, workspace_bytes);
    return run_jvp(h, x_t, t, r, labels, vx, vt, vr, out, jvp, batch, w, nb, (hipStream_t)stream);
}

size_t fg_edm_backward_workspace_bytes(const fg_edm* h, int batch) {
    if (!h || batch <= 0) return 0;
    Arena A;
    A.dry = true;
    Workspace w;
    plan_workspace(h, batch, A, w);
    NetBwd nb;
    return plan_net_bwd(h, batch, A, nb);
}

int fg_edm_forward_train(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, float* out,
                         float* const* features, int batch, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !x_t || !t || (!out && !features)) return fail(FG_EINVAL, "null argument");
    if (!h->packed) return fail(FG_ENOTREADY, "weights are not packed (call fg_edm_pack_weights)");
    if (!h->dtype) return fail(FG_EINVAL, "the backward pass runs in the bf16 compute mode only");
    if (features)
        for (const Block& b : h->enc)
            if (b.tap >= 0 && features[b.tap] && ((b.cout % 32) || ((b.res_out * b.res_out) % 32)))
                return fail(FG_EINVAL, "feature tap %d: channels and pixels must be multiples of 32", b.tap);
    if (h->cfg.r_timestep && !r) return fail(FG_EINVAL, "r is required by an r_timestep network");
    if (batch <= 0 || !workspace || (((uintptr_t)workspace) & 255)) return fail(FG_EINVAL, "bad batch / workspace");
    Arena A;
    A.base = (char*)workspace;
    Workspace w;
    plan_workspace(h, batch, A, w);
    NetBwd nb;
    const size_t need = plan_net_bwd(h, batch, A, nb);
    if (need > workspace_bytes) return fail(FG_ENOMEM, "workspace too small: need %zu bytes for batch %d, got %zu", need, batch, workspace_bytes);
    return run_forward(h, x_t, t, 1, r, 1, labels, out, batch, w, (hipStream_t)stream, features, out == nullptr, &nb.ts);
}

int fg_edm_backward(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* dout,
                    float* out, int have_forward, int batch, void* workspace, size_t workspace_bytes, void* stream) {
    return fg_edm_backward_ex(h, x_t, t, r, labels, dout, nullptr, out, nullptr, have_forward, batch, workspace, workspace_bytes, stream);
}

int fg_edm_backward_ex(fg_edm* h, const float* x_t, const double* t, const double* r, const float* labels, const float* dout,
                       const float* const* dfeatures, float* out, float* dx_t, int have_forward, int batch, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!h || !x_t || !t || (!dout && !dfeatures)) return fail(FG_EINVAL, "null argument");
    if (dout && !out && !have_forward) return fail(FG_EINVAL, "out is required when the forward runs here");
    if (!h->packed) return fail(FG_ENOTREADY, "weights are not packed (call fg_edm_pack_weights)");
    if (!h->dtype) return fail(FG_EINVAL, "the backward pass runs in the bf16 compute mode only");
    if (h->cfg.r_timestep && !r) return fail(FG_EINVAL, "r is required by an r_timestep network");
    if (batch <= 0 || !workspace || (((uintptr_t)workspace) & 255)) return fail(FG_EINVAL, "bad batch / workspace");
    Arena A;
    A.base = (char*)workspace;
    Workspace w;
    plan_workspace(h, batch, A, w);
    NetBwd nb;
    const size_t need = plan_net_bwd(h, batch, A, nb);
    if (need > workspace_bytes) return fail(FG_ENOMEM, "workspace too small: need %zu bytes for batch %d, got %zu", need, batch, workspace_bytes);
    return run_backward(h, x_t, t, r, labels, dout, out, batch, w, nb, (hipStream_t)stream, have_forward != 0, dfeatures, dx_t);
}

int fg_edm_bind_grad(fg_edm* h, const char* name, float* grad, int64_t numel) {
    if (!h || !name) return fail(FG_EINVAL, "null argument");
    const int i = h->find(name);
    if (i < 0) return fail(FG_EINVAL, "unknown parameter '%s'", name);
    if (grad && numel != h->params[i].numel) return fail(FG_EINVAL, "%s: expected %lld gradient elements, got %lld", name, (long long)h->params[i].numel, (long long)numel);
    h->params[i].grad = grad;
    return FG_OK;
}

size_t fg_edm_block_backward_workspace_bytes(const fg_edm* h, int index, int batch) {
    if (!h || batch <= 0 || index < 0 || index >= (int)h->blocks.size()) return 0;
    const Block& b = *h->blocks[index];
    Arena A;
    A.dry = true;
    Workspace w;
    plan_workspace(h, batch, A, w);
    BwdScratch q;
    plan_block_bwd(batch, b.res_in, b.res_out, b.cin, b.cout, A, q, b.attn ? b.res_out * b.res_out : 0);
    A.take((size_t)batch * b.res_out * b.res_out * b.cout * 2);  // gout in bf16
    return A.off;
}

int fg_edm_run_block_backward(fg_edm* h, int index, const float* x1, int c1, const float* x2, int c2, const float* emb,
                              const float* dout, float* dx1, float* dx2, float* demb, int batch, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (!h || !x1 || !emb || !dout) return fail(FG_EINVAL, "null argument");
    if (!h->packed) return fail(FG_ENOTREADY, "weights are not packed (call fg_edm_pack_weights)");
    if (!h->dtype) return fail(FG_EINVAL, "the backward pass runs in the bf16 compute mode only");
    if (index < 0 || index >= (int)h->blocks.size()) return fail(FG_EINVAL, "block index out of range");
    const Block& b = *h->blocks[index];
    if (c1 + c2 != b.cin || (c1 % 8) || (c2 % 8)) return fail(FG_EINVAL, "%s: bad input channel split %d+%d", b.key.c_str(), c1, c2);
    if (batch <= 0 || !workspace || (((uintptr_t)workspace) & 255)) return fail(FG_EINVAL, "bad batch / workspace");
    const int B = batch;
    Arena A;
    A.base = (char*)workspace;
    Workspace w;
    plan_workspace(h, B, A, w);
    BwdScratch q;
    plan_block_bwd(B, b.res_in, b.res_out, b.cin, b.cout, A, q, b.attn ? b.res_out * b.res_out : 0);
    const size_t npix = (size_t)B * b.res_out * b.res_out, npix_in = (size_t)B * b.res_in * b.res_in;
    void* gout = A.take(npix * b.cout * 2);
    if (A.off > workspace_bytes) return fail(FG_ENOMEM, "workspace too small: need %zu bytes, got %zu", A.off, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_linear(emb, h->aff_w, h->aff_b, w.temb, B, h->emb_ch, h->temb_total, 0, s));
    Act a1, a2;
    HIP_TRY(launch_to_act(1, x1, w.cvt1, (int64_t)npix_in * c1, s));
    a1.p = w.cvt1;
    if (c2) {
        HIP_TRY(launch_to_act(1, x2, w.cvt2, (int64_t)npix_in * c2, s));
        a2.p = w.cvt2;
    }
    HIP_TRY(launch_scale_to_bf16(dout, gout, 1.0f, (int64_t)npix * b.cout, s));
    int rc = block_backward(h, b, a1, c1, a2, c2, emb, w.temb, gout, q.dxin, demb, B, w, q, s);
    if (rc) return rc;
    if (dx1) HIP_TRY(launch_slice_to_f32(q.dxin, b.cin, 0, dx1, c1, (int64_t)npix_in, s));
    if (dx2 && c2) HIP_TRY(launch_slice_to_f32(q.dxin, b.cin, c1, dx2, c2, (int64_t)npix_in, s));
    return FG_OK;
}

int fg_op_gn_coeffs(const float* x1, int c1, const float* x2, int c2, const float* gamma, const float* beta, float eps,
                    float* ab_out, int batch, int hw, void* stream) {
    if (!x1 || !gamma || !beta || !ab_out) return fail(FG_EINVAL, "null argument");
    HIP_TRY(launch_gn_coeffs(0, x1, c1, c2 ? x2 : nullptr, c2, gamma, beta, eps, (float2*)ab_out, batch, hw, (hipStream_t)stream));
    return FG_OK;
}
int fg_op_latents(const float* noise, double t_init, float* out, int64_t total, void* stream) {
    HIP_TRY(launch_latents(noise, t_init, nullptr, 0, out, total, (hipStream_t)stream));
    return FG_OK;
}
int fg_op_forward_process(const float* x0, const float* eps, double t, int schedule, float* out, int64_t total,
                          void* stream) {
    HIP_TRY(launch_forward_process(x0, eps, t, nullptr, 0, schedule, out, total, (hipStream_t)stream));
    return FG_OK;
}
int fg_op_x0_to_eps(const float* xt, const float* x0, double t, int schedule, float* out, int64_t total, void* stream) {
    HIP_TRY(launch_x0_to_eps(xt, x0, t, nullptr, 0, schedule, 1e-6, out, total, (hipStream_t)stream));
    return FG_OK;
}
int fg_disc_edm_num_params(int res) { return disc_num_params(res); }
size_t fg_disc_edm_workspace_bytes(int res, int batch) { return disc_workspace_bytes(res, batch); }
int fg_disc_edm_run(const float* feat, int res, const float* const* params, float* logits, const float* dlogits, float* dfeat,
                    float* const* grads, int batch, void* workspace, size_t workspace_bytes, void* stream) {
    const int np = disc_num_params(res);
    if (!np) return fail(FG_EINVAL, "fg_disc_edm_run: resolution %d unsupported (8, 16, 32)", res);
    if (!feat || !params || !logits || batch <= 0 || !workspace || (((uintptr_t)workspace) & 255))
        return fail(FG_EINVAL, "fg_disc_edm_run: bad argument");
    for (int i = 0; i < np; ++i)
        if (!params[i]) return fail(FG_EINVAL, "fg_disc_edm_run: parameter %d is null", i);
    if (workspace_bytes < disc_workspace_bytes(res, batch))
        return fail(FG_ENOMEM, "fg_disc_edm_run: workspace too small (%zu < %zu bytes)", workspace_bytes, disc_workspace_bytes(res, batch));
    HIP_TRY(disc_run(feat, res, params, logits, dlogits, dfeat, grads, batch, workspace, (hipStream_t)stream));
    return FG_OK;
}
size_t fg_op_conv_wgrad_workspace_bytes(int batch, int res, int cin, int cout, int ks) {
    return conv_wgrad_supported(res, cin, cout, ks) && batch > 0 ? conv_wgrad_workspace_bytes(batch, res, cin, cout, ks) : 0;
}
int fg_op_conv_wgrad(const void* act, const void* dy, float* dw, int batch, int res, int cin, int cout, int ks,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!conv_wgrad_supported(res, cin, cout, ks))
        return fail(FG_EINVAL, "fg_op_conv_wgrad: unsupported shape res=%d cin=%d cout=%d ks=%d (res 8/16/32, cin %% 32, cout %% 128, ks 1/3)",
                    res, cin, cout, ks);
    if (batch <= 0 || !act || !dy || !dw || !workspace) return fail(FG_EINVAL, "fg_op_conv_wgrad: bad argument");
    if (workspace_bytes < conv_wgrad_workspace_bytes(batch, res, cin, cout, ks))
        return fail(FG_EINVAL, "fg_op_conv_wgrad: workspace too small (%zu < %zu bytes)", workspace_bytes,
                    conv_wgrad_workspace_bytes(batch, res, cin, cout, ks));
    HIP_TRY(launch_conv_wgrad(act, dy, dw, batch, res, cin, cout, ks, accumulate, workspace, (hipStream_t)stream));
    return FG_OK;
}
int fg_op_images_to_u8(const float* images, uint8_t* out, int64_t batch, int channels, int height, int width, void* stream) {
    if (batch < 0 || channels <= 0 || height <= 0 || width <= 0) return fail(FG_EINVAL, "fg_op_images_to_u8: bad shape");
    if (batch && (!images || !out)) return fail(FG_EINVAL, "fg_op_images_to_u8: null pointer");
    HIP_TRY(launch_images_to_u8(images, out, batch, channels, height * width, (hipStream_t)stream));
    return FG_OK;
}
int fg_op_randn(float* out, int64_t total, uint64_t seed, uint64_t offset, void* stream) {
    HIP_TRY(launch_randn(out, total, seed, offset, nullptr, (hipStream_t)stream));
    return FG_OK;
}

}  // extern "C"
