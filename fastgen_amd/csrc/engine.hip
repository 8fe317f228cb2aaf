// fastgen_amd engine: network plan, weight packing, workspace arena, forward orchestration, sampler loop with hipGraph
// replay, and the C ABI declared in include/fastgen_amd.h.  Host-side C++ only; every kernel lives in conv/attn/misc.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <vector>

#include "../../include/fastgen_amd.h"
#include "common.h"
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
    void* a1d = nullptr;   // kept (training) forward: where the block's conv1 operand (after dropout, if on) goes
    DropArgs drop;         // ... and its dropout parameters (p == 0: off)
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
    void* a1d = nullptr;   // conv1's operand silu(norm1(h)) [* keep], materialised by the kept forward
    uint32_t index = 0;    // position in fg_edm::blocks: the block's Philox stream
};
struct TrainStash {
    std::vector<Act> dec_store;      // decoder block outputs (the encoder's live on the skip stack anyway)
    std::vector<BlockStash> blocks;  // indexed like fg_edm::blocks
    float2* aux_ab = nullptr;        // coefficients of aux_norm
    float2* aux_mr = nullptr;        // its {mean, rstd}
    float* raw_out = nullptr;        // the network's own output F (before precond_output), [B, C, H, W] fp32
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
    int dtype = 0;  // storage type of the activation tensors and mode of every non-conv kernel: 0 fp32, 1 bf16
    int cmode = 0;  // arithmetic of the convolutions: FG_DTYPE_F32 / FG_DTYPE_BF16 / FG_DTYPE_BF16X3 (fp32 storage)
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
        void* packed_ws = nullptr;  // conv_ws.hip layout (3x3, 256 -> 256 at 32x32 / 16x16)
        uint64_t epoch = 0;
    };
    std::map<const float*, DgradW> dgrad_cache;
    uint64_t pack_epoch = 0;
    const float* augment = nullptr;  // [B][augment_dim] augmentation labels of the next calls (fg_edm_set_augment), or none
    bool training = false;           // the module is in train() mode: sigma_shift is not applied (fg_edm_set_training)
    double shift() const { return training ? 0.0 : cfg.sigma_shift; }  // EDM/network.py:956
    float dropout_p = 0.f;           // training-mode dropout of conv1's operand (fg_edm_set_dropout); 0 = off
    uint64_t dropout_seed = 0;
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
    const size_t st_elems = (size_t)B * 32 * 64;  // [B][<= 32 slots][256/4 quads] (8 tiles x 4 producer waves in conv_ws3.hip)
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
    const int rc = launch_conv_fused(h->cmode, ks, pro, res, outmode, a, s);
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
    w.a1d = st.a1d;
    w.drop.block = st.index;
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
    const int slots = conv_stat_slots(b.res_out);  // of the 1x1 convs; the 3x3 convs report theirs per launch
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
        w.h.slots = conv_launch_stat_slots(h->cmode, 3, PRO_NONE, RES_NONE, OUT_NHWC, a);
        HIP_TRY(conv_launch(h, 3, PRO_NONE, RES_NONE, OUT_NHWC, a, s));
    } else {
        w.h.slots = conv_launch_stat_slots(h->cmode, 3, PRO_GN_SILU, res_mode, OUT_NHWC, a);
        HIP_TRY(conv_launch(h, 3, PRO_GN_SILU, res_mode, OUT_NHWC, a, s));
    }
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
    HIP_TRY(launch_gn_finalize(w.h.st, b.cout, w.h.slots, nullptr, 0, 0, h->P(b.norm1_w), h->P(b.norm1_b), kBlockEps, w.ab1, B, hw, s, w.mr1));
    Act& x_mid = b.attn ? w.xattn : out;
    ConvArgs d{};
    d.src1 = w.h.p; d.C1 = b.cout; d.Hs = d.Ws = d.H = d.W = b.res_out; d.B = B;
    d.ab = w.ab1; d.wpack = b.p_conv1; d.wpack_ws = b.p_conv1_ws; d.bias = h->P(b.conv1_b);
    d.resid = resid; d.scale = kSkipScale; d.out = x_mid.p; d.Cout = b.cout; d.stats = x_mid.st;
    if (w.a1d) {
        // kept (training) forward: the operand silu(norm1(h)) [* keep with dropout, EDM/network.py:283-284] is materialised once —
        // the backward's weight gradient contracts with the same tensor — and the conv runs without a prologue
        HIP_TRY(launch_gn_act(h->dtype, 0, w.h.p, b.cout, nullptr, 0, w.ab1, w.a1d, B, b.res_out, 0, s, w.drop));
        d.src1 = w.a1d; d.ab = nullptr;
        x_mid.slots = conv_launch_stat_slots(h->cmode, 3, PRO_NONE, RES_NONE, OUT_NHWC, d);
        HIP_TRY(conv_launch(h, 3, PRO_NONE, RES_NONE, OUT_NHWC, d, s));
    } else {
        x_mid.slots = conv_launch_stat_slots(h->cmode, 3, PRO_GN_SILU, RES_NONE, OUT_NHWC, d);
        HIP_TRY(conv_launch(h, 3, PRO_GN_SILU, RES_NONE, OUT_NHWC, d, s));
    }
    if (b.attn) {
        HIP_TRY(launch_gn_finalize(x_mid.st, b.cout, x_mid.slots, nullptr, 0, 0, h->P(b.norm2_w), h->P(b.norm2_b), kBlockEps, w.ab2, B, hw, s, w.mr2));
        ConvArgs q{};
        q.src1 = x_mid.p; q.C1 = b.cout; q.Hs = q.Ws = q.H = q.W = b.res_out; q.B = B;
        q.ab = w.ab2; q.wpack = b.p_qkv; q.bias = b.qkv_bias; q.scale = 1.0f; q.Cout = 3 * b.cout;
        q.q_out = w.q; q.k_out = w.k; q.vt_out = w.vt;
        HIP_TRY(conv_launch(h, 1, PRO_GN, RES_NONE, OUT_QKV, q, s));
        HIP_TRY(launch_attention(h->cmode, w.q, w.k, w.vt, w.aout, B, hw, s));
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
                              h->cond_ch, h->noise_ch, s, h->augment, h->augment ? h->P(h->find("model.map_augment.weight")) : nullptr, c.augment_dim));
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
    HIP_TRY(launch_precond_coef(t, t_stride, c.r_timestep ? r : nullptr, r_stride, c.sigma_data, h->shift(), 1e-6,
                                c.drop_precond, w.coef, B, s));
    int rc = run_mapping(h, labels, B, w, s);
    if (rc) return rc;
    w.drop = DropArgs{};
    if (ts && h->dropout_p > 0.f) w.drop.p = h->dropout_p, w.drop.seed = h->dropout_seed;  // training forwards only
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
                HIP_TRY(launch_aux_head(h->cmode, x->p, aux_ab, b.p_aux, h->P(b.b), x_t, w.coef, out, B, b.cin, b.cout, s, ts ? ts->raw_out : nullptr));
            else
                HIP_TRY(launch_aux_out(h->dtype, x->p, aux_ab, h->P(b.w), h->P(b.b), x_t, w.coef, out, B, b.res_out, b.cin, b.cout, s, ts ? ts->raw_out : nullptr));
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
        const bool x3 = h->cmode == FG_DTYPE_BF16X3;  // conv_ws3.hip packing: two bf16 planes per weight = 4 bytes, as tsz
        if ((x3 ? conv_x3ws_shape_ok(b->cout, b->cin, b->res_out) : conv_ws_shape_ok(h->cmode, b->cout, b->cin, b->res_out)) && !b->down &&
            (rc = dev_alloc(h, &b->p_conv0_ws, conv_pack_elems(b->cout, b->cin, 3) * tsz)))
            return rc;
        if ((x3 ? conv_x3ws_shape_ok(b->cout, b->cout, b->res_out) : conv_ws_shape_ok(h->cmode, b->cout, b->cout, b->res_out)) &&
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
        if (b.kind == K_AUX_CONV && aux_head_supported(h->cmode, b.res_out, b.cin, b.cout))
            if ((rc = dev_alloc(h, &b.p_aux, aux_pack_elems(b.cin) * tsz))) return rc;
    if ((rc = dev_alloc(h, (void**)&h->aff_w, sizeof(float) * (size_t)h->temb_total * h->emb_ch))) return rc;
    if ((rc = dev_alloc(h, (void**)&h->aff_b, sizeof(float) * (size_t)h->temb_total))) return rc;
    // PositionalEmbedding(endpoint=True) frequencies in fp32, EDM/network.py:314-316
    const int half = h->noise_ch / 2;
    std::vector<float> fr(half);
    for (int j = 0; j < half; ++j) fr[j] = powf(1.0f / 10000.0f, (float)j / (float)(half - 1));
    if ((rc = dev_alloc(h, (void**)&h->freqs, sizeof(float) * half))) return rc;
    HIP_TRY(hipMemcpy(h->freqs, fr.data(), sizeof(float) * half, hipMemcpyHostToDevice));
    if (conv_prepare_all(h->cmode) != 0) return fail(FG_EHIP, "hipFuncSetAttribute(dynamic LDS) failed");
    if (launch_attention(h->cmode, nullptr, nullptr, nullptr, nullptr, 1, 256, nullptr) != 0) return fail(FG_EHIP, "attention prepare failed");
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
    if (cfg->compute_dtype != FG_DTYPE_F32 && cfg->compute_dtype != FG_DTYPE_BF16 && cfg->compute_dtype != FG_DTYPE_BF16X3)
        return fail(FG_EINVAL, "bad compute_dtype");
    if ((cfg->drop_precond & ~3) || (cfg->schedule != FG_SCHEDULE_EDM && cfg->schedule != FG_SCHEDULE_RF) ||
        (cfg->r_timestep & ~1))
        return fail(FG_EINVAL, "bad r_timestep / drop_precond / schedule");
    if (cfg->model_channels <= 0 || cfg->model_channels % 16 || cfg->channel_mult_noise < 1 || cfg->channel_mult_emb < 1)
        return fail(FG_EINVAL, "bad channel configuration");
    fg_edm* h = new fg_edm();
    h->cfg = *cfg;
    h->cmode = cfg->compute_dtype;
    h->dtype = cfg->compute_dtype == FG_DTYPE_BF16 ? 1 : 0;
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
        const bool unused = (p.name == "model.map_augment.weight" && !h->augment) || p.name.rfind("model.logvar_linear", 0) == 0;
        if (!p.ptr && !unused) return fail(FG_ENOTREADY, "parameter '%s' is not bound", p.name.c_str());
    }
    ++h->pack_epoch;
    for (Block* b : h->blocks) {
        HIP_TRY(launch_pack_conv_weights(h->cmode, h->P(b->conv0_w), b->p_conv0, b->cout, b->cin, 3, 0, s));
        HIP_TRY(launch_pack_conv_weights(h->cmode, h->P(b->conv1_w), b->p_conv1, b->cout, b->cout, 3, 0, s));
        if (h->cmode == FG_DTYPE_BF16X3) {
            if (b->p_conv0_ws) HIP_TRY(launch_pack_conv_weights_x3ws(h->P(b->conv0_w), b->p_conv0_ws, b->cout, b->cin, s));
            if (b->p_conv1_ws) HIP_TRY(launch_pack_conv_weights_x3ws(h->P(b->conv1_w), b->p_conv1_ws, b->cout, b->cout, s));
        } else {
            if (b->p_conv0_ws) HIP_TRY(launch_pack_conv_weights_ws(h->P(b->conv0_w), b->p_conv0_ws, b->cout, b->cin, s));
            if (b->p_conv1_ws) HIP_TRY(launch_pack_conv_weights_ws(h->P(b->conv1_w), b->p_conv1_ws, b->cout, b->cout, s));
        }
        if (b->has_skip) HIP_TRY(launch_pack_conv_weights(h->cmode, h->P(b->skip_w), b->p_skip, b->cout, b->cin, 1, 0, s));
        if (b->attn) {
            HIP_TRY(launch_pack_conv_weights(h->cmode, h->P(b->qkv_w), b->p_qkv, 3 * b->cout, b->cout, 1, 1, s));
            HIP_TRY(launch_pack_conv_weights(h->cmode, h->P(b->proj_w), b->p_proj, b->cout, b->cout, 1, 0, s));
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
        if (b.kind == K_AUX_CONV && b.p_aux) HIP_TRY(launch_pack_aux_weights(h->cmode, h->P(b.w), b.p_aux, b.cin, b.cout, s));
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
FG_API int fg_debug_conv_bench(int dtype, int batch, int cin, int res, int ks, int with_resid, int dbg, int iters, float* ms_out) {
    const size_t npix = (size_t)batch * res * res;
    float *x = nullptr, *out = nullptr, *resid = nullptr, *bias = nullptr;
    float2* ab = nullptr;
    void* wp = nullptr;
    if (dtype < 0 || dtype > 2) return fail(FG_EINVAL, "bad dtype");
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
            (rc = fill(wp, (size_t)256 * cin * ks * ks * (dtype == 1 ? 2 : 4))))
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

#include "engine_train.inc"  // block / network backward, forward-mode pass and their entry points
#include "engine_dit.inc"    // the DiT engine (fg_dit_*)
#include "engine_wan.inc"    // the causal video DiT engine (fg_wan_*)
#include "engine_sampler.inc"  // fg_dit_sampler_run / fg_wan_sampler_run: the student loops of the two transformer networks

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
int fg_edm_set_dropout(fg_edm* h, float p, uint64_t seed) {
    if (!h) return fail(FG_EINVAL, "null handle");
    if (!(p >= 0.f && p < 1.f)) return fail(FG_EINVAL, "dropout probability must be in [0, 1), got %g", (double)p);
    h->dropout_p = p;
    h->dropout_seed = seed;
    return FG_OK;
}
int fg_op_dropout_mask(float* out, int64_t total, float p, uint32_t block_index, uint64_t seed, void* stream) {
    if (!out || total <= 0 || (total % 8)) return fail(FG_EINVAL, "fg_op_dropout_mask: bad argument");
    DropArgs d;
    d.p = p, d.block = block_index, d.seed = seed;
    HIP_TRY(launch_dropout_mask(out, total, d, (hipStream_t)stream));
    return FG_OK;
}
int fg_edm_set_training(fg_edm* h, int training) {
    if (!h) return fail(FG_EINVAL, "null handle");
    if ((training != 0) != h->training && h->cfg.sigma_shift != 0.0) drop_graph(h);  // a captured sampler baked the old shift in
    h->training = training != 0;
    return FG_OK;
}
int fg_edm_set_augment(fg_edm* h, const float* augment_labels) {
    if (!h) return fail(FG_EINVAL, "null handle");
    if (augment_labels && h->cfg.augment_dim <= 0) return fail(FG_EINVAL, "this network was built without augment_dim");
    h->augment = augment_labels;
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
        return fail(FG_EINVAL, "fg_op_conv_wgrad: unsupported shape res=%d cin=%d cout=%d ks=%d (res 8/16/32, cin %% 32 [k3] / 128 [k1], cout %% 128, ks 1/3)",
                    res, cin, cout, ks);
    if (batch <= 0 || !act || !dy || !dw || !workspace) return fail(FG_EINVAL, "fg_op_conv_wgrad: bad argument");
    if (workspace_bytes < conv_wgrad_workspace_bytes(batch, res, cin, cout, ks))
        return fail(FG_EINVAL, "fg_op_conv_wgrad: workspace too small (%zu < %zu bytes)", workspace_bytes,
                    conv_wgrad_workspace_bytes(batch, res, cin, cout, ks));
    HIP_TRY(launch_conv_wgrad(FG_DTYPE_BF16, act, dy, dw, batch, res, cin, cout, ks, accumulate, workspace, (hipStream_t)stream));
    return FG_OK;
}
int fg_op_conv_wgrad_f32(const float* act, const float* dy, float* dw, int batch, int res, int cin, int cout, int ks,
                         int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    if (!conv_wgrad_supported(res, cin, cout, ks))
        return fail(FG_EINVAL, "fg_op_conv_wgrad_f32: unsupported shape res=%d cin=%d cout=%d ks=%d", res, cin, cout, ks);
    if (batch <= 0 || !act || !dy || !dw || !workspace) return fail(FG_EINVAL, "fg_op_conv_wgrad_f32: bad argument");
    if (workspace_bytes < conv_wgrad_workspace_bytes(batch, res, cin, cout, ks))
        return fail(FG_EINVAL, "fg_op_conv_wgrad_f32: workspace too small (%zu < %zu bytes)", workspace_bytes,
                    conv_wgrad_workspace_bytes(batch, res, cin, cout, ks));
    HIP_TRY(launch_conv_wgrad(FG_DTYPE_BF16X3, act, dy, dw, batch, res, cin, cout, ks, accumulate, workspace, (hipStream_t)stream));
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
int fg_op_attention_split(const void* q, const void* k, const void* v, void* out, int batch, int heads, int head_dim, int lq, int lkv, int nsplit,
                          void* stream) {
    if (!q || !k || !v || !out || batch <= 0 || heads <= 0 || lq <= 0 || lkv <= 0 || (head_dim != 128 && head_dim != 72))
        return fail(FG_EINVAL, "fg_op_attention: bad argument (head_dim 128 or 72)");
    if (nsplit < 0 || nsplit > 8) return fail(FG_EINVAL, "fg_op_attention_split: nsplit must be in [0, 8]");
    const int D = heads * head_dim;
    // key-split scratch for short grids (freed after the stream has drained: a test entry point, not a hot path)
    void* scratch = nullptr;
    HIP_TRY(hipMalloc(&scratch, fa128_scratch_bytes(batch, heads, lq)));
    const int rc = launch_fa(head_dim, q, D, (int64_t)lq * D, k, v, D, (int64_t)lkv * D, out, D, (int64_t)lq * D, batch, heads, lq, lkv,
                             (hipStream_t)stream, scratch, nsplit);
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(scratch);
    HIP_TRY(rc);
    return FG_OK;
}
int fg_op_attention(const void* q, const void* k, const void* v, void* out, int batch, int heads, int head_dim, int lq, int lkv, void* stream) {
    return fg_op_attention_split(q, k, v, out, batch, heads, head_dim, lq, lkv, 0, stream);
}
int fg_op_gemm_bf16(const void* a, const void* w, const float* bias, void* out, int m, int n, int k, int act, const float* gate,
                    int gate_stride, int gate_rows, const void* resid, int tile_order, void* stream) {
#ifdef FG_TIMING_BUILD
    const int act_ok = 1 | 4 | 8, order_ok = 1023;  // + act 4 / 8: no stores / no epilogue; tile_order 128: cycle stamps (gemm.hip, GM_TIMING)
#else
    const int act_ok = 1, order_ok = 127 | 256 | 512;
#endif
    if (act < 0 || (act & ~act_ok)) return fail(FG_EINVAL, "fg_op_gemm_bf16: act must be 0 (none) or 1 (GELU tanh), got %d", act);
    if (tile_order < 0 || (tile_order & ~order_ok)) return fail(FG_EINVAL, "fg_op_gemm_bf16: bad tile_order %d", tile_order);
    if (((tile_order >> 4) & 1) + ((tile_order >> 5) & 1) + ((tile_order >> 8) & 1) + ((tile_order >> 9) & 1) > 1)
        return fail(FG_EINVAL, "fg_op_gemm_bf16: tile_order bits 16, 32, 256 and 512 exclude each other");
    GemmArgs g;
    g.A = a; g.W = w; g.bias = bias; g.out = out; g.M = m; g.N = n; g.K = k; g.act = act;
    g.gate = gate; g.gate_stride = gate_stride; g.gate_rows = gate_rows > 0 ? gate_rows : 1; g.resid = resid; g.xn = tile_order & 15;
    g.variant = (tile_order & 16) ? 0 : (tile_order & 32) ? 1 : (tile_order & 256) ? 2 : (tile_order & 512) ? 3 : -1;
    if (!gemm_bf16_supported(g)) return fail(FG_EINVAL, "fg_op_gemm_bf16: unsupported shape (k %% 64, n %% 16, pointers)");
    HIP_TRY(launch_gemm_bf16(g, (hipStream_t)stream, true));
    // one scratch allocation, released on every path once the stream has drained (a test entry point, not a hot path)
    void* scratch = nullptr;
    hipError_t herr = hipSuccess;
#ifdef FG_TIMING_BUILD
    const bool stamps = (tile_order & 128) != 0;
    if (stamps && (tile_order & 64)) return fail(FG_EINVAL, "fg_op_gemm_bf16: tile_order bits 64 and 128 exclude each other");
    if (stamps) {  // cycle stamps of workgroup 0: printed per tile, see gemm_bf16_pp_kernel's `stamp`
        if ((herr = hipMalloc(&scratch, 65536)) == hipSuccess) herr = hipMemsetAsync(scratch, 0, 65536, (hipStream_t)stream);
        g.scratch = (float*)scratch;
        g.scratch_bytes = 0;
        g.act |= 16;
    }
#endif
    if (tile_order & 64) {  // let short grids split K
        g.scratch_bytes = (size_t)4 * m * n * 4;
        herr = hipMalloc(&scratch, g.scratch_bytes);
        g.scratch = (float*)scratch;
    }
    int rc = herr == hipSuccess ? launch_gemm_bf16(g, (hipStream_t)stream) : (int)herr;
    if (scratch) {
        (void)hipStreamSynchronize((hipStream_t)stream);
#ifdef FG_TIMING_BUILD
        if (stamps && rc == 0) {
            std::vector<unsigned long long> st(8192);
            (void)hipMemcpy(st.data(), scratch, 65536, hipMemcpyDeviceToHost);
            for (int ti = 0; ti < 16 && st[ti * 16]; ++ti) {
                fprintf(stderr, "tile %2d:", ti);
                for (int gq = 0; gq < 2; ++gq) {
                    const unsigned long long* q = st.data() + ti * 16 + gq;  // slot s at q[2 s]
                    fprintf(stderr, "  group %d: k-loop %6llu  last k-step %5llu  epilogue %6llu = quadrants %5llu %5llu %5llu %5llu  -> next tile %5llu |", gq,
                            q[2] - q[0], q[4] - q[2], q[6] - q[4], q[8] - q[4], q[10] - q[8], q[12] - q[10], q[6] - q[12], st[(ti + 1) * 16 + gq] ? st[(ti + 1) * 16 + gq] - q[6] : 0ull);
                }
                fprintf(stderr, "\n");
            }
        }
#endif
        (void)hipFree(scratch);
    }
    HIP_TRY(rc);
    return FG_OK;
}

}  // extern "C"
