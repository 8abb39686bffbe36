// Native layer loop of the CLIP ViT-B/32-shaped tower (clip_model.py:580-584 -> 230-247 -> 224-227):
//   x += out_proj(MHA(LN1 x)) ; x += c_proj(QuickGELU(c_fc(LN2 x)))
// forward and hand-written backward, launching the kernels of gemm/norm/attn/embed/optim.hip on one
// stream.  Activations needed by the backward live in a caller-allocated "saved" arena (per layer:
// x_in f32, x_mid f32, LN stats, h1/h2/O bf16, qkv bf16, fc pre-activation + GELU output bf16, lse2);
// transient buffers live in a "scratch" arena.  No allocation, no host sync: graph-capturable.
#include "../../include/mmvid_hip.h"
#include "common.h"
#include <vector>

#include "graphs.h"

namespace {

inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
constexpr int kMaxSplitK = 16;  // == the cap of mmvid_gemm_dw_pick_splitk
constexpr int kLnBwdBlocks = 512;   // grid of the LayerNorm backward (its dw/db/colsum partial rows live in the scratch arena)

struct Dims {
    int B, L, E, H, F, layers;
    int64_t M;
};
Dims dims_of(const mmvid_tower_cfg_t& c) {
    Dims d;
    d.B = c.B, d.L = c.L, d.E = c.E, d.H = c.H, d.F = c.F, d.layers = c.layers;
    d.M = (int64_t)c.B * c.L;
    return d;
}

// Grouped weight gradients: the backward keeps every layer's four dY tensors -- bf16(g) in front
// of c_proj (k_gpj) and of out_proj (k_gout), d_pre (k_dpre), dqkv (k_dqkv): M * (E + F + E + 3E) * 2 bytes per layer, 1.7 GB for
// the 12-layer training step -- in the layer's slice of the saved arena (which exists only for a forward that will be
// differentiated), so that the weight gradients of ALL layers of a kind go out as one launch after the layer loop.
struct SavedLayer {  // byte offsets inside one layer's slice of the saved arena
    int64_t x_in, x_mid, mean1, rstd1, mean2, rstd2, h1, qkv, o, lse2, h2, pre, act, k_gpj, k_dpre, k_gout, k_dqkv, k_ln1, k_ln2, total;
};
SavedLayer saved_layout(const Dims& d) {
    SavedLayer s;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += align256(bytes);
        return o;
    };
    s.x_in = take(d.M * d.E * 4);
    s.x_mid = take(d.M * d.E * 4);
    s.mean1 = take(d.M * 4), s.rstd1 = take(d.M * 4), s.mean2 = take(d.M * 4), s.rstd2 = take(d.M * 4);
    s.h1 = take(d.M * d.E * 2);
    s.qkv = take(d.M * 3 * d.E * 2);
    s.o = take(d.M * d.E * 2);
    s.lse2 = take((int64_t)d.B * d.H * d.L * 4);
    s.h2 = take(d.M * d.E * 2);
    s.pre = take(d.M * d.F * 2);
    s.act = take(d.M * d.F * 2);
    const int64_t keep = 1;  // (written by the backward only)
    s.k_gpj = take(keep * d.M * d.E * 2), s.k_dpre = take(keep * d.M * d.F * 2);
    s.k_gout = take(keep * d.M * d.E * 2), s.k_dqkv = take(keep * d.M * 3 * d.E * 2);
    // ... and the partial rows of its two LayerNorm backwards (weight / bias / column-sum gradients), reduced for all layers at once
    s.k_ln1 = take(keep * (int64_t)kLnBwdBlocks * 3 * d.E * 4), s.k_ln2 = take(keep * (int64_t)kLnBwdBlocks * 3 * d.E * 4);
    s.total = off;
    return s;
}

struct Scratch {  // byte offsets inside the scratch arena
    int64_t delta, d_h, d_o, splitk_ws, infer, total;
};
Scratch scratch_layout(const Dims& d) {
    Scratch s;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        int64_t o = off;
        off += align256(bytes);
        return o;
    };
    s.delta = take((int64_t)d.B * d.H * d.L * 4);
    s.d_h = take(d.M * d.E * 2);  // d(LN output), bf16: written by the dX GEMMs of c_fc / in_proj, read by the LayerNorm backward
    s.d_o = take(d.M * d.E * 2);
    s.splitk_ws = take((int64_t)kMaxSplitK * d.F * d.E * 4);  // largest weight ([F,E] >= [3E,E]) x splits (short calls: per-layer dW)
    s.infer = take(saved_layout(d).total);  // one layer's worth of activations for inference mode
    s.total = off;
    return s;
}

#define TRY(call)                   \
    do {                            \
        int rc__ = (call);          \
        if (rc__ != 0) return rc__; \
    } while (0)

int check_cfg(const mmvid_tower_cfg_t* c) {
    MMVID_REQUIRE(c, "tower: null config");
    MMVID_REQUIRE(c->B > 0 && c->L > 0 && c->layers > 0, "tower: bad B/L/layers");
    MMVID_REQUIRE(c->E == c->H * 64 && c->E % 8 == 0 && c->F % 8 == 0 && c->E <= 1024,
                  "tower: need E == 64*H <= 1024 and F %% 8 == 0 (E=%d H=%d F=%d)", c->E, c->H, c->F);
    return 0;
}

// Y = X W^T + b with epilogue options (A row-major [M,K], B row-major [N,K])
int linear_fwd(int64_t M, int N, int K, const void* X, const void* W, const float* bias, const float* residual,
               void* save_pre, int act, float* out_f32, void* out_bf16, void* st) {
    return mmvid_gemm_bf16(0, 0, (int)M, N, K, X, K, W, K, 1, 0, 0, 0, 1, 1.0f, bias, residual, N, nullptr, save_pre, N,
                           act, 0, out_f32, out_bf16, N, nullptr, st);
}
// dX = dY W (A = dY [M,N] row-major, B = W [N(red)][K(out)] k-major); out_colsum: [K] += column sums of dX, i.e. the
// bias gradient of the Linear that produced this layer's input
int linear_dx(int64_t M, int N, int K, const void* dY, const void* W, const void* dact_pre, float* out_f32,
              void* out_bf16, void* st, float* out_colsum = nullptr) {
    return mmvid_gemm_bf16(0, 1, (int)M, K, N, dY, N, W, K, 1, 0, 0, 0, 1, 1.0f, nullptr, nullptr, 0, dact_pre, nullptr, K,
                           0, 0, out_f32, out_bf16, K, out_colsum, st);
}
// dW[N,K] += dY^T X (both k-major over the token dimension), db[N] += colsum(dY)
int linear_dw(int64_t M, int N, int K, const void* dY, const void* X, float* dW, float* db, float* ws, void* st) {
    if (dW) {  // a frozen weight has no gradient buffer: its activations still need dX (a frozen tower under trainable embeddings)
        const int sk = mmvid_gemm_dw_pick_splitk(M, N, K);
        TRY(mmvid_gemm_bf16_dw(M, N, K, dY, N, X, K, sk, ws, dW, /*accumulate=*/1, st));
    }
    if (db) TRY(mmvid_colsum_bf16(dY, N, M, N, db, st));
    return 0;
}

}  // namespace

extern "C" int mmvid_tower_workspace(const mmvid_tower_cfg_t* cfg, int64_t* saved_bytes, int64_t* scratch_bytes) {
    TRY(check_cfg(cfg));
    const Dims d = dims_of(*cfg);
    if (saved_bytes) *saved_bytes = saved_layout(d).total * d.layers;
    if (scratch_bytes) *scratch_bytes = scratch_layout(d).total;
    return MMVID_OK;
}

static int tower_forward_enqueue(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                                 float* x_out, void* saved, void* scratch, void* stream, void* kv_cache = nullptr,
                                 int kv_lmax = 0);
static int tower_backward_enqueue(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, float* g,
                                  void* saved, void* scratch, void* stream);

static uint64_t tower_key(uint64_t seed, const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const void* a,
                          const void* b, const void* c, const void* d, const void* stream) {
    uint64_t k = mmvid_hash_bytes(cfg, sizeof(*cfg), 0xcbf29ce484222325ull ^ seed);
    k = mmvid_hash_bytes(layers, sizeof(mmvid_tower_layer_t) * (size_t)cfg->layers, k);
    return mmvid_hash_ptr(stream, mmvid_hash_ptr(d, mmvid_hash_ptr(c, mmvid_hash_ptr(b, mmvid_hash_ptr(a, k)))));
}

extern "C" int mmvid_tower_forward(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers,
                                   const float* x_in, float* x_out, void* saved, void* scratch, void* stream) {
    TRY(check_cfg(cfg));
    MMVID_REQUIRE(layers && x_in && x_out && scratch, "tower_forward: null pointer");
    const uint64_t key = tower_key(2, cfg, layers, x_in, x_out, saved, scratch, stream);
    return mmvid_run_cached(key, (hipStream_t)stream, [=](hipStream_t s) {
        return tower_forward_enqueue(cfg, layers, x_in, x_out, saved, scratch, (void*)s);
    });
}

static int tower_forward_enqueue(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                                 float* x_out, void* saved, void* scratch, void* stream, void* kv_cache, int kv_lmax) {
    const Dims d = dims_of(*cfg);
    const SavedLayer sl = saved_layout(d);
    const Scratch sc = scratch_layout(d);
    char* scr = (char*)scratch;
    const float scale = 0.125f;  // head_dim^-0.5, head_dim = 64
    const float* x = x_in;
    for (int i = 0; i < d.layers; ++i) {
        const mmvid_tower_layer_t& ly = layers[i];
        char* sv = saved ? (char*)saved + (int64_t)i * sl.total : scr + sc.infer;
        float* xin_s = (float*)(sv + sl.x_in);
        float* xmid = (float*)(sv + sl.x_mid);
        // the layer's input must survive for the backward: copy it into the arena (first layer) or it already
        // lives there (previous layer wrote its output into this layer's x_in slot).
        if (saved && i == 0) {
            if (hipMemcpyAsync(xin_s, x_in, d.M * d.E * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
                mmvid_set_error("tower_forward: memcpy failed");
                return MMVID_ERR_HIP;
            }
            x = xin_s;
        }
        float* xnext;
        if (i == d.layers - 1)
            xnext = x_out;
        else if (saved)
            xnext = (float*)((char*)saved + (int64_t)(i + 1) * sl.total + sl.x_in);
        else
            xnext = x_out;  // inference: ping through x_out (x_mid holds the intermediate)
        TRY(mmvid_layernorm_fwd(x, d.E, d.M, d.E, ly.ln1_w, ly.ln1_b, cfg->ln_eps, sv + sl.h1, nullptr, d.E,
                                (float*)(sv + sl.mean1), (float*)(sv + sl.rstd1), stream));
        TRY(linear_fwd(d.M, 3 * d.E, d.E, sv + sl.h1, ly.in_w, ly.in_b, nullptr, nullptr, 0, nullptr, sv + sl.qkv, stream));
        if (kv_cache)  // prefill of the incremental decoder: keep this layer's keys and values
            TRY(mmvid_kv_store(sv + sl.qkv, 3 * d.E, d.B, d.L, d.E, nullptr, 0, kv_lmax,
                               (char*)kv_cache + (int64_t)i * d.B * kv_lmax * 2 * d.E * 2, stream));
        TRY(mmvid_attention_fwd(sv + sl.qkv, 3 * d.E, d.B, d.L, d.H, d.E, scale, cfg->mask_mode, cfg->r0, cfg->c0, cfg->r1,
                                cfg->c1, sv + sl.o, d.E, (float*)(sv + sl.lse2), stream));
        TRY(linear_fwd(d.M, d.E, d.E, sv + sl.o, ly.out_w, ly.out_b, x, nullptr, 0, xmid, nullptr, stream));
        TRY(mmvid_layernorm_fwd(xmid, d.E, d.M, d.E, ly.ln2_w, ly.ln2_b, cfg->ln_eps, sv + sl.h2, nullptr, d.E,
                                (float*)(sv + sl.mean2), (float*)(sv + sl.rstd2), stream));
        TRY(linear_fwd(d.M, d.F, d.E, sv + sl.h2, ly.fc_w, ly.fc_b, nullptr, saved ? sv + sl.pre : nullptr, 1, nullptr,
                       sv + sl.act, stream));
        TRY(linear_fwd(d.M, d.E, d.F, sv + sl.act, ly.pj_w, ly.pj_b, xmid, nullptr, 0, xnext, nullptr, stream));
        x = xnext;
    }
    return MMVID_OK;
}

extern "C" int mmvid_tower_backward(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, float* g,
                                    void* saved, void* scratch, void* stream) {
    TRY(check_cfg(cfg));
    MMVID_REQUIRE(layers && g && saved && scratch, "tower_backward: null pointer");
    const uint64_t key = tower_key(3, cfg, layers, g, saved, scratch, nullptr, stream);
    return mmvid_run_cached(key, (hipStream_t)stream, [=](hipStream_t s) {
        return tower_backward_enqueue(cfg, layers, g, saved, scratch, (void*)s);
    });
}

// The layer loop with the weight gradients taken out of it: every layer writes its four dY tensors into its own slice of the
// scratch arena's `keep` region (no copies: the kernels that produce them are pointed there), and after the loop the weight
// gradients of each kind -- c_proj, c_fc, out_proj, in_proj -- are ONE launch over all layers of this call, each block reducing
// over all tokens (no split-K slabs, no reduce launches).  A kind whose group would leave the chip mostly idle (few layers per
// call: the chunked backward of the multi-GPU engine) keeps the per-layer split-K launches, run after the loop on the same data.
// Same arithmetic per element up to the fp32 summation order of the token reduction (one chain instead of split-K slabs).
static int tower_backward_enqueue(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, float* g, void* saved,
                                  void* scratch, void* stream) {
    const Dims d = dims_of(*cfg);
    const SavedLayer sl = saved_layout(d);
    const Scratch sc = scratch_layout(d);
    char* scr = (char*)scratch;
    char* keep = (char*)saved;  // (the kept tensors are the backward's own: written here, read by the launches after the loop)
    struct { int64_t g_pj, d_pre, g_out, dqkv, total; } kl = {sl.k_gpj, sl.k_dpre, sl.k_gout, sl.k_dqkv, sl.total};
    const float scale = 0.125f;
    void* d_h = scr + sc.d_h;
    float* ws = (float*)(scr + sc.splitk_ws);
    const int64_t ln_ws_floats = (int64_t)kLnBwdBlocks * 3 * d.E;
    std::vector<mmvid_ln_reduce_t> ln_items;
    int ln_blocks = 0;
    for (int i = d.layers - 1; i >= 0; --i) {
        const mmvid_tower_layer_t& ly = layers[i];
        const char* sv = (const char*)saved + (int64_t)i * sl.total;
        char* kp = keep + (int64_t)i * kl.total;
        void* g_pj = kp + kl.g_pj;  // bf16(g) in front of c_proj: the cast below (top layer of the call) or the layer above's LN1 backward
        if (i == d.layers - 1) {
            TRY(mmvid_cast_f32_to_bf16(g, g_pj, d.M * d.E, stream));
            if (ly.g_pj_b) TRY(mmvid_colsum_bf16(g_pj, d.E, d.M, d.E, ly.g_pj_b, stream));
        }
        // d_pre = (g W_proj) * QuickGELU'(pre); its column sums are c_fc's bias gradient: taken in this epilogue (unrounded fp32 sums)
        TRY(linear_dx(d.M, d.E, d.F, g_pj, ly.pj_w, sv + sl.pre, nullptr, kp + kl.d_pre, stream, ly.g_fc_b));
        TRY(linear_dx(d.M, d.F, d.E, kp + kl.d_pre, ly.fc_w, nullptr, nullptr, d_h, stream));
        {
            mmvid_ln_reduce_t r = {(const float*)(kp + sl.k_ln2), ly.g_ln2_w, ly.g_ln2_b, ly.g_out_b};
            int nb = 0;
            TRY(mmvid_layernorm_bwd_partial(d_h, 1, d.E, (const float*)(sv + sl.x_mid), d.E, (const float*)(sv + sl.mean2),
                                            (const float*)(sv + sl.rstd2), ly.ln2_w, d.M, d.E, g, d.E, 1, kp + kl.g_out, r.dw != nullptr,
                                            r.db != nullptr, r.dx_colsum != nullptr, (float*)(kp + sl.k_ln2), ln_ws_floats, &nb, stream));
            if (r.dw || r.db || r.dx_colsum) ln_items.push_back(r), ln_blocks = nb;
        }
        TRY(linear_dx(d.M, d.E, d.E, kp + kl.g_out, ly.out_w, nullptr, nullptr, scr + sc.d_o, stream));
        // (the in-projection's bias gradient -- column sums of dqkv -- comes out of the attention backward's registers)
        TRY(mmvid_attention_bwd_bias(sv + sl.qkv, 3 * d.E, sv + sl.o, d.E, scr + sc.d_o, d.E, (const float*)(sv + sl.lse2),
                                     (float*)(scr + sc.delta), d.B, d.L, d.H, d.E, scale, cfg->mask_mode, cfg->r0, cfg->c0,
                                     cfg->r1, cfg->c1, kp + kl.dqkv, 3 * d.E, ly.g_in_b, stream));
        TRY(linear_dx(d.M, 3 * d.E, d.E, kp + kl.dqkv, ly.in_w, nullptr, nullptr, d_h, stream));
        {
            mmvid_ln_reduce_t r = {(const float*)(kp + sl.k_ln1), ly.g_ln1_w, ly.g_ln1_b, i > 0 ? layers[i - 1].g_pj_b : nullptr};
            int nb = 0;
            TRY(mmvid_layernorm_bwd_partial(d_h, 1, d.E, (const float*)(sv + sl.x_in), d.E, (const float*)(sv + sl.mean1),
                                            (const float*)(sv + sl.rstd1), ly.ln1_w, d.M, d.E, g, d.E, 1,
                                            i > 0 ? (void*)(keep + (int64_t)(i - 1) * kl.total + kl.g_pj) : nullptr, r.dw != nullptr,
                                            r.db != nullptr, r.dx_colsum != nullptr, (float*)(kp + sl.k_ln1), ln_ws_floats, &nb, stream));
            if (r.dw || r.db || r.dx_colsum) ln_items.push_back(r), ln_blocks = nb;
        }
    }
    // ---- LayerNorm weight / bias gradients and the column sums that are the biases' gradients: one reduction for all layers
    if (!ln_items.empty()) TRY(mmvid_layernorm_bwd_reduce_multi((int)ln_items.size(), ln_items.data(), ln_blocks, d.E, stream));
    // ---- the weight gradients: dW[N][K] += dY^T X per kind
    struct Kind {
        int N, K;
        int64_t dy_off, x_off;  // inside a layer's keep slice / saved slice
        float* mmvid_tower_layer_t::*gw;
    };
    const Kind kinds[4] = {{d.E, d.F, kl.g_pj, sl.act, &mmvid_tower_layer_t::g_pj_w},
                           {d.F, d.E, kl.d_pre, sl.h2, &mmvid_tower_layer_t::g_fc_w},
                           {d.E, d.E, kl.g_out, sl.o, &mmvid_tower_layer_t::g_out_w},
                           {3 * d.E, d.E, kl.dqkv, sl.h1, &mmvid_tower_layer_t::g_in_w}};
    // ONE launch for the four kinds (one launch per kind left three partial last rounds; putting two of the four on a second stream
    // so that those overlap was measured SLOWER on the captured step, 15.69 vs 15.56 ms: profiles/r03_ab_whole_step_dw_grouped.log).
    std::vector<float*> outs((size_t)d.layers * 4);
    mmvid_dw_kind_t kd[4];
    int nk = 0;
    for (const Kind& k : kinds) {
        float** o = outs.data() + (size_t)nk * d.layers;
        bool any = false;
        for (int i = 0; i < d.layers; ++i) o[i] = layers[i].*(k.gw), any = any || o[i];
        if (!any) continue;  // a frozen weight (a frozen tower under trainable embeddings: none at all)
        kd[nk].N = k.N, kd[nk].K = k.K;
        kd[nk].dY = keep + k.dy_off, kd[nk].ldy = k.N, kd[nk].strideY = kl.total / 2;
        kd[nk].X = (const char*)saved + k.x_off, kd[nk].ldx = k.K, kd[nk].strideX = sl.total / 2;
        kd[nk].dW_list = o;
        ++nk;
    }
    if (nk == 0) return MMVID_OK;
    if (mmvid_gemm_dw_multi_fill(nk, kd, d.layers) >= 0.7) return mmvid_gemm_bf16_dw_multi(d.M, nk, kd, d.layers, /*accumulate=*/1, stream);
    for (int k = 0; k < nk; ++k)  // few tiles (short calls of a chunked backward on a small model): per layer, split-K
        for (int i = d.layers - 1; i >= 0; --i)
            TRY(linear_dw(d.M, kd[k].N, kd[k].K, (const char*)kd[k].dY + (int64_t)i * kl.total, (const char*)kd[k].X + (int64_t)i * sl.total,
                          kd[k].dW_list[i], nullptr, ws, stream));
    return MMVID_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Incremental decoding of the causal tower (decode.hip).  kv_cache: [layers][B][Lmax][2E] bf16.
extern "C" int mmvid_tower_prefill(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                                   float* x_out, void* kv_cache, int Lmax, void* scratch, void* stream) {
    TRY(check_cfg(cfg));
    MMVID_REQUIRE(layers && x_in && x_out && kv_cache && scratch, "tower_prefill: null pointer");
    MMVID_REQUIRE(cfg->mask_mode == 1 && cfg->L <= Lmax, "tower_prefill: needs the causal mask and L (%d) <= Lmax (%d)", cfg->L,
                  Lmax);
    return tower_forward_enqueue(cfg, layers, x_in, x_out, nullptr, scratch, stream, kv_cache, Lmax);
}

// One new position per sequence: x_in / x_out [B, E] fp32; its index comes from the device scalar pos_dev (so a
// captured step can be replayed for every position) or, when that is NULL, from pos.  cfg->L is ignored.
extern "C" int mmvid_tower_decode(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                                  float* x_out, void* kv_cache, int Lmax, const int32_t* pos_dev, int pos, void* scratch,
                                  void* stream) {
    TRY(check_cfg(cfg));
    MMVID_REQUIRE(layers && x_in && x_out && kv_cache && scratch, "tower_decode: null pointer");
    MMVID_REQUIRE(cfg->mask_mode == 1, "tower_decode: incremental decoding needs the causal mask");
    const int B = cfg->B, E = cfg->E, F = cfg->F, H = cfg->H;
    // a few [B, *] rows out of the scratch arena (it is sized for B*L tokens)
    char* p = (char*)scratch;
    auto take = [&](int64_t bytes) {
        char* o = p;
        p += align256(bytes);
        return o;
    };
    void* h = take((int64_t)B * E * 2);
    void* qkv = take((int64_t)B * 3 * E * 2);
    void* o = take((int64_t)B * E * 2);
    float* xmid = (float*)take((int64_t)B * E * 4);
    void* act = take((int64_t)B * F * 2);
    float* xa = (float*)take((int64_t)B * E * 4);
    float* xb = (float*)take((int64_t)B * E * 4);
    const float* x = x_in;
    for (int i = 0; i < cfg->layers; ++i) {
        const mmvid_tower_layer_t& ly = layers[i];
        void* cache = (char*)kv_cache + (int64_t)i * B * Lmax * 2 * E * 2;
        float* xnext = (i == cfg->layers - 1) ? x_out : ((i & 1) ? xb : xa);
        TRY(mmvid_layernorm_fwd(x, E, B, E, ly.ln1_w, ly.ln1_b, cfg->ln_eps, h, nullptr, E, nullptr, nullptr, stream));
        TRY(linear_fwd(B, 3 * E, E, h, ly.in_w, ly.in_b, nullptr, nullptr, 0, nullptr, qkv, stream));
        TRY(mmvid_kv_store(qkv, 3 * E, B, 1, E, pos_dev, pos, Lmax, cache, stream));
        TRY(mmvid_attention_decode(qkv, 3 * E, cache, B, Lmax, H, E, pos_dev, pos, 0.125f, o, E, stream));
        TRY(linear_fwd(B, E, E, o, ly.out_w, ly.out_b, x, nullptr, 0, xmid, nullptr, stream));
        TRY(mmvid_layernorm_fwd(xmid, E, B, E, ly.ln2_w, ly.ln2_b, cfg->ln_eps, h, nullptr, E, nullptr, nullptr, stream));
        TRY(linear_fwd(B, F, E, h, ly.fc_w, ly.fc_b, nullptr, nullptr, 1, nullptr, act, stream));
        TRY(linear_fwd(B, E, F, act, ly.pj_w, ly.pj_b, xmid, nullptr, 0, xnext, nullptr, stream));
        x = xnext;
    }
    return MMVID_OK;
}
