// One decode step of the causal tower (dalle_artv.py:236-304 over a key/value cache, SURVEY next-row N1) as ONE launch.
//
// Why.  csrc/decode.hip runs the step as 60 dependent launches (five per layer); each lasts 4.4 us although it moves a few MB, because
// every launch pays the same chain again -- dispatch ramp (0.9 us), kernel arguments (0.8), input rows (0.6), ..., completion
// (tools/decode_gemv_timeline.py) -- 270-370 us per token against a 27-us weight-streaming floor.  Here 256 co-resident blocks (one
// per CU) walk the 60 phases inside one kernel, and what a phase produces reaches the blocks that consume it WITHOUT a barrier:
//
//   * every value exchanged between blocks is an 8-byte word = fp32 payload | 32-bit tag, written and read with agent-scope relaxed
//     atomics (single-copy atomic: whoever sees the tag sees the payload).  The tag names (decode step, layer, phase); a consumer polls
//     the words it needs until they carry the tag it expects: the data is its own flag.  tools/chain_probe.py: 2.4 us per hop over 256
//     blocks (store -> memory side -> load; a counter barrier with release / acquire costs 8.7 us, a launch boundary 4.4).
//   * weights do not depend on anything: a wave requests the rows of its next phases' output features (16 B per lane, straight into
//     registers: 1 wave per SIMD, 512 VGPRs) right BEFORE a poll -- the poll is a memory round trip anyway, and they arrive under it.
//   * the attention over the cache is split over S key ranges per (sequence, head) (S = 4: 48 blocks per sequence instead of 12 stream the
//     cache; the first 256 keys / values of a range are requested before q exists), each writing (o[64], max, sum); the out-projection's
//     loader merges them.  fc activations travel as bf16 pairs (half the lines to poll); residual rows stay in LDS.
//
//   phase 1  q,k,v = LN1(x) W_in^T + b        9 output features per block; K|V appended to the cache (plain stores: for later steps)
//   phase 2  partial attention                 block u < B * 12 * S: (sequence, head, key range); the new position's K|V come from phase 1's words
//   phase 3  x_mid = x + merge(o) W_out^T + b  3 features per block
//   phase 4  a = QuickGELU(LN2(x_mid) W_fc^T + b)   12 features per block
//   phase 5  x' = x_mid + a W_proj^T + b       3 features per block (K = 3072)
// Operands are rounded to bf16 where the full forward stores bf16 (LayerNorm output, q/k/v, o, the GELU output), as csrc/decode.hip
// does: the two paths differ by fp32 summation order only.  What bounds it: five coherent round trips per layer.
// HBM/latency-bound; shape: E = 768, F = 3072, 12 heads, <= 12 layers, B <= 2 (at B = 4 the five-launch form is faster: every block
// polls every row, and the rows no longer fit the registers of the pollers).
//
// Safety: a poll gives up after PD_SPIN rounds (~0.3 s) and raises workspace word 1 instead of hanging the device -- it cannot happen
// while all 256 blocks are resident, which the launcher checks (CU count) and which holds when nothing else shares the device.
#include "../../include/mmvid_hip.h"
#include "common.h"
#include "wave_reduce.h"

namespace {

constexpr int PD_BLOCKS = 256, PD_E = 768, PD_F = 3072, PD_H = 12, PD_LAYERS = 12, PD_MAXB = 2, PD_S = 4;
constexpr int PD_REC = 72;         // words of one partial-attention record: o[64], max, sum, pad
constexpr int PD_MAXKEYS = 1024;   // keys of one attention unit
constexpr int PD_SPIN = 1 << 18;
constexpr int PD_MAXV = 2048;  // classes of the token step's head (a multiple of 1,024)
typedef unsigned long long u64;

struct PdLayer {
    const bf16_t *in_w, *out_w, *fc_w, *pj_w;
    const float *in_b, *out_b, *fc_b, *pj_b, *ln1_w, *ln1_b, *ln2_w, *ln2_b;
};
struct PdArgs {
    PdLayer ly[PD_LAYERS];
    const float* x_in;
    float* x_out;
    bf16_t* cache;  // [layers][B][Lmax][2E]
    u64* ws;        // [0] step counter, [1] failure flag, [8...] the tagged rows
    const int* pos_dev;
    int* pos_advance;  // = pos_dev when the step itself advances the position (block 0, after its last phase), or null
    int pos0, layers, Lmax, NB;
    float eps, scale_log2;
    // The sampler's whole token in the same launch (mmvid_artv_token_step_persistent; tk.table == null: the tower step alone): the input
    // row is the embedding of the token drawn last (table[tok] + pos_rows[pos + pos_off], dalle_artv.py:484-491), and after the last layer
    // come the head (LN + the image block of to_logits) and the draw of the next token (exponential race, csrc/sample.hip).
    struct Tok {
        long long* tok;  // [B]: read when the step starts, overwritten by the draw when it ends
        const float* table;
        long table_rows;
        const float* pos_rows;
        int pos_off;
        long long* record;  // [B][record_ld] or null: record[b][pos - record_pos0] = tok[b]
        long record_ld;
        int record_pos0;
        const float *lnf_w, *lnf_b;
        float lnf_eps;
        const bf16_t* head_w;  // [V][E]
        const float* head_b;
        int V;
        const float* E;  // [draws][B][V]: the draw made here is number (pos + 1 - e_pos0)
        long e_step_stride;
        int e_pos0;
        float inv_temp;
        long long tok_offset;
        float* logits_out;  // [B][V] or null
    } tk;
    int nowait;  // measurement only (MMVID_PD_NOWAIT bit 0: polls accept whatever they read -- the step without its dependency chain; bit 1: no weight loads; results void)
    u64* trace;  // measurement only (mmvid_decode_persistent_trace): [4 blocks][layers][16] wall-clock stamps of blocks 0, 1, 128, 255
};
u64* g_pd_trace = nullptr;

__device__ __forceinline__ u64 ld_word(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_word(u64* p, uint32_t tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The failure flag (workspace word 1).  A poll that runs out of patience raises it (agent scope: the other XCDs see it) and goes on with
// whatever it read; every OTHER poll looks at the flag every PD_FLAG_EVERY rounds and gives up at once when it is up -- one time-out
// ends the launch within a fraction of a millisecond instead of costing ~0.3 s in each of the launch's 60 dependent phases.
constexpr int PD_FLAG_EVERY = 1024;
__device__ __forceinline__ void raise_fail(u64* fail) {
    __hip_atomic_store(fail, (u64)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
__device__ __forceinline__ bool give_up(int spins, u64* fail) {
    if (spins > PD_SPIN) {  // (every later launch on this workspace returns at once: the step is void)
        raise_fail(fail);
        return true;
    }
    return (spins & (PD_FLAG_EVERY - 1)) == PD_FLAG_EVERY - 1 && ld_word(fail) != 0;
}
// N words at p, p + stride, ...: all requested at once, then re-requested one by one until they carry `tag`
template <int N>
__device__ __forceinline__ void poll_words(const u64* p, long stride, uint32_t tag, float (&v)[N], u64* fail) {
    u64 w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = ld_word(p + i * stride);
    for (int spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (tag && (uint32_t)(w[i] >> 32) != tag) ok = false, w[i] = ld_word(p + i * stride);
        if (ok) break;
        if (give_up(spins, fail)) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __uint_as_float((uint32_t)w[i]);
}

// the same with the word addresses given by a functor (index -> pointer)
template <int N, typename F>
__device__ __forceinline__ void poll_fn(F at, uint32_t tag, float (&v)[N], u64* fail) {
    u64 w[N];
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = ld_word(at(i));
    for (int spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (tag && (uint32_t)(w[i] >> 32) != tag) ok = false, w[i] = ld_word(at(i));
        if (ok) break;
        if (give_up(spins, fail)) break;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = __uint_as_float((uint32_t)w[i]);
}
// the same over a [NR][NC] block of words (row stride rs, column stride cs)
template <int NR, int NC>
__device__ __forceinline__ void poll_words2(const u64* p, long rs, long cs, uint32_t tag, float (&v)[NR * NC], u64* fail) {
    u64 w[NR * NC];
#pragma unroll
    for (int i = 0; i < NR * NC; ++i) w[i] = ld_word(p + (i / NC) * rs + (i % NC) * cs);
    for (int spins = 0;; ++spins) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NR * NC; ++i)
            if (tag && (uint32_t)(w[i] >> 32) != tag) ok = false, w[i] = ld_word(p + (i / NC) * rs + (i % NC) * cs);
        if (ok) break;
        if (give_up(spins, fail)) break;
    }
#pragma unroll
    for (int i = 0; i < NR * NC; ++i) v[i] = __uint_as_float((uint32_t)w[i]);
}

typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
__device__ __forceinline__ float dot2(uint32_t w, uint32_t x, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w), __builtin_bit_cast(bf2_t, x), acc, false);
}
// a weight row of 768 columns over the 64 lanes: columns [lane * 8, +8) and [512 + lane * 4, +4)
struct W768 {
    uint4 a;
    uint2 b;
};
typedef uint32_t pd_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pd_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ld_g16(const __attribute__((address_space(1))) bf16_t* p) {
    const pd_u32x4 v = *(const __attribute__((address_space(1))) pd_u32x4*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 ld_g8(const __attribute__((address_space(1))) bf16_t* p) {
    const pd_u32x2 v = *(const __attribute__((address_space(1))) pd_u32x2*)p;
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ W768 load_w768(const __attribute__((address_space(1))) bf16_t* row, int lane) {
    W768 w;
    w.a = ld_g16(row + lane * 8);
    w.b = ld_g8(row + 512 + lane * 4);
    return w;
}
__device__ __forceinline__ float dot768(const W768& w, const uint4& xa, const uint2& xb) {
    float s = dot2(w.a.x, xa.x, 0.f);
    s = dot2(w.a.y, xa.y, s), s = dot2(w.a.z, xa.z, s), s = dot2(w.a.w, xa.w, s);
    s = dot2(w.b.x, xb.x, s), s = dot2(w.b.y, xb.y, s);
    return s;
}
__device__ __forceinline__ float round_bf16_(float v) { return bf2f(f2bf(v)); }

// LayerNorm of one 768-wide row held as v[i] = x[i * 64 + lane] -> bf16 row in LDS (what the full forward feeds its GEMM)
__device__ __forceinline__ void ln_row_to_lds(const float (&v)[12], const float (&g)[12], const float (&bb)[12], float eps, bf16_t* dst, float* raw,
                                              int lane) {
#pragma unroll
    for (int i = 0; i < 12; ++i) raw[i * 64 + lane] = v[i];
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) t += v[i];
    const float mu = wave_sum_fast(t) * (1.0f / PD_E);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) q += (v[i] - mu) * (v[i] - mu);
    const float rs = rsqrtf(wave_sum_fast(q) * (1.0f / PD_E) + eps);
#pragma unroll
    for (int i = 0; i < 12; ++i) dst[i * 64 + lane] = f2bf((v[i] - mu) * rs * g[i] + bb[i]);
}

// A table entry read from LDS: moved to scalar registers (it is the same for every lane) and typed as a GLOBAL pointer -- a pointer
// that comes out of memory is generic to the compiler, and generic (flat_load) accesses make it wait for every outstanding load before
// each batch of requests and tie the LDS counter to them: the weight prefetch, the point of this kernel, would serialise.
#define PD_GLOBAL __attribute__((address_space(1)))
typedef const PD_GLOBAL bf16_t* gbf_p;
typedef const PD_GLOBAL float* gfl_p;
struct GLayer {
    gbf_p in_w, out_w, fc_w, pj_w;
    gfl_p in_b, out_b, fc_b, pj_b, ln1_w, ln1_b, ln2_w, ln2_b;
};
template <typename T>
__device__ __forceinline__ const PD_GLOBAL T* uniform_ptr(const T* p) {
    const u64 v = (u64)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const PD_GLOBAL T*)(((u64)hi << 32) | lo);
}
__device__ __forceinline__ GLayer uniform_layer(const PdLayer& s) {
    GLayer d;
    d.in_w = uniform_ptr(s.in_w), d.out_w = uniform_ptr(s.out_w), d.fc_w = uniform_ptr(s.fc_w), d.pj_w = uniform_ptr(s.pj_w);
    d.in_b = uniform_ptr(s.in_b), d.out_b = uniform_ptr(s.out_b), d.fc_b = uniform_ptr(s.fc_b), d.pj_b = uniform_ptr(s.pj_b);
    d.ln1_w = uniform_ptr(s.ln1_w), d.ln1_b = uniform_ptr(s.ln1_b), d.ln2_w = uniform_ptr(s.ln2_w), d.ln2_b = uniform_ptr(s.ln2_b);
    return d;
}

template <int NBT>
__global__ __launch_bounds__(256) void decode_persistent_kernel(PdArgs a) {
    constexpr int S = PD_S;  // key ranges per (sequence, head)
    constexpr bool PRE = NBT == 1;  // the first 256 keys / values of the range requested ahead (register budget: batch 1 only)
    __shared__ __attribute__((aligned(16))) bf16_t xs[NBT][PD_F];  // the staged input rows of the current phase (bf16-exact)
    __shared__ float xres[2][NBT][PD_E];  // fp32 rows of x (phase 3's residual) and x_mid (phase 5's): staged by the loader waves with the LayerNorm rows
    __shared__ float qkn[192];                                     // attention unit: q | new K | new V of its head
    __shared__ float sc[PD_MAXKEYS];
    __shared__ float red[4][64];
    __shared__ float stat[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), blk = blockIdx.x;
    const int NB = a.NB;
    u64* const fail = a.ws + 1;
    if (a.ws[1] != 0) return;  // a poll of an earlier step on this workspace timed out (block-uniform: written before this launch)
    const uint32_t seq = (uint32_t)a.ws[0] + 1u;
    u64* const X = a.ws + 8;
    u64* const QKV = X + (long)NBT * PD_E;
    u64* const PART = QKV + (long)NBT * 3 * PD_E;
    u64* const XMID = PART + (long)NBT * PD_H * S * PD_REC;
    u64* const ACT = XMID + (long)NBT * PD_E;
    u64* const LOGITS = ACT + (long)NBT * PD_F;  // (the token step's head output: up to 2,048 words per row)
    const int pos = a.pos_dev ? *a.pos_dev : a.pos0;
    const int n = pos + 1 < a.Lmax ? pos + 1 : a.Lmax;  // (a position beyond the cache attends the cache and is not appended: never out of bounds)
    // Output features are numbered so that a BLOCK's words of a row are contiguous: a 64-byte line whose eight words come from eight
    // blocks on eight XCDs reaches its readers 0.8 us later than one with two or three writers (tools/chain_probe.py, "writers
    // interleaved").  768-wide rows: feature 3 blk + wave (waves 0-2); q|k|v: 9 per block (wave 0: 3, waves 1-3: 2 each); fc pairs: 6 per
    // block (waves 0, 1: 2 each, waves 2, 3: 1 each).
    const int f0 = 3 * blk + wave;                                           // phases 3 and 5 (wave < 3)
    const int n1 = wave == 0 ? 3 : 2, f1 = 9 * blk + (wave == 0 ? 0 : 1 + 2 * wave);   // phase 1: features f1 .. f1 + n1 - 1
    const int n4 = wave < 2 ? 2 : 1, p4 = 6 * blk + (wave < 2 ? 2 * wave : 2 + wave);   // phase 4: pairs p4 .. p4 + n4 - 1
    const int tslot = blk == 0 ? 0 : (blk == 1 ? 1 : (blk == 128 ? 2 : (blk == 255 ? 3 : -1)));
    u64* const tr = (a.trace && tslot >= 0 && tid == 0) ? a.trace + (long)tslot * PD_LAYERS * 16 : nullptr;
#define PD_STAMP(i) \
    if (tr) tr[l * 16 + (i)] = clock64();  /* (shader clock: s_memtime is local to the CU; the wall clock costs a trip of its own) */

    // the layer table: from the kernel-argument segment to LDS once (indexing it by the layer is a scalar load per use otherwise, and
    // scalar loads share the counter of the LDS waits that follow them)
    __shared__ PdLayer lys[PD_LAYERS];
    {
        const u64* src = reinterpret_cast<const u64*>(&a.ly[0]);
        u64* dst = reinterpret_cast<u64*>(&lys[0]);
        for (int i = tid; i < (int)(sizeof(PdLayer) * PD_LAYERS / 8); i += 256) dst[i] = src[i];
    }
    __syncthreads();
    int vz = 0;  // an opaque zero in a vector register: `p[i + vz]` is a VECTOR load even when i is wave-uniform (biases), for the same reason
    asm volatile("" : "+v"(vz));
    // weights / biases / LayerNorm rows of the phases ahead (registers)
    W768 w1[3], w3, w4[4];
    uint4 w5[6];
    float b1[3], b3, b4[4], b5;
    float g1[12], h1[12], g2[12], h2[12];
    const bool noweights = (a.nowait & 2) != 0;  // measurement only: the step without its weight stream (operands are whatever the registers hold)
    auto fetch1 = [&](const GLayer& L) {
        if (noweights) return;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int f = f1 + j;
            const bool on = j < n1;
            w1[j] = load_w768(L.in_w + (long)(on ? f : 0) * PD_E, lane);
            b1[j] = L.in_b[(on ? f : 0) + vz];
        }
        if (wave < NB) {
#pragma unroll
            for (int i = 0; i < 12; ++i) g1[i] = L.ln1_w[i * 64 + lane], h1[i] = L.ln1_b[i * 64 + lane];
        }
    };
    auto fetch45 = [&](const GLayer& L) {  // the fc rows of this wave, LN2, the c_proj row
        if (noweights) return;
        // fc features come in adjacent PAIRS (pair p: features 2p, 2p + 1) so that the two bf16-exact activations
        // travel in one tagged word: phase 5 polls half the lines
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int pr = p4 + j, f = j < n4 ? 2 * pr + e : 0;
                w4[2 * j + e] = load_w768(L.fc_w + (long)f * PD_E, lane), b4[2 * j + e] = L.fc_b[f + vz];
            }
        if (wave < 3) {
#pragma unroll
            for (int i = 0; i < 6; ++i) w5[i] = ld_g16(L.pj_w + (long)f0 * PD_F + i * 512 + lane * 8);
            b5 = L.pj_b[f0 + vz];
        }
        if (wave < NB) {
#pragma unroll
            for (int i = 0; i < 12; ++i) g2[i] = L.ln2_w[i * 64 + lane], h2[i] = L.ln2_b[i * 64 + lane];
        }
    };
    fetch1(uniform_layer(lys[0]));
    // this block's attention unit (sequence, head, key range), if it has one
    const bool unit = blk < NB * PD_H * S;
    const int us = blk % S, uh = (blk / S) % PD_H, ub = unit ? blk / (S * PD_H) : 0;
    const int k_lo = (int)((long)us * n / S), nk = unit ? (int)((long)(us + 1) * n / S) - k_lo : 0;
    const int kg = tid >> 3, dc = tid & 7;  // the value pass: thread (kg, dc) owns 8 dims of every 32nd key

    for (int l = 0; l < a.layers; ++l) {
        const GLayer L = uniform_layer(lys[l]);
        const uint32_t tag = seq * 64u + (uint32_t)l * 5u;
        const uint32_t pm = (a.nowait & 1) ? 0u : 0xffffffffu;  // (nowait: the polls compare with tag 0 = accept anything)
        bf16_t* const cache = a.cache + (long)l * NB * a.Lmax * 2 * PD_E;
        // the cached keys / values of the unit do not depend on this step's q: the first 256 keys of the range are requested now and are
        // in registers when q arrives (they used to be requested after it: 2.9 us of attention for 33 keys, tools/decode_persistent_timeline.py)
        const bf16_t* const kv = cache + (long)ub * a.Lmax * 2 * PD_E + uh * 64;
        uint4 kpre[8], vpre[8];
        if (PRE && unit) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // thread (kg, dc): 16 bytes (8 dims) of every 32nd key -- a key's row is read by 8 adjacent lanes
                const int i = kg + 32 * j;
                const bool on = i < nk && k_lo + i != pos;
                const bf16_t* row = kv + (long)(on ? k_lo + i : 0) * 2 * PD_E + dc * 8;
                kpre[j] = on ? *reinterpret_cast<const uint4*>(row) : make_uint4(0u, 0u, 0u, 0u);
                vpre[j] = on ? *reinterpret_cast<const uint4*>(row + PD_E) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        // ================================================================= phase 1: q, k, v
        PD_STAMP(0)
        // Requests for later phases go out right BEFORE a poll: the poll is a memory round trip anyway, the weights arrive under it, and
        // whatever conservative full wait the compiler places afterwards (it cannot count across this control flow) finds nothing pending.
        if (wave < 3 && !noweights) w3 = load_w768(L.out_w + (long)f0 * PD_E, lane), b3 = L.out_b[f0 + vz];  // the out-projection row of this wave
        if (wave < NB) {
            float v[12];
            if (l == 0 && a.tk.table) {  // the embedding row of the token drawn last
                long long id = a.tk.tok[wave];
                if (id < 0 || id >= a.tk.table_rows) id = 0;
                const float* tr_ = a.tk.table + id * PD_E;
                const float* pr_ = a.tk.pos_rows + (long)(pos + a.tk.pos_off) * PD_E;
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = tr_[i * 64 + lane] + pr_[i * 64 + lane];
            } else if (l == 0) {
#pragma unroll
                for (int i = 0; i < 12; ++i) v[i] = a.x_in[(long)wave * PD_E + i * 64 + lane];
            } else {
                poll_words<12>(X + (long)wave * PD_E + lane, 64, tag & pm, v, fail);  // (tag of the previous layer's phase 5 = this layer's base)
            }
            PD_STAMP(1)
            ln_row_to_lds(v, g1, h1, a.eps, xs[wave], xres[0][wave], lane);
        }
        __syncthreads();
        {
            float acc[3][NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                const uint4 xa = *reinterpret_cast<const uint4*>(&xs[b][lane * 8]);
                const uint2 xb = *reinterpret_cast<const uint2*>(&xs[b][512 + lane * 4]);
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[j][b] = dot768(w1[j], xa, xb);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int f = f1 + j;
                if (j >= n1) break;  // wave-uniform
#pragma unroll
                for (int b = 0; b < NBT; ++b) {
                    if (b >= NB) break;
                    const float r = round_bf16_(wave_sum_fast(acc[j][b]) + b1[j]);
                    if (lane == 0) {
                        st_word(QKV + (long)b * 3 * PD_E + f, tag + 1, r);
                        if (f >= PD_E && pos < a.Lmax) cache[((long)b * a.Lmax + pos) * 2 * PD_E + (f - PD_E)] = f2bf(r);
                    }
                }
            }
        }
        PD_STAMP(2)
        __syncthreads();  // xs is rewritten by phase 3
        // ================================================================= phase 2: partial attention
        if (!unit) fetch45(L);  // (an attention unit asks after its attention: q must not queue behind 9 MB of weights)
        if (unit) {
            const bool last = us == S - 1;  // the range that ends with the new position
            if (tid < 64 || (last && tid < 192)) {
                const int col = tid < 64 ? uh * 64 + tid : (tid < 128 ? PD_E + uh * 64 + (tid - 64) : 2 * PD_E + uh * 64 + (tid - 128));
                float v[1];
                poll_words<1>(QKV + (long)ub * 3 * PD_E + col, 0, (tag + 1) & pm, v, fail);
                qkn[tid] = v[0];
            }
        }
        PD_STAMP(3)
        if (unit) {  // (block-uniform)
            __syncthreads();
            // thread (kg, dc) holds dims [8 dc, 8 dc + 8) of q in registers and takes that slice of every 32nd key; a key's score is the sum
            // over its 8 lanes (three DPP steps).  (One key per thread with q read from LDS 64 times: 1.0 us for 33 keys.)
            float qv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) qv[e] = qkn[dc * 8 + e];
            float dnew = 0.f;
            if (us == S - 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) dnew += qv[e] * qkn[64 + dc * 8 + e];
                dnew = group8_sum(dnew);
            }
            PD_STAMP(11)
            float mx = -INFINITY;
            for (int i0 = kg; i0 < nk; i0 += 256) {
                uint4 u[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + 32 * j, key = k_lo + i;
                    if (PRE && i0 == kg)
                        u[j] = kpre[j];
                    else
                        u[j] = (i < nk && key != pos) ? *reinterpret_cast<const uint4*>(kv + (long)key * 2 * PD_E + dc * 8) : make_uint4(0u, 0u, 0u, 0u);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + 32 * j;
                    float d = (bf_lo(u[j].x) * qv[0] + bf_hi(u[j].x) * qv[1]) + (bf_lo(u[j].y) * qv[2] + bf_hi(u[j].y) * qv[3]) +
                              (bf_lo(u[j].z) * qv[4] + bf_hi(u[j].z) * qv[5]) + (bf_lo(u[j].w) * qv[6] + bf_hi(u[j].w) * qv[7]);
                    d = group8_sum(d);
                    if (k_lo + i == pos) d = dnew;
                    d *= a.scale_log2;
                    if (i < nk) {
                        if (dc == 0) sc[i] = d;
                        mx = fmaxf(mx, d);
                    }
                }
            }
            PD_STAMP(12)
            mx = wave_max_fast(mx);
            if (lane == 0) stat[wave] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(stat[0], stat[1]), fmaxf(stat[2], stat[3]));
            float sum = 0.f;
            for (int i = tid; i < nk; i += 256) {
                const float p = __builtin_amdgcn_exp2f(sc[i] - mx);
                sc[i] = p;
                sum += p;
            }
            sum = wave_sum_fast(sum);
            if (lane == 0) stat[4 + wave] = sum;
            __syncthreads();
            sum = (stat[4] + stat[5]) + (stat[6] + stat[7]);
            PD_STAMP(13)
            // o[d] += p[k] V[k][d]: eight 16-byte loads in flight per thread (the first eight are already here)
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int i0 = kg; i0 < nk; i0 += 256) {
                uint4 u[8];
                float p[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = i0 + 32 * j, key = k_lo + i;
                    p[j] = i < nk ? sc[i] : 0.f;
                    if (i < nk && key == pos) {
                        const float* vn = qkn + 128 + dc * 8;
                        u[j] = make_uint4(pack_bf2(vn[0], vn[1]), pack_bf2(vn[2], vn[3]), pack_bf2(vn[4], vn[5]), pack_bf2(vn[6], vn[7]));
                    } else if (PRE && i0 == kg) {
                        u[j] = vpre[j];
                    } else {
                        u[j] = i < nk ? *reinterpret_cast<const uint4*>(kv + (long)key * 2 * PD_E + PD_E + dc * 8) : make_uint4(0u, 0u, 0u, 0u);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[0] += p[j] * bf_lo(u[j].x), acc[1] += p[j] * bf_hi(u[j].x), acc[2] += p[j] * bf_lo(u[j].y), acc[3] += p[j] * bf_hi(u[j].y);
                    acc[4] += p[j] * bf_lo(u[j].z), acc[5] += p[j] * bf_hi(u[j].z), acc[6] += p[j] * bf_lo(u[j].w), acc[7] += p[j] * bf_hi(u[j].w);
                }
            }
            PD_STAMP(14)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = over_groups_sum(acc[e]);  // the wave's 8 key groups
            if (lane < 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) red[wave][dc * 8 + e] = acc[e];
            }
            __syncthreads();
            u64* rec = PART + ((long)(ub * PD_H + uh) * S + us) * PD_REC;
            if (tid < 64) {
                const float o = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
                st_word(rec + tid, tag + 2, o);
            } else if (tid == 64) {
                st_word(rec + 64, tag + 2, mx);
            } else if (tid == 65) {
                st_word(rec + 65, tag + 2, sum);
            }
            PD_STAMP(15)
            fetch45(L);
        }
        // ================================================================= phase 3: merge, out-projection, residual
        PD_STAMP(4)
        if (l + 1 < a.layers) fetch1(uniform_layer(lys[l + 1]));  // the in-projection rows of the next layer, LN1
        {
            // wave w merges heads 3w .. 3w + 2 of every row: lane d takes o[d]; lanes 0 / 1 also take the record's max / sum
            constexpr int RB = PD_H * S * PD_REC;  // words per row of the partial records; 3 * S * NBT = 24 records per wave
            // one poll for both: lane d takes o[d] of each record, and even / odd lanes its max / sum word (used from lanes 0 / 1)
            constexpr int R = NBT * 3 * S;
            float pv[2 * R];
            const u64* base0 = PART + (long)(3 * wave * S) * PD_REC;
            auto at = [&](int i) {
                const int r = i % R;
                return base0 + (long)(r / (3 * S)) * RB + (r % (3 * S)) * PD_REC + (i < R ? lane : 64 + (lane & 1));
            };
            if (NB == NBT) {
                poll_fn<2 * R>(at, (tag + 2) & pm, pv, fail);
            } else {  // (rows >= NB are never written: poll the rows that exist, one by one)
#pragma unroll
                for (int b = 0; b < NBT; ++b) {
                    if (b >= NB) break;
                    float p1[6 * S];
                    auto at1 = [&](int i) { return at((i / (3 * S)) * R + b * 3 * S + i % (3 * S)); };
                    poll_fn<6 * S>(at1, (tag + 2) & pm, p1, fail);
#pragma unroll
                    for (int i = 0; i < 6 * S; ++i) pv[(i / (3 * S)) * R + b * 3 * S + i % (3 * S)] = p1[i];
                }
            }
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                if (b >= NB) break;
#pragma unroll
                for (int hh = 0; hh < 3; ++hh) {
                    float m[S], ls[S], M = -INFINITY;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        m[s] = __shfl(pv[R + (b * 3 + hh) * S + s], 0, 64), ls[s] = __shfl(pv[R + (b * 3 + hh) * S + s], 1, 64);
                        M = fmaxf(M, m[s]);
                    }
                    float num = 0.f, den = 0.f;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float wgt = __builtin_amdgcn_exp2f(m[s] - M);
                        num += wgt * pv[(b * 3 + hh) * S + s], den += wgt * ls[s];
                    }
                    xs[b][(3 * wave + hh) * 64 + lane] = f2bf(num / den);  // the full forward stores the attention output in bf16
                }
            }
        }
        PD_STAMP(5)
        __syncthreads();
        if (wave < 3) {
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                if (b >= NB) break;
                const uint4 xa = *reinterpret_cast<const uint4*>(&xs[b][lane * 8]);
                const uint2 xb = *reinterpret_cast<const uint2*>(&xs[b][512 + lane * 4]);
                const float r = wave_sum_fast(dot768(w3, xa, xb));
                if (lane == 0) st_word(XMID + (long)b * PD_E + f0, tag + 3, r + b3 + xres[0][b][f0]);
            }
        }
        __syncthreads();
        // ================================================================= phase 4: LN2, fc, QuickGELU
        PD_STAMP(6)
        if (wave < NB) {
            float v[12];
            poll_words<12>(XMID + (long)wave * PD_E + lane, 64, (tag + 3) & pm, v, fail);
            PD_STAMP(7)
            ln_row_to_lds(v, g2, h2, a.eps, xs[wave], xres[1][wave], lane);
        }
        __syncthreads();
        {
            float acc[4][NBT];
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                const uint4 xa = *reinterpret_cast<const uint4*>(&xs[b][lane * 8]);
                const uint2 xb = *reinterpret_cast<const uint2*>(&xs[b][512 + lane * 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j][b] = dot768(w4[j], xa, xb);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int pr = p4 + j;
                if (j >= n4) break;  // wave-uniform
#pragma unroll
                for (int b = 0; b < NBT; ++b) {
                    if (b >= NB) break;
                    float r0 = wave_sum_fast(acc[2 * j][b]) + b4[2 * j], r1 = wave_sum_fast(acc[2 * j + 1][b]) + b4[2 * j + 1];
                    // QuickGELU (clip_model.py:196-198), stored in bf16 by the full forward
                    r0 = r0 * sigmoidf_(1.702f * r0), r1 = r1 * sigmoidf_(1.702f * r1);
                    if (lane == 0) st_word(ACT + (long)b * (PD_F / 2) + pr, tag + 4, __uint_as_float(pack_bf2(r0, r1)));
                }
            }
        }
        __syncthreads();
        // ================================================================= phase 5: c_proj, residual
        PD_STAMP(8)
        {
            // wave w stages columns [768 w, 768 (w + 1)) of every row: 384 words of two bf16 each, copied to LDS as they are
            float v[NBT * 6];
            if (NB == NBT) {  // all rows in one poll: word i of row b at b * F / 2 + i * 64
                poll_words2<NBT, 6>(ACT + wave * (PD_E / 2) + lane, PD_F / 2, 64, (tag + 4) & pm, v, fail);
            } else {
#pragma unroll
                for (int b = 0; b < NBT; ++b) {
                    if (b >= NB) break;
                    float vb[6];
                    poll_words<6>(ACT + (long)b * (PD_F / 2) + wave * (PD_E / 2) + lane, 64, (tag + 4) & pm, vb, fail);
#pragma unroll
                    for (int i = 0; i < 6; ++i) v[b * 6 + i] = vb[i];
                }
            }
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                if (b >= NB) break;
#pragma unroll
                for (int i = 0; i < 6; ++i) reinterpret_cast<uint32_t*>(xs[b])[wave * (PD_E / 2) + i * 64 + lane] = __float_as_uint(v[b * 6 + i]);
            }
        }
        PD_STAMP(9)
        __syncthreads();
        if (wave < 3) {
            const bool final_layer = l + 1 == a.layers;
#pragma unroll
            for (int b = 0; b < NBT; ++b) {
                if (b >= NB) break;
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const uint4 x4 = *reinterpret_cast<const uint4*>(&xs[b][i * 512 + lane * 8]);
                    s = dot2(w5[i].x, x4.x, s), s = dot2(w5[i].y, x4.y, s), s = dot2(w5[i].z, x4.z, s), s = dot2(w5[i].w, x4.w, s);
                }
                const float r = wave_sum_fast(s);
                if (lane == 0) {
                    const float out = r + b5 + xres[1][b][f0];
                    if (final_layer && a.x_out) a.x_out[(long)b * PD_E + f0] = out;
                    if (!final_layer || a.tk.table) st_word(X + (long)b * PD_E + f0, tag + 5, out);  // = the next layer's (the head's) base tag
                }
            }
        }
        PD_STAMP(10)
        __syncthreads();
    }
#undef PD_STAMP
    if (a.tk.table) {
        // classes; head features per block (contiguous: ceil(V / 256)) and per wave (ceil of a quarter of those: 1 or 2) -- V = 1,024 and
        // 2,048 (the models' image vocabularies) fill every wave; smaller vocabularies (the tiny test models) leave waves / blocks idle
        const int V = a.tk.V, FPB = (V + 255) >> 8, NFW = (FPB + 3) >> 2;
        const uint32_t tagH = seq * 64u + (uint32_t)a.layers * 5u;
        const uint32_t pmh = (a.nowait & 1) ? 0u : 0xffffffffu;
        if (blk == 0 && tid < NB && a.tk.record && pos >= a.tk.record_pos0 && pos - a.tk.record_pos0 < a.tk.record_ld)
            a.tk.record[tid * a.tk.record_ld + (pos - a.tk.record_pos0)] = a.tk.tok[tid];  // (the token this step embedded)
        // ---- head: LN_f(x) . W_head^T + b over the image block of the vocabulary
        W768 wh[2];
        float bh[2] = {0.f, 0.f};
        const int fh0 = blk * FPB + wave * NFW;
        bool fon[2];  // (wave-uniform) this wave computes feature fh0 + j
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fon[j] = j < NFW && wave * NFW + j < FPB && fh0 + j < V;
            const int f = fon[j] ? fh0 + j : 0;
            wh[j] = load_w768((gbf_p)a.tk.head_w + (long)f * PD_E, lane), bh[j] = a.tk.head_b[f + vz];
        }
        // the race variates of this block's row (a draw block only): requested before anything is waited for
        float ev[8];
        if (blk < NB) {
            const float* e = a.tk.E + (long)(pos + 1 - a.tk.e_pos0) * a.tk.e_step_stride + (long)blk * V;
#pragma unroll
            for (int i = 0; i < 8; ++i) ev[i] = i * 256 + tid < V ? e[i * 256 + tid] : 0.f;
        }
        if (wave < NB) {
            float gf[12], hf[12], v[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) gf[i] = a.tk.lnf_w[i * 64 + lane], hf[i] = a.tk.lnf_b[i * 64 + lane];
            poll_words<12>(X + (long)wave * PD_E + lane, 64, tagH & pmh, v, fail);
            ln_row_to_lds(v, gf, hf, a.tk.lnf_eps, xs[wave], nullptr, lane);
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NBT; ++b) {
            if (b >= NB) break;
            const uint4 xa = *reinterpret_cast<const uint4*>(&xs[b][lane * 8]);
            const uint2 xb = *reinterpret_cast<const uint2*>(&xs[b][512 + lane * 4]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!fon[j]) continue;
                const float r = wave_sum_fast(dot768(wh[j], xa, xb)) + bh[j];
                if (lane == 0) {
                    st_word(LOGITS + (long)b * PD_MAXV + fh0 + j, tagH + 1, r);
                    if (a.tk.logits_out) a.tk.logits_out[(long)b * V + fh0 + j] = r;
                }
            }
        }
        // ---- draw: block b < B takes row b -- first argmin of E_c / expf(x_c / T - max), the rule of csrc/sample.hip
        if (blk < NB) {
            float xv[8];
            poll_fn<8>([&](int i) { return LOGITS + (long)blk * PD_MAXV + (i * 256 + tid < V ? i * 256 + tid : 0); }, (tagH + 1) & pmh, xv, fail);  // (word 0 always arrives)
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                xv[i] = i * 256 + tid < V ? xv[i] * a.tk.inv_temp : -INFINITY;
                mx = fmaxf(mx, xv[i]);
            }
            mx = wave_max_fast(mx);
            __syncthreads();  // (stat: the last attention unit of this block is long done)
            if (lane == 0) stat[wave] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(stat[0], stat[1]), fmaxf(stat[2], stat[3]));
            float bkey = INFINITY;
            int bidx = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i * 256 + tid >= V) continue;
                const float p = expf(xv[i] - mx);
                const float key = p > 0.f ? ev[i] / p : INFINITY;
                if (key < bkey || (key == bkey && i * 256 + tid < bidx)) bkey = key, bidx = i * 256 + tid;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ok = __shfl_xor(bkey, o, 64);
                const int oi = __shfl_xor(bidx, o, 64);
                if (ok < bkey || (ok == bkey && oi < bidx)) bkey = ok, bidx = oi;
            }
            __syncthreads();
            if (lane == 0) stat[wave] = bkey, stat[4 + wave] = __int_as_float(bidx);
            __syncthreads();
            if (tid == 0) {
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float ok = stat[w];
                    const int oi = __float_as_int(stat[4 + w]);
                    if (ok < bkey || (ok == bkey && oi < bidx)) bkey = ok, bidx = oi;
                }
                a.tk.tok[blk] = (long long)(bidx < V ? bidx : 0) + a.tk.tok_offset;
            }
        }
    }
    // every block has read the counter before it wrote anything this block waited for: block 0 may advance it now
    if (blk == 0 && tid == 0) {
        a.ws[0] = seq;
        if (a.pos_advance) *a.pos_advance = pos + 1;  // (every block read the position when it started)
    }
}

int64_t workspace_words(int nbt) {
    const int S = PD_S;
    return 8 + (int64_t)nbt * (PD_E + 3 * PD_E + PD_H * S * PD_REC + PD_E + PD_F + PD_MAXV);
}
int template_batch(int B) { return B <= 1 ? 1 : 2; }

}  // namespace

// measurement only: dev_buf = u64 [4][12][16] that the next steps stamp (tools/decode_persistent_timeline.py), or null
extern "C" int mmvid_decode_persistent_trace(void* dev_buf) {
    g_pd_trace = (u64*)dev_buf;
    return MMVID_OK;
}

// the launch shared by the two entry points
static int pd_launch(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, PdArgs& a, void* kv_cache, int Lmax, int32_t* pos_dev,
                     int pos, int advance_pos, void* workspace, void* stream) {
    for (int i = 0; i < cfg->layers; ++i) {
        const mmvid_tower_layer_t& s = layers[i];
        PdLayer& d = a.ly[i];
        d.in_w = (const bf16_t*)s.in_w, d.out_w = (const bf16_t*)s.out_w, d.fc_w = (const bf16_t*)s.fc_w, d.pj_w = (const bf16_t*)s.pj_w;
        d.in_b = s.in_b, d.out_b = s.out_b, d.fc_b = s.fc_b, d.pj_b = s.pj_b;
        d.ln1_w = s.ln1_w, d.ln1_b = s.ln1_b, d.ln2_w = s.ln2_w, d.ln2_b = s.ln2_b;
    }
    for (int i = cfg->layers; i < PD_LAYERS; ++i) a.ly[i] = a.ly[0];
    a.cache = (bf16_t*)kv_cache, a.ws = (u64*)workspace, a.pos_dev = pos_dev, a.pos0 = pos;
    a.pos_advance = (advance_pos && pos_dev) ? pos_dev : nullptr;
    a.layers = cfg->layers, a.Lmax = Lmax, a.NB = cfg->B, a.eps = cfg->ln_eps, a.scale_log2 = 0.125f * 1.4426950408889634f;
    a.trace = g_pd_trace;
    static const int nowait = getenv("MMVID_PD_NOWAIT") ? atoi(getenv("MMVID_PD_NOWAIT")) : 0;
    a.nowait = nowait;
    hipStream_t s = (hipStream_t)stream;
    switch (template_batch(cfg->B)) {
        case 1: hipLaunchKernelGGL(decode_persistent_kernel<1>, dim3(PD_BLOCKS), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(decode_persistent_kernel<2>, dim3(PD_BLOCKS), dim3(256), 0, s, a); break;
    }
    return MMVID_OK;
}

// 1 when mmvid_tower_decode_persistent takes this tower / cache (the caller uses mmvid_tower_decode_fused otherwise)
extern "C" int mmvid_tower_decode_persistent_supported(const mmvid_tower_cfg_t* cfg, int Lmax) {
    if (!cfg || cfg->mask_mode != 1 || cfg->E != PD_E || cfg->F != PD_F || cfg->H != PD_H || cfg->layers < 1 || cfg->layers > PD_LAYERS)
        return 0;
    if (cfg->B < 1 || cfg->B > PD_MAXB) return 0;
    const int S = PD_S;
    if (Lmax < 1 || (Lmax + S - 1) / S > PD_MAXKEYS) return 0;
    // the 256 blocks must be resident together: one block per CU at the kernel's own register / LDS footprint, on THIS device
    static int slots[16] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (slots[dev] < 0) {
        hipDeviceProp_t prop;
        int p1 = 0, p2 = 0;
        const bool ok = hipGetDeviceProperties(&prop, dev) == hipSuccess &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&p1, (const void*)decode_persistent_kernel<1>, 256, 0) == hipSuccess &&
                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&p2, (const void*)decode_persistent_kernel<2>, 256, 0) == hipSuccess;
        slots[dev] = ok ? (p1 < p2 ? p1 : p2) * prop.multiProcessorCount : 0;
    }
    return slots[dev] >= PD_BLOCKS ? 1 : 0;
}

// bytes of the workspace; it must be ZERO before its first use and is owned by the decode session from then on
extern "C" int64_t mmvid_tower_decode_persistent_workspace_bytes(int B) {
    return B >= 1 && B <= PD_MAXB ? workspace_words(template_batch(B)) * 8 : 0;
}

extern "C" int mmvid_tower_decode_persistent(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                                             float* x_out, void* kv_cache, int Lmax, int32_t* pos_dev, int pos, int advance_pos,
                                             void* workspace, void* stream) {
    MMVID_REQUIRE(cfg && layers && x_in && x_out && kv_cache && workspace, "tower_decode_persistent: null pointer");
    MMVID_REQUIRE(mmvid_tower_decode_persistent_supported(cfg, Lmax),
                  "tower_decode_persistent: needs the causal 768 / 3072 / 12-head tower, <= 12 layers, batch <= 2, a device with >= 256 CUs");
    PdArgs a;
    a.tk = {};
    a.x_in = x_in, a.x_out = x_out;
    pd_launch(cfg, layers, a, kv_cache, Lmax, pos_dev, pos, advance_pos, workspace, stream);
    MMVID_LAUNCH_CHECK("tower_decode_persistent");
    return MMVID_OK;
}

// The ART-V sampler's whole token as one launch (dalle_artv.py:252-293 over the key/value cache): embedding row of the token drawn last
// -> the tower step above -> LN + the image block of to_logits -> the draw of the next token (exponential race on pre-drawn variates) ->
// *pos_dev += 1.  t->tok is read at the start and overwritten at the end.  Needs V = 1,024 or 2,048 and what _supported needs.
extern "C" int mmvid_artv_token_step_persistent(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const mmvid_decode_token_t* t,
                                                float* x_out, void* kv_cache, int Lmax, int32_t* pos_dev, void* workspace, void* stream) {
    MMVID_REQUIRE(cfg && layers && t && kv_cache && workspace && pos_dev, "artv_token_step_persistent: null pointer");
    MMVID_REQUIRE(mmvid_tower_decode_persistent_supported(cfg, Lmax),
                  "artv_token_step_persistent: needs the causal 768 / 3072 / 12-head tower, <= 12 layers, batch <= 2, a device with >= 256 CUs");
    MMVID_REQUIRE(t->tok && t->table && t->pos_rows && t->lnf_w && t->lnf_b && t->head_w && t->head_b && t->E, "artv_token_step_persistent: null pointer in the token block");
    MMVID_REQUIRE(t->V >= 1 && t->V <= 2048 && t->temperature > 0.f, "artv_token_step_persistent: V = %d (1 .. 2048), temperature > 0", t->V);
    PdArgs a;
    a.x_in = nullptr, a.x_out = x_out;
    a.tk.tok = (long long*)t->tok, a.tk.table = t->table, a.tk.table_rows = (long)t->table_rows, a.tk.pos_rows = t->pos_rows, a.tk.pos_off = t->pos_off;
    a.tk.record = (long long*)t->record, a.tk.record_ld = (long)t->record_ld, a.tk.record_pos0 = t->record_pos0;
    a.tk.lnf_w = t->lnf_w, a.tk.lnf_b = t->lnf_b, a.tk.lnf_eps = t->lnf_eps, a.tk.head_w = (const bf16_t*)t->head_w, a.tk.head_b = t->head_b, a.tk.V = t->V;
    a.tk.E = t->E, a.tk.e_step_stride = (long)t->e_step_stride, a.tk.e_pos0 = t->e_pos0, a.tk.inv_temp = 1.0f / t->temperature;
    a.tk.tok_offset = (long long)t->tok_offset, a.tk.logits_out = t->logits_out;
    pd_launch(cfg, layers, a, kv_cache, Lmax, pos_dev, 0, 1, workspace, stream);
    MMVID_LAUNCH_CHECK("artv_token_step_persistent");
    return MMVID_OK;
}
