// Incremental (KV-cache) decoding for the causal tower: SURVEY next-row N1.  The reference's ART-V sampler
// (dalle_artv.py:236-304) re-runs the whole transformer over the growing prefix for each of its 1,024 tokens; with
// a causal mask the keys and values of earlier positions never change, so they are kept per layer and a step only
// computes the new position.  Same distribution as the full recomputation (bf16 rounding order aside).
//   cache layout: [layers][B][Lmax][2E] bf16, a row = K (E values) followed by V (E values) of one position.
// Both kernels are tiny and HBM/latency-bound (one query per (batch, head)); the step cost is the weight traffic of
// the twelve layers' GEMMs at M = B.
#include "../../include/mmvid_hip.h"
#include "common.h"
#include "wave_reduce.h"
#include <type_traits>

namespace {

// qkv rows [B*L, ldq] (K at column E, V at 2E) -> cache[b][p0 + l][0..2E).  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void kv_store_kernel(const bf16_t* __restrict__ qkv, long ldq, int B, int L, int E,
                                                       const int* __restrict__ pos_dev, int p0, int Lmax,
                                                       bf16_t* __restrict__ cache) {
    const int chunks = (2 * E) >> 3;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * L * chunks) return;
    const int c = (int)(i % chunks);
    const long row = i / chunks;
    const int b = (int)(row / L), l = (int)(row - (long)b * L);
    const int p = (pos_dev ? *pos_dev : p0) + l;
    if (p >= Lmax) return;
    *reinterpret_cast<uint4*>(cache + ((long)b * Lmax + p) * 2 * E + c * 8) =
        *reinterpret_cast<const uint4*>(qkv + row * ldq + E + c * 8);
}

// One block per (head, batch): softmax(q . K[0..t] * scale) V[0..t], head_dim 64.  Scores live in LDS (t < 4096).
constexpr int DEC_MAXL = 4096;
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ qkv, long ldq,
                                                          const bf16_t* __restrict__ cache, int Lmax, int E,
                                                          const int* __restrict__ pos_dev, int pos0, float scale_log2,
                                                          bf16_t* __restrict__ out, long ldo) {
    __shared__ float sc[DEC_MAXL];
    __shared__ float qs[64];
    __shared__ float red[4][64];
    __shared__ float stat[8];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (pos_dev ? *pos_dev : pos0) + 1;  // keys 0 .. t
    if (tid < 64) qs[tid] = bf2f(qkv[(long)b * ldq + h * 64 + tid]);
    __syncthreads();
    const bf16_t* kv = cache + (long)b * Lmax * 2 * E + h * 64;
    float mx = -INFINITY;
    for (int k = tid; k < n; k += 256) {
        const uint4* kr = reinterpret_cast<const uint4*>(kv + (long)k * 2 * E);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 u = kr[c];
            const float* q = qs + 8 * c;
            d += (bf_lo(u.x) * q[0] + bf_hi(u.x) * q[1]) + (bf_lo(u.y) * q[2] + bf_hi(u.y) * q[3]) +
                 (bf_lo(u.z) * q[4] + bf_hi(u.z) * q[5]) + (bf_lo(u.w) * q[6] + bf_hi(u.w) * q[7]);
        }
        d *= scale_log2;
        sc[k] = d;
        mx = fmaxf(mx, d);
    }
    mx = wave_max_fast(mx);
    if (lane == 0) stat[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(stat[0], stat[1]), fmaxf(stat[2], stat[3]));
    float sum = 0.f;
    for (int k = tid; k < n; k += 256) {
        const float p = __builtin_amdgcn_exp2f(sc[k] - mx);
        sc[k] = p;
        sum += p;
    }
    sum = wave_sum_fast(sum);
    if (lane == 0) stat[4 + wave] = sum;
    __syncthreads();
    sum = (stat[4] + stat[5]) + (stat[6] + stat[7]);
    // out[d] = sum_k p[k] V[k][d]: lane = d, the four waves take keys k = wave, wave + 4, ...
    const bf16_t* vv = kv + E + lane;
    float acc = 0.f;
    for (int k = wave; k < n; k += 4) acc += sc[k] * bf2f(vv[(long)k * 2 * E]);
    red[wave][lane] = acc;
    __syncthreads();
    if (tid < 64) out[(long)b * ldo + h * 64 + tid] = f2bf(((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) / sum);
}

}  // namespace

extern "C" int mmvid_kv_store(const void* qkv, int64_t ldq, int B, int L, int E, const int32_t* pos_dev, int pos0,
                              int Lmax, void* cache, void* stream) {
    MMVID_REQUIRE(qkv && cache && B > 0 && L > 0 && E % 8 == 0 && ldq % 8 == 0, "kv_store: bad arguments");
    MMVID_REQUIRE(pos_dev || (pos0 >= 0 && pos0 + L <= Lmax), "kv_store: positions %d..%d exceed the cache (%d)", pos0,
                  pos0 + L - 1, Lmax);
    const long n = (long)B * L * ((2 * E) >> 3);
    hipLaunchKernelGGL(kv_store_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ldq,
                       B, L, E, pos_dev, pos0, Lmax, (bf16_t*)cache);
    MMVID_LAUNCH_CHECK("kv_store");
    return MMVID_OK;
}

extern "C" int mmvid_attention_decode(const void* qkv, int64_t ldq, const void* cache, int B, int Lmax, int H, int E,
                                      const int32_t* pos_dev, int pos0, float scale, void* out, int64_t ldo, void* stream) {
    MMVID_REQUIRE(qkv && cache && out && B > 0 && H > 0 && E == H * 64, "attention_decode: need E == 64*H");
    MMVID_REQUIRE(Lmax <= DEC_MAXL && (pos_dev || (pos0 >= 0 && pos0 < Lmax)), "attention_decode: position / Lmax (<= %d)",
                  DEC_MAXL);
    hipLaunchKernelGGL(attn_decode_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ldq,
                       (const bf16_t*)cache, Lmax, E, pos_dev, pos0, scale * 1.4426950408889634f, (bf16_t*)out, (long)ldo);
    MMVID_LAUNCH_CHECK("attention_decode");
    return MMVID_OK;
}

// =====================================================================================================================
// Fused decode step (round 2).  The first version of mmvid_tower_decode reused the training GEMM (128x128 MFMA tiles) at
// M = B <= 8: ~13 launches per layer, each a handful of blocks -- 1.5 ms per token, 50x above the weight-streaming floor
// (12 layers x 14.2 MB of bf16 weights / 6.3 TB/s = 27 us).  A decode step is a chain of matrix-VECTOR products: every
// weight byte is used once per sequence of the batch, so the right kernel streams weight rows straight into registers
// (16 B per lane, several rows in flight), keeps the B input rows in LDS, and spreads the N output features over all
// 256 CUs.  Five launches per layer:
//   gemv<LN>   q,k,v = LN1(x) W_in^T + b          (+ K|V appended to the cache, q kept for the attention)
//   attn       o = softmax(q K^T / 8) V over the cached positions
//   gemv       x_mid = x + o W_out^T + b
//   gemv<LN>   a = QuickGELU(LN2(x_mid) W_fc^T + b)
//   gemv       x' = x_mid + a W_proj^T + b
// Operands are rounded to bf16 exactly where the full forward stores them in bf16 (LayerNorm output, q, o, the GELU
// output), so cached and recomputed logits differ only by fp32 summation order.  HBM/latency-bound.
namespace {

constexpr int GV_MAXB = 16;   // sequences per decode batch (9-16: the WIDE instance, rows staged as bf16 only)
constexpr int GV_COLS = 8;    // output features per block (2 per wave)
__device__ __forceinline__ float round_bf16(float v) { return bf2f(f2bf(v)); }

struct GemvArgs {
    const float* x;      // [NB][ldx] fp32
    long ldx;
    const float *ln_w, *ln_b;  // LayerNorm applied to the rows first, or null
    float eps;
    const bf16_t* W;     // [N][K] row-major
    const float* bias;   // [N] or null
    const float* residual;  // [NB][ldr] or null
    long ldr;
    float* out;          // [NB][ldo] fp32
    long ldo;
    int NB, N, K, act, round_in, round_out;
    // K|V append: features n in [kv_lo, kv_lo + 2E) are also written as bf16 to cache[b][pos][n - kv_lo]
    bf16_t* kv_cache;
    int kv_lo, kv_width, Lmax;
    const int* pos_dev;
    int pos0;
};

// KIT = ceil(K / 512): 16-byte weight loads per lane and output feature.  The kernel is latency-bound (a decode step is a
// chain of ~60 of these), so the two global round trips it needs are overlapped: every weight load of the wave is issued
// FIRST, into registers, and the input rows are fetched / normalised / staged in LDS while those are in flight.
unsigned long long* g_decode_trace = nullptr;  // measurement only (mmvid_decode_trace): [blocks][8] wall-clock stamps of the next gemv

// BF: the staged rows are bf16-exact (round_in) -> LDS holds them as bf16 (half the LDS traffic, 16-byte conflict-free reads) and the
// products go through v_dot2c_f32_bf16 (two multiply-adds per instruction, fp32 accumulate)
template <int KIT, bool TRACE = false, int FPW = 2, bool BF = false, bool WIDE = false>  // FPW: output features per wave (1 for the narrow outputs: twice the blocks); WIDE: 9-16 rows
__global__ __launch_bounds__(256) void gemv_rows_kernel(GemvArgs a, unsigned long long* trace = nullptr) {
#define GV_STAMP(i) \
    if constexpr (TRACE)  \
        if (threadIdx.x == 0 && blockIdx.x < 512) trace[blockIdx.x * 8 + (i)] = wall_clock64();
    GV_STAMP(0)
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [NB][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, NB = a.NB;
    const int n0 = blockIdx.x * (4 * FPW) + wave * FPW;
    const bool have = n0 < a.N, two = FPW == 2 && n0 + 1 < a.N;
    // ---- 1. weights of this wave's two output features -> registers
    uint4 w[2][KIT];
    {
        const bf16_t* w0 = a.W + (long)(have ? n0 : 0) * K;
        const bf16_t* w1 = a.W + (long)(two ? n0 + 1 : (have ? n0 : 0)) * K;
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int k0 = (it * 64 + lane) * 8;
            w[0][it] = k0 < K ? *reinterpret_cast<const uint4*>(w0 + k0) : make_uint4(0u, 0u, 0u, 0u);
            w[1][it] = (FPW == 2 && k0 < K) ? *reinterpret_cast<const uint4*>(w1 + k0) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // epilogue operands: value (feature c, row b) of the reduce-scatter below ends in the lanes [j << esh, (j + 1) << esh) with
    // j = c * NBP + b (NBP = NB rounded up to a power of two); the first lane of each group finishes that output.  Requested now,
    // consumed at the very end.
    const int elog = NB <= 1 ? 0 : (NB <= 2 ? 1 : (NB <= 4 ? 2 : 3)), esh = (FPW == 2 ? 5 : 6) - elog;
    const int ec = FPW == 2 ? lane >> 5 : 0, eb = (lane >> esh) & ((1 << elog) - 1);
    const bool elane = (lane & ((1 << esh) - 1)) == 0 && eb < NB && n0 + ec < a.N;
    const int en = n0 + ec;
    const float ebias = (elane && a.bias) ? a.bias[en] : 0.f;
    const float eres = (elane && a.residual) ? a.residual[eb * a.ldr + en] : 0.f;
    const int epos = a.kv_cache ? (a.pos_dev ? *a.pos_dev : a.pos0) : 0;
    // WIDE: one reduce-scatter of 16 rows per feature -- row b of a feature ends in lanes [4 b, 4 b + 4)
    const int wb = lane >> 2;
    const bool wlane = WIDE && (lane & 3) == 0 && wb < NB;
    float wbias[2] = {0.f, 0.f}, wres[2] = {0.f, 0.f};
    if constexpr (WIDE) {
#pragma unroll
        for (int c = 0; c < FPW; ++c) {
            const bool on = wlane && n0 + c < a.N;
            wbias[c] = (on && a.bias) ? a.bias[n0 + c] : 0.f;
            wres[c] = (on && a.residual) ? a.residual[wb * a.ldr + n0 + c] : 0.f;
        }
    }
    GV_STAMP(1)
    // ---- 2. input rows: wave w owns rows w and w + 4 (NB <= 8) -- a row is K/256 coalesced 16-byte loads per lane, its LayerNorm
    // statistics are two in-wave reductions (no cross-wave step, no barrier), and the normalised row goes to LDS.  (The first form
    // spread the rows' float4 over all 256 threads: per-row partial sums through eight-way selects, a cross-wave LDS reduction and
    // four barriers -- 3.8 us of an 11-us kernel at batch 4, tools/decode_gemv_timeline.py.)
    constexpr int XR = 2 * KIT;  // float4 per lane and row: K <= 512 * KIT
    constexpr int RW = WIDE ? 4 : 2;  // rows per wave (WIDE: one wave per SIMD, the register file is this block's)
    float4 xr[RW][XR];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int r = wave + 4 * rr;
#pragma unroll
        for (int jj = 0; jj < XR; ++jj) {
            const int k = (jj * 64 + lane) * 4;
            xr[rr][jj] = (r < NB && k < K) ? *reinterpret_cast<const float4*>(a.x + (long)r * a.ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float4 lg[XR], lb[XR];  // LayerNorm gain / bias of this lane's columns (requested with the rows)
#pragma unroll
    for (int jj = 0; jj < XR; ++jj) {
        const int k = (jj * 64 + lane) * 4;
        const bool on = a.ln_w && k < K;
        lg[jj] = on ? *reinterpret_cast<const float4*>(a.ln_w + k) : make_float4(1.f, 1.f, 1.f, 1.f);
        lb[jj] = on ? *reinterpret_cast<const float4*>(a.ln_b + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (TRACE) {  // (the stamp is taken when the input rows have ARRIVED: it consumes one of them)
        float probe = xr[0][0].x;
        asm volatile("" : "+v"(probe));
    }
    GV_STAMP(2)
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        const int r = wave + 4 * rr;
        if (r >= NB) break;  // wave-uniform
        float mu = 0.f, rs = 1.f;
        if (a.ln_w) {
            float t = 0.f;
#pragma unroll
            for (int jj = 0; jj < XR; ++jj) t += (xr[rr][jj].x + xr[rr][jj].y) + (xr[rr][jj].z + xr[rr][jj].w);  // (columns >= K hold 0)
            mu = wave_sum_fast(t) / (float)K;
            float q = 0.f;
#pragma unroll
            for (int jj = 0; jj < XR; ++jj) {
                if ((jj * 64 + lane) * 4 >= K) continue;
                const float d0 = xr[rr][jj].x - mu, d1 = xr[rr][jj].y - mu, d2 = xr[rr][jj].z - mu, d3 = xr[rr][jj].w - mu;
                q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            rs = rsqrtf(wave_sum_fast(q) / (float)K + a.eps);
        }
        if (rr == 0) { GV_STAMP(3) }
#pragma unroll
        for (int jj = 0; jj < XR; ++jj) {
            const int k = (jj * 64 + lane) * 4;
            if (k >= K) continue;
            float v[4] = {xr[rr][jj].x, xr[rr][jj].y, xr[rr][jj].z, xr[rr][jj].w};
            if (a.ln_w) {
                v[0] = (v[0] - mu) * rs * lg[jj].x + lb[jj].x, v[1] = (v[1] - mu) * rs * lg[jj].y + lb[jj].y;
                v[2] = (v[2] - mu) * rs * lg[jj].z + lb[jj].z, v[3] = (v[3] - mu) * rs * lg[jj].w + lb[jj].w;
            }
            if (a.round_in) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = round_bf16(v[e]);
            }
            if constexpr (BF)
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(xs) + (long)r * K + k) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            else
                *reinterpret_cast<float4*>(xs + (long)r * K + k) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    __syncthreads();
    GV_STAMP(4)
    if (!have) return;
    if constexpr (TRACE) {  // the weights have arrived
        asm volatile("" : "+v"(w[0][0].x), "+v"(w[0][KIT - 1].x));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    GV_STAMP(5)
    // ---- 3. dot products: weights from registers, rows from LDS
    constexpr int MAXB = WIDE ? 16 : 8;
    float acc[2][MAXB];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int b = 0; b < MAXB; ++b) acc[c][b] = 0.f;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
        const int k0 = (it * 64 + lane) * 8;
        if (k0 >= K) continue;
        const uint4 u0 = w[0][it], u1 = w[1][it];
        const float f0[8] = {bf_lo(u0.x), bf_hi(u0.x), bf_lo(u0.y), bf_hi(u0.y), bf_lo(u0.z), bf_hi(u0.z), bf_lo(u0.w), bf_hi(u0.w)};
        const float f1[8] = {bf_lo(u1.x), bf_hi(u1.x), bf_lo(u1.y), bf_hi(u1.y), bf_lo(u1.z), bf_hi(u1.z), bf_lo(u1.w), bf_hi(u1.w)};
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if constexpr (BF) {
                if (b < NB) {
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
                    const uint4 x8 = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(xs) + b * K + k0);
                    const uint32_t xw[4] = {x8.x, x8.y, x8.z, x8.w}, w0[4] = {u0.x, u0.y, u0.z, u0.w}, w1[4] = {u1.x, u1.y, u1.z, u1.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0][b] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w0[e]), __builtin_bit_cast(bf2_t, xw[e]), acc[0][b], false);
                        if constexpr (FPW == 2)
                            acc[1][b] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, w1[e]), __builtin_bit_cast(bf2_t, xw[e]), acc[1][b], false);
                    }
                }
                continue;
            }
            if (b < NB) {
                const float4 xa = *reinterpret_cast<const float4*>(xs + b * K + k0);
                const float4 xb = *reinterpret_cast<const float4*>(xs + b * K + k0 + 4);
                acc[0][b] += (f0[0] * xa.x + f0[1] * xa.y) + (f0[2] * xa.z + f0[3] * xa.w) + (f0[4] * xb.x + f0[5] * xb.y) +
                             (f0[6] * xb.z + f0[7] * xb.w);
                if constexpr (FPW == 2)
                    acc[1][b] += (f1[0] * xa.x + f1[1] * xa.y) + (f1[2] * xa.z + f1[3] * xa.w) + (f1[4] * xb.x + f1[5] * xb.y) +
                                 (f1[6] * xb.z + f1[7] * xb.w);
            }
        }
    }
    if constexpr (WIDE) {
#pragma unroll
        for (int c = 0; c < FPW; ++c) {
            float l16[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) l16[j] = acc[c][j];
            float v = reduce_scatter<4>(l16, lane);
            const int en = n0 + c;
            if (wlane && en < a.N) {
                v += wbias[c];
                if (a.act == 1) v = v * sigmoidf_(1.702f * v);  // QuickGELU (clip_model.py:196-198)
                v += wres[c];
                if (a.round_out) v = round_bf16(v);
                a.out[wb * a.ldo + en] = v;
                if (a.kv_cache && en >= a.kv_lo && en < a.kv_lo + a.kv_width && epos < a.Lmax)
                    a.kv_cache[((long)wb * a.Lmax + epos) * a.kv_width + (en - a.kv_lo)] = f2bf(v);
            }
        }
        GV_STAMP(6)
        return;
    }
    // 2 * NBP partial sums per lane -> the complete sum (c, b) in lane group c * NBP + b (see reduce_scatter above)
    float v;
    if constexpr (FPW == 2) {
        if (elog == 0) {
            float l2[2] = {acc[0][0], acc[1][0]};
            v = reduce_scatter<1>(l2, lane);
        } else if (elog == 1) {
            float l4[4] = {acc[0][0], acc[0][1], acc[1][0], acc[1][1]};
            v = reduce_scatter<2>(l4, lane);
        } else if (elog == 2) {
            float l8[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
            v = reduce_scatter<3>(l8, lane);
        } else {
            float l16[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) l16[j] = acc[j >> 3][j & 7];
            v = reduce_scatter<4>(l16, lane);
        }
    } else {
        if (elog == 0) {
            float l1[1] = {acc[0][0]};
            v = reduce_scatter<0>(l1, lane);
        } else if (elog == 1) {
            float l2[2] = {acc[0][0], acc[0][1]};
            v = reduce_scatter<1>(l2, lane);
        } else if (elog == 2) {
            float l4[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
            v = reduce_scatter<2>(l4, lane);
        } else {
            float l8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) l8[j] = acc[0][j];
            v = reduce_scatter<3>(l8, lane);
        }
    }
    if (elane) {
        v += ebias;
        if (a.act == 1) v = v * sigmoidf_(1.702f * v);  // QuickGELU (clip_model.py:196-198)
        v += eres;
        if (a.round_out) v = round_bf16(v);
        a.out[eb * a.ldo + en] = v;
        if (a.kv_cache && en >= a.kv_lo && en < a.kv_lo + a.kv_width && epos < a.Lmax)
            a.kv_cache[((long)eb * a.Lmax + epos) * a.kv_width + (en - a.kv_lo)] = f2bf(v);
    }
    GV_STAMP(6)
#undef GV_STAMP
}


// ---- Round 6: the same linear layer for 3..16 rows on the matrix pipe.  gemv_rows_kernel multiplies on the vector
// ALU: at 16 rows every wave reads the 16 staged rows from LDS for each pair of output features and ends in a 16-value reduce-scatter --
// 10.5 / 14.2 us per launch at batch 16 for a few MB of weights.  16 rows are exactly one v_mfma_f32_16x16x32_bf16 row block:
//   block = 16 output features x 16 rows, 8 waves, wave w owns the K slice [w K/8, (w+1) K/8): S = K/256 MFMAs per wave;
//   B operand = weights straight from HBM into registers (lane (n, g) holds chunk 4 s + g of feature row n's slice for MFMA step s: the
//   four lanes of a row read 64 contiguous bytes -- a first form gave every lane S CONSECUTIVE chunks, 64 separate lines per load
//   instruction: the K = 3,072 layer then spent its time in the texture addresser, not in HBM);
//   A operand = the rows: LayerNorm'd fp32 rows staged once per block in LDS as bf16 (in_proj, c_fc), or bf16 rows read as fragments
//   straight from global memory (out_proj reads the attention's bf16 output, c_proj the bf16 activation: no LDS staging at all);
//   the eight K-slice partial tiles meet in 8 KiB of LDS, 256 threads finish one output each (bias, QuickGELU, residual, K|V append).
// Same rounding points as gemv_rows_kernel (rows bf16-exact, fp32 accumulate); the summation order differs.
struct Gemv16Args {
    const float* x;      // fp32 rows [NB][ldx] (LayerNorm path) ...
    const bf16_t* xb;    // ... or bf16 rows [NB][ldx] (direct path)
    long ldx;
    const float *ln_w, *ln_b;
    float eps;
    const bf16_t* W;     // [N][K]
    const float* bias;
    const float* residual;
    long ldr;
    float* out;          // fp32 [NB][ldo] or null
    bf16_t* out_bf;      // bf16 [NB][ldo] or null (the value rounded once: what the next layer's operand is anyway)
    long ldo;
    int NB, N, K, act, round_out;
    bf16_t* kv_cache;
    int kv_lo, kv_width, Lmax;
    const int* pos_dev;
    int pos0;
    int* pos_inc;        // not null: *pos_inc += 1 when the launch is done with it (the step's last layer: a launch less per token)
};

// HALF: a block = 8 features x 8 rows (grid (N/8, 2)): the tile of the narrow, long-K layer (c_proj: N = 768, K = 3,072).  With 16 x 16 tiles
// that layer is 48 blocks which each pull 96 KiB of weights + the 96 KiB of all 16 rows through ONE CU's memory path (8.3 us against 4.8 for
// the K = 768 layers); 8 x 8 tiles are 192 blocks of 48 + 48 KiB.  The MFMA runs a quarter full, which costs nothing here.
// RB: 16-row blocks per launch (1, 2, 4: up to 64 rows -- the weights are streamed ONCE for all of them; a wave runs RB MFMA chains
// against the same weight fragments).  HALF keeps RB = 1 and covers more rows through grid.y.
// HR (HALF only): input rows per block, 8 or 16 -- more than 16 rows in all: a block keeps its 8 staged weight rows for 16 input rows.
template <int S, bool DIRECT, bool HALF = false, int RB = 1, int HR = 8>
__global__ __launch_bounds__(512) void gemv16_mfma_kernel(Gemv16Args a) {
    static_assert(!HALF || RB == 1, "the 8-feature tile covers rows through grid.y");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    float* red = reinterpret_cast<float*>(smem16);                       // [8 waves][16 RB rows][16 features]
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem16 + 8 * RB * 256 * 4);   // [16 RB][K + 8] (LayerNorm path); HALF: 8 input + 8 weight rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int FT = HALF ? 8 : 16;                 // features per block
    const int K = a.K, NB = a.NB, n0 = blockIdx.x * FT, rb0 = HALF ? blockIdx.y * HR : 0;
    const int r16 = lane & 15, g = lane >> 4;
    const int rf = HALF ? (r16 & 7) : r16;            // this lane's feature (B operand) inside the tile
    const int ra = (HALF && HR == 8) ? (r16 & 7) : r16;  // ... and its row (A operand)
    const int kw = wave * (S * 32) + g * 8;  // k of this lane's chunk in MFMA step 0; step s is 32 further (the four lanes of a row read 64 contiguous bytes)
    // ---- 1. weights -> registers (do not depend on the rows: requested first).  HALF: both operands are staged through LDS instead
    // (below): 12 fragment loads per lane, each touching sixteen 64-byte segments, kept the texture addresser busier than HBM.
    bf16x8_t wf[S];
    if constexpr (!HALF) {
        const bf16_t* wp = a.W + (long)(n0 + rf) * K + kw;
#pragma unroll
        for (int s = 0; s < S; ++s) wf[s] = *reinterpret_cast<const bf16x8_t*>(wp + s * 32);
    }
    // epilogue operands: thread t finishes outputs o = t + 512 e of the block's (16 RB rows) x 16 features (HALF: 8 x 8)
    constexpr int NE = HALF ? 1 : (RB * 256 + 511) / 512;
    int eb[NE], en[NE];
    bool elane[NE];
    float ebias[NE], eres[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int o = tid + 512 * e;
        eb[e] = rb0 + (HALF ? o >> 3 : o >> 4), en[e] = n0 + (HALF ? o & 7 : o & 15);
        elane[e] = o < (HALF ? HR * 8 : RB * 256) && eb[e] < NB;
        ebias[e] = (elane[e] && a.bias) ? a.bias[en[e]] : 0.f;
        eres[e] = (elane[e] && a.residual) ? a.residual[(long)eb[e] * a.ldr + en[e]] : 0.f;
    }
    const int epos = a.kv_cache ? (a.pos_dev ? *a.pos_dev : a.pos0) : 0;
    bf16x8_t af[RB][S];
    if constexpr (HALF) {
        // 8 weight rows + 8 input rows of K bf16 each, fetched as whole rows (a wave instruction = 1 KiB contiguous), parked in LDS at a
        // padded pitch, read back as MFMA fragments
        static_assert(DIRECT, "the 8 x 8 tile takes bf16 rows");
        const int ldsk = K + 8, cpr = K >> 3;         // 16-byte chunks per row
        bf16_t* wsm = xs + HR * ldsk;
        constexpr int NCH = S / 2;                    // chunks per thread for 8 rows: 8 rows x K / 8 chunks / 512 threads = K / 512
        constexpr int XM = HR / 8;                    // input rows are HR / 8 times the weight rows
        uint4 wv[NCH], xv[NCH * XM];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {  // (weight and input requests interleaved, as are the LDS stores below: the order measured fastest)
            const int c = tid + 512 * i, row = c / cpr, col = (c - row * cpr) * 8;
            wv[i] = *reinterpret_cast<const uint4*>(a.W + (long)(n0 + row) * K + col);
#pragma unroll
            for (int m = 0; m < XM; ++m) {
                const int cx = tid + 512 * (i * XM + m), rx = cx / cpr, colx = (cx - rx * cpr) * 8;
                xv[i * XM + m] = rb0 + rx < NB ? *reinterpret_cast<const uint4*>(a.xb + (long)(rb0 + rx) * a.ldx + colx) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 512 * i, row = c / cpr, col = (c - row * cpr) * 8;
            *reinterpret_cast<uint4*>(wsm + row * ldsk + col) = wv[i];
#pragma unroll
            for (int m = 0; m < XM; ++m) {
                const int cx = tid + 512 * (i * XM + m), rx = cx / cpr, colx = (cx - rx * cpr) * 8;
                *reinterpret_cast<uint4*>(xs + rx * ldsk + colx) = xv[i * XM + m];
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            wf[s] = *reinterpret_cast<const bf16x8_t*>(wsm + rf * ldsk + kw + s * 32);
            af[0][s] = *reinterpret_cast<const bf16x8_t*>(xs + ra * ldsk + kw + s * 32);
        }
    } else if constexpr (DIRECT) {
#pragma unroll
        for (int rbk = 0; rbk < RB; ++rbk) {
            const int xrow = rbk * 16 + rf;
            const bf16_t* xp = a.xb + (long)(xrow < NB ? xrow : 0) * a.ldx + kw;
#pragma unroll
            for (int s = 0; s < S; ++s) af[rbk][s] = *reinterpret_cast<const bf16x8_t*>(xp + s * 32);
        }
        // every request of the block is in flight before the first MFMA waits (hipcc sinks the row loads between the MFMAs otherwise,
        // three at a time: four dependent L2 round trips in the K = 3,072 instance, 9.8 us against 4.9 for K = 768)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rbk = 0; rbk < RB; ++rbk)
            if (rbk * 16 + rf >= NB) {
#pragma unroll
                for (int s = 0; s < S; ++s) af[rbk][s] = bf16x8_t{};
            }
    } else {
        // ---- 2. rows: wave w owns rows w, w + 8, ... (two per 16-row block); LayerNorm statistics are in-wave reductions; the
        // normalised row goes to LDS as bf16.
        const int ldsk = K + 8;
        float4 xr[2 * RB][S];  // every row of this wave is requested before the first is reduced (RB = 4: 24 requests in flight, not 4 x 6)
#pragma unroll
        for (int rr = 0; rr < 2 * RB; ++rr) {
            const int r = wave + 8 * rr;
#pragma unroll
            for (int jj = 0; jj < S; ++jj)
                xr[rr][jj] = r < NB ? *reinterpret_cast<const float4*>(a.x + (long)r * a.ldx + (jj * 64 + lane) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float4 lg[S], lb[S];
#pragma unroll
        for (int jj = 0; jj < S; ++jj) {
            const int k = (jj * 64 + lane) * 4;
            lg[jj] = a.ln_w ? *reinterpret_cast<const float4*>(a.ln_w + k) : make_float4(1.f, 1.f, 1.f, 1.f);
            lb[jj] = a.ln_w ? *reinterpret_cast<const float4*>(a.ln_b + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (RB > 1) __builtin_amdgcn_sched_barrier(0);  // (hipcc otherwise sinks the later rows' requests behind the first rows' reductions)
#pragma unroll
        for (int rr = 0; rr < 2 * RB; ++rr) {
            const int r = wave + 8 * rr;
            float mu = 0.f, rs = 1.f;
            if (a.ln_w && r < NB) {
                float t = 0.f;
#pragma unroll
                for (int jj = 0; jj < S; ++jj) t += (xr[rr][jj].x + xr[rr][jj].y) + (xr[rr][jj].z + xr[rr][jj].w);
                mu = wave_sum_fast(t) / (float)K;
                float q = 0.f;
#pragma unroll
                for (int jj = 0; jj < S; ++jj) {
                    const float d0 = xr[rr][jj].x - mu, d1 = xr[rr][jj].y - mu, d2 = xr[rr][jj].z - mu, d3 = xr[rr][jj].w - mu;
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
                rs = rsqrtf(wave_sum_fast(q) / (float)K + a.eps);
            }
#pragma unroll
            for (int jj = 0; jj < S; ++jj) {
                const int k = (jj * 64 + lane) * 4;
                float v0 = xr[rr][jj].x, v1 = xr[rr][jj].y, v2 = xr[rr][jj].z, v3 = xr[rr][jj].w;
                if (a.ln_w) {
                    v0 = (v0 - mu) * rs * lg[jj].x + lb[jj].x, v1 = (v1 - mu) * rs * lg[jj].y + lb[jj].y;
                    v2 = (v2 - mu) * rs * lg[jj].z + lb[jj].z, v3 = (v3 - mu) * rs * lg[jj].w + lb[jj].w;
                }
                const uint2 pk = r < NB ? make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3)) : make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(xs + (long)r * ldsk + k) = pk;
            }
        }
        __syncthreads();
#pragma unroll
        for (int rbk = 0; rbk < RB; ++rbk)
#pragma unroll
            for (int s = 0; s < S; ++s) af[rbk][s] = *reinterpret_cast<const bf16x8_t*>(xs + (long)(rbk * 16 + r16) * ldsk + kw + s * 32);
    }
    // ---- 3. S MFMAs per row block: C[row = 16 rbk + 4 g + i][feature = r16] over this wave's K slice
#pragma unroll
    for (int rbk = 0; rbk < RB; ++rbk) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rbk][s], wf[s], acc, 0, 0, 0);
        mfma_settle(acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave * (RB * 256) + (rbk * 16 + g * 4 + i) * 16 + r16] = acc[i];
    }
    __syncthreads();
    // (no block of this launch reads the position when pos_inc is set: the key/value append belongs to the in-projection)
    if (a.pos_inc && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *a.pos_inc += 1;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (!elane[e]) continue;
        const int o = tid + 512 * e;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += red[w * (RB * 256) + (HALF ? (o >> 3) * 16 + (o & 7) : o)];
        v += ebias[e];
        if (a.act == 1) v = v * sigmoidf_(1.702f * v);  // QuickGELU (clip_model.py:196-198)
        v += eres[e];
        if (a.round_out) v = round_bf16(v);
        if (a.out) a.out[(long)eb[e] * a.ldo + en[e]] = v;
        if (a.out_bf) a.out_bf[(long)eb[e] * a.ldo + en[e]] = f2bf(v);
        if (a.kv_cache && en[e] >= a.kv_lo && en[e] < a.kv_lo + a.kv_width && epos < a.Lmax)
            a.kv_cache[((long)eb[e] * a.Lmax + epos) * a.kv_width + (en[e] - a.kv_lo)] = f2bf(v);
    }
}

// from 3 rows on the matrix-pipe form is the faster one (batch 4: 315 -> 270 us per tower step, 8: 390 -> 307, 16: 568 -> 348; same box,
// profiles/r06_decode_step_vector_alu_vs_mfma_gemv.log); 1-2 rows stay on gemv_rows_kernel (the persistent step's fall-back form)
constexpr int GV16_MIN_ROWS = 3;
constexpr int GV16_MAXB = 64;  // rows per launch: four 16-row blocks
bool gemv16_supported(int NB, int N, int K) {
    const int S = K / 256;
    return NB >= 1 && NB <= GV16_MAXB && N % 16 == 0 && K % 256 == 0 && (S == 2 || S == 3 || ((S == 8 || S == 12) && N <= 1024));
}

int gemv16_launch(const Gemv16Args& a, hipStream_t s) {
    if (!gemv16_supported(a.NB, a.N, a.K) || a.ldx % 8 != 0 || (!a.x && !a.xb) || (!a.xb && a.K > 1024)) {  // (fp32 rows: K <= 1024)
        mmvid_set_error("decode gemv (MFMA form): NB=%d (<= %d), N=%d (multiple of 16), K=%d (512 / 768 / 2048 / 3072), ldx %% 8 == 0", a.NB, GV16_MAXB, a.N, a.K);
        return MMVID_ERR_ARG;
    }
    const bool direct = a.xb != nullptr;
    const bool half = direct && a.N <= 1024 && a.K >= 2048;  // (narrow and long: see HALF)
    if (!half && a.K > 1024) {
        mmvid_set_error("decode gemv (MFMA form): K=%d > 1024 needs N <= 1024 (the 8 x 8 tile)", a.K);
        return MMVID_ERR_ARG;
    }
    const int rb = half ? 1 : (a.NB <= 16 ? 1 : (a.NB <= 32 ? 2 : 4));
    const int hr = a.NB > 16 ? 16 : 8;  // (HALF: input rows per block)
    const dim3 grid(half ? a.N / 8 : a.N / 16, half ? cdiv(a.NB, hr) : 1);
    const size_t lds = half ? 8 * 256 * 4 + (size_t)(8 + hr) * (a.K + 8) * 2 : (size_t)8 * rb * 256 * 4 + (direct ? 0 : (size_t)16 * rb * (a.K + 8) * 2);
    auto go = [&](auto kern) {
        static bool attr = false;  // (per instantiation)
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 256 * 4 + 24 * (3072 + 8) * 2);  // (152 KiB: 8 weight + 16 input rows at K = 3,072; four row blocks at K = 768 need 129 KiB)
            attr = true;
        }
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
    };
    auto full = [&](auto s_c, auto d_c) {  // the 16 x 16-tile instances by row-block count
        constexpr int SC = decltype(s_c)::value;
        constexpr bool DC = decltype(d_c)::value != 0;
        if (rb == 1)
            go(gemv16_mfma_kernel<SC, DC, false, 1>);
        else if (rb == 2)
            go(gemv16_mfma_kernel<SC, DC, false, 2>);
        else
            go(gemv16_mfma_kernel<SC, DC, false, 4>);
    };
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using T1 = std::integral_constant<int, 1>;
    using T0 = std::integral_constant<int, 0>;
    switch (a.K / 256) {
        case 2: direct ? full(I2{}, T1{}) : full(I2{}, T0{}); break;
        case 3: direct ? full(I3{}, T1{}) : full(I3{}, T0{}); break;
        case 8: hr == 16 ? go(gemv16_mfma_kernel<8, true, true, 1, 16>) : go(gemv16_mfma_kernel<8, true, true>); break;
        default: hr == 16 ? go(gemv16_mfma_kernel<12, true, true, 1, 16>) : go(gemv16_mfma_kernel<12, true, true>); break;
    }
    return MMVID_OK;
}

// One block per (head, batch): q fp32 [B][ldq] (already bf16-rounded), K|V rows from the cache (position `pos` included).
// Thread (kg, dc) = (tid >> 3, tid & 7) holds dims [8 dc, 8 dc + 8) of q in registers and takes that 16-byte slice of the keys kg, kg + NG,
// ... (NG = 8 NW key groups): a key's row is read by 8 adjacent lanes, its score is the sum over them (three DPP steps), 8 loads per
// thread are in flight in both passes.  NW = 16 waves for long caches: the kernel is a chain of memory round trips (1,100 keys took
// three for K and five for V with 256 threads: 16 us of a 48-us layer at batch 8), and one block is all a (head, sequence) gets.
// NB: key / value batches (of 8 keys per thread = 64 NW keys per batch) held in registers: NB * 64 * NW positions are covered without a
// dependent round trip (NW = 8, NB = 3: 1,536)
template <int NW, int NB = 2>
__global__ __launch_bounds__(NW * 64) void attn_decode2_kernel(const float* __restrict__ q, long ldq, const bf16_t* __restrict__ cache,
                                                               int Lmax, int E, const int* __restrict__ pos_dev, int pos0,
                                                               float scale_log2, float* __restrict__ out, long ldo,
                                                               bf16_t* __restrict__ out_bf = nullptr) {
    __shared__ float sc[DEC_MAXL];
    __shared__ float red[NW][64];
    __shared__ float stat[2][NW];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = tid >> 3, dc = tid & 7;
    const int n_all = min((pos_dev ? *pos_dev : pos0) + 1, Lmax);  // (never beyond the cache)
    // a short cache is one round trip for four waves already: the others do nothing (the block size is fixed when the step is captured,
    // the position is not; 16 waves at position 129 cost the step 12 us).  They stay alive with an empty key range -- no loads, no
    // stores -- so that every wave of the block reaches every barrier below.
    const int nw = (NW > 4 && n_all <= 512) ? 4 : NW;
    const int n = wave < nw ? n_all : 0;
    const int NT = nw * 64, NG = nw * 8;
    const long rs = 2 * (long)E;  // elements between two positions
    const int vo = E;             // from a key row to its value row
    const bf16_t* kv = cache + (long)b * Lmax * 2 * E + h * 64 + dc * 8;
    // Round 6: the kernel is a chain of memory round trips (K batch -> scores -> V batch -> ...), 11 us on average at batch 16 for 31 MB.
    // Neither the keys nor the values depend on q or on the scores, so the first TWO key batches (2 x 8 NG keys: 2,048 with 16 waves) and the
    // first value batch are requested before anything else, the other value batches as soon as the key registers are free -- i.e. before
    // the softmax statistics and their two barriers: one round trip for the keys and one for the values of up to NB * 64 NW positions.
    // Eight waves with three batches (1,536 positions) instead of sixteen with two: 460 -> 450 us per token at batch 16 (same box).
    // (What is left is the fixed chain: 7.2 us at 129 keys, 4.8 TB/s marginal from there to 1,100; a head-major cache layout changes 3-6 %:
    // profiles/r06_decode_attention_cache_layout_experiment.log.)
    uint4 uk[NB][8], uv[NB][8];
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kg + NG * (j + 8 * bt);
            uk[bt][j] = k < n ? *reinterpret_cast<const uint4*>(kv + (long)k * rs) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kg + NG * j;
        uv[0][j] = k < n ? *reinterpret_cast<const uint4*>(kv + (long)k * rs + vo) : make_uint4(0u, 0u, 0u, 0u);
    }
    float qv[8];
    {
        const float4 q0 = *reinterpret_cast<const float4*>(q + (long)b * ldq + h * 64 + dc * 8);
        const float4 q1 = *reinterpret_cast<const float4*>(q + (long)b * ldq + h * 64 + dc * 8 + 4);
        qv[0] = q0.x, qv[1] = q0.y, qv[2] = q0.z, qv[3] = q0.w, qv[4] = q1.x, qv[5] = q1.y, qv[6] = q1.z, qv[7] = q1.w;
    }
    __builtin_amdgcn_sched_barrier(0);  // (all of the above in flight before the first score waits)
    float mx = -INFINITY;
    auto scores = [&](const uint4 (&u)[8], int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + NG * j;
            float d = (bf_lo(u[j].x) * qv[0] + bf_hi(u[j].x) * qv[1]) + (bf_lo(u[j].y) * qv[2] + bf_hi(u[j].y) * qv[3]) +
                      (bf_lo(u[j].z) * qv[4] + bf_hi(u[j].z) * qv[5]) + (bf_lo(u[j].w) * qv[6] + bf_hi(u[j].w) * qv[7]);
            d = group8_sum(d) * scale_log2;
            if (k < n) {
                if (dc == 0) sc[k] = d;
                mx = fmaxf(mx, d);
            }
        }
    };
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
        if (bt == 0 || kg + NG * 8 * bt < n) scores(uk[bt], kg + NG * 8 * bt);
    for (int k0 = kg + NG * 8 * NB; k0 < n; k0 += NG * 8) {  // caches beyond the register batches
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + NG * j;
            uk[0][j] = k < n ? *reinterpret_cast<const uint4*>(kv + (long)k * rs) : make_uint4(0u, 0u, 0u, 0u);
        }
        scores(uk[0], k0);
    }
    // the other value batches, under the softmax statistics (the key registers are free now)
#pragma unroll
    for (int bt = 1; bt < NB; ++bt)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kg + NG * (j + 8 * bt);
            uv[bt][j] = k < n ? *reinterpret_cast<const uint4*>(kv + (long)k * rs + vo) : make_uint4(0u, 0u, 0u, 0u);
        }
    mx = wave_max_fast(mx);
    if (lane == 0) stat[0][wave] = mx;
    __syncthreads();
    mx = stat[0][0];
    for (int w = 1; w < nw; ++w) mx = fmaxf(mx, stat[0][w]);
    float sum = 0.f;
    for (int k = tid; k < n; k += NT) {
        const float p = __builtin_amdgcn_exp2f(sc[k] - mx);
        sc[k] = p;
        sum += p;
    }
    sum = wave_sum_fast(sum);
    if (lane == 0) stat[1][wave] = sum;
    __syncthreads();
    sum = 0.f;
    for (int w = 0; w < nw; ++w) sum += stat[1][w];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto values = [&](const uint4 (&u)[8], int k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + NG * j;
            const float p = k < n ? sc[k] : 0.f;
            acc[0] += p * bf_lo(u[j].x), acc[1] += p * bf_hi(u[j].x), acc[2] += p * bf_lo(u[j].y), acc[3] += p * bf_hi(u[j].y);
            acc[4] += p * bf_lo(u[j].z), acc[5] += p * bf_hi(u[j].z), acc[6] += p * bf_lo(u[j].w), acc[7] += p * bf_hi(u[j].w);
        }
    };
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
        if (bt == 0 || kg + NG * 8 * bt < n) values(uv[bt], kg + NG * 8 * bt);
    for (int k0 = kg + NG * 8 * NB; k0 < n; k0 += NG * 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + NG * j;
            uv[0][j] = k < n ? *reinterpret_cast<const uint4*>(kv + (long)k * rs + vo) : make_uint4(0u, 0u, 0u, 0u);
        }
        values(uv[0], k0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = over_groups_sum(acc[e]);  // the wave's 8 key groups
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave][dc * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w][tid];
        // the full forward stores the attention output in bf16 (out_bf: as bf16 bits for the MFMA out-projection, same values)
        if (out_bf)
            out_bf[(long)b * ldo + h * 64 + tid] = f2bf(s / sum);
        else
            out[(long)b * ldo + h * 64 + tid] = round_bf16(s / sum);
    }
}

// x[b, :] = table[tok[b]] + pos_rows[*pos_dev + pos_off]: the embedding of the token just sampled (dalle_artv.py:484-491)
__global__ __launch_bounds__(256) void dec_embed_kernel(const long long* __restrict__ tok, const float* __restrict__ table,
                                                        long table_rows, const float* __restrict__ pos_rows,
                                                        const int* __restrict__ pos_dev, int pos_off, int E, float* __restrict__ x,
                                                        long long* __restrict__ record, long record_ld, int record_pos0) {
    const int b = blockIdx.x;
    long long id = tok[b];
    // the sampler's output row: token number (*pos_dev - record_pos0) of sequence b (a scatter_ and a counter update per token otherwise)
    if (record && threadIdx.x == 0 && *pos_dev >= record_pos0 && *pos_dev - record_pos0 < record_ld) record[b * record_ld + (*pos_dev - record_pos0)] = id;
    if (id < 0 || id >= table_rows) id = 0;
    const long p = (long)(*pos_dev) + pos_off;
    for (int e = threadIdx.x; e < E; e += 256) x[(long)b * E + e] = table[id * E + e] + pos_rows[p * E + e];
}

int gemv_launch(GemvArgs a, hipStream_t s) {
    const bool wide = a.NB > 8;  // 9-16 rows: bf16-exact staged rows only (96 KiB of LDS at 16 x 3,072)
    if (a.NB > GV_MAXB || a.K % 8 != 0 || (long)a.NB * a.K * (a.round_in ? 2 : 4) > 24576 * 4 || (wide && !a.round_in) || a.K > 3072 ||
        a.ldx % 4 != 0) {
        mmvid_set_error("decode gemv: NB=%d (<= %d; > 8 needs round_in), K=%d (multiple of 8, <= 3072, staged rows <= 96 KiB), ldx %% 4 == 0", a.NB,
                        GV_MAXB, a.K);
        return MMVID_ERR_ARG;
    }
    // 8 output features per block, or 4 (one per wave) for the narrow outputs (out-proj, c_proj: N = 768 would be 96 blocks on 256 CUs)
    const bool narrow = a.N <= 1024;
    const dim3 grid(cdiv(a.N, narrow ? 4 : GV_COLS));
    const size_t lds = (size_t)a.NB * a.K * (a.round_in ? 2 : 4);
    const int kit = cdiv(a.K, 512);
    static bool attr = false;
    if (!attr) {  // up to 8 rows of a 3,072-wide fp32 input: 96 KiB of LDS (48 KiB as bf16)
        (void)hipFuncSetAttribute((const void*)(gemv_rows_kernel<6, false, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
        (void)hipFuncSetAttribute((const void*)(gemv_rows_kernel<6, false, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
        (void)hipFuncSetAttribute((const void*)(gemv_rows_kernel<6, true, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
        (void)hipFuncSetAttribute((const void*)(gemv_rows_kernel<6, true, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
        attr = true;
    }
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, g_decode_trace); };
    if (wide) {
        static bool wattr = false;
        if (!wattr) {
            (void)hipFuncSetAttribute((const void*)(gemv_rows_kernel<2, false, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
            (void)hipFuncSetAttribute((const void*)(gemv_rows_kernel<2, false, 2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
            (void)hipFuncSetAttribute((const void*)(gemv_rows_kernel<6, false, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 24576 * 4);
            wattr = true;
        }
        if (kit <= 2)
            narrow ? go(gemv_rows_kernel<2, false, 1, true, true>) : go(gemv_rows_kernel<2, false, 2, true, true>);
        else if (narrow)
            go(gemv_rows_kernel<6, false, 1, true, true>);
        else {
            mmvid_set_error("decode gemv: NB=%d > 8 with K=%d > 1024 and N=%d > 1024 is not instantiated", a.NB, a.K, a.N);
            return MMVID_ERR_ARG;
        }
        return MMVID_OK;
    }
    const bool bf = a.round_in != 0;  // (the staged rows are then bf16-exact)
    auto pick = [&](auto kit_c) {
        constexpr int KITC = decltype(kit_c)::value;
        if (g_decode_trace) {
            if (narrow)
                bf ? go(gemv_rows_kernel<KITC, true, 1, true>) : go(gemv_rows_kernel<KITC, true, 1, false>);
            else
                bf ? go(gemv_rows_kernel<KITC, true, 2, true>) : go(gemv_rows_kernel<KITC, true, 2, false>);
        } else if (narrow) {
            bf ? go(gemv_rows_kernel<KITC, false, 1, true>) : go(gemv_rows_kernel<KITC, false, 1, false>);
        } else {
            bf ? go(gemv_rows_kernel<KITC, false, 2, true>) : go(gemv_rows_kernel<KITC, false, 2, false>);
        }
    };
    if (kit <= 2)
        pick(std::integral_constant<int, 2>{});
    else if (kit <= 4)
        pick(std::integral_constant<int, 4>{});
    else
        pick(std::integral_constant<int, 6>{});
    return MMVID_OK;
}

}  // namespace

extern "C" int mmvid_decode_trace(void* dev_buf) {
    g_decode_trace = (unsigned long long*)dev_buf;
    return MMVID_OK;
}

extern "C" int mmvid_gemv_rows(const float* x, int64_t ldx, int NB, int K, const float* ln_w, const float* ln_b, float eps,
                               const void* W, const float* bias, int N, int act, const float* residual, int64_t ldr,
                               int round_in, int round_out, float* out, int64_t ldo, void* stream) {
    MMVID_REQUIRE(x && W && out && NB > 0 && N > 0 && K > 0, "gemv_rows: bad arguments");
    GemvArgs a = {};
    a.x = x, a.ldx = ldx, a.ln_w = ln_w, a.ln_b = ln_b, a.eps = eps, a.W = (const bf16_t*)W, a.bias = bias;
    a.residual = residual, a.ldr = ldr, a.out = out, a.ldo = ldo, a.NB = NB, a.N = N, a.K = K, a.act = act;
    a.round_in = round_in, a.round_out = round_out;
    if (round_in && NB >= GV16_MIN_ROWS && K <= 1024 && ldx % 8 == 0 && gemv16_supported(NB, N, K)) {  // 3..64 bf16-exact rows: the matrix-pipe form
        Gemv16Args g = {};
        g.x = x, g.ldx = ldx, g.ln_w = ln_w, g.ln_b = ln_b, g.eps = eps, g.W = (const bf16_t*)W, g.bias = bias, g.residual = residual, g.ldr = ldr;
        g.out = out, g.ldo = ldo, g.NB = NB, g.N = N, g.K = K, g.act = act, g.round_out = round_out;
        int rc16 = gemv16_launch(g, (hipStream_t)stream);
        if (rc16) return rc16;
        MMVID_LAUNCH_CHECK("gemv_rows");
        return MMVID_OK;
    }
    int rc = gemv_launch(a, (hipStream_t)stream);
    if (rc) return rc;
    MMVID_LAUNCH_CHECK("gemv_rows");
    return MMVID_OK;
}

extern "C" int mmvid_decode_embed(const int64_t* tok, const float* table, int64_t table_rows, const float* pos_rows,
                                  const int32_t* pos_dev, int pos_off, int B, int E, float* x, void* stream) {
    return mmvid_decode_embed_record(tok, table, table_rows, pos_rows, pos_dev, pos_off, B, E, x, nullptr, 0, 0, stream);
}

// ... and record[b][*pos_dev - record_pos0] = tok[b] (record: int64 [B][record_ld] or null): the sampler's list of drawn tokens
extern "C" int mmvid_decode_embed_record(const int64_t* tok, const float* table, int64_t table_rows, const float* pos_rows,
                                         const int32_t* pos_dev, int pos_off, int B, int E, float* x, int64_t* record, int64_t record_ld,
                                         int record_pos0, void* stream) {
    MMVID_REQUIRE(tok && table && pos_rows && pos_dev && x && B > 0, "decode_embed: bad arguments");
    hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)tok, table, (long)table_rows,
                       pos_rows, pos_dev, pos_off, E, x, (long long*)record, (long)record_ld, record_pos0);
    MMVID_LAUNCH_CHECK("decode_embed");
    return MMVID_OK;
}

// One new position per sequence through all layers, five launches per layer (see above).  scratch: >= B*(3E+E+E+F+2E)*4 B.
extern "C" int mmvid_tower_decode_fused(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                                        float* x_out, void* kv_cache, int Lmax, const int32_t* pos_dev, int pos, void* scratch,
                                        void* stream) {
    MMVID_REQUIRE(cfg, "tower_decode_fused: null pointer");
    return mmvid_tower_decode_fused_slice(cfg, layers, x_in, x_out, kv_cache, Lmax, cfg->B, const_cast<int32_t*>(pos_dev), pos, 0, scratch, stream);
}

// The same for cfg->B consecutive sequences of a cache that holds cache_batch >= cfg->B of them: kv_cache points at the first of these
// sequences in layer 0, a layer is cache_batch * Lmax * 2E elements further.  (Batches above 16 run as slices of 16: the M = B corner of
// the training GEMM takes 2.2 ms per token at batch 16.)
__global__ void pos_advance_kernel(int* p) { *p += 1; }

extern "C" int mmvid_tower_decode_fused_slice(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                                              float* x_out, void* kv_cache, int Lmax, int cache_batch, int32_t* pos_dev, int pos,
                                              int advance_pos, void* scratch, void* stream) {
    MMVID_REQUIRE(!advance_pos || pos_dev, "tower_decode_fused: advance_pos needs the device position");
    MMVID_REQUIRE(cfg && layers && x_in && x_out && kv_cache && scratch, "tower_decode_fused: null pointer");
    MMVID_REQUIRE(cache_batch >= cfg->B, "tower_decode_fused: cache_batch %d < batch %d", cache_batch, cfg->B);
    MMVID_REQUIRE(cfg->mask_mode == 1 && cfg->E == cfg->H * 64 && cfg->B <= GV16_MAXB && Lmax <= DEC_MAXL,
                  "tower_decode_fused: causal tower, head_dim 64, batch <= %d, Lmax <= %d", GV16_MAXB, DEC_MAXL);
    const int B = cfg->B, E = cfg->E, F = cfg->F, H = cfg->H;
    hipStream_t s = (hipStream_t)stream;
    float* p = (float*)scratch;
    float* qkv = p;
    p += (long)B * 3 * E;
    float* o = p;
    p += (long)B * E;
    float* xmid = p;
    p += (long)B * E;
    float* act = p;
    p += (long)B * F;
    float* xa = p;
    p += (long)B * E;
    float* xb = p;
    const float* x = x_in;
    // 3..64 sequences: the linear layers on the matrix pipe (gemv16_mfma_kernel: the weights are streamed once for all of them); 1-2: the
    // vector-ALU form (what the persistent step falls back to)
    const bool mfma = B >= GV16_MIN_ROWS && gemv16_supported(B, 3 * E, E) && gemv16_supported(B, E, F) && gemv16_supported(B, F, E);
    MMVID_REQUIRE(mfma || B <= GV_MAXB, "tower_decode_fused: batch %d > %d needs the matrix-pipe form (widths 512 / 768, F = 4 E)", B, GV_MAXB);
    bf16_t* o_bf = (bf16_t*)o;      // (MFMA form: the attention output and the activation travel as bf16 -- the operand the next layer
    bf16_t* act_bf = (bf16_t*)act;  //  would round them to anyway)
    for (int i = 0; i < cfg->layers; ++i) {
        const mmvid_tower_layer_t& ly = layers[i];
        bf16_t* cache = (bf16_t*)kv_cache + (long)i * cache_batch * Lmax * 2 * E;
        float* xnext = (i == cfg->layers - 1) ? x_out : ((i & 1) ? xb : xa);
        if (mfma) {
            Gemv16Args g = {};
            g.NB = B, g.eps = cfg->ln_eps, g.x = x, g.ldx = E, g.ln_w = ly.ln1_w, g.ln_b = ly.ln1_b, g.W = (const bf16_t*)ly.in_w, g.bias = ly.in_b;
            g.N = 3 * E, g.K = E, g.out = qkv, g.ldo = 3 * E, g.round_out = 1, g.kv_cache = cache, g.kv_lo = E, g.kv_width = 2 * E, g.Lmax = Lmax;
            g.pos_dev = pos_dev, g.pos0 = pos;
            int rc = gemv16_launch(g, s);
            if (rc) return rc;
            if (Lmax > 512 && H * B > 512)  // more blocks than two rounds of the chip: two blocks per CU (124 registers) hide each other's chain
                hipLaunchKernelGGL((attn_decode2_kernel<8, 2>), dim3(H, B), dim3(512), 0, s, qkv, (long)3 * E, cache, Lmax, E, pos_dev, pos,
                                   0.125f * 1.4426950408889634f, (float*)nullptr, (long)E, o_bf);
            else if (Lmax > 512)
                hipLaunchKernelGGL((attn_decode2_kernel<8, 3>), dim3(H, B), dim3(512), 0, s, qkv, (long)3 * E, cache, Lmax, E, pos_dev, pos,
                                   0.125f * 1.4426950408889634f, (float*)nullptr, (long)E, o_bf);
            else
                hipLaunchKernelGGL(attn_decode2_kernel<4>, dim3(H, B), dim3(256), 0, s, qkv, (long)3 * E, cache, Lmax, E, pos_dev, pos,
                                   0.125f * 1.4426950408889634f, (float*)nullptr, (long)E, o_bf);
            Gemv16Args g2 = {};
            g2.NB = B, g2.xb = o_bf, g2.ldx = E, g2.W = (const bf16_t*)ly.out_w, g2.bias = ly.out_b, g2.N = E, g2.K = E, g2.residual = x, g2.ldr = E;
            g2.out = xmid, g2.ldo = E;
            rc = gemv16_launch(g2, s);
            if (rc) return rc;
            Gemv16Args g3 = {};
            g3.NB = B, g3.eps = cfg->ln_eps, g3.x = xmid, g3.ldx = E, g3.ln_w = ly.ln2_w, g3.ln_b = ly.ln2_b, g3.W = (const bf16_t*)ly.fc_w;
            g3.bias = ly.fc_b, g3.N = F, g3.K = E, g3.act = 1, g3.out_bf = act_bf, g3.ldo = F;
            rc = gemv16_launch(g3, s);
            if (rc) return rc;
            Gemv16Args g4 = {};
            g4.NB = B, g4.xb = act_bf, g4.ldx = F, g4.W = (const bf16_t*)ly.pj_w, g4.bias = ly.pj_b, g4.N = E, g4.K = F, g4.residual = xmid, g4.ldr = E;
            g4.out = xnext, g4.ldo = E;
            if (advance_pos && i == cfg->layers - 1) g4.pos_inc = pos_dev;
            rc = gemv16_launch(g4, s);
            if (rc) return rc;
            x = xnext;
            continue;
        }
        GemvArgs g = {};
        g.NB = B, g.eps = cfg->ln_eps;
        // q,k,v (+ cache append)
        g.x = x, g.ldx = E, g.ln_w = ly.ln1_w, g.ln_b = ly.ln1_b, g.W = (const bf16_t*)ly.in_w, g.bias = ly.in_b, g.N = 3 * E, g.K = E;
        g.out = qkv, g.ldo = 3 * E, g.round_in = 1, g.round_out = 1, g.kv_cache = cache, g.kv_lo = E, g.kv_width = 2 * E, g.Lmax = Lmax;
        g.pos_dev = pos_dev, g.pos0 = pos;
        int rc = gemv_launch(g, s);
        if (rc) return rc;
        if (Lmax > 512)
            hipLaunchKernelGGL((attn_decode2_kernel<8, 3>), dim3(H, B), dim3(512), 0, s, qkv, (long)3 * E, cache, Lmax, E, pos_dev, pos,
                               0.125f * 1.4426950408889634f, o, (long)E);
        else
            hipLaunchKernelGGL(attn_decode2_kernel<4>, dim3(H, B), dim3(256), 0, s, qkv, (long)3 * E, cache, Lmax, E, pos_dev, pos,
                               0.125f * 1.4426950408889634f, o, (long)E);
        GemvArgs g2 = {};
        g2.NB = B, g2.x = o, g2.ldx = E, g2.W = (const bf16_t*)ly.out_w, g2.bias = ly.out_b, g2.N = E, g2.K = E, g2.residual = x,
        g2.ldr = E, g2.out = xmid, g2.ldo = E, g2.round_in = 1;  // (o is bf16-exact already: rounding is a no-op, LDS holds bf16)
        rc = gemv_launch(g2, s);
        if (rc) return rc;
        GemvArgs g3 = {};
        g3.NB = B, g3.eps = cfg->ln_eps, g3.x = xmid, g3.ldx = E, g3.ln_w = ly.ln2_w, g3.ln_b = ly.ln2_b, g3.W = (const bf16_t*)ly.fc_w,
        g3.bias = ly.fc_b, g3.N = F, g3.K = E, g3.act = 1, g3.out = act, g3.ldo = F, g3.round_in = 1, g3.round_out = 1;
        rc = gemv_launch(g3, s);
        if (rc) return rc;
        GemvArgs g4 = {};
        g4.NB = B, g4.x = act, g4.ldx = F, g4.W = (const bf16_t*)ly.pj_w, g4.bias = ly.pj_b, g4.N = E, g4.K = F, g4.residual = xmid,
        g4.ldr = E, g4.out = xnext, g4.ldo = E, g4.round_in = 1;  // (the rounded GELU output)
        rc = gemv_launch(g4, s);
        if (rc) return rc;
        x = xnext;
    }
    if (advance_pos && !mfma) hipLaunchKernelGGL(pos_advance_kernel, dim3(1), dim3(1), 0, s, pos_dev);
    MMVID_LAUNCH_CHECK("tower_decode_fused");
    return MMVID_OK;
}
