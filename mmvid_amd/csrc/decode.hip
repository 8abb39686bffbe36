// Incremental (KV-cache) decoding for the causal tower: SURVEY next-row N1.  The reference's ART-V sampler
// (dalle_artv.py:236-304) re-runs the whole transformer over the growing prefix for each of its 1,024 tokens; with
// a causal mask the keys and values of earlier positions never change, so they are kept per layer and a step only
// computes the new position.  Same distribution as the full recomputation (bf16 rounding order aside).
//   cache layout: [layers][B][Lmax][2E] bf16, a row = K (E values) followed by V (E values) of one position.
// Both kernels are tiny and HBM/latency-bound (one query per (batch, head)); the step cost is the weight traffic of
// the twelve layers' GEMMs at M = B.
#include "../../include/mmvid_hip.h"
#include "common.h"

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
    mx = wave_max(mx);
    if (lane == 0) stat[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(stat[0], stat[1]), fmaxf(stat[2], stat[3]));
    float sum = 0.f;
    for (int k = tid; k < n; k += 256) {
        const float p = __builtin_amdgcn_exp2f(sc[k] - mx);
        sc[k] = p;
        sum += p;
    }
    sum = wave_sum(sum);
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
