// VQ codebook lookup for gfx950.
//
// vq_argmin: replaces taming/modules/vqvae/quantize.py:306-310 of the reference
//     d = sum(z^2) + sum(e^2) - 2 z.e^T ; idx = argmin(d)
// with the FIXED fp32 operation order of oracle/vq_argmin.c (bit-exact contract):
//     zz, ee, dot = fmaf chains over k ascending from 0;  d = (zz + ee) - 2*dot;  first minimum wins.
// The dot products run on the f32-input matrix pipe (v_mfma_f32_16x16x4_f32), which the CDNA4
// guide documents as bit-for-bit a k-ordered fmaf chain -- so exactness costs nothing.
//
// Geometry: block = 2 waves, each wave owns 16 rows of z held entirely in registers
// (lane (i=l&15, g=l>>4) keeps z[i][4s+g], s=0..DIM/4-1, i.e. exactly its MFMA A operand);
// the codebook streams through LDS in 32-code tiles (two independent 16-code accumulators
// per wave cover the 40-cycle dependent-MFMA latency), double buffered with register
// prefetch.  LDS rows are padded to DIM+2 floats: conflict-free ds_read_b32 for the
// (2j+g) bank pattern.  Roofline: fp32 matrix pipe (157 TFLOP/s), 512 FLOP per byte.
#include "common.h"

namespace {

constexpr int kWaves = 2;
constexpr int kRowsPerWave = 16;
constexpr int kTileCodes = 32;

__global__ void vq_sqnorm_kernel(const float* __restrict__ e, int n, int dim, float* __restrict__ ee) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const float* r = e + (long)j * dim;
    float s = 0.0f;
    for (int k = 0; k < dim; ++k) s = __fmaf_rn(r[k], r[k], s);
    ee[j] = s;
}

// (distance, index) -> one 64-bit key whose unsigned order is (distance ascending, then index ascending).  Distances are
// sums of squares minus a dot product: never -0.0 (x - x rounds to +0.0), NaN only from non-finite input.
__device__ __forceinline__ unsigned long long vq_key(float d, int j) {
    unsigned u = __float_as_uint(d + 0.0f);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned)j;
}
__global__ void vq_keys_init_kernel(long long* __restrict__ idx, long rows) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r < rows) reinterpret_cast<unsigned long long*>(idx)[r] = vq_key(INFINITY, 0);  // what an all-NaN row ends with
}
__global__ void vq_unpack_kernel(long long* __restrict__ idx, float* __restrict__ dmin, long rows) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const unsigned long long k = reinterpret_cast<unsigned long long*>(idx)[r];
    unsigned u = (unsigned)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xffffffffu;
    idx[r] = (long long)(unsigned)(k & 0xffffffffu);
    if (dmin) dmin[r] = __uint_as_float(u);
}

template <int DIM>
__global__ __launch_bounds__(kWaves * 64) void vq_argmin_kernel(const float* __restrict__ z,
                                                                 const float* __restrict__ e,
                                                                 const float* __restrict__ ee, long rows, int n,
                                                                 long long* __restrict__ idx_out,
                                                                 float* __restrict__ dmin_out) {
    constexpr int LD = DIM + 2;                    // floats per LDS row
    constexpr int TILE_FLOATS = kTileCodes * LD;   // one buffer
    constexpr int F4_PER_ROW = DIM / 4;
    constexpr int F4_PER_TILE = kTileCodes * F4_PER_ROW;
    constexpr int F4_PER_THREAD = F4_PER_TILE / (kWaves * 64);
    static_assert(F4_PER_TILE % (kWaves * 64) == 0, "tile/threads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* buf0 = smem;
    float* buf1 = smem + TILE_FLOATS;
    float* zzs = smem + 2 * TILE_FLOATS;  // [kWaves*16]

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int li = lane & 15;
    const int g = lane >> 4;
    const long row0 = (long)blockIdx.x * (kWaves * kRowsPerWave);

    // ---- stage the block's 32 z rows through buf1 (same shape as a code tile) ----
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
        int f = tid + i * (kWaves * 64);
        int r = f / F4_PER_ROW, c = f % F4_PER_ROW;
        long gr = row0 + r;
        if (gr >= rows) gr = rows - 1;
        float4 v = *reinterpret_cast<const float4*>(z + gr * DIM + c * 4);
        float2* dst = reinterpret_cast<float2*>(buf1 + r * LD + c * 4);
        dst[0] = make_float2(v.x, v.y);
        dst[1] = make_float2(v.z, v.w);
    }
    __syncthreads();
    float a[DIM / 4];
    {
        const float* zr = buf1 + (wave * kRowsPerWave + li) * LD + g;
#pragma unroll
        for (int s = 0; s < DIM / 4; ++s) a[s] = zr[4 * s];
    }
    if (lane < 16) {  // exact sequential chain for ||z||^2, one lane per row
        const float* zr = buf1 + (wave * kRowsPerWave + lane) * LD;
        float s = 0.0f;
        for (int k = 0; k < DIM; ++k) s = __fmaf_rn(zr[k], zr[k], s);
        zzs[wave * kRowsPerWave + lane] = s;
    }
    __syncthreads();
    float zz[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) zz[r] = zzs[wave * kRowsPerWave + 4 * g + r];

    // ---- stream the codebook (this block's share of it: gridDim.y code ranges, merged by atomicMin on (distance, index)) ----
    const int tiles_total = n / kTileCodes, per_split = (tiles_total + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_begin = (int)blockIdx.y * per_split;
    const int ntiles = t_begin + per_split < tiles_total ? t_begin + per_split : tiles_total;  // one past this block's last tile
    float4 pre[F4_PER_THREAD];
    auto gload = [&](int t) {
#pragma unroll
        for (int i = 0; i < F4_PER_THREAD; ++i) {
            int f = tid + i * (kWaves * 64);
            int r = f / F4_PER_ROW, c = f % F4_PER_ROW;
            pre[i] = *reinterpret_cast<const float4*>(e + ((long)t * kTileCodes + r) * DIM + c * 4);
        }
    };
    auto lstore = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < F4_PER_THREAD; ++i) {
            int f = tid + i * (kWaves * 64);
            int r = f / F4_PER_ROW, c = f % F4_PER_ROW;
            float2* dst = reinterpret_cast<float2*>(buf + r * LD + c * 4);
            dst[0] = make_float2(pre[i].x, pre[i].y);
            dst[1] = make_float2(pre[i].z, pre[i].w);
        }
    };
    if (t_begin < ntiles) {
        gload(t_begin);
        lstore((t_begin & 1) ? buf1 : buf0);
    }
    __syncthreads();  // also: everyone is done reading z from buf1

    float best_d[4];
    int best_j[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        best_d[r] = INFINITY;
        best_j[r] = 0;
    }

    for (int t = t_begin; t < ntiles; ++t) {
        float* cur = (t & 1) ? buf1 : buf0;
        float* nxt = (t & 1) ? buf0 : buf1;
        if (t + 1 < ntiles) gload(t + 1);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* eb = cur + li * LD + g;
#pragma unroll
        for (int s = 0; s < DIM / 4; ++s) {
            float b0 = eb[4 * s];
            float b1 = eb[16 * LD + 4 * s];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1, acc1, 0, 0, 0);
        }
        const int j0 = t * kTileCodes + li;
        const float ee0 = ee[j0], ee1 = ee[j0 + 16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float d0 = __fsub_rn(__fadd_rn(zz[r], ee0), 2.0f * acc0[r]);
            if (d0 < best_d[r]) {
                best_d[r] = d0;
                best_j[r] = j0;
            }
            float d1 = __fsub_rn(__fadd_rn(zz[r], ee1), 2.0f * acc1[r]);
            if (d1 < best_d[r]) {
                best_d[r] = d1;
                best_j[r] = j0 + 16;
            }
        }
        if (t + 1 < ntiles) lstore(nxt);
        __syncthreads();
    }

    // ---- first-minimum across the 16 lanes that share a row group ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float d = best_d[r];
        int j = best_j[r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            float d2 = __shfl_xor(d, o, 64);
            int j2 = __shfl_xor(j, o, 64);
            if (d2 < d || (d2 == d && j2 < j)) {
                d = d2;
                j = j2;
            }
        }
        long gr = row0 + wave * kRowsPerWave + 4 * g + r;
        if (li == 0 && gr < rows) {
            if (gridDim.y == 1) {
                idx_out[gr] = j;
                if (dmin_out) dmin_out[gr] = d;
            } else {
                // several code ranges per row: the smallest (distance, index) pair wins -- exactly the first minimum of a
                // sequential scan -- whatever order the blocks arrive in (idx_out holds the packed key until vq_unpack_kernel)
                atomicMin(reinterpret_cast<unsigned long long*>(idx_out) + gr, vq_key(d, j));
            }
        }
    }
}

// out[r, :] = table[idx[r], :]   (one wave per row, 16 B per lane per step)
__global__ void gather_rows_kernel(const float* __restrict__ table, const long long* __restrict__ idx, long rows,
                                   int dim, long table_rows, float* __restrict__ out_f32,
                                   bf16_t* __restrict__ out_bf16) {
    long r = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    int lane = threadIdx.x & 63;
    long src = idx[r];
    if (src < 0 || src >= table_rows) src = 0;  // host validates; never fault
    const float4* s = reinterpret_cast<const float4*>(table + src * dim);
    for (int c = lane; c < dim / 4; c += 64) {
        float4 v = s[c];
        if (out_f32) reinterpret_cast<float4*>(out_f32 + r * dim)[c] = v;
        if (out_bf16) {
            uint2 p = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
            reinterpret_cast<uint2*>(out_bf16 + r * dim)[c] = p;
        }
    }
}

}  // namespace

extern "C" int mmvid_vq_sqnorm(const float* codebook, int n, int dim, float* ee, void* stream) {
    MMVID_REQUIRE(codebook && ee && n > 0 && dim > 0, "vq_sqnorm: bad arguments");
    hipLaunchKernelGGL(vq_sqnorm_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, codebook, n, dim, ee);
    MMVID_LAUNCH_CHECK("vq_sqnorm");
    return MMVID_OK;
}

extern "C" int mmvid_vq_argmin_l2(const float* z, const float* codebook, const float* ee, int64_t rows, int n,
                                  int dim, int64_t* idx, float* dmin, void* stream) {
    MMVID_REQUIRE(z && codebook && ee && idx, "vq_argmin_l2: null pointer");
    MMVID_REQUIRE(dim == 256, "vq_argmin_l2: dim %d unsupported (embed_dim is 256 on this path)", dim);
    MMVID_REQUIRE(n > 0 && n % kTileCodes == 0, "vq_argmin_l2: n_embed %d must be a positive multiple of 32", n);
    if (rows == 0) return MMVID_OK;
    MMVID_REQUIRE(rows > 0, "vq_argmin_l2: negative row count");
    constexpr int DIM = 256;
    size_t lds = (size_t)(2 * kTileCodes * (DIM + 2) + kWaves * kRowsPerWave) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)vq_argmin_kernel<DIM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    int blocks = cdiv(rows, kWaves * kRowsPerWave);
    // A wave scans its 16 rows against every code: 64 us for 1,024 codes however few rows there are.  When the rows alone
    // cannot fill the chip (a training step's 54 frames are 108 blocks) the codes are split over gridDim.y as well and the
    // per-range minima merged with atomicMin on (distance, index) keys: same result bit for bit, a quarter of the latency.
    int splits = 1;
    while (splits < 8 && (long)blocks * splits < 256 && n / kTileCodes >= 2 * splits) splits *= 2;
    hipStream_t s = (hipStream_t)stream;
    if (splits > 1) hipLaunchKernelGGL(vq_keys_init_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, (long long*)idx, (long)rows);
    hipLaunchKernelGGL(vq_argmin_kernel<DIM>, dim3(blocks, splits), dim3(kWaves * 64), lds, s, z, codebook, ee, (long)rows, n,
                       (long long*)idx, dmin);
    if (splits > 1) hipLaunchKernelGGL(vq_unpack_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, (long long*)idx, dmin, (long)rows);
    MMVID_LAUNCH_CHECK("vq_argmin_l2");
    return MMVID_OK;
}

extern "C" int mmvid_gather_rows(const float* table, int64_t table_rows, const int64_t* idx, int64_t rows, int dim,
                                 float* out_f32, void* out_bf16, void* stream) {
    MMVID_REQUIRE(table && idx && (out_f32 || out_bf16), "gather_rows: null pointer");
    MMVID_REQUIRE(dim > 0 && dim % 4 == 0, "gather_rows: dim %d must be a multiple of 4", dim);
    if (rows == 0) return MMVID_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, table,
                       (const long long*)idx, (long)rows, dim, (long)table_rows, out_f32, (bf16_t*)out_bf16);
    MMVID_LAUNCH_CHECK("gather_rows");
    return MMVID_OK;
}
