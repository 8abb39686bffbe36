// Row normalisations on the path (HBM-bound, fp32 statistics):
//   LayerNorm forward/backward     clip_model.py:188-193 (eps 1e-5, fp32 math), dalle_bert.py:414-425 heads
//   GroupNorm(32, eps 1e-6)+swish  taming/modules/diffusionmodules/model.py:38-42, 33-35 (NHWC here)
// One wave per row for LayerNorm (E <= 64*4*MAXV lanes*float4), 16-B accesses.
#include "../../include/mmvid_hip.h"
#include "common.h"
#include <type_traits>

namespace {

constexpr int LN_MAXV = 4;  // float4 per lane -> E <= 1024

// y = (x - mean) * rstd * w + b ; writes bf16 and/or f32; saves mean/rstd.  A wave owns TWO rows (r, r + rows/2 rounded): both rows'
// loads are in flight before either is reduced and the two reductions interleave -- the kernel is bound by a wave's load -> reduce
// -> reduce -> store chain (2,606 short blocks), not by HBM.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, long ldx, long rows, int E,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ b, float eps,
                                                            bf16_t* __restrict__ y_bf16, float* __restrict__ y_f32,
                                                            long ldy, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out) {
    const long half = (rows + 1) >> 1;
    const long r0 = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r0 >= half) return;
    const long rr[2] = {r0, r0 + half};
    const bool live1 = rr[1] < rows;
    const int lane = threadIdx.x & 63;
    const int nv = E >> 2;  // float4 count
    float4 v[2][LN_MAXV];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float4* xr = reinterpret_cast<const float4*>(x + (k == 0 || live1 ? rr[k] : rr[0]) * ldx);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            v[k][i] = c < nv ? xr[c] : make_float4(0, 0, 0, 0);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) s[k] += (v[k][i].x + v[k][i].y) + (v[k][i].z + v[k][i].w);  // (columns >= E hold 0)
    const float mean[2] = {wave_sum_fast(s[0]) / (float)E, wave_sum_fast(s[1]) / (float)E};  // (DPP / permlane reductions, common.h)
    float q[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float a = v[k][i].x - mean[k], b2 = v[k][i].y - mean[k], c2 = v[k][i].z - mean[k], d = v[k][i].w - mean[k];
                q[k] += (a * a + b2 * b2) + (c2 * c2 + d * d);
            }
        }
    const float rstd[2] = {rsqrtf(wave_sum_fast(q[0]) / (float)E + eps), rsqrtf(wave_sum_fast(q[1]) / (float)E + eps)};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && !live1) break;
        const long r = rr[k];
        if (lane == 0) {
            if (mean_out) mean_out[r] = mean[k];
            if (rstd_out) rstd_out[r] = rstd[k];
        }
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 w4 = reinterpret_cast<const float4*>(w)[c];
                const float4 b4 = reinterpret_cast<const float4*>(b)[c];
                float4 o;
                o.x = (v[k][i].x - mean[k]) * rstd[k] * w4.x + b4.x;
                o.y = (v[k][i].y - mean[k]) * rstd[k] * w4.y + b4.y;
                o.z = (v[k][i].z - mean[k]) * rstd[k] * w4.z + b4.z;
                o.w = (v[k][i].w - mean[k]) * rstd[k] * w4.w + b4.w;
                if (y_f32) reinterpret_cast<float4*>(y_f32 + r * ldy)[c] = o;
                if (y_bf16) reinterpret_cast<uint2*>(y_bf16 + r * ldy)[c] = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
            }
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*w ;  dx is ADDED into dx_accum (residual-stream
// gradient) when add != 0, else stored.  dw += sum dy*xhat, db += sum dy  (fp32 atomics, one per block/column).
// dx_colsum (optional) += column sums of the UPDATED dx: that is the bias gradient of the Linear whose output
// gradient this tensor is (out_proj / c_proj of the tower), so no separate column-sum pass over it is needed.
// Each block walks rows blockIdx.x, +gridDim.x, ... with 4 waves; per-lane partial dw/db stay in registers.
// DY = float, or bf16_t: the dX GEMM that produces dy then writes half the bytes (packed 16-B stores) and this kernel reads half
template <typename DY>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const DY* __restrict__ dy, long lddy,
                                                            const float* __restrict__ x, long ldx,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ w, long rows, int E,
                                                            float* __restrict__ dx, long lddx, int add,
                                                            bf16_t* __restrict__ dx_bf16,
                                                            float* __restrict__ dw, float* __restrict__ db,
                                                            float* __restrict__ dx_colsum, float* __restrict__ partial) {
    __shared__ float red[2][4][LN_MAXV * 64 * 4];  // [dw|db][wave][E]  = 32 KiB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = E >> 2;
    float4 pw[LN_MAXV], pb[LN_MAXV], pc[LN_MAXV], w4[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        pw[i] = pb[i] = pc[i] = make_float4(0, 0, 0, 0);
        const int c = lane + 64 * i;
        w4[i] = (c < nv) ? reinterpret_cast<const float4*>(w)[c] : make_float4(0, 0, 0, 0);
    }
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const float mu = mean[r], rs = rstd[r];
        float4 g[LN_MAXV], xh[LN_MAXV], prev[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                // the residual-stream gradient this row's dx is added to: requested with the other operands (it used to be loaded
                // after the two wave reductions: a third dependent memory round trip per row)
                prev[i] = add ? reinterpret_cast<const float4*>(dx + r * lddx)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 d4;
                if constexpr (sizeof(DY) == 4) {
                    d4 = reinterpret_cast<const float4*>(dy + r * lddy)[c];
                } else {
                    const uint2 u = reinterpret_cast<const uint2*>(dy + r * lddy)[c];
                    d4 = make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
                }
                const float4 x4 = reinterpret_cast<const float4*>(x + r * ldx)[c];
                xh[i] = make_float4((x4.x - mu) * rs, (x4.y - mu) * rs, (x4.z - mu) * rs, (x4.w - mu) * rs);
                g[i] = make_float4(d4.x * w4[i].x, d4.y * w4[i].y, d4.z * w4[i].z, d4.w * w4[i].w);
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
                pw[i].x += d4.x * xh[i].x, pw[i].y += d4.y * xh[i].y, pw[i].z += d4.z * xh[i].z, pw[i].w += d4.w * xh[i].w;
                pb[i].x += d4.x, pb[i].y += d4.y, pb[i].z += d4.z, pb[i].w += d4.w;
            }
        }
        const float m1 = wave_sum_fast(s1) / (float)E, m2 = wave_sum_fast(s2) / (float)E;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o;
                o.x = rs * (g[i].x - m1 - xh[i].x * m2);
                o.y = rs * (g[i].y - m1 - xh[i].y * m2);
                o.z = rs * (g[i].z - m1 - xh[i].z * m2);
                o.w = rs * (g[i].w - m1 - xh[i].w * m2);
                float4* d = reinterpret_cast<float4*>(dx + r * lddx) + c;
                o.x += prev[i].x, o.y += prev[i].y, o.z += prev[i].z, o.w += prev[i].w;
                *d = o;
                pc[i].x += o.x, pc[i].y += o.y, pc[i].z += o.z, pc[i].w += o.w;
                if (dx_bf16)  // bf16 copy of the updated residual gradient: the next backward GEMMs' operand
                    reinterpret_cast<uint2*>(dx_bf16 + r * lddx)[c] = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
            }
        }
    }
    if (dw || db) {
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                reinterpret_cast<float4*>(red[0][wave])[c] = pw[i];
                reinterpret_cast<float4*>(red[1][wave])[c] = pb[i];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < E; e += 256) {
            const float sw = (red[0][0][e] + red[0][1][e]) + (red[0][2][e] + red[0][3][e]);
            const float sb = (red[1][0][e] + red[1][1][e]) + (red[1][2][e] + red[1][3][e]);
            if (partial) {  // two-stage reduction: this block's row of the [blocks][3][E] workspace (no atomics)
                partial[((long)blockIdx.x * 3 + 0) * E + e] = sw;
                partial[((long)blockIdx.x * 3 + 1) * E + e] = sb;
            } else {
                if (dw) unsafeAtomicAdd(dw + e, sw);
                if (db) unsafeAtomicAdd(db + e, sb);
            }
        }
    }
    if (dx_colsum) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) reinterpret_cast<float4*>(red[0][wave])[c] = pc[i];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < E; e += 256) {
            const float sc = (red[0][0][e] + red[0][1][e]) + (red[0][2][e] + red[0][3][e]);
            if (partial)
                partial[((long)blockIdx.x * 3 + 2) * E + e] = sc;
            else
                unsafeAtomicAdd(dx_colsum + e, sc);
        }
    }
}

// Round 5: the same backward for E = 256 * NVI (768: the tower; 512: the text tower), software-pipelined.  The generic kernel's wave runs
// load -> two wave reductions -> store per row with nothing in flight in between (2 waves per SIMD: 36 us for 128 MB, 3.6 TB/s); here
// the NEXT row's operands are requested before this row is reduced and stored, the row loop is branch-free (no per-lane column guard:
// every lane owns NVI float4 of the row) so the compiler's waits are counted (vmcnt(n) for the row in use, the prefetch stays in
// flight), and the row index is wave-uniform (scalar loads of mean / rstd).  The arithmetic and its order are the generic kernel's:
// bit-identical results.
template <typename DY, int NVI>
__global__ __launch_bounds__(256) void layernorm_bwd_fast_kernel(const DY* __restrict__ dy, long lddy, const float* __restrict__ x,
                                                                 long ldx, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ w, long rows,
                                                                 float* __restrict__ dx, long lddx, int add,
                                                                 bf16_t* __restrict__ dx_bf16, float* __restrict__ dw,
                                                                 float* __restrict__ db, float* __restrict__ dx_colsum,
                                                                 float* __restrict__ partial) {
    constexpr int E = 256 * NVI;
    __shared__ float red[2][4][E];
    typedef typename std::conditional<sizeof(DY) == 4, float4, uint2>::type dy_raw_t;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 pw[NVI], pb[NVI], pc[NVI], w4[NVI];
#pragma unroll
    for (int i = 0; i < NVI; ++i) {
        pw[i] = pb[i] = pc[i] = make_float4(0, 0, 0, 0);
        w4[i] = reinterpret_cast<const float4*>(w)[lane + 64 * i];
    }
    struct RowIn {
        float4 prev[NVI], x4[NVI];
        dy_raw_t d[NVI];
        float mu, rs;
    };
    const long rstep = (long)gridDim.x * 4;
    auto load_row = [&](RowIn& in, long r) {
        in.mu = mean[r], in.rs = rstd[r];
#pragma unroll
        for (int i = 0; i < NVI; ++i) {
            const int c = lane + 64 * i;
            in.prev[i] = reinterpret_cast<const float4*>(dx + r * lddx)[c];
            in.d[i] = reinterpret_cast<const dy_raw_t*>(dy + r * lddy)[c];
            in.x4[i] = reinterpret_cast<const float4*>(x + r * ldx)[c];
        }
    };
    RowIn cur, nxt;
    long r = (long)blockIdx.x * 4 + wave;
    if (r < rows) load_row(nxt, r);
    for (; r < rows; r += rstep) {
        cur = nxt;
        load_row(nxt, r + rstep < rows ? r + rstep : r);  // (the last row of a wave re-requests itself: no branch around the loads)
        const float mu = cur.mu, rs = cur.rs;
        float4 g[NVI], xh[NVI];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NVI; ++i) {
            float4 d4;
            if constexpr (sizeof(DY) == 4) {
                d4 = cur.d[i];
            } else {
                const uint2 u = cur.d[i];
                d4 = make_float4(bf_lo(u.x), bf_hi(u.x), bf_lo(u.y), bf_hi(u.y));
            }
            const float4 x4 = cur.x4[i];
            xh[i] = make_float4((x4.x - mu) * rs, (x4.y - mu) * rs, (x4.z - mu) * rs, (x4.w - mu) * rs);
            g[i] = make_float4(d4.x * w4[i].x, d4.y * w4[i].y, d4.z * w4[i].z, d4.w * w4[i].w);
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
            pw[i].x += d4.x * xh[i].x, pw[i].y += d4.y * xh[i].y, pw[i].z += d4.z * xh[i].z, pw[i].w += d4.w * xh[i].w;
            pb[i].x += d4.x, pb[i].y += d4.y, pb[i].z += d4.z, pb[i].w += d4.w;
        }
        const float m1 = wave_sum_fast(s1) / (float)E, m2 = wave_sum_fast(s2) / (float)E;
#pragma unroll
        for (int i = 0; i < NVI; ++i) {
            const int c = lane + 64 * i;
            float4 o;
            o.x = rs * (g[i].x - m1 - xh[i].x * m2);
            o.y = rs * (g[i].y - m1 - xh[i].y * m2);
            o.z = rs * (g[i].z - m1 - xh[i].z * m2);
            o.w = rs * (g[i].w - m1 - xh[i].w * m2);
            if (add) o.x += cur.prev[i].x, o.y += cur.prev[i].y, o.z += cur.prev[i].z, o.w += cur.prev[i].w;
            reinterpret_cast<float4*>(dx + r * lddx)[c] = o;
            pc[i].x += o.x, pc[i].y += o.y, pc[i].z += o.z, pc[i].w += o.w;
            if (dx_bf16) reinterpret_cast<uint2*>(dx_bf16 + r * lddx)[c] = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
        }
    }
    if (dw || db) {
#pragma unroll
        for (int i = 0; i < NVI; ++i) {
            reinterpret_cast<float4*>(red[0][wave])[lane + 64 * i] = pw[i];
            reinterpret_cast<float4*>(red[1][wave])[lane + 64 * i] = pb[i];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < E; e += 256) {
            const float sw = (red[0][0][e] + red[0][1][e]) + (red[0][2][e] + red[0][3][e]);
            const float sb = (red[1][0][e] + red[1][1][e]) + (red[1][2][e] + red[1][3][e]);
            if (partial) {
                partial[((long)blockIdx.x * 3 + 0) * E + e] = sw;
                partial[((long)blockIdx.x * 3 + 1) * E + e] = sb;
            } else {
                if (dw) unsafeAtomicAdd(dw + e, sw);
                if (db) unsafeAtomicAdd(db + e, sb);
            }
        }
    }
    if (dx_colsum) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NVI; ++i) reinterpret_cast<float4*>(red[0][wave])[lane + 64 * i] = pc[i];
        __syncthreads();
        for (int e = threadIdx.x; e < E; e += 256) {
            const float sc = (red[0][0][e] + red[0][1][e]) + (red[0][2][e] + red[0][3][e]);
            if (partial)
                partial[((long)blockIdx.x * 3 + 2) * E + e] = sc;
            else
                unsafeAtomicAdd(dx_colsum + e, sc);
        }
    }
}

// second stage: out_which[e] += sum over blocks of partial[b][which][e], fixed order (deterministic).  The sum is latency
// bound (a few MB, one dependent chain per thread), so the block is wide and shallow: 32 columns x 32 row groups, every
// thread keeps 16 independent loads in flight; grid = (ceil(E/32), 3).  (36 blocks of 4 row groups took 66 us per call.)
__global__ __launch_bounds__(1024) void layernorm_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int E,
                                                                    float* __restrict__ dw, float* __restrict__ db,
                                                                    float* __restrict__ dx_colsum) {
    __shared__ float sh[32][33];
    const int which = blockIdx.y;
    float* out = which == 0 ? dw : (which == 1 ? db : dx_colsum);
    if (!out) return;
    const int lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + lane;
    float a = 0.f;
    if (col < E) {
        const float* src = partial + (long)which * E + col;
        int b = rg;
        for (; b + 15 * 32 < nblocks; b += 16 * 32) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = src[(long)(b + i * 32) * 3 * E];
#pragma unroll
            for (int i = 0; i < 16; ++i) a += v[i];
        }
        for (; b < nblocks; b += 32) a += src[(long)b * 3 * E];
    }
    sh[rg][lane] = a;
    __syncthreads();
    if (rg == 0 && col < E) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += sh[i][lane];
        out[col] += t;
    }
}

// the same second stage for SEVERAL LayerNorm backwards in one launch (blockIdx.z = which one): the tower backward defers the
// reductions of its 2 x layers LayerNorms to the end of the layer loop (24 launches of ~5 us, each a dependent step of the chain,
// become one); the entries travel in the kernel arguments
constexpr int LN_MULTI_MAX = 32;
struct LnReduceTable {
    mmvid_ln_reduce_t item[LN_MULTI_MAX];
};
__global__ __launch_bounds__(1024) void layernorm_bwd_reduce_multi_kernel(LnReduceTable t, int nblocks, int E) {
    __shared__ float sh[32][33];
    const int which = blockIdx.y;
    const float* partial = t.item[blockIdx.z].partial;
    float* out = which == 0 ? t.item[blockIdx.z].dw : (which == 1 ? t.item[blockIdx.z].db : t.item[blockIdx.z].dx_colsum);
    if (!out) return;
    const int lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + lane;
    float a = 0.f;
    if (col < E) {
        const float* src = partial + (long)which * E + col;
        int b = rg;
        for (; b + 15 * 32 < nblocks; b += 16 * 32) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = src[(long)(b + i * 32) * 3 * E];
#pragma unroll
            for (int i = 0; i < 16; ++i) a += v[i];
        }
        for (; b < nblocks; b += 32) a += src[(long)b * 3 * E];
    }
    sh[rg][lane] = a;
    __syncthreads();
    if (rg == 0 && col < E) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += sh[i][lane];
        out[col] += s;
    }
}

// ---------------------------------------------------------------- GroupNorm(32) on NHWC
// stats pass: one block per (image n, pixel chunk); accumulates per-group sum / sumsq with fp32 atomics
// into stats[n][32][2].  C = 32*cpg channels, cpg in {1,4,8,16}.
template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&f)[8]);
template <>
__device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&f)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    f[0] = bf_lo(u.x), f[1] = bf_hi(u.x), f[2] = bf_lo(u.y), f[3] = bf_hi(u.y);
    f[4] = bf_lo(u.z), f[5] = bf_hi(u.z), f[6] = bf_lo(u.w), f[7] = bf_hi(u.w);
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&f)[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    f[0] = a.x, f[1] = a.y, f[2] = a.z, f[3] = a.w, f[4] = b.x, f[5] = b.y, f[6] = b.z, f[7] = b.w;
}

// Thread t handles channel chunk (t % (C/8)) for pixels t / (C/8) + k * (256 / (C/8)).  Requires C%8==0,
// C/8 <= 256 and 256 % (C/8) == 0  (C in {32, 64, 128, 256, 512}).
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const T* __restrict__ x, long hw, int C,
                                                              int pix_per_block, float* __restrict__ partial) {
    // Deterministic (no atomics): per-thread partials -> LDS [pixel row][channel] -> fixed-order column sums
    // -> per-group sums -> partial[n][block][32][2]; groupnorm_finalize_kernel adds the blocks in order.
    __shared__ float red[2][2048];  // [sum|sumsq][prow * C + channel], pstep * C == 2048
    __shared__ float sh[2][512];    // per-channel sums
    const int n = blockIdx.y;
    const int cchunks = C >> 3;
    const int cc = threadIdx.x % cchunks, prow = threadIdx.x / cchunks, pstep = 256 / cchunks;
    const long p0 = (long)blockIdx.x * pix_per_block;
    long p1 = p0 + pix_per_block;
    if (p1 > hw) p1 = hw;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    const T* base = x + ((long)n * hw) * C + cc * 8;
    for (long p = p0 + prow; p < p1; p += pstep) {
        float f[8];
        load8<T>(base + p * C, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e], q[e] += f[e] * f[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[0][prow * C + cc * 8 + e] = s[e];
        red[1][prow * C + cc * 8 + e] = q[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        const int which = c / C, ch = c - which * C;
        float a = 0.f;
        for (int r = 0; r < pstep; ++r) a += red[which][r * C + ch];
        sh[which][ch] = a;
    }
    __syncthreads();
    const int cpg = C / 32;
    if (threadIdx.x < 64) {
        const int grp = threadIdx.x & 31, which = threadIdx.x >> 5;
        float a = 0.f;
        for (int e = 0; e < cpg; ++e) a += sh[which][grp * cpg + e];
        partial[(((long)n * gridDim.x + blockIdx.x) * 32 + grp) * 2 + which] = a;
    }
}

// Per (image, channel) affine of the normalisation: y = x*a + b with a = rstd*w, b = bias - mean*a.  One wave per
// (image, group): lanes stride over the partial blocks (partial[n][blk][grp][which]) and are combined by a fixed
// xor-butterfly, so the result does not depend on scheduling.  ab[n][C][2].
__global__ __launch_bounds__(256) void groupnorm_finalize_kernel(const float* __restrict__ partial, int nblk, int N,
                                                                 int C, float cnt, float eps,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ b, float* __restrict__ ab) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);  // (n, grp)
    if (item >= N * 32) return;
    const int lane = threadIdx.x & 63;
    const int n = item >> 5, grp = item & 31;
    float sm = 0.f, sq = 0.f;
    for (int k = lane; k < nblk; k += 64) {
        const float2 v = *reinterpret_cast<const float2*>(partial + (((long)n * nblk + k) * 32 + grp) * 2);
        sm += v.x, sq += v.y;
    }
    sm = wave_sum(sm), sq = wave_sum(sq);
    const float mu = sm / cnt;
    float var = sq / cnt - mu * mu;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + eps);
    const int cpg = C >> 5;
    if (lane < cpg) {
        const int ch = grp * cpg + lane;
        const float a = rstd * w[ch];
        *reinterpret_cast<float2*>(ab + ((long)n * C + ch) * 2) = make_float2(a, b[ch] - mu * a);
    }
}

// apply: y = swish?(x*a + b) -> bf16 NHWC (and/or f32).  8 channels per thread, 16-B accesses.
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const T* __restrict__ x, long hw, int C,
                                                              const float* __restrict__ ab, int swish,
                                                              bf16_t* __restrict__ y_bf16,
                                                              float* __restrict__ y_f32, long total_chunks) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total_chunks) return;
    const int cchunks = C >> 3;
    const int cc = (int)(t % cchunks);
    const long pix = t / cchunks;  // n*hw + p
    const int n = (int)(pix / hw);
    float f[8];
    load8<T>(x + pix * C + cc * 8, f);
    const float4* q = reinterpret_cast<const float4*>(ab + ((long)n * C + cc * 8) * 2);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const float av[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
    const float bv[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = f[e] * av[e] + bv[e];
        // swish on the hardware exp2 / rcp (1 ulp each; the result is rounded to bf16 right after): the IEEE division of sigmoidf_ is ~10
        // VALU per element, and with 16-B loads the bf16-input pass was VALU-bound (4.46 TB/s against 5.86 for the fp32-input pass)
        if (swish) v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
        o[e] = v;
    }
    if (y_bf16)
        *reinterpret_cast<uint4*>(y_bf16 + pix * C + cc * 8) =
            make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
    if (y_f32) {
        float4* d = reinterpret_cast<float4*>(y_f32 + pix * C + cc * 8);
        d[0] = make_float4(o[0], o[1], o[2], o[3]);
        d[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// Small maps (hw <= 256 pixels: the 8x8 and 16x16 levels of the VQGAN): statistics, finalisation and the apply pass in ONE launch -- the
// three launches of the general path were 15 us per GroupNorm for 4 us of work, ten times per encode.  GroupNorm's groups are independent,
// so a block takes one image's channels [64 j, 64 j + 64) (whole groups: C / 32 channels each) -- C / 64 blocks per image instead of one
// (round 5: with one block per image the 54-frame encode ran 54 blocks on 256 CUs, 13.4 us per launch for 3.5 MB; the kernel is a chain
// of two global round trips and three block barriers, so more, smaller blocks shorten it).  Same formulas as the general path
// (groupnorm_stats_kernel's sums, groupnorm_finalize_kernel's statistics, groupnorm_apply_kernel's element math); the order of the
// per-channel pixel sums is this kernel's own (32 pixel rows per block), fixed: bit-reproducible, independent of the batch.
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_small_fused_kernel(const T* __restrict__ x, long hw, int C, float cnt, float eps,
                                                                    const float* __restrict__ w, const float* __restrict__ b, int swish,
                                                                    bf16_t* __restrict__ y_bf16, float* __restrict__ y_f32) {
    __shared__ float red[2][2048];  // [sum | sum of squares][pixel row][channel of the block]
    __shared__ float sh[2][64];
    __shared__ float ab[64][2];
    const int n = blockIdx.x;
    const int CB = C < 64 ? C : 64, c0 = blockIdx.y * CB;  // this block's channels
    const int cchunks = CB >> 3;
    const int cc = threadIdx.x % cchunks, prow = threadIdx.x / cchunks, pstep = 256 / cchunks;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    const T* base = x + ((long)n * hw) * C + c0 + cc * 8;
    for (long p = prow; p < hw; p += pstep) {
        float f[8];
        load8<T>(base + p * C, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e], q[e] += f[e] * f[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[0][prow * CB + cc * 8 + e] = s[e];
        red[1][prow * CB + cc * 8 + e] = q[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * CB; c += 256) {
        const int which = c / CB, ch = c - which * CB;
        float a = 0.f;
        for (int r = 0; r < pstep; ++r) a += red[which][r * CB + ch];
        sh[which][ch] = a;
    }
    __syncthreads();
    const int cpg = C / 32;
    if (threadIdx.x < CB / cpg) {  // one lane per group of this block
        const int grp = threadIdx.x;
        float sm = 0.f, sq = 0.f;
        for (int e = 0; e < cpg; ++e) sm += sh[0][grp * cpg + e];
        for (int e = 0; e < cpg; ++e) sq += sh[1][grp * cpg + e];
        const float mu = sm / cnt;
        float var = sq / cnt - mu * mu;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + eps);
        for (int e = 0; e < cpg; ++e) {
            const int ch = grp * cpg + e;
            const float a = rstd * w[c0 + ch];
            ab[ch][0] = a, ab[ch][1] = b[c0 + ch] - mu * a;
        }
    }
    __syncthreads();
    float av[8], bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) av[e] = ab[cc * 8 + e][0], bv[e] = ab[cc * 8 + e][1];
    for (long p = prow; p < hw; p += pstep) {
        float f[8], o[8];
        load8<T>(base + p * C, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = f[e] * av[e] + bv[e];
            if (swish) v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
            o[e] = v;
        }
        const long at = ((long)n * hw + p) * C + c0 + cc * 8;
        if (y_bf16)
            *reinterpret_cast<uint4*>(y_bf16 + at) = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
        if (y_f32) {
            float4* d = reinterpret_cast<float4*>(y_f32 + at);
            d[0] = make_float4(o[0], o[1], o[2], o[3]);
            d[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// ---- split operator (vae.strict = 'split', conv.hip): fp32 input, statistics finalised in fp64, output as a bf16 PAIR
// (mean, rstd) per (image, channel), stored per channel so the apply kernel reads them with the channel chunk: mr[n][C][2]
__global__ __launch_bounds__(256) void groupnorm_finalize_f64_kernel(const float* __restrict__ partial, int nblk, int N, int C,
                                                                     double cnt, float eps, float* __restrict__ mr) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);  // (n, grp)
    if (item >= N * 32) return;
    const int lane = threadIdx.x & 63;
    const int n = item >> 5, grp = item & 31;
    double sm = 0.0, sq = 0.0;
    for (int k = lane; k < nblk; k += 64) {
        const float2 v = *reinterpret_cast<const float2*>(partial + (((long)n * nblk + k) * 32 + grp) * 2);
        sm += (double)v.x, sq += (double)v.y;
    }
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64), sq += __shfl_xor(sq, o, 64);  // fixed butterfly
    const double mu = sm / cnt;
    double var = sq / cnt - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const int cpg = C >> 5;
    if (lane < cpg) *reinterpret_cast<float2*>(mr + ((long)n * C + grp * cpg + lane) * 2) = make_float2((float)mu, (float)rstd);
}

// y = swish?(((x - mean) * rstd) * w + b) evaluated like ATen (expf, true division), stored as hi = bf16(y), lo = bf16(y - hi)
// in planes [2][N*hw*C]; 8 channels per thread.  F16: ONE plane of IEEE-half values instead (the input of the fp16 single-product convolutions)
template <bool F16>
__global__ __launch_bounds__(256) void groupnorm_apply_split_kernel(const float* __restrict__ x, long hw, int C,
                                                                    const float* __restrict__ mr, const float* __restrict__ w,
                                                                    const float* __restrict__ b, int swish,
                                                                    bf16_t* __restrict__ planes, long total_chunks) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total_chunks) return;
    const int cchunks = C >> 3;
    const int cc = (int)(t % cchunks);
    const long pix = t / cchunks;
    const int n = (int)(pix / hw);
    float f[8], wv[8], bv[8];
    load8<float>(x + pix * C + cc * 8, f);
    load8<float>(w + cc * 8, wv);
    load8<float>(b + cc * 8, bv);
    const float4* q = reinterpret_cast<const float4*>(mr + ((long)n * C + cc * 8) * 2);
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const float mu[8] = {q0.x, q0.z, q1.x, q1.z, q2.x, q2.z, q3.x, q3.z};
    const float rs[8] = {q0.y, q0.w, q1.y, q1.w, q2.y, q2.w, q3.y, q3.w};
    uint32_t h[4], l[4];
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = ((f[e] - mu[e]) * rs[e]) * wv[e] + bv[e];
        if (swish) v = v * (1.0f / (1.0f + expf(-v)));
        o[e] = v;
    }
    if constexpr (F16) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
            const h2_t p2 = {(_Float16)o[2 * e], (_Float16)o[2 * e + 1]};  // round to nearest even
            h[e] = __builtin_bit_cast(uint32_t, p2);
        }
        *reinterpret_cast<uint4*>(planes + t * 8) = make_uint4(h[0], h[1], h[2], h[3]);
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = pack_bf2(o[2 * e], o[2 * e + 1]);
        l[e] = pack_bf2(o[2 * e] - bf_lo(h[e]), o[2 * e + 1] - bf_hi(h[e]));
    }
    *reinterpret_cast<uint4*>(planes + t * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(planes + (total_chunks + t) * 8) = make_uint4(l[0], l[1], l[2], l[3]);
}

}  // namespace

extern "C" int mmvid_layernorm_fwd(const float* x, int64_t ldx, int64_t rows, int E, const float* w, const float* b,
                                   float eps, void* y_bf16, float* y_f32, int64_t ldy, float* mean, float* rstd,
                                   void* stream) {
    MMVID_REQUIRE(x && w && b && (y_bf16 || y_f32), "layernorm_fwd: null pointer");
    MMVID_REQUIRE(E % 4 == 0 && E <= 64 * 4 * LN_MAXV && ldx % 4 == 0 && ldy % 4 == 0,
                  "layernorm_fwd: E=%d must be a multiple of 4 and <= %d; row strides multiples of 4", E,
                  64 * 4 * LN_MAXV);
    if (rows == 0) return MMVID_OK;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv((rows + 1) / 2, 4)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx,
                       (long)rows, E, w, b, eps, (bf16_t*)y_bf16, y_f32, (long)ldy, mean, rstd);
    MMVID_LAUNCH_CHECK("layernorm_fwd");
    return MMVID_OK;
}

extern "C" int mmvid_layernorm_bwd_ws(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                      const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                                      int add_into_dx, void* dx_bf16, float* dw, float* db, float* dx_colsum,
                                      float* workspace, int64_t workspace_floats, void* stream);
extern "C" int mmvid_layernorm_bwd_ex(const void* dy, int dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                      const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                                      int add_into_dx, void* dx_bf16, float* dw, float* db, float* dx_colsum,
                                      float* workspace, int64_t workspace_floats, void* stream);

extern "C" int mmvid_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                   const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                                   int add_into_dx, void* dx_bf16, float* dw, float* db, float* dx_colsum,
                                   void* stream) {
    return mmvid_layernorm_bwd_ws(dy, lddy, x, ldx, mean, rstd, w, rows, E, dx, lddx, add_into_dx, dx_bf16, dw, db, dx_colsum,
                                  nullptr, 0, stream);
}

// workspace (optional): fp32 scratch of workspace_floats >= 3 * E * blocks; with it the weight / bias / column-sum
// gradients are reduced in two stages (per-block rows, then a fixed-order column reduction): no atomics, deterministic,
// and the grid no longer has to be kept small to bound the atomic traffic (more waves in flight -> closer to HBM speed).
extern "C" int mmvid_layernorm_bwd_ws(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                      const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                                      int add_into_dx, void* dx_bf16, float* dw, float* db, float* dx_colsum,
                                      float* workspace, int64_t workspace_floats, void* stream) {
    return mmvid_layernorm_bwd_ex(dy, 0, lddy, x, ldx, mean, rstd, w, rows, E, dx, lddx, add_into_dx, dx_bf16, dw, db, dx_colsum,
                                  workspace, workspace_floats, stream);
}

// stage 1 (+ stage 2 unless blocks_out is given: then the caller reduces the partial rows later, mmvid_layernorm_bwd_reduce_multi)
static int layernorm_bwd_impl(const void* dy, int dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                              const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx, int add_into_dx,
                              void* dx_bf16, float* dw, float* db, float* dx_colsum, float* workspace, int64_t workspace_floats,
                              int* blocks_out, void* stream) {
    MMVID_REQUIRE(dy && x && mean && rstd && w && dx, "layernorm_bwd: null pointer");
    MMVID_REQUIRE(E % 4 == 0 && E <= 64 * 4 * LN_MAXV && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0,
                  "layernorm_bwd: bad E/strides");
    if (rows == 0) return MMVID_OK;
    int blocks = cdiv(rows, 4);
    const bool any_red = dw || db || dx_colsum;
    float* partial = nullptr;
    if (workspace && any_red) {
        int cap = (int)(workspace_floats / (3 * (int64_t)E));
        if (cap > 2048) cap = 2048;
        if (cap >= 64) {
            if (blocks > cap) blocks = cap;
            partial = workspace;
        }
    }
    if (!partial) {
        // one-stage form (no workspace): the dw / db atomics scale with the grid -- measured on the whole step 2048 blocks: +0.6 ms,
        // 1024: +0.11 ms, 256: +0.25 ms against 512
        if (blocks > 512) blocks = 512;
    }
    // in two-stage mode the kernel needs non-null dw/db to take the reduction branch at all
    float* const a_dw = partial ? (dw ? dw : workspace) : dw;
    float* const a_db = partial ? (db ? db : workspace) : db;
    float* const a_cs = partial ? (dx_colsum ? dx_colsum : nullptr) : dx_colsum;
    // E = 512 / 768 (the towers): the software-pipelined instance (the same arithmetic as the generic kernel up to fma contraction: dx
    // within 1 ulp, dw / db partial rows equal); the prev row is always read, so `dx` must be readable memory even when it is only
    // stored to (add_into_dx = 0) -- it is the output buffer: it is
    if (E == 768 || E == 512) {
#define MMVID_LN_FAST(DYT, NVI)                                                                                                            \
    hipLaunchKernelGGL((layernorm_bwd_fast_kernel<DYT, NVI>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const DYT*)dy, (long)lddy, x, \
                       (long)ldx, mean, rstd, w, (long)rows, dx, (long)lddx, add_into_dx, (bf16_t*)dx_bf16, a_dw, a_db, a_cs, partial)
        if (dy_is_bf16) {
            if (E == 768) MMVID_LN_FAST(bf16_t, 3); else MMVID_LN_FAST(bf16_t, 2);
        } else {
            if (E == 768) MMVID_LN_FAST(float, 3); else MMVID_LN_FAST(float, 2);
        }
#undef MMVID_LN_FAST
    } else if (dy_is_bf16)
        hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (long)lddy, x,
                           (long)ldx, mean, rstd, w, (long)rows, E, dx, (long)lddx, add_into_dx, (bf16_t*)dx_bf16,
                           partial ? (dw ? dw : workspace) : dw, partial ? (db ? db : workspace) : db,
                           partial ? (dx_colsum ? dx_colsum : nullptr) : dx_colsum, partial);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (long)lddy, x,
                           (long)ldx, mean, rstd, w, (long)rows, E, dx, (long)lddx, add_into_dx, (bf16_t*)dx_bf16,
                           partial ? (dw ? dw : workspace) : dw, partial ? (db ? db : workspace) : db,
                           partial ? (dx_colsum ? dx_colsum : nullptr) : dx_colsum, partial);
    if (blocks_out) {
        MMVID_REQUIRE(partial || !any_red, "layernorm_bwd_partial: needs a workspace of at least 64 * 3 * E floats");
        *blocks_out = blocks;
    } else if (partial) {
        hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3(cdiv(E, 32), 3), dim3(1024), 0, (hipStream_t)stream, partial, blocks, E,
                           dw, db, dx_colsum);
    }
    MMVID_LAUNCH_CHECK("layernorm_bwd");
    return MMVID_OK;
}

extern "C" int mmvid_layernorm_bwd_ex(const void* dy, int dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                      const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                                      int add_into_dx, void* dx_bf16, float* dw, float* db, float* dx_colsum,
                                      float* workspace, int64_t workspace_floats, void* stream) {
    return layernorm_bwd_impl(dy, dy_is_bf16, lddy, x, ldx, mean, rstd, w, rows, E, dx, lddx, add_into_dx, dx_bf16, dw, db, dx_colsum,
                              workspace, workspace_floats, nullptr, stream);
}

// Stage 1 only: dx (and dx_bf16) are final; the weight / bias / column-sum gradients stay as `*blocks_out` partial rows
// [blocks][3][E] in `workspace` (which must therefore be this call's own).  want_dw / want_db / want_colsum say which of the three
// the later reduction will take (the kernel skips the others).
extern "C" int mmvid_layernorm_bwd_partial(const void* dy, int dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                           const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                                           int add_into_dx, void* dx_bf16, int want_dw, int want_db, int want_colsum,
                                           float* workspace, int64_t workspace_floats, int* blocks_out, void* stream) {
    MMVID_REQUIRE(workspace && blocks_out, "layernorm_bwd_partial: null workspace / blocks_out");
    // (non-null markers: in two-stage mode the kernel only tests these pointers, everything goes to the partial rows)
    return layernorm_bwd_impl(dy, dy_is_bf16, lddy, x, ldx, mean, rstd, w, rows, E, dx, lddx, add_into_dx, dx_bf16,
                              want_dw ? workspace : nullptr, want_db ? workspace : nullptr, want_colsum ? workspace : nullptr, workspace,
                              workspace_floats, blocks_out, stream);
}

// out[e] += sum over the partial rows, for n LayerNorm backwards in one launch (fixed order: the arithmetic of the single reduction)
extern "C" int mmvid_layernorm_bwd_reduce_multi(int n, const mmvid_ln_reduce_t* items, int blocks, int E, void* stream) {
    MMVID_REQUIRE(n >= 0 && (n == 0 || items) && blocks > 0 && E > 0, "layernorm_bwd_reduce_multi: bad arguments");
    for (int i0 = 0; i0 < n; i0 += LN_MULTI_MAX) {
        const int m = n - i0 < LN_MULTI_MAX ? n - i0 : LN_MULTI_MAX;
        LnReduceTable t;
        for (int i = 0; i < LN_MULTI_MAX; ++i) {
            if (i < m) {
                MMVID_REQUIRE(items[i0 + i].partial, "layernorm_bwd_reduce_multi: entry %d has no partial rows", i0 + i);
                t.item[i] = items[i0 + i];
            } else {
                t.item[i].partial = nullptr, t.item[i].dw = t.item[i].db = t.item[i].dx_colsum = nullptr;
            }
        }
        hipLaunchKernelGGL(layernorm_bwd_reduce_multi_kernel, dim3(cdiv(E, 32), 3, m), dim3(1024), 0, (hipStream_t)stream, t, blocks, E);
    }
    MMVID_LAUNCH_CHECK("layernorm_bwd_reduce_multi");
    return MMVID_OK;
}

// x NHWC [N, hw, C] (bf16 when x_is_bf16 else f32).  stats_scratch: fp32 [N*(2*C + 64*ceil(hw/128))]
// (per-channel affine first, then the per-block partial sums).  partial_blocks > 0: the producer (conv epilogue)
// already wrote that many partial blocks per image; 0: a statistics pass runs here.  Deterministic: no atomics.
extern "C" int mmvid_groupnorm_swish_nhwc(const void* x, int x_is_bf16, int N, int64_t hw, int C, const float* w,
                                          const float* b, float eps, int swish, float* stats_scratch,
                                          int partial_blocks, void* y_bf16, float* y_f32, void* stream) {
    MMVID_REQUIRE(x && w && b && stats_scratch && (y_bf16 || y_f32), "groupnorm: null pointer");
    MMVID_REQUIRE(C % 32 == 0 && C <= 512 && 256 % (C / 8) == 0, "groupnorm: C=%d unsupported", C);
    MMVID_REQUIRE(partial_blocks >= 0 && partial_blocks <= cdiv(hw, 64), "groupnorm: partial_blocks %d", partial_blocks);
    if (N == 0 || hw == 0) return MMVID_OK;
    hipStream_t s = (hipStream_t)stream;
    const int pix_per_block = 256;
    int nblk = partial_blocks;
    float* ab = stats_scratch;                         // [N][C][2]
    float* partial = stats_scratch + (long)N * C * 2;  // [N][nblk][32][2]
    const long chunks = (long)N * hw * (C / 8);
    if (partial_blocks == 0 && hw <= pix_per_block && (C & (C - 1)) == 0) {  // one launch does all three steps (whole groups per 64-channel block)
        if (x_is_bf16)
            hipLaunchKernelGGL(groupnorm_small_fused_kernel<bf16_t>, dim3(N, C < 64 ? 1 : C / 64), dim3(256), 0, s, (const bf16_t*)x, (long)hw, C,
                               (float)hw * (float)(C / 32), eps, w, b, swish, (bf16_t*)y_bf16, y_f32);
        else
            hipLaunchKernelGGL(groupnorm_small_fused_kernel<float>, dim3(N, C < 64 ? 1 : C / 64), dim3(256), 0, s, (const float*)x, (long)hw, C,
                               (float)hw * (float)(C / 32), eps, w, b, swish, (bf16_t*)y_bf16, y_f32);
        MMVID_LAUNCH_CHECK("groupnorm");
        return MMVID_OK;
    }
    if (partial_blocks == 0) {
        nblk = cdiv(hw, pix_per_block);
        dim3 g1(nblk, N);
        if (x_is_bf16)
            hipLaunchKernelGGL(groupnorm_stats_kernel<bf16_t>, g1, dim3(256), 0, s, (const bf16_t*)x, (long)hw, C,
                               pix_per_block, partial);
        else
            hipLaunchKernelGGL(groupnorm_stats_kernel<float>, g1, dim3(256), 0, s, (const float*)x, (long)hw, C,
                               pix_per_block, partial);
    }
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(cdiv((long)N * 32, 4)), dim3(256), 0, s, partial, nblk, N, C,
                       (float)hw * (float)(C / 32), eps, w, b, ab);
    if (x_is_bf16)
        hipLaunchKernelGGL(groupnorm_apply_kernel<bf16_t>, dim3(cdiv(chunks, 256)), dim3(256), 0, s,
                           (const bf16_t*)x, (long)hw, C, ab, swish, (bf16_t*)y_bf16, y_f32, chunks);
    else
        hipLaunchKernelGGL(groupnorm_apply_kernel<float>, dim3(cdiv(chunks, 256)), dim3(256), 0, s, (const float*)x,
                           (long)hw, C, ab, swish, (bf16_t*)y_bf16, y_f32, chunks);
    MMVID_LAUNCH_CHECK("groupnorm");
    return MMVID_OK;
}

// GroupNorm(32) [+ swish] of the split operator: x fp32 NHWC -> bf16 pair planes [2][N,hw,C].  Partial sums in fp32 -- per 256
// pixels by a statistics pass here (partial_blocks = 0), or per 128 / 64 pixels by the producing convolution's epilogue
// (partial_blocks = hw/128 or hw/64) -- combined and finalised in fp64.  stats_scratch: fp32 [N*(2*C + 64*ceil(hw/64))].
static int groupnorm_split_launch(const float* x, int N, int64_t hw, int C, const float* w, const float* b, float eps, int swish,
                                  float* stats_scratch, int partial_blocks, void* planes_bf16, void* stream, bool f16);

extern "C" int mmvid_groupnorm_swish_nhwc_split(const float* x, int N, int64_t hw, int C, const float* w, const float* b, float eps,
                                                int swish, float* stats_scratch, int partial_blocks, void* planes_bf16, void* stream) {
    return groupnorm_split_launch(x, N, hw, C, w, b, eps, swish, stats_scratch, partial_blocks, planes_bf16, stream, false);
}
// the same statistics and arithmetic, the result stored as ONE plane of IEEE-half values [N*hw*C]
extern "C" int mmvid_groupnorm_swish_nhwc_f16out(const float* x, int N, int64_t hw, int C, const float* w, const float* b, float eps,
                                                 int swish, float* stats_scratch, int partial_blocks, void* y_f16, void* stream) {
    return groupnorm_split_launch(x, N, hw, C, w, b, eps, swish, stats_scratch, partial_blocks, y_f16, stream, true);
}

static int groupnorm_split_launch(const float* x, int N, int64_t hw, int C, const float* w, const float* b, float eps, int swish,
                                  float* stats_scratch, int partial_blocks, void* planes_bf16, void* stream, bool f16) {
    MMVID_REQUIRE(x && w && b && stats_scratch && planes_bf16, "groupnorm_split: null pointer");
    MMVID_REQUIRE(C % 32 == 0 && C <= 512 && 256 % (C / 8) == 0, "groupnorm_split: C=%d unsupported", C);
    MMVID_REQUIRE(partial_blocks >= 0 && partial_blocks <= cdiv(hw, 64), "groupnorm_split: partial_blocks %d", partial_blocks);
    if (N == 0 || hw == 0) return MMVID_OK;
    hipStream_t s = (hipStream_t)stream;
    const int pix_per_block = 256;
    int nblk = partial_blocks;  // > 0: the producing convolution's epilogue already wrote that many partial blocks per image
    float* mr = stats_scratch;                         // [N][C][2]
    float* partial = stats_scratch + (long)N * C * 2;  // [N][nblk][32][2]
    if (partial_blocks == 0) {
        nblk = cdiv(hw, pix_per_block);
        hipLaunchKernelGGL(groupnorm_stats_kernel<float>, dim3(nblk, N), dim3(256), 0, s, x, (long)hw, C, pix_per_block, partial);
    }
    hipLaunchKernelGGL(groupnorm_finalize_f64_kernel, dim3(cdiv((long)N * 32, 4)), dim3(256), 0, s, partial, nblk, N, C,
                       (double)hw * (double)(C / 32), eps, mr);
    const long chunks = (long)N * hw * (C / 8);
    if (f16)
        hipLaunchKernelGGL(groupnorm_apply_split_kernel<true>, dim3(cdiv(chunks, 256)), dim3(256), 0, s, x, (long)hw, C, mr, w, b, swish,
                           (bf16_t*)planes_bf16, chunks);
    else
        hipLaunchKernelGGL(groupnorm_apply_split_kernel<false>, dim3(cdiv(chunks, 256)), dim3(256), 0, s, x, (long)hw, C, mr, w, b, swish,
                           (bf16_t*)planes_bf16, chunks);
    MMVID_LAUNCH_CHECK("groupnorm_split");
    return MMVID_OK;
}
