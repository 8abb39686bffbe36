// Sequence assembly and losses around the tower -- HBM-bound gathers/scatters.
//   assemble_sequence      dalle_bert.py:899-973,1030-1035 : x[b,l,:] = table[seg[l]][ids[b,l], :] + pos[l, :]
//   assemble_sequence_bwd  scatter-add of dx rows into the table gradients (fp32 atomics) + dpos = sum_b dx
//   cross_entropy fwd/bwd  dalle_bert.py:1040 F.cross_entropy(logits[~mask1], target[~mask1]) (mean over selected rows)
//   colsum                 bias gradients: db[n] += sum_m dY[m][n]
#include "../../include/mmvid_hip.h"
#include "common.h"

namespace {

constexpr int MAX_TABLES = 4;
struct Tables {
    const float* t[MAX_TABLES];
    long rows[MAX_TABLES];
};
struct GradTables {
    float* t[MAX_TABLES];
    long rows[MAX_TABLES];
};

// Device-side fault counters: kernels never fault on bad indices (an out-of-range id reads row 0), they COUNT them;
// mmvid_device_faults() reads and optionally clears the counters (a synchronising call: between steps, not during capture).
//   [0] embedding id outside its table (assemble_sequence), [1] cross-entropy target outside [0, V)
__device__ unsigned long long g_faults[4];

// one wave per (b,l) row; E % 4 == 0
__global__ __launch_bounds__(256) void assemble_fwd_kernel(Tables tb, const long long* __restrict__ ids,
                                                           const int* __restrict__ seg,
                                                           const float* __restrict__ pos, long nrows, int L, int E,
                                                           float* __restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    const int lane = threadIdx.x & 63;
    const int l = (int)(row % L);
    const int s = seg[l];
    long id = ids[row];
    if (id < 0 || id >= tb.rows[s]) {  // never fault; counted (mmvid_device_faults)
        if (lane == 0) atomicAdd(&g_faults[0], 1ull);
        id = 0;
    }
    const float4* src = reinterpret_cast<const float4*>(tb.t[s] + id * E);
    const float4* pp = reinterpret_cast<const float4*>(pos + (long)l * E);
    float4* dst = reinterpret_cast<float4*>(out + row * E);
    for (int c = lane; c < (E >> 2); c += 64) {
        const float4 a = src[c], p = pp[c];
        dst[c] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    }
}

// Scatter-add of dx rows into the table gradients.  A block owns 32 consecutive rows and first merges the rows that
// hit the SAME table row (the [MASK] id fills most of the target segment, special tokens repeat in every sequence):
// one burst of atomics per distinct (table, id) of the chunk instead of one per row.  Per-row atomics measured 265 us
// at 10,422 rows: ~5,000 of them serialised on the 768 addresses of the [MASK] embedding.
constexpr int SC_ROWS = 32;
__global__ __launch_bounds__(256) void assemble_bwd_scatter_kernel(GradTables tb, const long long* __restrict__ ids,
                                                                   const int* __restrict__ seg,
                                                                   const float* __restrict__ dx, long nrows, int L,
                                                                   int E) {
    __shared__ long key[SC_ROWS];  // table * 2^40 + id, or -1 when the row has no gradient to deliver
    __shared__ int lead[SC_ROWS];
    const long base = (long)blockIdx.x * SC_ROWS;
    const int tid = threadIdx.x;
    if (tid < SC_ROWS) {
        const long row = base + tid;
        long k = -1;
        if (row < nrows) {
            const int s = seg[(int)(row % L)];
            const long id = ids[row];
            if (tb.t[s] && id >= 0 && id < tb.rows[s]) k = ((long)s << 40) + id;
        }
        key[tid] = k;
    }
    __syncthreads();
    if (tid < SC_ROWS) {
        int l = tid;
        const long k = key[tid];
        if (k < 0) {
            l = -1;
        } else {
            for (int j = 0; j < tid; ++j)
                if (key[j] == k) {
                    l = j;
                    break;
                }
        }
        lead[tid] = l;
    }
    __syncthreads();
    // every row of the chunk is loaded first (32 independent 16-B loads per thread in flight), then merged by leader in registers:
    // the round-2 form loaded inside the leader loop, i.e. 32 dependent memory latencies per block (92 us for 32 MB)
    for (int c = tid; c < (E >> 2); c += 256) {
        float4 v[SC_ROWS];
#pragma unroll
        for (int r = 0; r < SC_ROWS; ++r)
            v[r] = lead[r] >= 0 ? *reinterpret_cast<const float4*>(dx + (base + r) * E + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < SC_ROWS; ++r) {
            if (lead[r] != r) continue;  // block-uniform
            float4 a = v[r];
#pragma unroll
            for (int r2 = r + 1; r2 < SC_ROWS; ++r2) {
                if (lead[r2] != r) continue;
                a.x += v[r2].x, a.y += v[r2].y, a.z += v[r2].z, a.w += v[r2].w;
            }
            const long k = key[r];
            float* dst = tb.t[(int)(k >> 40)] + (k & ((1l << 40) - 1)) * E;
            unsafeAtomicAdd(dst + 4 * c, a.x), unsafeAtomicAdd(dst + 4 * c + 1, a.y);
            unsafeAtomicAdd(dst + 4 * c + 2, a.z), unsafeAtomicAdd(dst + 4 * c + 3, a.w);
        }
    }
}

// dpos[l, e] (+)= sum_b dx[b, l, e]
__global__ __launch_bounds__(256) void batch_reduce_kernel(const float* __restrict__ dx, int B, long LE,
                                                           float* __restrict__ dpos, int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= LE) return;
    float4 a = make_float4(0, 0, 0, 0);
    for (int b = 0; b < B; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(dx + (long)b * LE + i);
        a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
    }
    float4* d = reinterpret_cast<float4*>(dpos + i);
    if (accumulate) {
        const float4 p = *d;
        a.x += p.x, a.y += p.y, a.z += p.z, a.w += p.w;
    }
    *d = a;
}

// ---- cross entropy over selected rows.  One wave per row, V % 4 == 0.
// fwd: lse[row] saved; loss_sum += (lse - logit[target]) for selected rows (select[row] != 0).
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, long ldl,
                                                     const long long* __restrict__ target,
                                                     const unsigned char* __restrict__ select, long rows, int V,
                                                     float* __restrict__ lse, float* __restrict__ loss_sum) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    if (select && !select[row]) {
        if (lane == 0) lse[row] = 0.f;
        return;
    }
    const float4* p = reinterpret_cast<const float4*>(logits + row * ldl);
    float mx = -INFINITY;
    for (int c = lane; c < (V >> 2); c += 64) {
        const float4 v = p[c];
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < (V >> 2); c += 64) {
        const float4 v = p[c];
        s += (__expf(v.x - mx) + __expf(v.y - mx)) + (__expf(v.z - mx) + __expf(v.w - mx));
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float l = mx + __logf(s);
        lse[row] = l;
        long t = target[row];
        if (t < 0 || t >= V) {
            atomicAdd(&g_faults[1], 1ull);
            t = 0;
        }
        unsafeAtomicAdd(loss_sum, l - logits[row * ldl + t]);
    }
}

// bwd: dlogits[row] = (softmax - onehot) * (*gscale) for selected rows, 0 otherwise; bf16 output.
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, long ldl,
                                                     const long long* __restrict__ target,
                                                     const unsigned char* __restrict__ select,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ gscale, long rows, int V,
                                                     bf16_t* __restrict__ dlogits, long ldd) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    uint2* d = reinterpret_cast<uint2*>(dlogits + row * ldd);
    if (select && !select[row]) {
        for (int c = lane; c < (V >> 2); c += 64) d[c] = make_uint2(0, 0);
        return;
    }
    const float gs = *gscale;
    const float l = lse[row];
    long t = target[row];
    if (t < 0 || t >= V) t = 0;
    const float4* p = reinterpret_cast<const float4*>(logits + row * ldl);
    for (int c = lane; c < (V >> 2); c += 64) {
        const float4 v = p[c];
        float o[4] = {__expf(v.x - l), __expf(v.y - l), __expf(v.z - l), __expf(v.w - l)};
        const long base = (long)c * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (base + e == t) o[e] -= 1.0f;
        d[c] = make_uint2(pack_bf2(o[0] * gs, o[1] * gs), pack_bf2(o[2] * gs, o[3] * gs));
    }
}

// db[n] += sum_m dY[m][n] (bf16 in, fp32 atomic out).  Thread = 8 columns (one 16-B load per row), block = 256
// rows as 8 row-groups x 32 column-chunks reduced through LDS: one atomic per column per 256 rows.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ dy, long ld, long M, int N,
                                                          float* __restrict__ db) {
    __shared__ float red[8][256];
    const int cc = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int n = (blockIdx.x * 32 + cc) * 8;
    const long m0 = (long)blockIdx.y * 256;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    if (n < N) {
        long m1 = m0 + 256;
        if (m1 > M) m1 = M;
        for (long m = m0 + rg; m < m1; m += 8) {
            const uint4 w = *reinterpret_cast<const uint4*>(dy + m * ld + n);
            a[0] += bf_lo(w.x), a[1] += bf_hi(w.x), a[2] += bf_lo(w.y), a[3] += bf_hi(w.y);
            a[4] += bf_lo(w.z), a[5] += bf_hi(w.z), a[6] += bf_lo(w.w), a[7] += bf_hi(w.w);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rg][cc * 8 + e] = a[e];
    __syncthreads();
    const int col = threadIdx.x;  // 256 columns of this block
    const int gn = blockIdx.x * 256 + col;
    if (gn < N) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][col];
        unsafeAtomicAdd(db + gn, s);
    }
}

}  // namespace

extern "C" int mmvid_assemble_sequence(const float* const* tables, const int64_t* table_rows, int ntables,
                                       const int64_t* ids, const int32_t* seg, const float* pos, int64_t B, int L,
                                       int E, float* out, void* stream) {
    MMVID_REQUIRE(tables && table_rows && ids && seg && pos && out, "assemble_sequence: null pointer");
    MMVID_REQUIRE(ntables >= 1 && ntables <= MAX_TABLES && E % 4 == 0, "assemble_sequence: ntables=%d E=%d", ntables, E);
    Tables tb;
    for (int i = 0; i < MAX_TABLES; ++i) {
        tb.t[i] = i < ntables ? tables[i] : tables[0];
        tb.rows[i] = i < ntables ? table_rows[i] : table_rows[0];
    }
    const long nrows = (long)B * L;
    if (nrows == 0) return MMVID_OK;
    hipLaunchKernelGGL(assemble_fwd_kernel, dim3(cdiv(nrows, 4)), dim3(256), 0, (hipStream_t)stream, tb,
                       (const long long*)ids, seg, pos, nrows, L, E, out);
    MMVID_LAUNCH_CHECK("assemble_sequence");
    return MMVID_OK;
}

extern "C" int mmvid_assemble_sequence_bwd(float* const* grad_tables, const int64_t* table_rows, int ntables,
                                           const int64_t* ids, const int32_t* seg, const float* dx, int64_t B, int L,
                                           int E, float* dpos, int accumulate_dpos, void* stream) {
    MMVID_REQUIRE(grad_tables && table_rows && ids && seg && dx, "assemble_sequence_bwd: null pointer");
    MMVID_REQUIRE(ntables >= 1 && ntables <= MAX_TABLES && E % 4 == 0, "assemble_sequence_bwd: ntables=%d E=%d", ntables, E);
    GradTables tb;
    for (int i = 0; i < MAX_TABLES; ++i) {
        tb.t[i] = i < ntables ? grad_tables[i] : nullptr;
        tb.rows[i] = i < ntables ? table_rows[i] : 0;
    }
    const long nrows = (long)B * L;
    if (nrows == 0) return MMVID_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(assemble_bwd_scatter_kernel, dim3(cdiv(nrows, SC_ROWS)), dim3(256), 0, s, tb, (const long long*)ids,
                       seg, dx, nrows, L, E);
    if (dpos) {
        const long LE = (long)L * E;
        hipLaunchKernelGGL(batch_reduce_kernel, dim3(cdiv(LE / 4, 256)), dim3(256), 0, s, dx, (int)B, LE, dpos,
                           accumulate_dpos);
    }
    MMVID_LAUNCH_CHECK("assemble_sequence_bwd");
    return MMVID_OK;
}

extern "C" int mmvid_cross_entropy_fwd(const float* logits, int64_t ldl, const int64_t* target,
                                       const uint8_t* select, int64_t rows, int V, float* lse, float* loss_sum,
                                       void* stream) {
    MMVID_REQUIRE(logits && target && lse && loss_sum, "cross_entropy_fwd: null pointer");
    MMVID_REQUIRE(V % 4 == 0 && ldl % 4 == 0, "cross_entropy_fwd: V and ldl must be multiples of 4");
    if (rows == 0) return MMVID_OK;
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, (long)ldl,
                       (const long long*)target, select, (long)rows, V, lse, loss_sum);
    MMVID_LAUNCH_CHECK("cross_entropy_fwd");
    return MMVID_OK;
}

extern "C" int mmvid_cross_entropy_bwd(const float* logits, int64_t ldl, const int64_t* target,
                                       const uint8_t* select, const float* lse, const float* gscale, int64_t rows,
                                       int V, void* dlogits_bf16, int64_t ldd, void* stream) {
    MMVID_REQUIRE(logits && target && lse && gscale && dlogits_bf16, "cross_entropy_bwd: null pointer");
    MMVID_REQUIRE(V % 4 == 0 && ldl % 4 == 0 && ldd % 4 == 0, "cross_entropy_bwd: V/ld must be multiples of 4");
    if (rows == 0) return MMVID_OK;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, logits, (long)ldl,
                       (const long long*)target, select, lse, gscale, (long)rows, V, (bf16_t*)dlogits_bf16, (long)ldd);
    MMVID_LAUNCH_CHECK("cross_entropy_bwd");
    return MMVID_OK;
}

extern "C" int mmvid_colsum_bf16(const void* dy, int64_t ld, int64_t M, int N, float* db, void* stream) {
    MMVID_REQUIRE(dy && db, "colsum_bf16: null pointer");
    MMVID_REQUIRE(N % 8 == 0 && ld % 8 == 0, "colsum_bf16: N and ld must be multiples of 8");
    if (M == 0) return MMVID_OK;
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cdiv(N, 256), cdiv(M, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, (long)ld, (long)M, N, db);
    MMVID_LAUNCH_CHECK("colsum_bf16");
    return MMVID_OK;
}

// ---- dense positional table of a BERT sequence (dalle_bert.py:903-973: special / text / visual-axial / target-axial
// positional embeddings laid out along the sequence; axial_positional_embedding in summed mode) in ONE launch, and its
// backward in one launch.  A segment maps `rows` consecutive table rows [dst0, dst0 + rows) to either consecutive rows
// of one parameter (naxes = 0: w[0][src0 + i]) or the sum of up to three axial weights (row i -> indices (i / (d1 d2),
// (i / d2) % d1, i % d2) with the trailing dims; w[a] is [d_a][E]).  Rows no segment covers are zero ([SEP] slots).
namespace {
struct PosSeg {
    const float* w[3];
    float* gw[3];
    int dst0, rows, naxes, src0, d[3];
};
constexpr int POS_MAXSEG = 12;
struct PosSegs {
    PosSeg s[POS_MAXSEG];
    int n;
};
__device__ __forceinline__ void axial_index(const PosSeg& sg, int i, int (&ix)[3]) {
    if (sg.naxes == 3)
        ix[0] = i / (sg.d[1] * sg.d[2]), ix[1] = (i / sg.d[2]) % sg.d[1], ix[2] = i % sg.d[2];
    else if (sg.naxes == 2)
        ix[0] = i / sg.d[1], ix[1] = i % sg.d[1], ix[2] = 0;
    else
        ix[0] = i, ix[1] = 0, ix[2] = 0;
}
// one wave per table row
__global__ __launch_bounds__(256) void pos_table_fwd_kernel(PosSegs ps, int L, int E, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= L) return;
    int which = -1;
    for (int k = 0; k < ps.n; ++k)
        if (row >= ps.s[k].dst0 && row < ps.s[k].dst0 + ps.s[k].rows) which = k;
    float4* dst = reinterpret_cast<float4*>(out + (long)row * E);
    if (which < 0) {
        for (int c = lane; c < (E >> 2); c += 64) dst[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const PosSeg& sg = ps.s[which];
    const int i = row - sg.dst0;
    if (sg.naxes == 0) {
        const float4* src = reinterpret_cast<const float4*>(sg.w[0] + (long)(sg.src0 + i) * E);
        for (int c = lane; c < (E >> 2); c += 64) dst[c] = src[c];
        return;
    }
    int ix[3];
    axial_index(sg, i, ix);
    for (int c = lane; c < (E >> 2); c += 64) {
        float4 a = reinterpret_cast<const float4*>(sg.w[0] + (long)ix[0] * E)[c];  // ((w0 + w1) + w2): the order of AxialPositionalEmbedding.table()
        if (sg.naxes >= 2) {
            const float4 b = reinterpret_cast<const float4*>(sg.w[1] + (long)ix[1] * E)[c];
            a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        }
        if (sg.naxes >= 3) {
            const float4 b = reinterpret_cast<const float4*>(sg.w[2] + (long)ix[2] * E)[c];
            a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        }
        dst[c] = a;
    }
}
// blockIdx.x enumerates (segment, axis, index) parameter rows in the order of `first`; fixed-order sums (deterministic)
__global__ __launch_bounds__(256) void pos_table_bwd_kernel(PosSegs ps, int E, const float* __restrict__ g) {
    int job = blockIdx.x, k = 0, a = 0;
    for (k = 0; k < ps.n; ++k) {
        const PosSeg& sg = ps.s[k];
        const int na = sg.naxes == 0 ? 1 : sg.naxes;
        bool found = false;
        for (a = 0; a < na; ++a) {
            const int cnt = sg.naxes == 0 ? sg.rows : sg.d[a];
            if (job < cnt) {
                found = true;
                break;
            }
            job -= cnt;
        }
        if (found) break;
    }
    if (k >= ps.n) return;
    const PosSeg& sg = ps.s[k];
    float* dstp = sg.gw[a];
    if (!dstp) return;  // a frozen parameter
    for (int c = threadIdx.x; c < (E >> 2); c += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sg.naxes == 0) {
            acc = reinterpret_cast<const float4*>(g + (long)(sg.dst0 + job) * E)[c];
            float4* d = reinterpret_cast<float4*>(dstp + (long)(sg.src0 + job) * E) + c;
            const float4 o = *d;
            *d = make_float4(o.x + acc.x, o.y + acc.y, o.z + acc.z, o.w + acc.w);
            continue;
        }
        // the rows whose axis-a index is `job`, enumerated directly (rows / d[a] of them, fixed order: deterministic): the
        // round-2 form scanned all rows of the segment with a division per row (154 us for a 1.7-MB table)
        const int d1 = sg.naxes > 1 ? sg.d[1] : 1, d2 = sg.naxes > 2 ? sg.d[2] : 1;
        const int inner = a == 0 ? d1 * d2 : (a == 1 ? d2 : 1);  // rows sharing the index are `inner` apart in blocks of ...
        const int cnt = (sg.d[0] * d1 * d2) / sg.d[a];  // (rows beyond sg.rows -- a truncated table -- are skipped below)
        for (int j0 = 0; j0 < cnt; j0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                // j -> (outer, within): row = outer * (d[a] * inner) + job * inner + within
                const int outer = j / inner, within = j - outer * inner;
                const int i = outer * (sg.d[a] * inner) + job * inner + within;
                v[u] = (j < cnt && i < sg.rows) ? reinterpret_cast<const float4*>(g + (long)(sg.dst0 + i) * E)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
        }
        float4* d = reinterpret_cast<float4*>(dstp + (long)job * E) + c;
        const float4 o = *d;
        *d = make_float4(o.x + acc.x, o.y + acc.y, o.z + acc.z, o.w + acc.w);
    }
}
// out = wa * a + wb * b + wc * c (device scalars; null = absent)   |   ga, gb, gc = w * g
__global__ void lincomb3_kernel(const float* a, const float* b, const float* c, float wa, float wb, float wc, float* out) {
    if (threadIdx.x == 0) out[0] = (a ? wa * a[0] : 0.f) + (b ? wb * b[0] : 0.f) + (c ? wc * c[0] : 0.f);
}
__global__ void scale3_kernel(const float* g, float wa, float wb, float wc, float* ga, float* gb, float* gc) {
    if (threadIdx.x == 0) {
        const float v = g[0];
        if (ga) ga[0] = wa * v;
        if (gb) gb[0] = wb * v;
        if (gc) gc[0] = wc * v;
    }
}
int fill_segs(PosSegs& ps, const mmvid_pos_segment_t* segs, int nseg) {
    MMVID_REQUIRE(segs && nseg >= 1 && nseg <= POS_MAXSEG, "pos_table: 1..%d segments", POS_MAXSEG);
    ps.n = nseg;
    for (int k = 0; k < nseg; ++k) {
        const mmvid_pos_segment_t& in = segs[k];
        MMVID_REQUIRE(in.naxes >= 0 && in.naxes <= 3 && in.rows > 0 && in.w[0], "pos_table: bad segment %d", k);
        PosSeg& o = ps.s[k];
        for (int a = 0; a < 3; ++a) o.w[a] = in.w[a], o.gw[a] = in.gw[a], o.d[a] = in.d[a] > 0 ? in.d[a] : 1;
        o.dst0 = in.dst0, o.rows = in.rows, o.naxes = in.naxes, o.src0 = in.src0;
        if (in.naxes > 0) {
            long prod = 1;
            for (int a = 0; a < in.naxes; ++a) prod *= o.d[a];
            MMVID_REQUIRE(in.rows <= prod, "pos_table: segment %d has %d rows but its axes hold %ld", k, in.rows, prod);
        }
    }
    return MMVID_OK;
}
}  // namespace

extern "C" int mmvid_pos_table_fwd(const mmvid_pos_segment_t* segs, int nseg, int L, int E, float* out, void* stream) {
    MMVID_REQUIRE(out && L > 0 && E % 4 == 0, "pos_table_fwd: bad arguments");
    PosSegs ps;
    int rc = fill_segs(ps, segs, nseg);
    if (rc) return rc;
    hipLaunchKernelGGL(pos_table_fwd_kernel, dim3(cdiv(L, 4)), dim3(256), 0, (hipStream_t)stream, ps, L, E, out);
    MMVID_LAUNCH_CHECK("pos_table_fwd");
    return MMVID_OK;
}

// gw[a] += the gradient of w[a] given g = dL/d(table) [L][E]; segments whose gw is null are skipped
extern "C" int mmvid_pos_table_bwd(const mmvid_pos_segment_t* segs, int nseg, int E, const float* g, void* stream) {
    MMVID_REQUIRE(g && E % 4 == 0, "pos_table_bwd: bad arguments");
    PosSegs ps;
    int rc = fill_segs(ps, segs, nseg);
    if (rc) return rc;
    int jobs = 0;
    for (int k = 0; k < nseg; ++k) {
        if (ps.s[k].naxes == 0)
            jobs += ps.s[k].rows;
        else
            for (int a = 0; a < ps.s[k].naxes; ++a) jobs += ps.s[k].d[a];
    }
    hipLaunchKernelGGL(pos_table_bwd_kernel, dim3(jobs), dim3(256), 0, (hipStream_t)stream, ps, E, g);
    MMVID_LAUNCH_CHECK("pos_table_bwd");
    return MMVID_OK;
}

extern "C" int mmvid_lincomb3(const float* a, const float* b, const float* c, float wa, float wb, float wc, float* out, void* stream) {
    MMVID_REQUIRE(out && (a || b || c), "lincomb3: null pointer");
    hipLaunchKernelGGL(lincomb3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c, wa, wb, wc, out);
    MMVID_LAUNCH_CHECK("lincomb3");
    return MMVID_OK;
}
extern "C" int mmvid_scale3(const float* g, float wa, float wb, float wc, float* ga, float* gb, float* gc, void* stream) {
    MMVID_REQUIRE(g, "scale3: null pointer");
    hipLaunchKernelGGL(scale3_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g, wa, wb, wc, ga, gb, gc);
    MMVID_LAUNCH_CHECK("scale3");
    return MMVID_OK;
}

// ---- row-wise exchange of a sparse table gradient (engine.FlatTrainer._exchange_sparse; train.py:28-35 all-reduces the whole table).
// pack: ids [n] (any order, repeats allowed; the text ids of this rank's batch) -> uid [n] ascending with repeats blanked to -1, and
// rows [n][E] = the table-gradient row of every kept id (zeros where blanked): one rank's fixed-shape message.  One block sorts the
// ids in LDS (bitonic, n <= 4096), a second launch gathers the rows.  merge: W[ids[i]] += rows[i] for ids[i] >= 0 -- ids are unique within
// one rank's message, so there are no atomics; the host calls it once per peer in rank order (a fixed summation order on every rank).
namespace {
constexpr int PACK_MAX = 4096;
__global__ __launch_bounds__(1024) void rows_pack_ids_kernel(const long long* __restrict__ ids, int n, int np2, long long* __restrict__ uid) {
    __shared__ long long key[PACK_MAX];
    for (int i = threadIdx.x; i < np2; i += 1024) key[i] = i < n ? ids[i] : 0x7fffffffffffffffll;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const long long a = key[i], b = key[p];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) key[i] = b, key[p] = a;
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < n; i += 1024) uid[i] = (i > 0 && key[i] == key[i - 1]) ? -1ll : key[i];
}
__global__ __launch_bounds__(256) void rows_pack_gather_kernel(const float* __restrict__ W, long long V, int E, const long long* __restrict__ uid,
                                                               float* __restrict__ rows) {
    const long long id = uid[blockIdx.x];
    const bool ok = id >= 0 && id < V;
    for (int c = threadIdx.x; c < (E >> 2); c += 256)
        reinterpret_cast<float4*>(rows + (long)blockIdx.x * E)[c] =
            ok ? reinterpret_cast<const float4*>(W + id * E)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void rows_merge_kernel(float* __restrict__ W, long long V, int E, const long long* __restrict__ ids,
                                                         const float* __restrict__ rows) {
    const long long id = ids[blockIdx.x];
    if (id < 0 || id >= V) return;
    for (int c = threadIdx.x; c < (E >> 2); c += 256) {
        float4 a = reinterpret_cast<float4*>(W + id * E)[c];
        const float4 b = reinterpret_cast<const float4*>(rows + (long)blockIdx.x * E)[c];
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        reinterpret_cast<float4*>(W + id * E)[c] = a;
    }
}
}  // namespace

extern "C" int mmvid_rows_pack(const float* W, int64_t V, int E, const int64_t* ids, int n, int64_t* uid, float* rows, void* stream) {
    MMVID_REQUIRE(W && ids && uid && rows && n > 0 && n <= PACK_MAX && E % 4 == 0, "rows_pack: bad arguments (n <= %d, E %% 4 == 0)", PACK_MAX);
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    hipLaunchKernelGGL(rows_pack_ids_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const long long*)ids, n, np2, (long long*)uid);
    hipLaunchKernelGGL(rows_pack_gather_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, W, (long long)V, E, (const long long*)uid, rows);
    MMVID_LAUNCH_CHECK("rows_pack");
    return MMVID_OK;
}
extern "C" int mmvid_rows_merge(float* W, int64_t V, int E, const int64_t* ids, const float* rows, int n, void* stream) {
    MMVID_REQUIRE(W && ids && rows && n > 0 && E % 4 == 0, "rows_merge: bad arguments");
    hipLaunchKernelGGL(rows_merge_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, W, (long long)V, E, (const long long*)ids, rows);
    MMVID_LAUNCH_CHECK("rows_merge");
    return MMVID_OK;
}

// counts[4] <- the device fault counters (see g_faults); reset != 0 clears them.  Waits for the device.
extern "C" int mmvid_device_faults(int64_t* counts, int reset) {
    MMVID_REQUIRE(counts, "device_faults: null pointer");
    unsigned long long h[4] = {0, 0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_faults), sizeof(h)) != hipSuccess) {
        mmvid_set_error("device_faults: %s", hipGetErrorString(hipGetLastError()));
        return MMVID_ERR_HIP;
    }
    for (int i = 0; i < 4; ++i) counts[i] = (int64_t)h[i];
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_faults), z, sizeof(z)) != hipSuccess) {
            mmvid_set_error("device_faults: reset failed: %s", hipGetErrorString(hipGetLastError()));
            return MMVID_ERR_HIP;
        }
    }
    return MMVID_OK;
}
