// Multi-head attention core of the CLIP tower, forward and backward, head_dim 64, for gfx950.
// Replaces nn.MultiheadAttention's softmax(QK^T/8 + mask)V at clip_model.py:217-222 for the three mask
// shapes the reference builds (clip_model.py:561-578): none, causal (ART-V), and "mask_prev" = up to two
// query rows that may not look at earlier columns (BERT).  The mask is a predicate, never an L x L tensor.
//
// Flash-style on v_mfma_f32_32x32x16_bf16; a wave owns 32 queries (or 32 keys) and keeps them LANE-LOCAL:
//   forward : S^T = K Q^T -> lane holds 16 keys x its query (col = lane&31); row max/sum need one shfl_xor(32);
//             O^T = V^T P^T with P taken straight from the S registers (a fixed key permutation shared with the
//             V^T fragment), so the online-softmax rescale is lane-local too.
//   dQ      : S^T, dP^T = V dO^T, dS^T = P (dP - delta), dQ^T = K^T dS^T              (loop over key tiles)
//   dK, dV  : S = Q K^T -> lane holds 16 queries x its key; dV^T = dO^T P, dK^T = Q^T dS (loop over query tiles)
// Operands whose reduction index is the sequence position come from [B,H,64,Lp] transposed copies
// (head_transpose) so every fragment is an 8/16-byte contiguous LDS read.  64-position tiles of K/V (or Q/dO)
// are staged once per block (4 waves x 32 rows share them), double-buffered with register prefetch; row tiles
// use the GEMM's XOR swizzle (conflict-free ds_read_b128), transposed tiles 136-byte rows (17*d mod 32:
// conflict-free ds_read_b64).  exp2 with the 1/sqrt(d)*log2(e) scale folded in; lse2 = m + log2(sum) is kept
// for the backward.  No atomics: bit-reproducible.
#include "common.h"
#include "prof.h"

namespace {

struct MaskSpec {
    int mode;  // 0 none, 1 causal, 2 restricted rows
    int r0, c0, r1, c1;
};

__device__ __forceinline__ bool is_masked(const MaskSpec& m, int q, int key, int L) {
    // branch-free on purpose (bitwise ops, no short circuit): it is evaluated per accumulator element
    const bool pad = key >= L;
    const bool causal = (m.mode == 1) & (key > q);
    const bool rows = (m.mode == 2) & (((q == m.r0) & (key < m.c0)) | ((q == m.r1) & (key < m.c1)));
    return pad | causal | rows;
}

constexpr int ROW_TILE_BYTES = 64 * 128;  // [64 pos][64 d] bf16, swizzled 16-B chunks
constexpr int T_ROW = 136;                // bytes per row of a transposed tile ([64 d][64 pos] + 8 pad)
constexpr int T_TILE_BYTES = 64 * T_ROW;
constexpr int ROWS_PER_BLOCK = 128;       // 4 waves x 32

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// row tile: rows = sequence positions [p0, p0+64) of a token-major matrix (zero beyond L)
__device__ __forceinline__ void row_tile_load(uint4 (&v)[2], const bf16_t* base, long ld, int p0, int L, int tid) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pos = p0 + (tid >> 3) + 32 * i;
        v[i] = (pos < L) ? *reinterpret_cast<const uint4*>(base + (long)pos * ld + c * 8) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void row_tile_store(const uint4 (&v)[2], char* tile, int tid) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint4*>(tile + lds_off((tid >> 3) + 32 * i, c)) = v[i];
}
// transposed tile: rows = d (64), cols = positions [p0, p0+64) of a [64][Lp] matrix (Lp % 64 == 0, zero padded)
__device__ __forceinline__ void t_tile_load(uint4 (&v)[2], const bf16_t* base, long Lp, int p0, int tid) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int d = (tid >> 3) + 32 * i;
        v[i] = *reinterpret_cast<const uint4*>(base + (long)d * Lp + p0 + c * 8);
    }
}
__device__ __forceinline__ void t_tile_store(const uint4 (&v)[2], char* tile, int tid) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        char* p = tile + ((tid >> 3) + 32 * i) * T_ROW + c * 16;
        *reinterpret_cast<uint2*>(p) = make_uint2(v[i].x, v[i].y);
        *reinterpret_cast<uint2*>(p + 8) = make_uint2(v[i].z, v[i].w);
    }
}

// MFMA 32x32x16 operand from a row tile: row `row` (= lane&31 + base), k-step s (16 d), half h
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int row, int s, int h) {
    return *reinterpret_cast<const bf16x8_t*>(tile + lds_off(row, 2 * s + h));
}
// MFMA operand from a transposed tile for reduction step m (16 positions) of sub-step ss (32 positions):
// row d, positions pos(m,h,e) = 32 ss + 16 m + 4 h + (e&3) + 8 (e>>2): two 8-byte reads 16 bytes apart.
__device__ __forceinline__ bf16x8_t t_frag(const char* tile, int d, int ss, int m, int h) {
    const char* p = tile + d * T_ROW + (32 * ss + 16 * m + 4 * h) * 2;
    const uint2 lo = *reinterpret_cast<const uint2*>(p);
    const uint2 hi = *reinterpret_cast<const uint2*>(p + 16);
    const uint4 u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    return __builtin_bit_cast(bf16x8_t, u);
}
// The matching B operand: accumulator registers 8m..8m+7 of a 32x32 tile whose ROW index is the reduction
// position: row(r, h) = (r&3) + 8 (r>>2) + 4 h  ==  pos(m, h, e) - 32 ss with r = 8 m + e.
__device__ __forceinline__ bf16x8_t pack_half(const f32x16& a, int m) {
    const int o = 8 * m;
    const uint4 u = make_uint4(pack_bf2(a[o], a[o + 1]), pack_bf2(a[o + 2], a[o + 3]), pack_bf2(a[o + 4], a[o + 5]),
                               pack_bf2(a[o + 6], a[o + 7]));
    return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8_t a, bf16x8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Does any (q, key) pair of a 32x32 sub-tile need the mask predicate?  Wave-uniform.
__device__ __forceinline__ bool tile_needs_mask(const MaskSpec& m, int q_lane, int key0, int L) {
    bool need = key0 + 32 > L;
    if (m.mode == 1) need = need || (key0 + 31 > q_lane);
    if (m.mode == 2) need = need || q_lane == m.r0 || q_lane == m.r1;
    return __any(need);
}

// ------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                       const bf16_t* __restrict__ VT, int L, int Lp, int H, int E,
                                                       float scale_log2, MaskSpec mask, bf16_t* __restrict__ out,
                                                       long ldo, float* __restrict__ lse2) {
    __shared__ __attribute__((aligned(16))) char smem[2][ROW_TILE_BYTES + T_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int qt = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
    const int q = qt * ROWS_PER_BLOCK + wave * 32 + l32;
    const int qc = q < L ? q : L - 1;
    const bf16_t* Qp = qkv + ((long)b * L + qc) * ld + hd * 64;
    bf16x8_t qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(Qp + 16 * s + 8 * h);
    const bf16_t* Kbase = qkv + (long)b * L * ld + E + hd * 64;
    const bf16_t* VTb = VT + ((long)(b * H + hd) * 64) * Lp;
    int kv_end = L;
    if (mask.mode == 1 && (qt + 1) * ROWS_PER_BLOCK < L) kv_end = (qt + 1) * ROWS_PER_BLOCK;
    const int ntiles = (kv_end + 63) >> 6;

    uint4 kr[2], vr[2];
    row_tile_load(kr, Kbase, ld, 0, L, tid);
    t_tile_load(vr, VTb, Lp, 0, tid);
    row_tile_store(kr, smem[0], tid);
    t_tile_store(vr, smem[0] + ROW_TILE_BYTES, tid);
    __syncthreads();

    float m = -INFINITY, lsum = 0.f;
    f32x16 oacc[2] = {zero16(), zero16()};

    for (int t = 0; t < ntiles; ++t) {
        const char* Kt = smem[t & 1];
        const char* Vt = smem[t & 1] + ROW_TILE_BYTES;
        const bool more = t + 1 < ntiles;
        if (more) {
            row_tile_load(kr, Kbase, ld, (t + 1) * 64, L, tid);
            t_tile_load(vr, VTb, Lp, (t + 1) * 64, tid);
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int key0 = t * 64 + 32 * ss;
            f32x16 s = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) s = mfma32(row_frag(Kt, 32 * ss + l32, ks, h), qf[ks], s);
            mfma_settle(s);
            float mx = -INFINITY;
            if (tile_needs_mask(mask, q, key0, L)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = is_masked(mask, q, key0 + acc_row(r, h), L) ? -INFINITY : s[r] * scale_log2;
                    s[r] = v;
                    mx = fmaxf(mx, v);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] *= scale_log2;
                    mx = fmaxf(mx, s[r]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m - m_use);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = fast_exp2(s[r] - m_use);
                psum += s[r];
            }
            lsum = lsum * alpha + psum;
            m = m_new;
            if (!__all(alpha == 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
            const bf16x8_t pf[2] = {pack_half(s, 0), pack_half(s, 1)};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) oacc[dt] = mfma32(t_frag(Vt, 32 * dt + l32, ss, mm, h), pf[mm], oacc[dt]);
        }
        if (more) {
            row_tile_store(kr, smem[(t + 1) & 1], tid);
            t_tile_store(vr, smem[(t + 1) & 1] + ROW_TILE_BYTES, tid);
        }
        __syncthreads();
    }
    lsum += __shfl_xor(lsum, 32, 64);
    mfma_settle(oacc[0]), mfma_settle(oacc[1]);
    if (q < L) {
        const float inv = 1.0f / lsum;
        bf16_t* op = out + ((long)b * L + q) * ldo + hd * 64 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<uint2*>(op + 32 * dt + 8 * g4) =
                    make_uint2(pack_bf2(oacc[dt][4 * g4] * inv, oacc[dt][4 * g4 + 1] * inv),
                               pack_bf2(oacc[dt][4 * g4 + 2] * inv, oacc[dt][4 * g4 + 3] * inv));
        if (h == 0) lse2[((long)b * H + hd) * L + q] = m + log2f(lsum);
    }
}

// ------------------------------------------------------------------------------------------ dQ
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                          const bf16_t* __restrict__ KT,
                                                          const bf16_t* __restrict__ dO, long lddo,
                                                          const float* __restrict__ lse2,
                                                          const float* __restrict__ delta, int L, int Lp, int H,
                                                          int E, float scale, float scale_log2, MaskSpec mask,
                                                          bf16_t* __restrict__ dqkv, long ldg) {
    __shared__ __attribute__((aligned(16))) char smem[2][2 * ROW_TILE_BYTES + T_TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int qt = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
    const int q = qt * ROWS_PER_BLOCK + wave * 32 + l32;
    const int qc = q < L ? q : L - 1;
    const bf16_t* Qp = qkv + ((long)b * L + qc) * ld + hd * 64;
    const bf16_t* dOp = dO + ((long)b * L + qc) * lddo + hd * 64;
    bf16x8_t qf[4], dof[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const bf16x8_t*>(Qp + 16 * s + 8 * h);
        dof[s] = *reinterpret_cast<const bf16x8_t*>(dOp + 16 * s + 8 * h);
    }
    const float my_lse = lse2[((long)b * H + hd) * L + qc];
    const float my_delta = delta[((long)b * H + hd) * L + qc];
    const bf16_t* Kbase = qkv + (long)b * L * ld + E + hd * 64;
    const bf16_t* Vbase = qkv + (long)b * L * ld + 2 * E + hd * 64;
    const bf16_t* KTb = KT + ((long)(b * H + hd) * 64) * Lp;
    int kv_end = L;
    if (mask.mode == 1 && (qt + 1) * ROWS_PER_BLOCK < L) kv_end = (qt + 1) * ROWS_PER_BLOCK;
    const int ntiles = (kv_end + 63) >> 6;

    uint4 kr[2], vr[2], tr[2];
    row_tile_load(kr, Kbase, ld, 0, L, tid);
    row_tile_load(vr, Vbase, ld, 0, L, tid);
    t_tile_load(tr, KTb, Lp, 0, tid);
    row_tile_store(kr, smem[0], tid);
    row_tile_store(vr, smem[0] + ROW_TILE_BYTES, tid);
    t_tile_store(tr, smem[0] + 2 * ROW_TILE_BYTES, tid);
    __syncthreads();

    f32x16 dq[2] = {zero16(), zero16()};
    for (int t = 0; t < ntiles; ++t) {
        const char* Kt = smem[t & 1];
        const char* Vt = Kt + ROW_TILE_BYTES;
        const char* KTt = Kt + 2 * ROW_TILE_BYTES;
        const bool more = t + 1 < ntiles;
        if (more) {
            row_tile_load(kr, Kbase, ld, (t + 1) * 64, L, tid);
            row_tile_load(vr, Vbase, ld, (t + 1) * 64, L, tid);
            t_tile_load(tr, KTb, Lp, (t + 1) * 64, tid);
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int key0 = t * 64 + 32 * ss;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(row_frag(Kt, 32 * ss + l32, ks, h), qf[ks], s);
                dp = mfma32(row_frag(Vt, 32 * ss + l32, ks, h), dof[ks], dp);
            }
            mfma_settle(s), mfma_settle(dp);
            const bool nm = tile_needs_mask(mask, q, key0, L);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = fast_exp2(s[r] * scale_log2 - my_lse);
                if (nm && is_masked(mask, q, key0 + acc_row(r, h), L)) p = 0.f;
                s[r] = p * (dp[r] - my_delta);  // dS
            }
            const bf16x8_t dsf[2] = {pack_half(s, 0), pack_half(s, 1)};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) dq[dt] = mfma32(t_frag(KTt, 32 * dt + l32, ss, mm, h), dsf[mm], dq[dt]);
        }
        if (more) {
            char* nx = smem[(t + 1) & 1];
            row_tile_store(kr, nx, tid);
            row_tile_store(vr, nx + ROW_TILE_BYTES, tid);
            t_tile_store(tr, nx + 2 * ROW_TILE_BYTES, tid);
        }
        __syncthreads();
    }
    mfma_settle(dq[0]), mfma_settle(dq[1]);
    if (q < L) {
        bf16_t* op = dqkv + ((long)b * L + q) * ldg + hd * 64 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<uint2*>(op + 32 * dt + 8 * g4) =
                    make_uint2(pack_bf2(dq[dt][4 * g4] * scale, dq[dt][4 * g4 + 1] * scale),
                               pack_bf2(dq[dt][4 * g4 + 2] * scale, dq[dt][4 * g4 + 3] * scale));
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
constexpr int DKV_BUF = 2 * ROW_TILE_BYTES + 2 * T_TILE_BYTES + 512;

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, long ld,
                                                           const bf16_t* __restrict__ QT,
                                                           const bf16_t* __restrict__ dO, long lddo,
                                                           const bf16_t* __restrict__ dOT,
                                                           const float* __restrict__ lse2,
                                                           const float* __restrict__ delta, int L, int Lp, int H,
                                                           int E, float scale, float scale_log2, MaskSpec mask,
                                                           bf16_t* __restrict__ dqkv, long ldg) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];  // [2][DKV_BUF]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int kt = blockIdx.x, hd = blockIdx.y, b = blockIdx.z;
    const int key = kt * ROWS_PER_BLOCK + wave * 32 + l32;
    const int keyc = key < L ? key : L - 1;
    const bf16_t* Kp = qkv + ((long)b * L + keyc) * ld + E + hd * 64;
    const bf16_t* Vp = qkv + ((long)b * L + keyc) * ld + 2 * E + hd * 64;
    bf16x8_t kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kf[s] = *reinterpret_cast<const bf16x8_t*>(Kp + 16 * s + 8 * h);
        vf[s] = *reinterpret_cast<const bf16x8_t*>(Vp + 16 * s + 8 * h);
    }
    const bf16_t* Qbase = qkv + (long)b * L * ld + hd * 64;
    const bf16_t* dObase = dO + (long)b * L * lddo + hd * 64;
    const bf16_t* QTb = QT + ((long)(b * H + hd) * 64) * Lp;
    const bf16_t* dOTb = dOT + ((long)(b * H + hd) * 64) * Lp;
    const float* lse_b = lse2 + ((long)b * H + hd) * L;
    const float* del_b = delta + ((long)b * H + hd) * L;
    const int nq_tiles = (L + 63) >> 6;
    const int t0 = (mask.mode == 1) ? (kt * ROWS_PER_BLOCK) >> 6 : 0;  // causal: only queries >= keys contribute

    uint4 qr[2], dor[2], qtr[2], dotr[2];
    float stat = 0.f;
    auto gload = [&](int t) {
        row_tile_load(qr, Qbase, ld, t * 64, L, tid);
        row_tile_load(dor, dObase, lddo, t * 64, L, tid);
        t_tile_load(qtr, QTb, Lp, t * 64, tid);
        t_tile_load(dotr, dOTb, Lp, t * 64, tid);
        if (tid < 128) {
            const int qq = t * 64 + (tid & 63);
            if (tid < 64)
                stat = qq < L ? lse_b[qq] : INFINITY;  // exp2(-inf) = 0 for padded queries
            else
                stat = qq < L ? del_b[qq] : 0.f;
        }
    };
    auto lstore = [&](char* buf) {
        row_tile_store(qr, buf, tid);
        row_tile_store(dor, buf + ROW_TILE_BYTES, tid);
        t_tile_store(qtr, buf + 2 * ROW_TILE_BYTES, tid);
        t_tile_store(dotr, buf + 2 * ROW_TILE_BYTES + T_TILE_BYTES, tid);
        if (tid < 128) reinterpret_cast<float*>(buf + 2 * ROW_TILE_BYTES + 2 * T_TILE_BYTES)[tid] = stat;
    };
    gload(t0);
    lstore(dsm);
    __syncthreads();

    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    for (int t = t0; t < nq_tiles; ++t) {
        const int bi = (t - t0) & 1;
        const char* Qt = dsm + bi * DKV_BUF;
        const char* dOt = Qt + ROW_TILE_BYTES;
        const char* QTt = Qt + 2 * ROW_TILE_BYTES;
        const char* dOTt = QTt + T_TILE_BYTES;
        const float* st_lse = reinterpret_cast<const float*>(dOTt + T_TILE_BYTES);
        const float* st_del = st_lse + 64;
        const bool more = t + 1 < nq_tiles;
        if (more) gload(t + 1);
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int q0 = t * 64 + 32 * ss;
            f32x16 s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = mfma32(row_frag(Qt, 32 * ss + l32, ks, h), kf[ks], s);
                dp = mfma32(row_frag(dOt, 32 * ss + l32, ks, h), vf[ks], dp);
            }
            mfma_settle(s), mfma_settle(dp);
            // mask needed?  (wave-uniform) key padding, causal diagonal region, or a restricted query row in range
            bool nm = key >= L;
            if (mask.mode == 1) nm = nm || (key > q0);
            if (mask.mode == 2) nm = nm || (mask.r0 >= q0 && mask.r0 < q0 + 32) || (mask.r1 >= q0 && mask.r1 < q0 + 32);
            nm = __any(nm);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ql = 32 * ss + 8 * g4 + 4 * h;  // 4 consecutive query rows: registers 4 g4 .. 4 g4 + 3
                const float4 l4 = *reinterpret_cast<const float4*>(st_lse + ql);
                const float4 d4 = *reinterpret_cast<const float4*>(st_del + ql);
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g4 + e;
                    float p = fast_exp2(s[r] * scale_log2 - lv[e]);
                    if (nm && is_masked(mask, t * 64 + ql + e, key, L)) p = 0.f;
                    s[r] = p;
                    dp[r] = p * (dp[r] - dl[e]);
                }
            }
            const bf16x8_t pf[2] = {pack_half(s, 0), pack_half(s, 1)};
            const bf16x8_t dsf[2] = {pack_half(dp, 0), pack_half(dp, 1)};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    dv[dt] = mfma32(t_frag(dOTt, 32 * dt + l32, ss, mm, h), pf[mm], dv[dt]);
                    dk[dt] = mfma32(t_frag(QTt, 32 * dt + l32, ss, mm, h), dsf[mm], dk[dt]);
                }
        }
        if (more) lstore(dsm + (bi ^ 1) * DKV_BUF);
        __syncthreads();
    }
    mfma_settle(dk[0]), mfma_settle(dk[1]), mfma_settle(dv[0]), mfma_settle(dv[1]);
    if (key < L) {
        bf16_t* kp = dqkv + ((long)b * L + key) * ldg + E + hd * 64 + 4 * h;
        bf16_t* vp = dqkv + ((long)b * L + key) * ldg + 2 * E + hd * 64 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                *reinterpret_cast<uint2*>(kp + 32 * dt + 8 * g4) =
                    make_uint2(pack_bf2(dk[dt][4 * g4] * scale, dk[dt][4 * g4 + 1] * scale),
                               pack_bf2(dk[dt][4 * g4 + 2] * scale, dk[dt][4 * g4 + 3] * scale));
                *reinterpret_cast<uint2*>(vp + 32 * dt + 8 * g4) =
                    make_uint2(pack_bf2(dv[dt][4 * g4], dv[dt][4 * g4 + 1]), pack_bf2(dv[dt][4 * g4 + 2], dv[dt][4 * g4 + 3]));
            }
    }
}

// delta[b][h][q] = sum_d dO[q][h*64+d] * O[q][h*64+d]   (8 lanes x 8 elements per (token, head))
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ O, long ldo,
                                                         const bf16_t* __restrict__ dO, long lddo, int B, int L,
                                                         int H, float* __restrict__ delta) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long item = gid >> 3;  // (token, head)
    const int sub = gid & 7;
    const long total = (long)B * L * H;
    float acc = 0.f;
    long tok = 0;
    int h = 0;
    if (item < total) {
        tok = item / H;
        h = (int)(item % H);
        const uint4 a = *reinterpret_cast<const uint4*>(O + tok * ldo + h * 64 + sub * 8);
        const uint4 c = *reinterpret_cast<const uint4*>(dO + tok * lddo + h * 64 + sub * 8);
        acc = bf_lo(a.x) * bf_lo(c.x) + bf_hi(a.x) * bf_hi(c.x) + bf_lo(a.y) * bf_lo(c.y) + bf_hi(a.y) * bf_hi(c.y) +
              bf_lo(a.z) * bf_lo(c.z) + bf_hi(a.z) * bf_hi(c.z) + bf_lo(a.w) * bf_lo(c.w) + bf_hi(a.w) * bf_hi(c.w);
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (item < total && sub == 0) {
        const long bb = tok / L, q = tok % L;
        delta[(bb * H + h) * L + q] = acc;
    }
}

// src token-major [B*L, ld] columns [col0 + h*64 + d]  ->  dst [B, H, 64, Lp] (zero for pos >= L)
__global__ __launch_bounds__(256) void head_transpose_kernel(const bf16_t* __restrict__ src, long ld, int col0,
                                                             int L, int Lp, int H, bf16_t* __restrict__ dst) {
    __shared__ bf16_t tile[64][66];
    const int tid = threadIdx.x;
    const int pt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (tid >> 3) + 32 * i;
        const int pos = pt * 64 + r;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (pos < L) v = *reinterpret_cast<const uint4*>(src + ((long)b * L + pos) * ld + col0 + h * 64 + c * 8);
        uint32_t* t32 = reinterpret_cast<uint32_t*>(&tile[r][c * 8]);
        t32[0] = v.x, t32[1] = v.y, t32[2] = v.z, t32[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int d = (tid >> 3) + 32 * i;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[c * 8 + 2 * e][d] | ((uint32_t)tile[c * 8 + 2 * e + 1][d] << 16);
        *reinterpret_cast<uint4*>(dst + (((long)b * H + h) * 64 + d) * Lp + pt * 64 + c * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

static MaskSpec make_mask(int mode, int r0, int c0, int r1, int c1) {
    MaskSpec m;
    m.mode = mode, m.r0 = r0, m.c0 = c0, m.r1 = r1, m.c1 = c1;
    return m;
}

}  // namespace

#define ATTN_COMMON_CHECKS(name)                                                                            \
    MMVID_REQUIRE(B > 0 && L > 0 && H > 0 && E == H * 64, name ": need E == H*64 (head_dim 64), got E=%d H=%d", E, H); \
    MMVID_REQUIRE(Lp % 64 == 0 && Lp >= L, name ": Lp (%d) must be a multiple of 64 and >= L (%d)", Lp, L);  \
    MMVID_REQUIRE(mask_mode >= 0 && mask_mode <= 2, name ": mask_mode %d", mask_mode)

extern "C" int mmvid_head_transpose(const void* src, int64_t ld, int col0, int B, int L, int Lp, int H, void* dst,
                                    void* stream) {
    MMVID_REQUIRE(src && dst, "head_transpose: null pointer");
    MMVID_REQUIRE(Lp % 64 == 0 && Lp >= L && ld % 8 == 0 && col0 % 8 == 0, "head_transpose: bad Lp/ld/col0");
    hipLaunchKernelGGL(head_transpose_kernel, dim3(Lp / 64, H, B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (long)ld, col0, L, Lp, H, (bf16_t*)dst);
    MMVID_LAUNCH_CHECK("head_transpose");
    return MMVID_OK;
}

extern "C" int mmvid_attention_fwd(const void* qkv, int64_t ld, const void* VT, int B, int L, int Lp, int H, int E,
                                   float scale, int mask_mode, int r0, int c0, int r1, int c1, void* out,
                                   int64_t ldo, float* lse2, void* stream) {
    MMVID_REQUIRE(qkv && VT && out && lse2, "attention_fwd: null pointer");
    ATTN_COMMON_CHECKS("attention_fwd");
    MMVID_REQUIRE(ld % 8 == 0 && ldo % 4 == 0, "attention_fwd: bad leading dims");
    MmvidProfScope prof(PROF_ATTN_FWD, 4.0 * B * H * (double)L * L * 64, (hipStream_t)stream);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(L, ROWS_PER_BLOCK), H, B), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)qkv, (long)ld, (const bf16_t*)VT, L, Lp, H, E, scale * 1.4426950408889634f,
                       make_mask(mask_mode, r0, c0, r1, c1), (bf16_t*)out, (long)ldo, lse2);
    MMVID_LAUNCH_CHECK("attention_fwd");
    return MMVID_OK;
}

extern "C" int mmvid_attention_bwd(const void* qkv, int64_t ld, const void* QT, const void* KT, const void* O,
                                   int64_t ldo, const void* dO, int64_t lddo, const void* dOT, const float* lse2,
                                   float* delta, int B, int L, int Lp, int H, int E, float scale, int mask_mode,
                                   int r0, int c0, int r1, int c1, void* dqkv, int64_t ldg, void* stream) {
    MMVID_REQUIRE(qkv && QT && KT && O && dO && dOT && lse2 && delta && dqkv, "attention_bwd: null pointer");
    ATTN_COMMON_CHECKS("attention_bwd");
    MMVID_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && ldg % 4 == 0, "attention_bwd: bad leading dims");
    hipStream_t s = (hipStream_t)stream;
    const MaskSpec m = make_mask(mask_mode, r0, c0, r1, c1);
    const float sl2 = scale * 1.4426950408889634f;
    const long items = (long)B * L * H * 8;
    hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(items, 256)), dim3(256), 0, s, (const bf16_t*)O, (long)ldo,
                       (const bf16_t*)dO, (long)lddo, B, L, H, delta);
    MmvidProfScope prof(PROF_ATTN_BWD, 10.0 * B * H * (double)L * L * 64, s);  // 5 GEMM-equivalents (recompute counted once)
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(cdiv(L, ROWS_PER_BLOCK), H, B), dim3(256), 0, s, (const bf16_t*)qkv,
                       (long)ld, (const bf16_t*)KT, (const bf16_t*)dO, (long)lddo, lse2, delta, L, Lp, H, E, scale, sl2, m,
                       (bf16_t*)dqkv, (long)ldg);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * DKV_BUF);
        attr = true;
    }
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(cdiv(L, ROWS_PER_BLOCK), H, B), dim3(256), 2 * DKV_BUF, s,
                       (const bf16_t*)qkv, (long)ld, (const bf16_t*)QT, (const bf16_t*)dO, (long)lddo, (const bf16_t*)dOT,
                       lse2, delta, L, Lp, H, E, scale, sl2, m, (bf16_t*)dqkv, (long)ldg);
    MMVID_LAUNCH_CHECK("attention_bwd");
    return MMVID_OK;
}
