// Multi-head attention core of the CLIP tower, forward and backward, head_dim 64, for gfx950.
// Replaces nn.MultiheadAttention's softmax(QK^T/8 + mask)V at clip_model.py:217-222 for the three mask
// shapes the reference builds (clip_model.py:561-578): none, causal (ART-V), and "mask_prev" = up to two
// query rows that may not look at earlier columns (BERT).  The mask is a predicate, never an L x L tensor.
//
// Flash-style on v_mfma_f32_32x32x16_bf16; a wave owns 32 queries (or 32 keys) and keeps them LANE-LOCAL:
//   forward : S^T = K Q^T -> lane holds 16 keys x its query (col = lane&31); row max/sum need one half-swap;
//             O^T = V^T P^T with P taken straight from the S registers, so the softmax rescale is lane-local.
//   dQ      : delta = rowsum(dO * O); S^T, dP^T = V dO^T, dS^T = P (dP - delta), dQ^T = K^T dS^T  (loop over key tiles)
//   dK, dV  : S = Q K^T -> lane holds 16 queries x its key; dV^T = dO^T P, dK^T = Q^T dS (loop over query tiles)
// Every operand tile is the plain token-major [64 positions][64 d] slice of qkv / dO, brought into LDS by
// LDS-DMA (no VGPR staging, no ds_write), double-buffered.  Operands whose MFMA rows are positions are read
// with ds_read_b128; operands whose REDUCTION index is the position (V^T, K^T, Q^T, dO^T) are read from the
// same row-major tile with the hardware transpose read ds_read_b64_tr_b16 -- there are no transposed copies
// in HBM.  One 16-B-chunk XOR swizzle (chunk ^= swz(row), applied on the DMA source address) serves both, conflict-free for both.
// The inner loops are written for the vector ALU (at head dimension 64 a key tile needs as many vector-ALU cycles for its softmax as
// matrix-pipe cycles for its products: PMC rounds 4-5, profiles/r05_pmc_attention.txt): single-lane fp32 fma / add for the scale-and-shift
// and the row sum (the packed forms halve the instruction count and were measured SLOWER: a packed fp32 instruction takes two passes),
// hardware bf16 pack, v_max3, v_permlane32_swap for the cross-half max, a LAZY softmax rescale (the running
// max is only raised when a tile exceeds it by 2^8; P <= 256 is exact enough in bf16), and -- round 5 -- tile loops unrolled by the
// LDS stage so that every fragment address is a per-lane base computed once + an immediate, a mask-free body for the tiles that
// cannot contain a masked pair, a two-instruction-per-element padding mask, and no padding mask at all where padded positions
// cannot reach a stored result (dQ: their K rows are zero-filled; dK/dV: a padded key's lane is never stored).
// exp2 with 1/sqrt(d)*log2(e) folded in; lse2 = m + log2(sum) is kept for the backward.  No atomics except the bias-gradient sums.
#include <type_traits>

#include "../../include/mmvid_hip.h"
#include "gemm_core.h"
#include "prof.h"

namespace {
using mmvid_core::blds16;
using mmvid_core::dma_publish_barrier;
using mmvid_core::make_rsrc;
using mmvid_core::rsrc_t;
using mmvid_core::bf16x4_t;
using mmvid_core::ds_read_tr16;
using mmvid_core::lds_addr;
using mmvid_core::lgkm_wait_tied;
using mmvid_core::xcd_remap;

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

constexpr int TILE = 64 * 128;       // [64 pos][64 d] bf16, 16-B chunks swizzled
constexpr int ROWS_PER_BLOCK = 128;  // 4 waves x 32
constexpr float RESCALE_THR = 8.0f;  // log2 domain

// 16-B chunk swizzle of a [64 pos][64 d] tile: chunk ^= swz(row), swz = row bits (1, 3, 2) -> chunk bits (2, 1, 0).  Any bijection of
// (row >> 1) & 7 keeps the ds_read_b128 lane groups of row_frag() conflict-free (8 even / 8 odd rows of a group need 8 distinct values);
// sending row bit 1 to the 64-B-half bit (chunk bit 2) is what the transpose reads need: a half-wave of ds_read_b64_tr_b16 reads 4 rows x
// 64 B, and with the round-3 swizzle ((row >> 1) & 7: row bit 1 -> chunk bit 0) rows r and r + 2 landed on the same 16 banks -- two extra
// LDS cycles per transpose read (PMC round 4: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33 / 0.25 / 0.25 in fwd / dQ / dK,dV).
__device__ __forceinline__ int swz(int row) {
    const int x = (row >> 1) & 7;
    return ((x & 1) << 2) | (x >> 1);
}

// division of a small non-negative number by a launch constant (block coordinates): q = (n * m) >> 32 with m = floor(2^32 / d) + 1 is
// exact while n * d < 2^32 (checked by the launcher); the kernels' integer divisions were ~40 instructions each
struct FastDiv {
    uint32_t d, m;
};
static FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = (uint32_t)d, f.m = d > 1 ? (uint32_t)((1ull << 32) / (uint32_t)d) + 1u : 0u;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv f) { return f.d == 1 ? n : __umulhi(n, f.m); }

// LDS-DMA of positions [p0, p0+64) x 64 d of one (batch, head) slice of a token-major matrix: 8 pieces of 1 KiB, 2 per
// wave, through a buffer descriptor that ends at row L-1 -- positions >= L are zero-filled by the range check.
// Per-lane offsets are computed once; a tile costs its two buffer_load ... lds and a scalar offset.
struct TileStage {
    rsrc_t rsrc;
    uint32_t voff[2], rowbytes;
    __device__ __forceinline__ void init(const bf16_t* slice, long ld, int L, int wave, int lane) {
        rsrc = make_rsrc(slice, (uint32_t)(((long)(L - 1) * ld + 64) * 2));
        rowbytes = (uint32_t)ld * 2u;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = (wave * 2 + jj) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz(row);
            voff[jj] = (uint32_t)row * rowbytes + (uint32_t)chunk * 16u;  // (L * ld * 2 < 2^31: checked by the launcher)
        }
    }
    __device__ __forceinline__ void issue(int p0, char* tile, int wave) const {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) blds16(rsrc, voff[jj], (uint32_t)p0 * rowbytes, tile + (wave * 2 + jj) * 1024);
    }
};

// MFMA 32x32x16 operand whose rows are positions: row 32 ss + (lane & 31) of a tile, 8 consecutive d = 16 s + 8 h ..  The swizzle only
// looks at row bits 1-3, so a lane's four byte offsets (s = 0..3) are the same in every tile, stage and 32-row half: computed once.
struct RowLane {
    uint32_t o[4];
};
__device__ __forceinline__ RowLane row_lane_offs(int lane) {
    const int l32 = lane & 31, h = lane >> 5;
    RowLane r;
#pragma unroll
    for (int s = 0; s < 4; ++s) r.o[s] = (uint32_t)(l32 * 128 + (((2 * s + h) ^ swz(l32)) << 4));
    return r;
}
template <int SS, int S>
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, const RowLane& rl) {
    return *reinterpret_cast<const bf16x8_t*>(tile + SS * 4096 + rl.o[S]);
}
// Per-lane parts of the transpose-read address (see tr_frag): lane (G = lane >> 4, si = lane & 15) reads position rowl = 4 (G >> 1) +
// (si >> 2) of an 8-row group, 16-B chunk cl = 2 (G & 1) + ((si & 3) >> 1) of a 32-d half dt, 8-B half si & 1.  With chunk = 4 dt + cl and
// swz(row) = 4 x0 + 2 x2 + x1 (x0 = rowl bit 1, x1 = rowl bit 2, x2 = row bit 3 = which 8-row group of a 16-position reduction step):
//   chunk ^ swz = 4 (dt ^ x0) + (cl ^ x1 ^ 2 x2)  ->  one lane offset per (dt, x2): four registers.
struct TrLane {
    uint32_t o[2][2];  // [dt][x2], byte offsets inside a tile
};
__device__ __forceinline__ TrLane tr_lane_offs(int lane) {
    const int G = lane >> 4, si = lane & 15;
    const int rowl = 4 * (G >> 1) + (si >> 2);
    const int cl = 2 * (G & 1) + ((si & 3) >> 1);
    const int x0 = (rowl >> 1) & 1, x1 = (rowl >> 2) & 1;
    TrLane t;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int x2 = 0; x2 < 2; ++x2) t.o[dt][x2] = (uint32_t)(rowl * 128 + ((4 * (dt ^ x0) + (cl ^ x1 ^ (2 * x2))) << 4) + 8 * (si & 1));
    return t;
}
// (the same offsets as absolute LDS addresses of a tile at `base`: the transpose reads then need no per-tile address arithmetic)
__device__ __forceinline__ TrLane tr_lane_at(const TrLane& t, uint32_t base) {
    TrLane a;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int x2 = 0; x2 < 2; ++x2) a.o[dt][x2] = t.o[dt][x2] + base;
    return a;
}
// MFMA operand whose rows are d (= 32 dt + lane&31) and whose reduction index is the position: reduction step
// m (16 positions) of sub-tile ss (32 positions); element e of half h <-> position 32 ss + 16 m + 4 h + (e&3) +
// 8 (e>>2), the order in which pack_half() lays out accumulator rows.  Two transpose reads, 8 positions apart (x2 = 0, 1).
// Issued as inline asm (see gemm_core.h: the builtin would drain the LDS-DMA prefetch); await with lgkm_wait_tied.
// OFF: byte offset of the tile from the address the lane offsets were made absolute for (a compile-time stage offset).
template <int DT, int SS, int M, int OFF>
__device__ __forceinline__ bf16x8_t tr_frag(const TrLane& tl) {
    const bf16x4_t lo = ds_read_tr16<OFF + (32 * SS + 16 * M) * 128>(tl.o[DT][0]);
    const bf16x4_t hi = ds_read_tr16<OFF + (32 * SS + 16 * M + 8) * 128>(tl.o[DT][1]);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// the four fragments of sub-tile SS in MFMA order: (dt 0, m 0), (dt 1, m 0), (dt 0, m 1), (dt 1, m 1)
template <int SS, int OFF>
__device__ __forceinline__ void tr_frags4(const TrLane& tl, bf16x8_t (&f)[4]) {
    f[0] = tr_frag<0, SS, 0, OFF>(tl), f[1] = tr_frag<1, SS, 0, OFF>(tl);
    f[2] = tr_frag<0, SS, 1, OFF>(tl), f[3] = tr_frag<1, SS, 1, OFF>(tl);
}
// The matching B operand: accumulator registers 8m..8m+7 of a 32x32 tile whose ROW index is the reduction
// position: row(r, h) = (r&3) + 8 (r>>2) + 4 h.
__device__ __forceinline__ bf16x8_t pack_half(const f32x16& a, int m) {
    const int o = 8 * m;
    const uint4 u = make_uint4(pack_bf2(a[o], a[o + 1]), pack_bf2(a[o + 2], a[o + 3]), pack_bf2(a[o + 4], a[o + 5]),
                               pack_bf2(a[o + 6], a[o + 7]));
    return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8_t a, bf16x8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// first MFMA of an accumulation chain: C = 0 as an inline operand instead of 16 zeroing moves per tile (the loops are VALU-bound)
__device__ __forceinline__ f32x16 mfma32z(bf16x8_t a, bf16x8_t b) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
__device__ __forceinline__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// max of the 16 accumulator registers of a 32x32 tile in 8 instructions: plain fmaxf() on MFMA outputs makes hipcc emit a
// canonicalising v_max(x, x) per operand first (28 instructions for 16 values; the kernel is VALU-bound: 76 % VALU-busy per SIMD,
// profiles/r03_pmc_attention_before_interleave.txt)
__device__ __forceinline__ float max16(const f32x16& a) {
    float m0, m1;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(a[0]), "v"(a[1]), "v"(a[2]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(a[3]), "v"(a[4]), "v"(a[5]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(m0), "v"(a[6]), "v"(a[7]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(m1), "v"(a[8]), "v"(a[9]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(m0), "v"(a[10]), "v"(a[11]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(m1), "v"(a[12]), "v"(a[13]));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(m0), "v"(a[14]), "v"(a[15]));
    asm("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(m0), "v"(m1));
    return m0;
}
// combine a value with the other 32-lane half's (v_permlane32_swap: VALU, no LDS round trip).  After the swap of
// (v, v) one result register holds the lane's own value and the other its partner's, in every lane.
__device__ __forceinline__ float half_max(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    float m;
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(__uint_as_float(r[0])), "v"(__uint_as_float(r[1])));
    return m;
}
__device__ __forceinline__ float half_sum(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// a[r] = exp2(a[r] * sc + nb) for all 16 registers, returns their sum.  Single-lane v_fma_f32 / v_add_f32, written as asm so that -O3
// does not re-pack them: MI355X_MICROARCH prices a packed fp32 instruction beside MFMAs above the two plain ones it replaces, and the
// packed form (v_pk_fma_f32 / v_pk_add_f32, half the instructions) measured 3 us slower in the forward and 10 us in the backward
// at the training shape (profiles/r05_micro_attention_pk_tail_ln_gn.log), +0.04 ms on the step.
__device__ __forceinline__ float exp2_affine_sum(f32x16& a, float sc, float nb) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float x, y;
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(a[2 * i]), "v"(sc), "v"(nb));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(y) : "v"(a[2 * i + 1]), "v"(sc), "v"(nb));
        x = fast_exp2(x), y = fast_exp2(y);
        a[2 * i] = x, a[2 * i + 1] = y;
        asm("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(s0), "v"(x));
        asm("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(s1), "v"(y));
    }
    return s0 + s1;
}
// the padding mask of a key sub-tile that straddles L, for kernels whose lane column is a query: register r of half h holds key
// key0 + acc_row(r, h); kl = L - key0 - 4 h per lane, one compare against a constant + one select per register (the general predicate
// is_masked() is ~12 instructions per register)
__device__ __forceinline__ void mask_pad_keys(f32x16& s, int kl) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = (acc_row(r, 0) >= kl) ? -INFINITY : s[r];
}

// Row-per-lane epilogue stores, widened (cdna_hip_programming.md T21): a lane holds columns 8 g + 4 h .. +3 of its row for g = 0..3;
// one v_permlane32_swap per dword on (g = 2k, 2k+1) leaves lanes 0-31 with columns 16k .. 16k+7 and lanes 32-63 with 16k+8 .. 16k+15,
// so a 64-wide bf16 row segment goes out as 4 stores of 16 B per lane instead of 8 of 8 B (a store costs its issue slot).
__device__ __forceinline__ uint4 widen_pair(uint2 lo, uint2 hi) {
    const auto a = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
    return make_uint4(a[0], b[0], a[1], b[1]);
}
// acc[dt][r] * scale for one row of 64 values -> bf16 at `row_ptr` (the row's first element); h = lane >> 5
__device__ __forceinline__ void store_row64(bf16_t* row_ptr, const f32x16 (&acc)[2], float scale, int h) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint2 pc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int g4 = 2 * k + u;
                pc[u] = make_uint2(pack_bf2(acc[dt][4 * g4] * scale, acc[dt][4 * g4 + 1] * scale),
                                   pack_bf2(acc[dt][4 * g4 + 2] * scale, acc[dt][4 * g4 + 3] * scale));
            }
            *reinterpret_cast<uint4*>(row_ptr + 32 * dt + 16 * k + 8 * h) = widen_pair(pc[0], pc[1]);
        }
}

// Column sums of a wave's 32 rows x 64 values (row-per-lane accumulators, rows `live` only) added to dst[0..64): the bias gradient
// of the in-projection is the column sum of dqkv, and each backward kernel holds its rows of dq / dk / dv in registers right
// before storing them (the separate colsum pass over dqkv was 13 us per layer).  A halving exchange: after step s a lane keeps
// half of its remaining columns, summed with its partner's; 31 cross-lane moves for 32 columns, then one atomic per lane.
__device__ __forceinline__ void colsum_rows64(const f32x16 (&acc)[2], float scale, bool live, float* dst, int lane) {
    float cs[32];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[16 * dt + r] = live ? acc[dt][r] * scale : 0.f;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int m = 1 << s;
        const bool up = (lane & m) != 0;
#pragma unroll
        for (int t = 0; t < (16 >> s); ++t) {
            const float keep = up ? cs[2 * t + 1] : cs[2 * t];
            const float send = up ? cs[2 * t] : cs[2 * t + 1];
            cs[t] = keep + __shfl_xor(send, m, 64);
        }
    }
    const int c = lane & 31, h = lane >> 5;  // lane l of half h ends with column index c = l: d = 32 dt + 8 g4 + 4 h + e
    unsafeAtomicAdd(dst + 32 * (c >> 4) + 8 * ((c >> 2) & 3) + 4 * h + (c & 3), cs[0]);
}

// Block coordinates: (row tile, head, batch) of a dispatch id; blocks of one (batch, head) are consecutive AND on one XCD: they share
// K/V (or Q/dO) through that XCD's L2.  (Round 5 built and measured a split of the last, partly filled round's blocks over the streamed
// dimension with a combine launch -- L = 579, 18 sequences: 1,080 blocks against 1,024 / 768 / 512 resident slots in forward / dQ /
// dK,dV: 3-4 us per backward pass in isolation, nothing on the whole step (profiles/r05_ab_whole_step_pk_tail.log); removed.)
struct BlockCoords {
    int rt, hd, b;
};
__device__ __forceinline__ BlockCoords block_coords(FastDiv nrt, FastDiv H) {
    BlockCoords c;
    const uint32_t id = (uint32_t)xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t bh = fdiv(id, nrt);
    c.rt = (int)(id - bh * nrt.d);
    c.b = (int)fdiv(bh, H);
    c.hd = (int)(bh - (uint32_t)c.b * H.d);
    return c;
}

template <int V>
using ic = std::integral_constant<int, V>;

// ------------------------------------------------------------------------------------------ forward
// Three blocks per CU (round 6; four until then): at four the 128-register cap left 26 registers of the masked tile body in scratch -- the body
// that runs for the tail tile and the two restricted rows of EVERY block.  154 registers, no scratch: 40.3 -> 38.0 us per layer at the training
// shape, captured step 15.21 -> 15.12 ms, same box, alternating runs (profiles/r06_same_box_attention_forward_occupancy.log).
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld, int L, int H, int E, FastDiv nrt_d,
                                                           FastDiv h_d, float scale_log2, MaskSpec mask, bf16_t* __restrict__ out,
                                                           long ldo, float* __restrict__ lse2) {
    __shared__ __attribute__((aligned(16))) char smem[2][2 * TILE];  // K tile, V tile, two stages
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, h = lane >> 5;
    const BlockCoords bc = block_coords(nrt_d, h_d);
    const int qt = bc.rt, hd = bc.hd, b = bc.b;
    const bf16_t* Kbase = qkv + (long)b * L * ld + E + hd * 64;
    const bf16_t* Vbase = Kbase + E;
    const RowLane rl = row_lane_offs(lane);
    const TrLane trv = tr_lane_at(tr_lane_offs(lane), lds_addr(smem[0] + TILE));  // V tile of stage 0
    const int q_wave0 = qt * ROWS_PER_BLOCK + wave * 32;  // first query of the wave (wave-uniform)
    const int q = q_wave0 + l32;
    const int qc = q < L ? q : L - 1;
    const bool wave_active = q_wave0 < L;
    bf16x8_t qf[4];
    {
        const bf16_t* Qp = qkv + ((long)b * L + qc) * ld + hd * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(Qp + 16 * s + 8 * h);
    }
    int kv_end = L;
    if (mask.mode == 1) {
        const int blk_end = (qt + 1) * ROWS_PER_BLOCK;  // causal: the last key any of these queries may see, + 1
        if (blk_end < L) kv_end = blk_end;
    }
    const int t_begin = 0, t_end = (kv_end + 63) >> 6;

    TileStage stK, stV;
    stK.init(Kbase, ld, L, wave, lane), stV.init(Vbase, ld, L, wave, lane);
    stK.issue(t_begin * 64, smem[0], wave), stV.issue(t_begin * 64, smem[0] + TILE, wave);

    // running maximum (log2 domain, scaled), its negation as the exponent's reference (0 while it is -inf) and the raw-score value a
    // sub-tile's maximum must exceed to raise it: the last two change only inside the (rare) rescale branch, the common path compares
    // and subtracts without recomputing them
    float m_run = -INFINITY, nref = 0.f, thr_raw = -INFINITY, lsum = 0.f;
    const float inv_scale_log2 = 1.0f / scale_log2;
    f32x16 oacc[2] = {zero16(), zero16()};
    // restricted rows (mode 2): row r0 / r1 may not see keys below c0 / c1 -- only the wave that holds such a row, and only on key
    // sub-tiles that begin below that bound, evaluates the predicate (it is ~200 instructions per sub-tile: applied to all 19
    // sub-tiles it made the blocks holding rows 65 / 66 run 40 % longer than the rest, profiles/r03_attention_timeline*.log)
    const int row_kmax = mask.mode != 2 ? 0
                                        : max((mask.r0 >= q_wave0 && mask.r0 < q_wave0 + 32) ? mask.c0 : 0,
                                              (mask.r1 >= q_wave0 && mask.r1 < q_wave0 + 32) ? mask.c1 : 0);
    // Tiles [lo, hi) run the mask-free body: both 32-key sub-tiles are below L, no pair of the wave's queries with the tile's keys can
    // be masked, and a further tile follows (so its requests are unconditional).  They run as whole PAIRS of tiles in a loop of their
    // own (lo at an even distance from t_begin: the LDS stage is a compile-time constant of each body); the tiles in front of and behind
    // that range run the general body.  (One loop with both bodies as alternatives made the register allocator give the accumulators
    // different homes on the two paths: 16 v_mov_b64 per tile at the join.)  Every tile of every wave passes exactly one barrier.
    int lo = (row_kmax + 63) >> 6, hi = min(t_end - 1, L >> 6);
    if (mask.mode == 1) hi = min(hi, (q_wave0 + 1) >> 6);
    lo = max(lo, t_begin);
    lo += (lo - t_begin) & 1;
    if (lo >= t_end) lo = t_end;
    hi = hi > lo ? lo + ((hi - lo) & ~1) : lo;
    const int kl0 = L - 4 * h;  // (padding mask: key0 + acc_row(r, 0) >= kl0 - key0)

    // one 64-key tile; ST = LDS stage (compile time: every fragment address is lane base + immediate), PLAIN = mask-free body
    auto tile = [&](auto st_c, auto plain_c, int t) {
        constexpr int ST = decltype(st_c)::value;
        constexpr bool PLAIN = decltype(plain_c)::value != 0;
        constexpr int VOFF = ST * 2 * TILE;
        const char* Kt = smem[ST];
        char* nxt = smem[ST ^ 1];
        // the next tile's LDS-DMA requests cost the wave 60-185 issue cycles per piece (4 pieces): they are issued right AFTER this
        // tile's S MFMAs, so that this cost runs under the matrix pipe's 8 x 32 cycles instead of in front of them
        const bool more = PLAIN || t + 1 < t_end;
        // Both 32-key sub-tiles of the tile are in flight at once: the eight S = K Q^T MFMAs are issued back to back, and each
        // sub-tile's softmax arithmetic (VALU: exp2 is quarter rate) runs while the matrix pipe still works on the other
        // sub-tile's S or PV products.  Issued one sub-tile after the other (round 2), a wave sat in MFMA-result waits for 31 % of
        // its cycles and parked for 41 % (profiles/r03_pmc_attention_before_interleave.txt).
        const bool two = PLAIN || t * 64 + 32 < L;  // the second sub-tile holds live keys (block-uniform)
        f32x16 sv[2];
        auto s_mfmas = [&](auto ss_c) {
            constexpr int SS = decltype(ss_c)::value;
            sv[SS] = mfma32z(row_frag<SS, 0>(Kt, rl), qf[0]);
            sv[SS] = mfma32(row_frag<SS, 1>(Kt, rl), qf[1], sv[SS]);
            sv[SS] = mfma32(row_frag<SS, 2>(Kt, rl), qf[2], sv[SS]);
            sv[SS] = mfma32(row_frag<SS, 3>(Kt, rl), qf[3], sv[SS]);
        };
        s_mfmas(ic<0>{});
        // (the general body -- a few tiles per block -- computes the second sub-tile's S after the first sub-tile is done: sixteen
        //  registers fewer in flight, where the mask predicate needs them)
        if constexpr (PLAIN) s_mfmas(ic<1>{});
        if (more) {
            __builtin_amdgcn_sched_barrier(0);
            stK.issue((t + 1) * 64, nxt, wave), stV.issue((t + 1) * 64, nxt + TILE, wave);
        }
        auto sub = [&](auto ss_c) {
            constexpr int SS = decltype(ss_c)::value;
            const int key0 = t * 64 + 32 * SS;
            f32x16& s = sv[SS];
            // (the general body reaches the first reader of s through conditional branches: see mfma_settle; the mask-free body is
            //  straight-line code from the MFMAs to that reader, with at least four MFMAs or the softmax of sub-tile 0 in between)
            if constexpr (!PLAIN) mfma_settle(s);
            bf16x8_t vt[4];  // V^T fragments: requested now, consumed after the softmax arithmetic
            tr_frags4<SS, VOFF>(trv, vt);
            if constexpr (!PLAIN) {
                // the general predicate where the causal diagonal band or (wave-constant) a restricted query row of this wave can
                // apply; otherwise, in the sub-tile that straddles L, the padding compare alone
                if ((mask.mode == 1 && key0 + 31 > q_wave0) || key0 < row_kmax) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = is_masked(mask, q, key0 + acc_row(r, h), L) ? -INFINITY : s[r];
                } else if (key0 + 32 > L) {
                    mask_pad_keys(s, kl0 - key0);
                }
            }
            const float mx_raw = half_max(max16(s));
            if (__any(mx_raw > thr_raw)) {  // rare after the first tiles (scale > 0: the comparison in the raw-score domain)
                const float m_new = fmaxf(m_run, mx_raw * scale_log2);
                const float m_ref = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = fast_exp2(m_run - m_ref);
                lsum *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                m_run = m_new, nref = -m_ref, thr_raw = (m_new + RESCALE_THR) * inv_scale_log2;
            }
            lsum += exp2_affine_sum(s, scale_log2, nref);
            bf16x8_t pf[2] = {pack_half(s, 0), pack_half(s, 1)};
            lgkm_wait_tied<0>(vt[0], vt[1], vt[2], vt[3], pf[0], pf[1]);
            oacc[0] = mfma32(vt[0], pf[0], oacc[0]);
            oacc[1] = mfma32(vt[1], pf[0], oacc[1]);
            oacc[0] = mfma32(vt[2], pf[1], oacc[0]);
            oacc[1] = mfma32(vt[3], pf[1], oacc[1]);
        };
        sub(ic<0>{});
        if (two) {
            if constexpr (!PLAIN) s_mfmas(ic<1>{});
            sub(ic<1>{});
        }
    };
    if (!wave_active) {  // a wave without queries only stages its share of the tiles
        for (int t = t_begin; t < t_end; ++t) {
            dma_publish_barrier();
            char* nxt = smem[(t + 1 - t_begin) & 1];
            if (t + 1 < t_end) stK.issue((t + 1) * 64, nxt, wave), stV.issue((t + 1) * 64, nxt + TILE, wave);
        }
        return;
    }
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        const int a = ph == 0 ? t_begin : hi, z = ph == 0 ? lo : t_end;
#pragma unroll 1
        for (int t = a; t < z; ++t) {
            dma_publish_barrier();  // tile t has landed for every wave; everyone is done with tile t-1
            if ((t - t_begin) & 1)
                tile(ic<1>{}, ic<0>{}, t);
            else
                tile(ic<0>{}, ic<0>{}, t);
        }
        if (ph == 0) {
#pragma unroll 1
            for (int t = lo; t < hi; t += 2) {
                dma_publish_barrier();
                tile(ic<0>{}, ic<1>{}, t);
                dma_publish_barrier();
                tile(ic<1>{}, ic<1>{}, t + 1);
            }
        }
    }
    lsum = half_sum(lsum);
    mfma_settle(oacc[0]), mfma_settle(oacc[1]);
    if (q < L) {  // (lanes l and l + 32 hold the same row: the half-wave exchange inside store_row64 pairs two active lanes)
        store_row64(out + ((long)b * L + q) * ldo + hd * 64, oacc, 1.0f / lsum, h);
        if (h == 0) lse2[((long)b * H + hd) * L + q] = m_run + log2f(lsum);
    }
}

// ------------------------------------------------------------------------------------------ dQ
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ O, long ldo,
                                                              const bf16_t* __restrict__ dO, long lddo, const float* __restrict__ lse2,
                                                              float* __restrict__ delta, int L, int H, int E, FastDiv nrt_d, FastDiv h_d,
                                                              float scale, float scale_log2, MaskSpec mask, bf16_t* __restrict__ dqkv,
                                                              long ldg, float* __restrict__ dbias) {
    __shared__ __attribute__((aligned(16))) char smem[2][2 * TILE];  // K tile, V tile, two stages
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, h = lane >> 5;
    const BlockCoords bc = block_coords(nrt_d, h_d);
    const int qt = bc.rt, hd = bc.hd, b = bc.b;
    const bf16_t* Kbase = qkv + (long)b * L * ld + E + hd * 64;
    const bf16_t* Vbase = Kbase + E;
    const RowLane rl = row_lane_offs(lane);
    const TrLane trk = tr_lane_at(tr_lane_offs(lane), lds_addr(smem[0]));  // K tile of stage 0
    const int q_wave0 = qt * ROWS_PER_BLOCK + wave * 32;
    const int q = q_wave0 + l32;
    const int qc = q < L ? q : L - 1;
    const bool wave_active = q_wave0 < L;
    // the per-query operands: Q and dO fragments of the lane's row, -lse2, delta = rowsum(dO * O) (also stored for the dK/dV kernel)
    bf16x8_t qf[4], dof[4];
    float neg_lse, my_delta;
    {
        const bf16_t* Qp = qkv + ((long)b * L + qc) * ld + hd * 64;
        const bf16_t* dOp = dO + ((long)b * L + qc) * lddo + hd * 64;
        const bf16_t* Op = O + ((long)b * L + qc) * ldo + hd * 64;
        uint4 o4[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8_t*>(Qp + 16 * s + 8 * h);
            dof[s] = *reinterpret_cast<const bf16x8_t*>(dOp + 16 * s + 8 * h);
            o4[s] = *reinterpret_cast<const uint4*>(Op + 16 * s + 8 * h);
        }
        neg_lse = -lse2[((long)b * H + hd) * L + qc];
        float d = 0.f;  // each half-lane holds 32 of the row's 64 d
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const uint4 d4 = __builtin_bit_cast(uint4, dof[s]);
            d += (bf_lo(o4[s].x) * bf_lo(d4.x) + bf_hi(o4[s].x) * bf_hi(d4.x)) + (bf_lo(o4[s].y) * bf_lo(d4.y) + bf_hi(o4[s].y) * bf_hi(d4.y)) +
                 (bf_lo(o4[s].z) * bf_lo(d4.z) + bf_hi(o4[s].z) * bf_hi(d4.z)) + (bf_lo(o4[s].w) * bf_lo(d4.w) + bf_hi(o4[s].w) * bf_hi(d4.w));
        }
        my_delta = half_sum(d);
        if (h == 0 && q < L) delta[((long)b * H + hd) * L + q] = my_delta;
    }
    int kv_end = L;
    if (mask.mode == 1) {
        const int blk_end = (qt + 1) * ROWS_PER_BLOCK;
        if (blk_end < L) kv_end = blk_end;
    }
    const int t_begin = 0, t_end = (kv_end + 63) >> 6;

    TileStage sk, sv;
    sk.init(Kbase, ld, L, wave, lane), sv.init(Vbase, ld, L, wave, lane);
    sk.issue(t_begin * 64, smem[0], wave), sv.issue(t_begin * 64, smem[0] + TILE, wave);

    f32x16 dq[2] = {zero16(), zero16()};
    const int row_kmax = mask.mode != 2 ? 0
                                        : max((mask.r0 >= q_wave0 && mask.r0 < q_wave0 + 32) ? mask.c0 : 0,
                                              (mask.r1 >= q_wave0 && mask.r1 < q_wave0 + 32) ? mask.c1 : 0);
    // No padding mask here: the K rows of positions >= L are zero-filled by the descriptor's range check, so whatever dS holds for a
    // padded key is multiplied by zero in dQ^T += K^T dS^T (S = 0 there; the sub-tile that straddles L sets P = 0, see below).  Mask-free body: no causal /
    // restricted pair possible and a further tile follows.
    // (tile ranges as in the forward kernel: [lo, hi) = whole pairs of mask-free tiles)
    int lo = (row_kmax + 63) >> 6, hi = min(t_end - 1, L >> 6);
    if (mask.mode == 1) hi = min(hi, (q_wave0 + 1) >> 6);
    lo = max(lo, t_begin);
    lo += (lo - t_begin) & 1;
    if (lo >= t_end) lo = t_end;
    hi = hi > lo ? lo + ((hi - lo) & ~1) : lo;

    auto tile = [&](auto st_c, auto plain_c, int t) {
        constexpr int ST = decltype(st_c)::value;
        constexpr bool PLAIN = decltype(plain_c)::value != 0;
        constexpr int KOFF = ST * 2 * TILE;
        const char* Kt = smem[ST];
        const char* Vt = Kt + TILE;
        char* nxt = smem[ST ^ 1];
        const bool more = PLAIN || t + 1 < t_end;  // (the next tile's requests go out after the first eight MFMAs: see the forward kernel)
        auto sub = [&](auto ss_c) {
            constexpr int SS = decltype(ss_c)::value;
            const int key0 = t * 64 + 32 * SS;
            f32x16 s = mfma32z(row_frag<SS, 0>(Kt, rl), qf[0]);
            f32x16 dp = mfma32z(row_frag<SS, 0>(Vt, rl), dof[0]);
            s = mfma32(row_frag<SS, 1>(Kt, rl), qf[1], s), dp = mfma32(row_frag<SS, 1>(Vt, rl), dof[1], dp);
            s = mfma32(row_frag<SS, 2>(Kt, rl), qf[2], s), dp = mfma32(row_frag<SS, 2>(Vt, rl), dof[2], dp);
            s = mfma32(row_frag<SS, 3>(Kt, rl), qf[3], s), dp = mfma32(row_frag<SS, 3>(Vt, rl), dof[3], dp);
            if (SS == 0 && more) {
                __builtin_amdgcn_sched_barrier(0);
                sk.issue((t + 1) * 64, nxt, wave), sv.issue((t + 1) * 64, nxt + TILE, wave);
            }
            // (the softmax arithmetic reads s through asm statements, for which hipcc inserts no MFMA -> VALU wait states at all: the
            //  settle stays in every body)
            mfma_settle(s), mfma_settle(dp);
            bf16x8_t kt4[4];  // K^T fragments
            tr_frags4<SS, KOFF>(trk, kt4);
            if constexpr (!PLAIN) {
                if ((mask.mode == 1 && key0 + 31 > q_wave0) || key0 < row_kmax) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = is_masked(mask, q, key0 + acc_row(r, h), L) ? -INFINITY : s[r];
                }
                // the one sub-tile that straddles L: a padded key has S = 0, i.e. P = 2^(-lse2) -- finite unless EVERY score
                // of the row lies below -128 in the log2 domain, where it overflows and 0 x inf would put NaN into dQ; P = 0 there instead
                if (key0 + 32 > L) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = (key0 + acc_row(r, h) >= L) ? -INFINITY : s[r];
                }
            }
            (void)exp2_affine_sum(s, scale_log2, neg_lse);  // P
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] *= dp[r] - my_delta;  // dS
            bf16x8_t dsf[2] = {pack_half(s, 0), pack_half(s, 1)};
            lgkm_wait_tied<0>(kt4[0], kt4[1], kt4[2], kt4[3], dsf[0], dsf[1]);
            dq[0] = mfma32(kt4[0], dsf[0], dq[0]);
            dq[1] = mfma32(kt4[1], dsf[0], dq[1]);
            dq[0] = mfma32(kt4[2], dsf[1], dq[0]);
            dq[1] = mfma32(kt4[3], dsf[1], dq[1]);
        };
        sub(ic<0>{});
        if (PLAIN || t * 64 + 32 < L) sub(ic<1>{});  // (else: a padding-only sub-tile)
    };
    if (!wave_active) {  // a wave without queries only stages its share of the tiles
        for (int t = t_begin; t < t_end; ++t) {
            dma_publish_barrier();
            char* nxt = smem[(t + 1 - t_begin) & 1];
            if (t + 1 < t_end) sk.issue((t + 1) * 64, nxt, wave), sv.issue((t + 1) * 64, nxt + TILE, wave);
        }
        return;
    }
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        const int a = ph == 0 ? t_begin : hi, z = ph == 0 ? lo : t_end;
#pragma unroll 1
        for (int t = a; t < z; ++t) {
            dma_publish_barrier();
            if ((t - t_begin) & 1)
                tile(ic<1>{}, ic<0>{}, t);
            else
                tile(ic<0>{}, ic<0>{}, t);
        }
        if (ph == 0) {
#pragma unroll 1
            for (int t = lo; t < hi; t += 2) {
                dma_publish_barrier();
                tile(ic<0>{}, ic<1>{}, t);
                dma_publish_barrier();
                tile(ic<1>{}, ic<1>{}, t + 1);
            }
        }
    }
    mfma_settle(dq[0]), mfma_settle(dq[1]);
    if (q < L) store_row64(dqkv + ((long)b * L + q) * ldg + hd * 64, dq, scale, h);
    if (dbias) colsum_rows64(dq, scale, q < L, dbias + hd * 64, lane);
}

// ------------------------------------------------------------------------------------------ dK, dV
constexpr int DKV_BUF = 2 * TILE + 512;  // Q tile, dO tile, lse2[64], delta[64]

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ dO, long lddo,
                                                               const float* __restrict__ lse2, const float* __restrict__ delta, int L, int H,
                                                               int E, FastDiv nrt_d, FastDiv h_d, float scale, float scale_log2, MaskSpec mask,
                                                               bf16_t* __restrict__ dqkv, long ldg, float* __restrict__ dbias) {
    __shared__ __attribute__((aligned(16))) char dsm[2][DKV_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, h = lane >> 5;
    const BlockCoords bc = block_coords(nrt_d, h_d);
    const int kt = bc.rt, hd = bc.hd, b = bc.b;
    const bf16_t* Qbase = qkv + (long)b * L * ld + hd * 64;
    const bf16_t* dObase = dO + (long)b * L * lddo + hd * 64;
    const float* lse_b = lse2 + ((long)b * H + hd) * L;
    const float* del_b = delta + ((long)b * H + hd) * L;
    const int nq_tiles = (L + 63) >> 6;
    const RowLane rl = row_lane_offs(lane);
    const TrLane tr0 = tr_lane_offs(lane);
    const TrLane trq = tr_lane_at(tr0, lds_addr(dsm[0])), trdo = tr_lane_at(tr0, lds_addr(dsm[0] + TILE));
    const int key_wave0 = kt * ROWS_PER_BLOCK + wave * 32;
    const int key = key_wave0 + l32;
    const bool wave_active = key_wave0 < L;
    bf16x8_t kf[4], vf[4];  // the lane's key row (clamped: a padded key's lane computes a copy of key L - 1 and stores nothing)
    {
        const int keyc = key < L ? key : L - 1;
        const bf16_t* Kp = qkv + ((long)b * L + keyc) * ld + E + hd * 64;
        const bf16_t* Vp = Kp + E;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8_t*>(Kp + 16 * s + 8 * h);
            vf[s] = *reinterpret_cast<const bf16x8_t*>(Vp + 16 * s + 8 * h);
        }
    }
    // per-query statistics of a tile: threads 0..63 carry -lse2 (-inf for padded queries: exp2(-inf) = 0),
    // threads 64..127 carry delta
    auto load_stat = [&](int t) -> float {
        if (tid >= 128) return 0.f;
        const int qq = t * 64 + (tid & 63);
        if (tid < 64) return qq < L ? -lse_b[qq] : -INFINITY;
        return qq < L ? del_b[qq] : 0.f;
    };
    const int t_begin = (mask.mode == 1) ? (kt * ROWS_PER_BLOCK) >> 6 : 0, t_end = nq_tiles;  // causal: only queries >= keys contribute

    TileStage sq, sdo;
    sq.init(Qbase, ld, L, wave, lane), sdo.init(dObase, lddo, L, wave, lane);
    sq.issue(t_begin * 64, dsm[0], wave), sdo.issue(t_begin * 64, dsm[0] + TILE, wave);
    float stat = load_stat(t_begin);
    if (tid < 128) reinterpret_cast<float*>(dsm[0] + 2 * TILE)[tid] = stat;

    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    // No padding mask: a padded QUERY has -lse2 = -inf (P = 0), and a padded KEY is a lane whose dk / dv are never stored or summed
    // (the lanes of a wave are independent columns of S); the predicate is evaluated for causal / restricted pairs only.  Round 4 ran
    // it on every sub-tile of the wave that holds the keys around L -- ~200 instructions per sub-tile in one block of every head.
    const bool rows_hit0 = mask.mode == 2 && key_wave0 < mask.c0, rows_hit1 = mask.mode == 2 && key_wave0 < mask.c1;

    auto tile = [&](auto st_c, int t) {
        constexpr int ST = decltype(st_c)::value;
        constexpr int OFF = ST * DKV_BUF;
        const char* Qt = dsm[ST];
        const char* dOt = Qt + TILE;
        const float* st_nlse = reinterpret_cast<const float*>(Qt + 2 * TILE);
        const float* st_del = st_nlse + 64;
        char* nx = dsm[ST ^ 1];
        const bool more = t + 1 < t_end;
        dma_publish_barrier();
        if (!wave_active) {
            if (more) {
                sq.issue((t + 1) * 64, nx, wave), sdo.issue((t + 1) * 64, nx + TILE, wave);
                stat = load_stat(t + 1);
                if (tid < 128) reinterpret_cast<float*>(nx + 2 * TILE)[tid] = stat;
            }
            return;
        }
        auto sub = [&](auto ss_c) {
            constexpr int SS = decltype(ss_c)::value;
            const int q0 = t * 64 + 32 * SS;
            f32x16 s = mfma32z(row_frag<SS, 0>(Qt, rl), kf[0]);
            f32x16 dp = mfma32z(row_frag<SS, 0>(dOt, rl), vf[0]);
            s = mfma32(row_frag<SS, 1>(Qt, rl), kf[1], s), dp = mfma32(row_frag<SS, 1>(dOt, rl), vf[1], dp);
            s = mfma32(row_frag<SS, 2>(Qt, rl), kf[2], s), dp = mfma32(row_frag<SS, 2>(dOt, rl), vf[2], dp);
            s = mfma32(row_frag<SS, 3>(Qt, rl), kf[3], s), dp = mfma32(row_frag<SS, 3>(dOt, rl), vf[3], dp);
            if (SS == 0 && more) {  // the next tile's requests, under the first eight MFMAs (see the forward kernel)
                __builtin_amdgcn_sched_barrier(0);
                sq.issue((t + 1) * 64, nx, wave), sdo.issue((t + 1) * 64, nx + TILE, wave);
                stat = load_stat(t + 1);
            }
            mfma_settle(s), mfma_settle(dp);
            bf16x8_t dot4[4], qt4[4];  // dO^T and Q^T fragments, in consumption order
            tr_frags4<SS, OFF>(trdo, dot4);
            tr_frags4<SS, OFF>(trq, qt4);
            // mask needed?  (wave-uniform: the wave's keys are key_wave0 .. key_wave0 + 31) the causal diagonal region, or a restricted
            // query row among these queries AND some of this wave's keys below its bound
            bool nm = mask.mode == 1 && key_wave0 + 31 > q0;
            nm = nm || (rows_hit0 && mask.r0 >= q0 && mask.r0 < q0 + 32) || (rows_hit1 && mask.r1 >= q0 && mask.r1 < q0 + 32);
            if (nm) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = is_masked(mask, q0 + acc_row(r, h), key, L) ? -INFINITY : s[r];
            }
            const f32x2 sc2 = {scale_log2, scale_log2};
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int ql = 32 * SS + 8 * g4 + 4 * h;  // 4 consecutive query rows: registers 4 g4 .. 4 g4 + 3
                const float4 l4 = *reinterpret_cast<const float4*>(st_nlse + ql);
                const float4 d4 = *reinterpret_cast<const float4*>(st_del + ql);
                f32x2 a = {s[4 * g4], s[4 * g4 + 1]}, c = {s[4 * g4 + 2], s[4 * g4 + 3]};
                a = __builtin_elementwise_fma(a, sc2, (f32x2){l4.x, l4.y});
                c = __builtin_elementwise_fma(c, sc2, (f32x2){l4.z, l4.w});
                const float p0 = fast_exp2(a[0]), p1 = fast_exp2(a[1]), p2 = fast_exp2(c[0]), p3 = fast_exp2(c[1]);
                s[4 * g4] = p0, s[4 * g4 + 1] = p1, s[4 * g4 + 2] = p2, s[4 * g4 + 3] = p3;
                dp[4 * g4] = p0 * (dp[4 * g4] - d4.x);
                dp[4 * g4 + 1] = p1 * (dp[4 * g4 + 1] - d4.y);
                dp[4 * g4 + 2] = p2 * (dp[4 * g4 + 2] - d4.z);
                dp[4 * g4 + 3] = p3 * (dp[4 * g4 + 3] - d4.w);
            }
            bf16x8_t pf[2] = {pack_half(s, 0), pack_half(s, 1)};
            bf16x8_t dsf[2] = {pack_half(dp, 0), pack_half(dp, 1)};
            lgkm_wait_tied<8>(dot4[0], dot4[1], dot4[2], dot4[3], pf[0], pf[1]);  // the 8 reads of Q^T may still be in flight
            dv[0] = mfma32(dot4[0], pf[0], dv[0]);
            dv[1] = mfma32(dot4[1], pf[0], dv[1]);
            dv[0] = mfma32(dot4[2], pf[1], dv[0]);
            dv[1] = mfma32(dot4[3], pf[1], dv[1]);
            lgkm_wait_tied<0>(qt4[0], qt4[1], qt4[2], qt4[3], dsf[0], dsf[1]);
            dk[0] = mfma32(qt4[0], dsf[0], dk[0]);
            dk[1] = mfma32(qt4[1], dsf[0], dk[1]);
            dk[0] = mfma32(qt4[2], dsf[1], dk[0]);
            dk[1] = mfma32(qt4[3], dsf[1], dk[1]);
        };
        sub(ic<0>{});
        if (t * 64 + 32 < L) sub(ic<1>{});  // (else: a padding-only query sub-tile: P = 0 there)
        if (more && tid < 128) reinterpret_cast<float*>(nx + 2 * TILE)[tid] = stat;
    };
    for (int t = t_begin; t < t_end; t += 2) {
        tile(ic<0>{}, t);
        if (t + 1 >= t_end) break;
        tile(ic<1>{}, t + 1);
    }
    if (!wave_active) return;
    mfma_settle(dk[0]), mfma_settle(dk[1]), mfma_settle(dv[0]), mfma_settle(dv[1]);
    if (key < L) {
        bf16_t* kp = dqkv + ((long)b * L + key) * ldg + E + hd * 64;
        store_row64(kp, dk, scale, h);
        store_row64(kp + E, dv, 1.0f, h);
    }
    if (dbias) {
        colsum_rows64(dk, scale, key < L, dbias + E + hd * 64, lane);
        colsum_rows64(dv, 1.0f, key < L, dbias + 2 * E + hd * 64, lane);
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
    MMVID_REQUIRE(mask_mode >= 0 && mask_mode <= 2, name ": mask_mode %d", mask_mode);                             \
    MMVID_REQUIRE((int64_t)cdiv(L, ROWS_PER_BLOCK) * H * B * (H > 64 ? H : 64) < (1ll << 31), name ": too many (batch, head, row block) units")

extern "C" int mmvid_attention_fwd(const void* qkv, int64_t ld, int B, int L, int H, int E, float scale, int mask_mode,
                                   int r0, int c0, int r1, int c1, void* out, int64_t ldo, float* lse2, void* stream) {
    MMVID_REQUIRE(qkv && out && lse2, "attention_fwd: null pointer");
    ATTN_COMMON_CHECKS("attention_fwd");
    MMVID_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out & 15) == 0, "attention_fwd: leading dims must be multiples of 8, out 16-byte aligned");
    MMVID_REQUIRE((int64_t)L * ld * 2 < (1ll << 31), "attention_fwd: one batch entry of qkv must be smaller than 2 GiB");
    hipStream_t s = (hipStream_t)stream;
    MmvidProfScope prof(PROF_ATTN_FWD, 4.0 * B * H * (double)L * L * 64, s);
    const int nrt = cdiv(L, ROWS_PER_BLOCK), nblocks = nrt * H * B;
    const FastDiv nrt_d = make_fastdiv(nrt), h_d = make_fastdiv(H);
    const MaskSpec m = make_mask(mask_mode, r0, c0, r1, c1);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(nblocks), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld, L, H, E, nrt_d, h_d,
                       scale * 1.4426950408889634f, m, (bf16_t*)out, (long)ldo, lse2);
    MMVID_LAUNCH_CHECK("attention_fwd");
    return MMVID_OK;
}

extern "C" int mmvid_attention_bwd(const void* qkv, int64_t ld, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                                   const float* lse2, float* delta, int B, int L, int H, int E, float scale,
                                   int mask_mode, int r0, int c0, int r1, int c1, void* dqkv, int64_t ldg,
                                   void* stream) {
    return mmvid_attention_bwd_bias(qkv, ld, O, ldo, dO, lddo, lse2, delta, B, L, H, E, scale, mask_mode, r0, c0, r1, c1, dqkv, ldg,
                                    nullptr, stream);
}

extern "C" int mmvid_attention_bwd_bias(const void* qkv, int64_t ld, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                                        const float* lse2, float* delta, int B, int L, int H, int E, float scale,
                                        int mask_mode, int r0, int c0, int r1, int c1, void* dqkv, int64_t ldg,
                                        float* dbias, void* stream) {
    MMVID_REQUIRE(qkv && O && dO && lse2 && delta && dqkv, "attention_bwd: null pointer");
    ATTN_COMMON_CHECKS("attention_bwd");
    MMVID_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && ldg % 8 == 0 && ((uintptr_t)dqkv & 15) == 0,
                  "attention_bwd: leading dims must be multiples of 8, dqkv 16-byte aligned");
    MMVID_REQUIRE((int64_t)L * ld * 2 < (1ll << 31) && (int64_t)L * lddo * 2 < (1ll << 31),
                  "attention_bwd: one batch entry of qkv / dO must be smaller than 2 GiB");
    hipStream_t s = (hipStream_t)stream;
    const MaskSpec m = make_mask(mask_mode, r0, c0, r1, c1);
    const float sl2 = scale * 1.4426950408889634f;
    MmvidProfScope prof(PROF_ATTN_BWD, 10.0 * B * H * (double)L * L * 64, s);  // 5 GEMM-equivalents (recompute counted once)
    const int nrt = cdiv(L, ROWS_PER_BLOCK), nblocks = nrt * H * B;
    const FastDiv nrt_d = make_fastdiv(nrt), h_d = make_fastdiv(H);
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(nblocks), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld, (const bf16_t*)O, (long)ldo,
                       (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt_d, h_d, scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(nblocks), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld, (const bf16_t*)dO, (long)lddo, lse2,
                       delta, L, H, E, nrt_d, h_d, scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias);
    MMVID_LAUNCH_CHECK("attention_bwd");
    return MMVID_OK;
}
