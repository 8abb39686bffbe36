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
// The inner loops are written for instruction count: packed fp32 fma/add, hardware bf16 pack, v_permlane32_swap for
// the cross-half max, and a LAZY softmax rescale (the running max is only raised when a tile exceeds it by 2^8;
// P <= 256 is exact enough in bf16).  What bounds them, measured with time stamps from inside (tools/attn_timeline.py,
// DESIGN.md section 3): a wave's own MFMA -> softmax -> MFMA chain, 1.65-2.45 us per 64-position tile whatever shares
// its SIMD (matrix pipe ~20 %, VALU ~25 %, LDS ~50 % busy), so a block lasts ~25 us and every partly filled round of
// the 1,080-block grid costs a whole one.
// exp2 with 1/sqrt(d)*log2(e) folded in; lse2 = m + log2(sum) is kept for the backward.  No atomics.
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
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ swz(row)) << 4); }

// LDS-DMA of positions [p0, p0+64) x 64 d of one (batch, head) slice of a token-major matrix: 8 pieces of 1 KiB, 2 per
// wave, through a buffer descriptor that ends at row L-1 -- positions >= L are zero-filled by the range check.
// Per-lane offsets are computed once; a tile costs its two buffer_load ... lds and a scalar offset.
struct TileStage {
    rsrc_t rsrc;
    uint32_t voff[2], rowbytes;
    __device__ __forceinline__ void init(const bf16_t* slice, long ld, int L, int wave, int lane) {
        rsrc = make_rsrc(slice, (uint32_t)(((long)(L - 1) * ld + 64) * 2));
        rowbytes = (uint32_t)(ld * 2);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = (wave * 2 + jj) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ swz(row);
            voff[jj] = (uint32_t)(((long)row * ld + chunk * 8) * 2);
        }
    }
    __device__ __forceinline__ void issue(int p0, char* tile, int wave) const {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) blds16(rsrc, voff[jj], (uint32_t)p0 * rowbytes, tile + (wave * 2 + jj) * 1024);
    }
};

// MFMA 32x32x16 operand whose rows are positions: row `row`, 8 consecutive d = 16 s + 8 h ..
__device__ __forceinline__ bf16x8_t row_frag(const char* tile, int row, int s, int h) {
    return *reinterpret_cast<const bf16x8_t*>(tile + lds_off(row, 2 * s + h));
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
// MFMA operand whose rows are d (= 32 dt + lane&31) and whose reduction index is the position: reduction step
// m (16 positions) of sub-tile ss (32 positions); element e of half h <-> position 32 ss + 16 m + 4 h + (e&3) +
// 8 (e>>2), the order in which pack_half() lays out accumulator rows.  Two transpose reads, 8 positions apart (x2 = 0, 1).
// Issued as inline asm (see gemm_core.h: the builtin would drain the LDS-DMA prefetch); await with lgkm_wait_tied.
template <int DT, int SS, int M>
__device__ __forceinline__ bf16x8_t tr_frag(uint32_t base, const TrLane& tl) {
    const bf16x4_t lo = ds_read_tr16<(32 * SS + 16 * M) * 128>(base + tl.o[DT][0]);
    const bf16x4_t hi = ds_read_tr16<(32 * SS + 16 * M + 8) * 128>(base + tl.o[DT][1]);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// the four fragments of sub-tile SS in MFMA order: (dt 0, m 0), (dt 1, m 0), (dt 0, m 1), (dt 1, m 1)
template <int SS>
__device__ __forceinline__ void tr_frags4(uint32_t base, const TrLane& tl, bf16x8_t (&f)[4]) {
    f[0] = tr_frag<0, SS, 0>(base, tl), f[1] = tr_frag<1, SS, 0>(base, tl);
    f[2] = tr_frag<0, SS, 1>(base, tl), f[3] = tr_frag<1, SS, 1>(base, tl);
}
__device__ __forceinline__ void tr_frags4(int ss, uint32_t base, const TrLane& tl, bf16x8_t (&f)[4]) {
    if (ss == 0)
        tr_frags4<0>(base, tl, f);
    else
        tr_frags4<1>(base, tl, f);
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
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
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
// a[r] = exp2(a[r] * sc + nb) for all 16 registers, returns their sum (packed fma / add)
__device__ __forceinline__ float exp2_affine_sum(f32x16& a, float sc, float nb) {
    const f32x2 sc2 = {sc, sc}, nb2 = {nb, nb};
    f32x2 sum = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x2 v = {a[2 * i], a[2 * i + 1]};
        v = __builtin_elementwise_fma(v, sc2, nb2);
        v[0] = fast_exp2(v[0]), v[1] = fast_exp2(v[1]);
        a[2 * i] = v[0], a[2 * i + 1] = v[1];
        sum += v;
    }
    return sum[0] + sum[1];
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

// Does any (q, key) pair of a 32x32 sub-tile need the mask predicate?  Wave-uniform.
__device__ __forceinline__ bool tile_needs_mask(const MaskSpec& m, int q_lane, int key0, int L) {
    bool need = key0 + 32 > L;
    if (m.mode == 1) need = need || (key0 + 31 > q_lane);
    if (m.mode == 2) need = need || q_lane == m.r0 || q_lane == m.r1;
    return __any(need);
}

// blocks of one (batch, head) are consecutive AND on one XCD: they share K/V (or Q/dO) through that XCD's L2
// ---- tail split.  The grids of these kernels do not divide into the resident block slots (L = 579, 18 sequences: 1,080 blocks against
// 1,024 / 768 / 512), and a block lasts 25-30 us whatever shares its CU: the 56 blocks of the last, 5-%-full round cost a whole round.
// With a workspace the launcher cuts exactly those blocks -- the last `tail` in dispatch order -- into `parts` blocks over disjoint ranges
// of the streamed dimension (query tiles in dK/dV); they write their fp32 accumulators to the workspace and a small second launch adds
// the parts in order (a fixed summation order: still bit-reproducible) and runs the normal epilogue.  The last round then lasts a
// quarter as long.  nfull = blocks that run whole; dispatch ids >= nfull are (tail block j, part p) = ((id - nfull) / parts, % parts).
struct TailSplit {
    int nfull, parts, nlogical;
    float* ws;  // [tail][parts][4 waves][64 accumulator registers][64 lanes] fp32
};
__device__ __forceinline__ void block_coords_of(int dispatch_id, int nblocks, int nrt, int H, int& rt, int& hd, int& b) {
    const int id = xcd_remap(dispatch_id, nblocks);
    rt = id % nrt;
    const int bh = id / nrt;
    hd = bh % H;
    b = bh / H;
}
__device__ __forceinline__ void block_coords(int nrt, int H, int& rt, int& hd, int& b) {
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    rt = id % nrt;
    const int bh = id / nrt;
    hd = bh % H;
    b = bh / H;
}

// ---- "resident" form of the three kernels (L <= 608, i.e. the BERT configurations): ONE block of 8 waves per (batch, head) stages
// the whole K and V (or Q and dO) of that head into LDS once -- 2 x 76 KiB of the 160 KiB -- and every wave then runs its 32-row units
// (u = wave, wave + 8, wave + 16) against it without any further barrier or staging.  Why: the streaming kernels re-stage K/V once
// per 128-row block (5x per head), park at one barrier per 64-key tile (41 % of a wave's cycles, profiles/
// r03_pmc_attention_before_interleave.txt), and their grid of 1,080 blocks does not divide into the 1,024 / 768 / 512 resident
// slots (a second or third round that is 5 % full).  Here the grid is B*H = 216 blocks, one round.  The first unit of every wave
// starts while the later tiles are still in flight: all LDS-DMA requests are issued up front in tile order, and the first pass
// waits per tile (counted vmcnt + barrier); passes two and three are barrier-free.  The per-row arithmetic and its order are the
// streaming kernels' own (same code), so the results are bit-identical.
constexpr int RES_WAVES = 8, RES_UNITS = 3, RES_MAX_L = 608;  // 19 units of 32 rows: 8 waves x up to 3 units

// s_waitcnt vmcnt(n) for a wave-uniform even n in 0..40 (any other value: wait for everything, which never under-waits)
__device__ __forceinline__ void wait_vm_even(int n) {
    switch (n) {
#define MMVID_VM_CASE(N) \
    case N: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); break;
        MMVID_VM_CASE(2) MMVID_VM_CASE(4) MMVID_VM_CASE(6) MMVID_VM_CASE(8) MMVID_VM_CASE(10) MMVID_VM_CASE(12) MMVID_VM_CASE(14)
        MMVID_VM_CASE(16) MMVID_VM_CASE(18) MMVID_VM_CASE(20) MMVID_VM_CASE(22) MMVID_VM_CASE(24) MMVID_VM_CASE(26) MMVID_VM_CASE(28)
        MMVID_VM_CASE(30) MMVID_VM_CASE(32) MMVID_VM_CASE(34) MMVID_VM_CASE(36) MMVID_VM_CASE(38) MMVID_VM_CASE(40)
#undef MMVID_VM_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Requests every 8-row piece of two token-major [rows x 64 d] operands (rows % 32 == 0) into resA / resB, in tile order: wave w owns
// piece 8 t + w of tile t.  Returns whether this wave issued for the LAST tile (it may be a half tile: 4 pieces) -- the count of
// requests that are younger than tile t's is then 2 * (ntiles - 1 - t), minus 2 if not.
template <int W = RES_WAVES>
__device__ __forceinline__ bool res_stage(const bf16_t* A, long lda, const bf16_t* Bm, long ldb, int L, int rows, char* resA, char* resB,
                                          int wave, int lane) {
    if (W > 8 && wave >= 8) return false;  // (a tile is 8 pieces: waves 8.. of a 16-wave block request nothing)
    const rsrc_t ra = make_rsrc(A, (uint32_t)(((long)(L - 1) * lda + 64) * 2));
    const rsrc_t rb = make_rsrc(Bm, (uint32_t)(((long)(L - 1) * ldb + 64) * 2));
    const int npieces = rows >> 3, ntiles = (rows + 63) >> 6;
    bool last = false;
    for (int t = 0; t < ntiles; ++t) {
        const int p = 8 * t + wave;  // wave-uniform
        if (p >= npieces) break;
        const int row = p * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        blds16(ra, (uint32_t)(((long)row * lda + chunk * 8) * 2), 0, resA + p * 1024);
        blds16(rb, (uint32_t)(((long)row * ldb + chunk * 8) * 2), 0, resB + p * 1024);
        last = t == ntiles - 1;
    }
    return last;
}
// first pass, before tile t: this wave's requests for tiles 0..t have landed, then every wave's (barrier)
template <int W = RES_WAVES>
__device__ __forceinline__ void res_wait_tile(int t, int ntiles, bool issued_last) {
    const int younger = 2 * (ntiles - 1 - t) - ((issued_last || t == ntiles - 1) ? 0 : 2);
    wait_vm_even(younger < 0 ? 0 : younger);  // (a wave that requested nothing passes at once: its counter is 0)
    __builtin_amdgcn_s_barrier();
}

// ------------------------------------------------------------------------------------------ forward
template <int MINB, bool RES, int W = RES_WAVES, bool TRACE = false>  // W: waves of the resident form (8: up to 3 units per wave; 16: up to 2)
__global__ __launch_bounds__(RES ? W * 64 : 256, RES ? 1 : MINB) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, long ld, int L,
                                                                                         int H, int E, int nrt, float scale_log2,
                                                                                         MaskSpec mask, bf16_t* __restrict__ out,
                                                                                         long ldo, float* __restrict__ lse2,
                                                                                         unsigned long long* trace) {
    __shared__ __attribute__((aligned(16))) char smem[RES ? 1 : 2][RES ? 16 : 2 * TILE];  // streaming: K tile, V tile, two stages
    extern __shared__ __attribute__((aligned(16))) char rsm[];                             // resident: K rows, V rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, h = lane >> 5;
    int qt = 0, hd, b;
    if constexpr (RES)
        hd = blockIdx.x % H, b = blockIdx.x / H;
    else
        block_coords(nrt, H, qt, hd, b);
    const bf16_t* Kbase = qkv + (long)b * L * ld + E + hd * 64;
    const bf16_t* Vbase = Kbase + E;
    const TrLane trl = tr_lane_offs(lane);
    const int rows = ((L + 31) >> 5) * 32, ntiles_all = (L + 63) >> 6;
    // measurement only (mmvid_attention_trace): blocks 100 and gridDim.x - 8 (a first-round and a tail-round block) stamp the
    // 100-MHz wall clock at 7 points of every tile, per wave: [2 blocks][4 waves][16 tiles][8]
    unsigned long long* tr = nullptr;
    if (TRACE && trace && lane == 0 && (blockIdx.x == 100 || blockIdx.x == gridDim.x - 8))
        tr = trace + ((blockIdx.x == 100 ? 0 : 4) + wave) * 16 * 8;
    if (TRACE && trace && tid == 0) trace[1024 + 2 * blockIdx.x] = wall_clock64();  // block entry ([1024 + 2 b], exit at + 1)
    if (TRACE && trace && tid == 0 && blockIdx.x == 100) trace[1000] = clock64(), trace[1001] = wall_clock64();  // shader clock vs wall clock
#define ATTN_STAMP(i) \
    if constexpr (TRACE)  \
        if (tr && t < 16) tr[t * 8 + (i)] = wall_clock64();
    const char* Kres = rsm;
    const char* Vres = rsm + rows * 128;
    bool issued_last = false;
    // resident: the Q fragments of all of this wave's units first (they complete before the DMA requests are issued, so that no
    // compiler-placed wait for them can end up waiting for the whole K/V as well)
    constexpr int NU = (RES_MAX_L / 32 + W - 1) / W;  // units per wave
    constexpr int NPRE = W >= 16 ? 1 : NU;            // (16 waves: 128 registers per wave -- only the first unit's Q is preloaded)
    bf16x8_t qres[RES ? NPRE : 1][4];
    if constexpr (RES) {
#pragma unroll
        for (int ui = 0; ui < NPRE; ++ui) {
            int qq = (wave + W * ui) * 32 + l32;
            qq = qq < L ? qq : L - 1;
            const bf16_t* Qp = qkv + ((long)b * L + qq) * ld + hd * 64;
#pragma unroll
            for (int s = 0; s < 4; ++s) qres[ui][s] = *reinterpret_cast<const bf16x8_t*>(Qp + 16 * s + 8 * h);
        }
#pragma unroll
        for (int ui = 0; ui < NPRE; ++ui)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(qres[ui][0]), "+v"(qres[ui][1]), "+v"(qres[ui][2]), "+v"(qres[ui][3])::"memory");
        issued_last = res_stage<W>(Kbase, ld, Vbase, ld, L, rows, rsm, rsm + rows * 128, wave, lane);
    }
#pragma unroll
    for (int ui = 0; ui < (RES ? NU : 1); ++ui) {
    const int q_wave0 = RES ? (wave + W * ui) * 32 : qt * ROWS_PER_BLOCK + wave * 32;  // first query of the unit (wave-uniform)
    if (RES && ui > 0 && q_wave0 >= L) break;
    const int q = q_wave0 + l32;
    const int qc = q < L ? q : L - 1;
    const bool wave_active = q_wave0 < L;
    bf16x8_t qf[4];
    if (RES && ui < NPRE) {
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = qres[ui < NPRE ? ui : 0][s];
    } else {
        const bf16_t* Qp = qkv + ((long)b * L + qc) * ld + hd * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(Qp + 16 * s + 8 * h);
    }
    int kv_end = L;
    if (mask.mode == 1) {
        const int blk_end = RES ? q_wave0 + 32 : (qt + 1) * ROWS_PER_BLOCK;  // causal: the last key any of these queries may see, + 1
        if (blk_end < L) kv_end = blk_end;
    }
    const int ntiles = (kv_end + 63) >> 6;

    TileStage stK, stV;
    if constexpr (!RES) {
        stK.init(Kbase, ld, L, wave, lane), stV.init(Vbase, ld, L, wave, lane);
        stK.issue(0, smem[0], wave), stV.issue(0, smem[0] + TILE, wave);
    }

    float m_run = -INFINITY, lsum = 0.f;
    f32x16 oacc[2] = {zero16(), zero16()};
    // restricted rows (mode 2): row r0 / r1 may not see keys below c0 / c1 -- only the wave that holds such a row, and only on key
    // sub-tiles that begin below that bound, evaluates the predicate (it is ~200 instructions per sub-tile: applied to all 19
    // sub-tiles it made the blocks holding rows 65 / 66 run 40 % longer than the rest, tools/attn_timeline.py)
    const int row_kmax = mask.mode != 2 ? 0
                                        : max((mask.r0 >= q_wave0 && mask.r0 < q_wave0 + 32) ? mask.c0 : 0,
                                              (mask.r1 >= q_wave0 && mask.r1 < q_wave0 + 32) ? mask.c1 : 0);

    const int tloop = (RES && ui == 0) ? ntiles_all : ntiles;  // (first resident pass: every wave attends every tile's barrier)
    for (int t = 0; t < tloop; ++t) {
        const char* Kt = RES ? Kres + t * TILE : smem[t & 1];
        const char* Vt = RES ? Vres + t * TILE : Kt + TILE;
        if constexpr (RES) {
            if (ui == 0) res_wait_tile<W>(t, ntiles_all, issued_last);
            if (t >= ntiles) continue;
        } else {
            ATTN_STAMP(0)
            dma_publish_barrier();  // tile t has landed for every wave; everyone is done with tile t-1
            ATTN_STAMP(1)
        }
        // the next tile's LDS-DMA requests cost the wave 60-185 issue cycles per piece (4 pieces): they are issued right AFTER this
        // tile's S MFMAs, so that this cost runs under the matrix pipe's 8 x 32 cycles instead of in front of them
        const bool more = !RES && t + 1 < ntiles;
        if (!wave_active) {
            if (more) stK.issue((t + 1) * 64, smem[(t + 1) & 1], wave), stV.issue((t + 1) * 64, smem[(t + 1) & 1] + TILE, wave);
            continue;
        }
        // Both 32-key sub-tiles of the tile are in flight at once: the eight S = K Q^T MFMAs are issued back to back, and each
        // sub-tile's softmax arithmetic (VALU: exp2 is quarter rate) runs while the matrix pipe still works on the other
        // sub-tile's S or PV products.  Issued one sub-tile after the other (round 2), a wave sat in MFMA-result waits for 31 % of
        // its cycles and parked for 41 % (profiles/r03_pmc_attention_before_interleave.txt).
        const bool two = t * 64 + 32 < L;  // the second sub-tile holds live keys (block-uniform)
        f32x16 sv[2];
        sv[0] = mfma32z(row_frag(Kt, l32, 0, h), qf[0]);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) sv[0] = mfma32(row_frag(Kt, l32, ks, h), qf[ks], sv[0]);
        if (two) {
            sv[1] = mfma32z(row_frag(Kt, 32 + l32, 0, h), qf[0]);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) sv[1] = mfma32(row_frag(Kt, 32 + l32, ks, h), qf[ks], sv[1]);
        }
        if (more) {
            __builtin_amdgcn_sched_barrier(0);
            stK.issue((t + 1) * 64, smem[(t + 1) & 1], wave), stV.issue((t + 1) * 64, smem[(t + 1) & 1] + TILE, wave);
        }
        ATTN_STAMP(2)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            if (ss == 1 && !two) break;
            const int key0 = t * 64 + 32 * ss;
            f32x16& s = sv[ss];
            mfma_settle(s);
            bf16x8_t vt[4];  // V^T fragments: requested now, consumed after the softmax arithmetic
            tr_frags4(ss, lds_addr(Vt), trl, vt);
            // mask needed?  padding keys, the causal diagonal band, or (wave-constant) a restricted query row in this wave
            if (key0 + 32 > L || (mask.mode == 1 && key0 + 31 > q_wave0) || key0 < row_kmax) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = is_masked(mask, q, key0 + acc_row(r, h), L) ? -INFINITY : s[r];
            }
            float mx = half_max(max16(s)) * scale_log2;  // scale > 0
            if (__any(mx > m_run + RESCALE_THR)) {        // rare after the first tiles
                const float m_new = fmaxf(m_run, mx);
                const float alpha = fast_exp2(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
                lsum *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                m_run = m_new;
            }
            const float m_ref = (m_run == -INFINITY) ? 0.f : m_run;
            lsum += exp2_affine_sum(s, scale_log2, -m_ref);
            bf16x8_t pf[2] = {pack_half(s, 0), pack_half(s, 1)};
            lgkm_wait_tied<0>(vt[0], vt[1], vt[2], vt[3], pf[0], pf[1]);
            ATTN_STAMP(3 + 2 * ss)
            oacc[0] = mfma32(vt[0], pf[0], oacc[0]);
            oacc[1] = mfma32(vt[1], pf[0], oacc[1]);
            oacc[0] = mfma32(vt[2], pf[1], oacc[0]);
            oacc[1] = mfma32(vt[3], pf[1], oacc[1]);
            ATTN_STAMP(4 + 2 * ss)
        }
    }
#undef ATTN_STAMP
    if (!wave_active) {
        if constexpr (RES) continue; else return;
    }
    lsum = half_sum(lsum);
    mfma_settle(oacc[0]), mfma_settle(oacc[1]);
    if (q < L) {  // (lanes l and l + 32 hold the same row: the half-wave exchange inside store_row64 pairs two active lanes)
        store_row64(out + ((long)b * L + q) * ldo + hd * 64, oacc, 1.0f / lsum, h);
        if (h == 0) lse2[((long)b * H + hd) * L + q] = m_run + log2f(lsum);
    }
    if (TRACE && trace && tid == 0) trace[1024 + 2 * blockIdx.x + 1] = wall_clock64();
    if (TRACE && trace && tid == 0 && blockIdx.x == 100) trace[1002] = clock64(), trace[1003] = wall_clock64();
    }  // unit
}

// ------------------------------------------------------------------------------------------ dQ
// the per-query operands of a dQ unit: Q and dO fragments of the lane's row, -lse2, delta = rowsum(dO * O)
struct DqRow {
    bf16x8_t qf[4], dof[4];
    float neg_lse, delta;
};
// requests the row's operands (q clamped to L - 1); delta is finished by dq_row_finish() once the loads have landed
__device__ __forceinline__ void dq_row_load(DqRow& r, uint4 (&o4)[4], const bf16_t* qkv, long ld, const bf16_t* O, long ldo, const bf16_t* dO,
                                            long lddo, const float* lse2, int L, int H, int b, int hd, int q, int h) {
    const int qc = q < L ? q : L - 1;
    const bf16_t* Qp = qkv + ((long)b * L + qc) * ld + hd * 64;
    const bf16_t* dOp = dO + ((long)b * L + qc) * lddo + hd * 64;
    const bf16_t* Op = O + ((long)b * L + qc) * ldo + hd * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        r.qf[s] = *reinterpret_cast<const bf16x8_t*>(Qp + 16 * s + 8 * h);
        r.dof[s] = *reinterpret_cast<const bf16x8_t*>(dOp + 16 * s + 8 * h);
        o4[s] = *reinterpret_cast<const uint4*>(Op + 16 * s + 8 * h);
    }
    r.neg_lse = -lse2[((long)b * H + hd) * L + qc];
}
// delta[q] = sum_d dO[q][d] * O[q][d]: each half-lane holds 32 of the row's 64 d; also stored for the dK/dV kernel
__device__ __forceinline__ void dq_row_finish(DqRow& r, const uint4 (&o4)[4], float* delta, int L, int H, int b, int hd, int q, int h) {
    float d = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint4 d4 = __builtin_bit_cast(uint4, r.dof[s]);
        d += (bf_lo(o4[s].x) * bf_lo(d4.x) + bf_hi(o4[s].x) * bf_hi(d4.x)) + (bf_lo(o4[s].y) * bf_lo(d4.y) + bf_hi(o4[s].y) * bf_hi(d4.y)) +
             (bf_lo(o4[s].z) * bf_lo(d4.z) + bf_hi(o4[s].z) * bf_hi(d4.z)) + (bf_lo(o4[s].w) * bf_lo(d4.w) + bf_hi(o4[s].w) * bf_hi(d4.w));
    }
    r.delta = half_sum(d);
    if (h == 0 && q < L) delta[((long)b * H + hd) * L + q] = r.delta;
}

template <int MINB, bool RES, bool TRACE = false>
__global__ __launch_bounds__(RES ? RES_WAVES * 64 : 256, RES ? 1 : MINB) void attn_bwd_dq_kernel(
    const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ O, long ldo, const bf16_t* __restrict__ dO, long lddo,
    const float* __restrict__ lse2, float* __restrict__ delta, int L, int H, int E, int nrt, float scale, float scale_log2, MaskSpec mask,
    bf16_t* __restrict__ dqkv, long ldg, float* __restrict__ dbias, unsigned long long* trace) {
    // measurement only (mmvid_attention_trace): block entry / exit at trace[1024 + 4096 + 2 b], tile starts of blocks 100 and grid - 8
    if (TRACE && trace && threadIdx.x == 0) trace[1024 + 4096 + 2 * blockIdx.x] = wall_clock64();
    __shared__ __attribute__((aligned(16))) char smem[RES ? 1 : 2][RES ? 16 : 2 * TILE];  // streaming: K tile, V tile, two stages
    extern __shared__ __attribute__((aligned(16))) char rsm[];                             // resident: K rows, V rows
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, h = lane >> 5;
    int qt = 0, hd, b;
    if constexpr (RES)
        hd = blockIdx.x % H, b = blockIdx.x / H;
    else
        block_coords(nrt, H, qt, hd, b);
    const bf16_t* Kbase = qkv + (long)b * L * ld + E + hd * 64;
    const bf16_t* Vbase = Kbase + E;
    const TrLane trl = tr_lane_offs(lane);
    const int rows = ((L + 31) >> 5) * 32, ntiles_all = (L + 63) >> 6;
    const char* Kres = rsm;
    const char* Vres = rsm + rows * 128;
    bool issued_last = false;
    auto unit_q0 = [&](int ui) { return RES ? (wave + RES_WAVES * ui) * 32 : qt * ROWS_PER_BLOCK + wave * 32; };
    // row operands: `cur` for the unit being computed, `nxt` requested one unit ahead (resident form)
    DqRow cur, nxt;
    uint4 o4[4];
    dq_row_load(cur, o4, qkv, ld, O, ldo, dO, lddo, lse2, L, H, b, hd, unit_q0(0) + l32, h);
    dq_row_finish(cur, o4, delta, L, H, b, hd, unit_q0(0) + l32, h);
    if constexpr (RES) {
        dq_row_load(nxt, o4, qkv, ld, O, ldo, dO, lddo, lse2, L, H, b, hd, unit_q0(1) + l32, h);
        dq_row_finish(nxt, o4, delta, L, H, b, hd, unit_q0(1) + l32, h);  // (every load has landed: none is left to be waited for
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  //  behind the DMA requests that follow)
        issued_last = res_stage(Kbase, ld, Vbase, ld, L, rows, rsm, rsm + rows * 128, wave, lane);
    }
#pragma unroll 1
    for (int ui = 0; ui < (RES ? RES_UNITS : 1); ++ui) {
    const int q_wave0 = unit_q0(ui);
    if (RES && ui > 0 && q_wave0 >= L) break;
    const int q = q_wave0 + l32;
    const bool wave_active = q_wave0 < L;
    bool pend = false;  // the next unit's operands were requested at the start of this one: they fly under its work
    if (RES && ui > 0) {
        cur = nxt;
        pend = ui + 1 < RES_UNITS && unit_q0(ui + 1) < L;
        if (pend) dq_row_load(nxt, o4, qkv, ld, O, ldo, dO, lddo, lse2, L, H, b, hd, unit_q0(ui + 1) + l32, h);
    }
    const bf16x8_t (&qf)[4] = cur.qf;
    const bf16x8_t (&dof)[4] = cur.dof;
    const float neg_lse = cur.neg_lse, my_delta = cur.delta;
    int kv_end = L;
    if (mask.mode == 1) {
        const int blk_end = RES ? q_wave0 + 32 : (qt + 1) * ROWS_PER_BLOCK;
        if (blk_end < L) kv_end = blk_end;
    }
    const int ntiles = (kv_end + 63) >> 6;

    TileStage sk, sv;
    if constexpr (!RES) {
        sk.init(Kbase, ld, L, wave, lane), sv.init(Vbase, ld, L, wave, lane);
        sk.issue(0, smem[0], wave), sv.issue(0, smem[0] + TILE, wave);
    }

    f32x16 dq[2] = {zero16(), zero16()};
    // restricted rows (mode 2): row r0 / r1 may not see keys below c0 / c1 -- only the wave that holds such a row, and only on key
    // sub-tiles that begin below that bound, evaluates the predicate (it is ~200 instructions per sub-tile: applied to all 19
    // sub-tiles it made the blocks holding rows 65 / 66 run 40 % longer than the rest, tools/attn_timeline.py)
    const int row_kmax = mask.mode != 2 ? 0
                                        : max((mask.r0 >= q_wave0 && mask.r0 < q_wave0 + 32) ? mask.c0 : 0,
                                              (mask.r1 >= q_wave0 && mask.r1 < q_wave0 + 32) ? mask.c1 : 0);
    const int tloop = (RES && ui == 0) ? ntiles_all : ntiles;
    for (int t = 0; t < tloop; ++t) {
        const char* Kt = RES ? Kres + t * TILE : smem[t & 1];
        const char* Vt = RES ? Vres + t * TILE : Kt + TILE;
        if constexpr (RES) {
            if (ui == 0) res_wait_tile(t, ntiles_all, issued_last);
            if (t >= ntiles) continue;
        } else {
            if (TRACE && trace && (threadIdx.x & 63) == 0 && t < 16 && (blockIdx.x == 100 || blockIdx.x == gridDim.x - 8))
                trace[256 + ((blockIdx.x == 100 ? 0 : 4) + wave) * 16 + t] = wall_clock64();
            dma_publish_barrier();
        }
        const bool more = !RES && t + 1 < ntiles;  // (the next tile's requests go out after the first eight MFMAs: see the forward kernel)
        if (!wave_active) {
            if (more) sk.issue((t + 1) * 64, smem[(t + 1) & 1], wave), sv.issue((t + 1) * 64, smem[(t + 1) & 1] + TILE, wave);
            continue;
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const int key0 = t * 64 + 32 * ss;
            if (key0 >= L) continue;  // padding-only sub-tile
            f32x16 s = mfma32z(row_frag(Kt, 32 * ss + l32, 0, h), qf[0]);
            f32x16 dp = mfma32z(row_frag(Vt, 32 * ss + l32, 0, h), dof[0]);
#pragma unroll
            for (int ks = 1; ks < 4; ++ks) {
                s = mfma32(row_frag(Kt, 32 * ss + l32, ks, h), qf[ks], s);
                dp = mfma32(row_frag(Vt, 32 * ss + l32, ks, h), dof[ks], dp);
            }
            if (ss == 0 && more) {
                __builtin_amdgcn_sched_barrier(0);
                sk.issue((t + 1) * 64, smem[(t + 1) & 1], wave), sv.issue((t + 1) * 64, smem[(t + 1) & 1] + TILE, wave);
            }
            mfma_settle(s), mfma_settle(dp);
            bf16x8_t kt4[4];  // K^T fragments
            tr_frags4(ss, lds_addr(Kt), trl, kt4);
            if (key0 + 32 > L || (mask.mode == 1 && key0 + 31 > q_wave0) || key0 < row_kmax) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = is_masked(mask, q, key0 + acc_row(r, h), L) ? -INFINITY : s[r];
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
        }
    }
    if (RES && pend) dq_row_finish(nxt, o4, delta, L, H, b, hd, unit_q0(ui + 1) + l32, h);
    if (!wave_active) {
        if constexpr (RES) continue; else return;
    }
    mfma_settle(dq[0]), mfma_settle(dq[1]);
    if (q < L) store_row64(dqkv + ((long)b * L + q) * ldg + hd * 64, dq, scale, h);
    if (dbias) colsum_rows64(dq, scale, q < L, dbias + hd * 64, lane);
    if (TRACE && trace && threadIdx.x == 0) trace[1024 + 4096 + 2 * blockIdx.x + 1] = wall_clock64();
    }  // unit
}

// ------------------------------------------------------------------------------------------ dK, dV
constexpr int DKV_BUF = 2 * TILE + 512;  // Q tile, dO tile, lse2[64], delta[64]

template <int MINB, bool RES, bool TRACE = false>
__global__ __launch_bounds__(RES ? RES_WAVES * 64 : 256, RES ? 1 : MINB) void attn_bwd_dkv_kernel(
    const bf16_t* __restrict__ qkv, long ld, const bf16_t* __restrict__ dO, long lddo, const float* __restrict__ lse2,
    const float* __restrict__ delta, int L, int H, int E, int nrt, float scale, float scale_log2, MaskSpec mask,
    bf16_t* __restrict__ dqkv, long ldg, float* __restrict__ dbias, unsigned long long* trace, TailSplit ts) {
    if (TRACE && trace && threadIdx.x == 0) trace[1024 + 8192 + 2 * blockIdx.x] = wall_clock64();
    __shared__ __attribute__((aligned(16))) char dsm[RES ? 16 : 2 * DKV_BUF];
    extern __shared__ __attribute__((aligned(16))) char rsm[];  // resident: Q rows, dO rows, -lse2[rows], delta[rows]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l32 = lane & 31, h = lane >> 5;
    int kt = 0, hd, b;
    int part = -1, tail_j = 0;  // tail split: this block covers query tiles [part * nq / parts, (part + 1) * nq / parts) of tail block tail_j
    if constexpr (RES)
        hd = blockIdx.x % H, b = blockIdx.x / H;
    else if (ts.parts > 1) {
        int did = blockIdx.x;
        if (did >= ts.nfull) tail_j = (did - ts.nfull) / ts.parts, part = (did - ts.nfull) % ts.parts, did = ts.nfull + tail_j;
        block_coords_of(did, ts.nlogical, nrt, H, kt, hd, b);
    } else
        block_coords(nrt, H, kt, hd, b);
    const bf16_t* Qbase = qkv + (long)b * L * ld + hd * 64;
    const bf16_t* dObase = dO + (long)b * L * lddo + hd * 64;
    const float* lse_b = lse2 + ((long)b * H + hd) * L;
    const float* del_b = delta + ((long)b * H + hd) * L;
    const int nq_tiles = (L + 63) >> 6;
    const TrLane trl = tr_lane_offs(lane);
    const int rows = ((L + 31) >> 5) * 32;
    const char* Qres = rsm;
    const char* dOres = rsm + rows * 128;
    float* nlse_res = reinterpret_cast<float*>(rsm + rows * 256);
    float* del_res = nlse_res + rows;
    bool issued_last = false;
    auto unit_k0 = [&](int ui) { return RES ? (wave + RES_WAVES * ui) * 32 : kt * ROWS_PER_BLOCK + wave * 32; };
    auto load_kv = [&](int key, bf16x8_t (&kf)[4], bf16x8_t (&vf)[4]) {
        const int keyc = key < L ? key : L - 1;
        const bf16_t* Kp = qkv + ((long)b * L + keyc) * ld + E + hd * 64;
        const bf16_t* Vp = Kp + E;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8_t*>(Kp + 16 * s + 8 * h);
            vf[s] = *reinterpret_cast<const bf16x8_t*>(Vp + 16 * s + 8 * h);
        }
    };
    bf16x8_t kf[4], vf[4], kn[4], vn[4];  // this unit's key rows; the next unit's (resident form), requested one unit ahead
    load_kv(unit_k0(0) + l32, kf, vf);

    // per-query statistics of a tile: threads 0..63 carry -lse2 (-inf for padded queries: exp2(-inf) = 0),
    // threads 64..127 carry delta
    auto load_stat = [&](int t) -> float {
        if (tid >= 128) return 0.f;
        const int qq = t * 64 + (tid & 63);
        if (tid < 64) return qq < L ? -lse_b[qq] : -INFINITY;
        return qq < L ? del_b[qq] : 0.f;
    };
    if constexpr (RES) {
        load_kv(unit_k0(1) + l32, kn, vn);
        for (int r = tid; r < rows; r += RES_WAVES * 64) {
            nlse_res[r] = r < L ? -lse_b[r] : -INFINITY;
            del_res[r] = r < L ? del_b[r] : 0.f;
        }
        // every load has landed and this wave's statistics are in LDS before the DMA requests are issued (the first tile's barrier
        // publishes them); nothing is left to be waited for behind the requests
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]), "+v"(kn[0]),
                       "+v"(kn[1]), "+v"(kn[2]), "+v"(kn[3]), "+v"(vn[0]), "+v"(vn[1]), "+v"(vn[2]), "+v"(vn[3])
                     :
                     : "memory");
        issued_last = res_stage(Qbase, ld, dObase, lddo, L, rows, rsm, rsm + rows * 128, wave, lane);
    }
#pragma unroll 1
    for (int ui = 0; ui < (RES ? RES_UNITS : 1); ++ui) {
    const int key_wave0 = unit_k0(ui);
    if (RES && ui > 0 && key_wave0 >= L) break;
    const int key = key_wave0 + l32;
    const bool wave_active = key_wave0 < L;
    if (RES && ui > 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = kn[s], vf[s] = vn[s];
        if (ui + 1 < RES_UNITS && unit_k0(ui + 1) < L) load_kv(unit_k0(ui + 1) + l32, kn, vn);  // flies under this unit's work
    }
    const int t0 = (mask.mode == 1) ? key_wave0 >> 6 : 0;  // causal: only queries >= keys contribute
    const int t_begin = RES ? (ui == 0 ? 0 : t0) : (part >= 0 ? part * nq_tiles / ts.parts : ((mask.mode == 1) ? (kt * ROWS_PER_BLOCK) >> 6 : 0));
    const int t_end = (!RES && part >= 0) ? (part + 1) * nq_tiles / ts.parts : nq_tiles;

    TileStage sq, sdo;
    float stat = 0.f;
    if constexpr (!RES) {
        sq.init(Qbase, ld, L, wave, lane), sdo.init(dObase, lddo, L, wave, lane);
        sq.issue(t_begin * 64, dsm, wave), sdo.issue(t_begin * 64, dsm + TILE, wave);
        stat = load_stat(t_begin);
        if (tid < 128) reinterpret_cast<float*>(dsm + 2 * TILE)[tid] = stat;
    }

    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    for (int t = t_begin; t < t_end; ++t) {
        const int bi = (t - t_begin) & 1;
        const char* Qt = RES ? Qres + t * TILE : dsm + bi * DKV_BUF;
        const char* dOt = RES ? dOres + t * TILE : Qt + TILE;
        const float* st_nlse = RES ? nlse_res + t * 64 : reinterpret_cast<const float*>(Qt + 2 * TILE);
        const float* st_del = RES ? del_res + t * 64 : st_nlse + 64;
        char* nx = dsm + (RES ? 0 : (bi ^ 1) * DKV_BUF);
        const bool more = t + 1 < t_end;
        if constexpr (RES) {
            if (ui == 0) res_wait_tile(t, nq_tiles, issued_last);
            if (t < t0) continue;
        } else {
            if (TRACE && trace && (threadIdx.x & 63) == 0 && t < 16 && (blockIdx.x == 100 || blockIdx.x == gridDim.x - 8))
                trace[512 + ((blockIdx.x == 100 ? 0 : 4) + wave) * 16 + t] = wall_clock64();
            dma_publish_barrier();
            if (more && !wave_active) {
                sq.issue((t + 1) * 64, nx, wave), sdo.issue((t + 1) * 64, nx + TILE, wave);
                stat = load_stat(t + 1);
            }
        }
        if (wave_active) {
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int q0 = t * 64 + 32 * ss;
                if (q0 >= L) continue;  // padding-only query sub-tile: P = 0 there
                f32x16 s = mfma32z(row_frag(Qt, 32 * ss + l32, 0, h), kf[0]);
                f32x16 dp = mfma32z(row_frag(dOt, 32 * ss + l32, 0, h), vf[0]);
#pragma unroll
                for (int ks = 1; ks < 4; ++ks) {
                    s = mfma32(row_frag(Qt, 32 * ss + l32, ks, h), kf[ks], s);
                    dp = mfma32(row_frag(dOt, 32 * ss + l32, ks, h), vf[ks], dp);
                }
                if (!RES && ss == 0 && more) {  // the next tile's requests, under the first eight MFMAs (see the forward kernel)
                    __builtin_amdgcn_sched_barrier(0);
                    sq.issue((t + 1) * 64, nx, wave), sdo.issue((t + 1) * 64, nx + TILE, wave);
                    stat = load_stat(t + 1);
                }
                mfma_settle(s), mfma_settle(dp);
                bf16x8_t dot4[4], qt4[4];  // dO^T and Q^T fragments, in consumption order
                tr_frags4(ss, lds_addr(dOt), trl, dot4);
                tr_frags4(ss, lds_addr(Qt), trl, qt4);
                // mask needed?  (wave-uniform) key padding, causal diagonal region, or a restricted query row in range
                // (all wave-uniform: the wave's keys are key_wave0 .. key_wave0 + 31)
                bool nm = key_wave0 + 32 > L;
                if (mask.mode == 1) nm = nm || (key_wave0 + 31 > q0);
                if (mask.mode == 2)  // a restricted row among these queries AND some of this wave's keys below its bound
                    nm = nm || (mask.r0 >= q0 && mask.r0 < q0 + 32 && key_wave0 < mask.c0) || (mask.r1 >= q0 && mask.r1 < q0 + 32 && key_wave0 < mask.c1);
                if (nm) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] = is_masked(mask, q0 + acc_row(r, h), key, L) ? -INFINITY : s[r];
                }
                const f32x2 sc2 = {scale_log2, scale_log2};
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int ql = 32 * ss + 8 * g4 + 4 * h;  // 4 consecutive query rows: registers 4 g4 .. 4 g4 + 3
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
            }
        }
        if constexpr (!RES) {
            if (more && tid < 128) reinterpret_cast<float*>(nx + 2 * TILE)[tid] = stat;
        }
    }
    if (!wave_active) {
        if constexpr (RES) continue; else return;
    }
    mfma_settle(dk[0]), mfma_settle(dk[1]), mfma_settle(dv[0]), mfma_settle(dv[1]);
    if (!RES && part >= 0) {  // a part of a tail block: raw accumulators to the workspace, [register][lane] so that a store is 256 contiguous bytes
        float* w = ts.ws + ((long)(tail_j * ts.parts + part) * 4 + wave) * 64 * 64 + lane;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) w[(16 * dt + r) * 64] = dk[dt][r], w[(32 + 16 * dt + r) * 64] = dv[dt][r];
        return;
    }
    if (key < L) {
        bf16_t* kp = dqkv + ((long)b * L + key) * ldg + E + hd * 64;
        store_row64(kp, dk, scale, h);
        store_row64(kp + E, dv, 1.0f, h);
    }
    if (dbias) {
        colsum_rows64(dk, scale, key < L, dbias + E + hd * 64, lane);
        colsum_rows64(dv, 1.0f, key < L, dbias + 2 * E + hd * 64, lane);
    }
    if (TRACE && trace && threadIdx.x == 0) trace[1024 + 8192 + 2 * blockIdx.x + 1] = wall_clock64();
    }  // unit
}

// second launch of a tail-split dK/dV pass: one block per tail block; a wave adds the parts of its 32 keys in part order and runs the
// epilogue of attn_bwd_dkv_kernel (bf16 rows of dK, dV and their share of the in-projection's bias gradient)
__global__ __launch_bounds__(256) void attn_dkv_combine_kernel(int L, int H, int E, int nrt, float scale, bf16_t* __restrict__ dqkv, long ldg,
                                                               float* __restrict__ dbias, TailSplit ts) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, h = lane >> 5;
    int kt, hd, b;
    block_coords_of(ts.nfull + blockIdx.x, ts.nlogical, nrt, H, kt, hd, b);
    const int key_wave0 = kt * ROWS_PER_BLOCK + wave * 32, key = key_wave0 + l32;
    if (key_wave0 >= L) return;
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    for (int p = 0; p < ts.parts; ++p) {
        const float* w = ts.ws + ((long)(blockIdx.x * ts.parts + p) * 4 + wave) * 64 * 64 + lane;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dk[dt][r] += w[(16 * dt + r) * 64], dv[dt][r] += w[(32 + 16 * dt + r) * 64];
    }
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

// the resident kernels hold 2 x ceil32(L) x 128 B in LDS (+ 8 B per row of statistics in dK/dV): L <= 608.  Geometry only.
static bool attn_resident(int L) { return L <= RES_MAX_L && mmvid_option(MMVID_OPT_ATTN_RES) != 0; }  // (default 0: measured slower)

unsigned long long* g_attn_trace = nullptr;

// resident 256-thread blocks of a kernel on the whole device (occupancy query x CU count; cached per kernel)
static int attn_block_slots(const void* kernel) {
    static const void* k_cached = nullptr;
    static int slots_cached = 0;
    if (kernel != k_cached) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || hipGetDevice(&dev) != hipSuccess ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return 0;
        k_cached = kernel, slots_cached = per_cu * prop.multiProcessorCount;
    }
    return slots_cached;
}

static MaskSpec make_mask(int mode, int r0, int c0, int r1, int c1) {
    MaskSpec m;
    m.mode = mode, m.r0 = r0, m.c0 = c0, m.r1 = r1, m.c1 = c1;
    return m;
}

}  // namespace

// Measurement only: device buffer of [2 blocks][4 waves][16 tiles][8] uint64 wall-clock stamps written by the next streaming
// forward launches (tools/attn_timeline.py); NULL switches it off.
extern "C" int mmvid_attention_trace(void* dev_buf) {
    g_attn_trace = (unsigned long long*)dev_buf;
    return MMVID_OK;
}

#define ATTN_COMMON_CHECKS(name)                                                                            \
    MMVID_REQUIRE(B > 0 && L > 0 && H > 0 && E == H * 64, name ": need E == H*64 (head_dim 64), got E=%d H=%d", E, H); \
    MMVID_REQUIRE(mask_mode >= 0 && mask_mode <= 2, name ": mask_mode %d", mask_mode)

extern "C" int mmvid_attention_fwd(const void* qkv, int64_t ld, int B, int L, int H, int E, float scale, int mask_mode,
                                   int r0, int c0, int r1, int c1, void* out, int64_t ldo, float* lse2, void* stream) {
    MMVID_REQUIRE(qkv && out && lse2, "attention_fwd: null pointer");
    ATTN_COMMON_CHECKS("attention_fwd");
    MMVID_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out & 15) == 0, "attention_fwd: leading dims must be multiples of 8, out 16-byte aligned");
    MMVID_REQUIRE((int64_t)L * ld * 2 < (1ll << 31), "attention_fwd: one batch entry of qkv must be smaller than 2 GiB");
    MmvidProfScope prof(PROF_ATTN_FWD, 4.0 * B * H * (double)L * L * 64, (hipStream_t)stream);
    const int nrt = cdiv(L, ROWS_PER_BLOCK);
    if (attn_resident(L)) {
        const size_t lds = (size_t)cdiv(L, 32) * 32 * 256;
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RES_MAX_L * 256);
            attr = true;
        }
        if (mmvid_option(MMVID_OPT_ATTN_RES) == 2) {  // 16 waves (4 per SIMD), up to 2 units each
            static bool attr16 = false;
            if (!attr16) {
                (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<1, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          RES_MAX_L * 256);
                attr16 = true;
            }
            hipLaunchKernelGGL((attn_fwd_kernel<1, true, 16>), dim3(H * B), dim3(1024), lds, (hipStream_t)stream, (const bf16_t*)qkv,
                               (long)ld, L, H, E, nrt, scale * 1.4426950408889634f, make_mask(mask_mode, r0, c0, r1, c1), (bf16_t*)out,
                               (long)ldo, lse2, nullptr);
        } else
            hipLaunchKernelGGL((attn_fwd_kernel<1, true>), dim3(H * B), dim3(RES_WAVES * 64), lds, (hipStream_t)stream,
                               (const bf16_t*)qkv, (long)ld, L, H, E, nrt, scale * 1.4426950408889634f,
                               make_mask(mask_mode, r0, c0, r1, c1), (bf16_t*)out, (long)ldo, lse2, nullptr);
    } else if (mmvid_option(MMVID_OPT_ATTN_OCC) & 1)
        hipLaunchKernelGGL((attn_fwd_kernel<5, false>), dim3(nrt * H * B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ld,
                       L, H, E, nrt, scale * 1.4426950408889634f, make_mask(mask_mode, r0, c0, r1, c1), (bf16_t*)out,
                       (long)ldo, lse2, nullptr);
    else
        if (g_attn_trace)  // (measurement build of the same kernel, held to the production kernel's 4 waves per SIMD)
            hipLaunchKernelGGL((attn_fwd_kernel<4, false, RES_WAVES, true>), dim3(nrt * H * B), dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)qkv, (long)ld, L, H, E, nrt, scale * 1.4426950408889634f,
                               make_mask(mask_mode, r0, c0, r1, c1), (bf16_t*)out, (long)ldo, lse2, g_attn_trace);
        else
        hipLaunchKernelGGL((attn_fwd_kernel<2, false>), dim3(nrt * H * B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, (long)ld,
                       L, H, E, nrt, scale * 1.4426950408889634f, make_mask(mask_mode, r0, c0, r1, c1), (bf16_t*)out,
                       (long)ldo, lse2, nullptr);
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
    return mmvid_attention_bwd_ws(qkv, ld, O, ldo, dO, lddo, lse2, delta, B, L, H, E, scale, mask_mode, r0, c0, r1, c1, dqkv, ldg, dbias,
                                  nullptr, 0, stream);
}

extern "C" int mmvid_attention_bwd_ws(const void* qkv, int64_t ld, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                                      const float* lse2, float* delta, int B, int L, int H, int E, float scale,
                                      int mask_mode, int r0, int c0, int r1, int c1, void* dqkv, int64_t ldg,
                                      float* dbias, void* workspace, int64_t workspace_bytes, void* stream) {
    MMVID_REQUIRE(qkv && O && dO && lse2 && delta && dqkv, "attention_bwd: null pointer");
    const TailSplit no_split = {0, 1, 0, nullptr};
    ATTN_COMMON_CHECKS("attention_bwd");
    MMVID_REQUIRE(ld % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && ldg % 8 == 0 && ((uintptr_t)dqkv & 15) == 0,
                  "attention_bwd: leading dims must be multiples of 8, dqkv 16-byte aligned");
    MMVID_REQUIRE((int64_t)L * ld * 2 < (1ll << 31) && (int64_t)L * lddo * 2 < (1ll << 31),
                  "attention_bwd: one batch entry of qkv / dO must be smaller than 2 GiB");
    hipStream_t s = (hipStream_t)stream;
    const MaskSpec m = make_mask(mask_mode, r0, c0, r1, c1);
    const float sl2 = scale * 1.4426950408889634f;
    MmvidProfScope prof(PROF_ATTN_BWD, 10.0 * B * H * (double)L * L * 64, s);  // 5 GEMM-equivalents (recompute counted once)
    const int nrt = cdiv(L, ROWS_PER_BLOCK);
    const bool res = attn_resident(L);
    if (res) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, RES_MAX_L * 256);
            attr = true;
        }
        hipLaunchKernelGGL((attn_bwd_dq_kernel<1, true>), dim3(H * B), dim3(RES_WAVES * 64), (size_t)cdiv(L, 32) * 32 * 256, s,
                           (const bf16_t*)qkv, (long)ld, (const bf16_t*)O, (long)ldo, (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt,
                           scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias, nullptr);
    } else if (mmvid_option(MMVID_OPT_ATTN_OCC) & 2)
        hipLaunchKernelGGL((attn_bwd_dq_kernel<4, false>), dim3(nrt * H * B), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld,
                       (const bf16_t*)O, (long)ldo, (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt, scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias, nullptr);
    else if (g_attn_trace)
        hipLaunchKernelGGL((attn_bwd_dq_kernel<3, false, true>), dim3(nrt * H * B), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld,
                       (const bf16_t*)O, (long)ldo, (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt, scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias, g_attn_trace);
    else
        hipLaunchKernelGGL((attn_bwd_dq_kernel<2, false>), dim3(nrt * H * B), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld,
                       (const bf16_t*)O, (long)ldo, (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt, scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias, nullptr);
    if (res) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      RES_MAX_L * 264);
            attr = true;
        }
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<1, true>), dim3(H * B), dim3(RES_WAVES * 64), (size_t)cdiv(L, 32) * 32 * 264, s,
                           (const bf16_t*)qkv, (long)ld, (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt, scale, sl2, m,
                           (bf16_t*)dqkv, (long)ldg, dbias, nullptr, no_split);
    } else if (mmvid_option(MMVID_OPT_ATTN_OCC) & 4)
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<3, false>), dim3(nrt * H * B), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld,
                       (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt, scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias, nullptr, no_split);
    else if (g_attn_trace)
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, false, true>), dim3(nrt * H * B), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld,
                       (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt, scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias, g_attn_trace, no_split);
    else {
        // tail split (see TailSplit): needs the caller's workspace; not for the causal mask (its query range depends on the keys)
        const int nblocks = nrt * H * B, slots = attn_block_slots((const void*)attn_bwd_dkv_kernel<2, false>);
        const int nq = (L + 63) >> 6, parts = nq >= 8 ? 4 : (nq >= 4 ? 2 : 1);
        const int tail = slots > 0 ? nblocks % slots : 0;
        TailSplit ts = no_split;
        if (workspace && mask_mode != 1 && parts > 1 && nblocks > slots && tail > 0 && tail * parts <= slots / 2 &&
            (int64_t)tail * parts * 4 * 64 * 64 * 4 <= workspace_bytes && mmvid_option(MMVID_OPT_ATTN_TAIL)) {
            ts.nfull = nblocks - tail, ts.parts = parts, ts.nlogical = nblocks, ts.ws = (float*)workspace;
        }
        const int grid = ts.parts > 1 ? ts.nfull + tail * ts.parts : nblocks;
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, false>), dim3(grid), dim3(256), 0, s, (const bf16_t*)qkv, (long)ld,
                           (const bf16_t*)dO, (long)lddo, lse2, delta, L, H, E, nrt, scale, sl2, m, (bf16_t*)dqkv, (long)ldg, dbias, nullptr, ts);
        if (ts.parts > 1)
            hipLaunchKernelGGL(attn_dkv_combine_kernel, dim3(tail), dim3(256), 0, s, L, H, E, nrt, scale, (bf16_t*)dqkv, (long)ldg, dbias, ts);
    }
    MMVID_LAUNCH_CHECK("attention_bwd");
    return MMVID_OK;
}

// bytes of workspace with which mmvid_attention_bwd_ws can split the blocks of its last, partly filled round (0: nothing to split)
extern "C" int64_t mmvid_attention_bwd_workspace_bytes(int B, int L, int H) {
    const int nrt = cdiv(L, ROWS_PER_BLOCK);
    return (int64_t)nrt * H * B > 0 ? (int64_t)256 * 4 * 4 * 64 * 64 * 4 : 0;  // at most slots / 2 = 256 part blocks of 64 KiB
}
