// Shared tile machinery of the bf16 MFMA kernels (gemm.hip, conv.hip) for gfx950.
//
// Block tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves in 2x2, each wave 64x64 as 2x2
// v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Both operand tiles are 16 KiB in LDS and are filled by
// LDS-DMA (global_load_lds_dwordx4: 16 B per lane, no VGPR round trip, no ds_write).  The DMA destination is
// lane-linear (1 KiB per wave-instruction), so the bank-conflict swizzle is applied to the per-lane SOURCE
// address and again on the fragment read (guide rule 21):
//
//   row-major operand  [rows][K]  -> LDS [128 rows][64 k] (128-B rows), 16-B chunk ^= (row>>1)&7,
//                                    fragment = one ds_read_b128 (8 consecutive k of one row);
//   k-major operand    [K][rows]  -> LDS [64 k][128 rows] (256-B rows, i.e. the global layout as is),
//                                    64-B block ^= (k&3), fragment = two ds_read_b64_tr_b16 (the hardware
//                                    transpose read: lane t of a 16-lane group receives src[4j+(t>>2)][t&3],
//                                    j=0..3 -- verified on hardware by tools/gpu_probe.py), so dX / dW GEMMs need
//                                    neither transposed copies in HBM nor register transposes.
//
// Out-of-range chunks (row >= rows, k >= K) are zero-filled by the buffer range check (OperandStage below).
#pragma once
#include <type_traits>

#include "common.h"

namespace mmvid_core {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;

static __device__ uint4 g_zero16[1];  // source of out-of-range chunks (zero-initialised)

__device__ __forceinline__ void glds16(const void* src, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)lds_dst_wave_uniform, 16, 0, 0);
}

// ---- XCD-aware tile order ----------------------------------------------------------------------------------
// MI355X dispatches workgroup i to XCD i % 8, each XCD with its own L2.  Give every XCD a CONTIGUOUS chunk of
// the (n fastest, then m) tile order, so that tiles sharing an A row-block (and conv tiles sharing halo rows)
// hit the same L2 instead of pulling the operand through all eight.  Bijective for any grid size (guide T1).
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
    const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (id >> 3);
}

// ---- operand staging by LDS-DMA through a buffer descriptor ---------------------------------------------------
// buffer_load_dwordx4 ... lds: per-lane 32-bit byte offset (VGPR, computed ONCE per block) + uniform per-tile advance
// (SGPR soffset) against a bounds-checked descriptor.  Measured with the plain global_load_lds form: ~20 VALU/SALU
// instructions of 64-bit address arithmetic per 1-KiB piece made the K loop ISSUE-bound (PMC: 170 non-MFMA
// instructions per 16 MFMAs); here a piece costs its s_mov m0 and the load.  Out-of-range lanes (offset + soffset
// beyond num_records, or the explicit OOB marker) write ZEROS to LDS -- verified on hardware by
// tools/gpu_probe_buffer.py -- which pads ragged M / N / K edges without a zero page or per-lane pointer selects.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((ext_vector_type(4))) unsigned core_u32x4_t;
constexpr uint32_t OOB = 0x80000000u;  // with num_records <= 0x7fffffff: always out of range
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);  // raw, stride 0
}
__device__ __forceinline__ void blds16(rsrc_t r, uint32_t voff, uint32_t soff, char* lds_dst_wave_uniform) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)lds_dst_wave_uniform, 16, voff, soff, 0, 0);
}

// The barrier that publishes LDS-DMA'd tiles.  A buffer_load ... lds is complete for OTHER waves only after the ISSUING
// wave's vmcnt has counted it down and both waves have passed a barrier.  hipcc (ROCm 7.2) usually puts an s_waitcnt
// vmcnt(0) in front of __syncthreads() while LDS-DMA is in flight, but not always: attn_fwd_kernel's K/V loop was compiled
// with lgkmcnt(0) only, and under hipGraph replay (kernels back to back, busier memory system) a late piece was read as
// stale LDS once in a few hundred launches -> NaN rows (tools/stress_nan2.py).  So the wait is written out.
__device__ __forceinline__ void dma_publish_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ int rm_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ bf16x8_t frag_rowmajor(const char* tile, int row, int s, int fh) {
    return *reinterpret_cast<const bf16x8_t*>(tile + rm_off(row, 2 * s + fh));
}

// One GEMM operand = NSUB 128-row sub-tiles per K tile; this wave owns PPW of the 16 DMA pieces of each.
//   row-major [rows][ld] : LDS [128 rows][64 k], piece j = rows 8j..8j+7, slot lane&7 holds chunk (lane&7)^swz(row)
//   k-major   [K][ld]    : LDS [64 k][128 rows], piece j = k rows 4j..4j+3, slot lane&15 holds chunk (lane&15)^swz(k)
// The descriptor covers exactly the operand, so rows >= `rows` (row-major) and k >= K (k-major) read as zeros by the
// range check; the K tail of a row-major operand and the column tail of a k-major one use the OOB marker.
template <bool KM, int NSUB, int PPW>
struct OperandStage {
    rsrc_t rsrc;
    uint32_t voff[NSUB * PPW];
    uint32_t kstep;  // bytes per unit of k
    __device__ __forceinline__ void init(const bf16_t* base, long ld, int rows, int K, int r0, int wave, int lane) {
        if constexpr (KM) {
            rsrc = make_rsrc(base, (uint32_t)(((long)(K - 1) * ld + rows) * 2));
            kstep = (uint32_t)(ld * 2);
        } else {
            rsrc = make_rsrc(base, (uint32_t)(((long)(rows - 1) * ld + K) * 2));
            kstep = 2;
        }
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int jj = 0; jj < PPW; ++jj) {
                const int j = wave * PPW + jj;
                uint32_t v;
                if constexpr (KM) {
                    const int kr = j * 4 + (lane >> 4);
                    const int c = (lane & 15) ^ ((kr & 3) << 2);
                    const int r = r0 + sub * 128 + c * 8;
                    v = r < rows ? (uint32_t)(((long)kr * ld + r) * 2) : OOB;
                } else {
                    const int row = j * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((row >> 1) & 7);
                    const int gr = r0 + sub * 128 + row;
                    v = gr < rows ? (uint32_t)(((long)gr * ld + c * 8) * 2) : OOB;
                }
                voff[sub * PPW + jj] = v;
            }
    }
    // request tile k [k0, k0+64) into `stage` (this operand's first sub-tile); `wave` must be wave-uniform (SGPR)
    __device__ __forceinline__ void issue(int k0, int K, char* stage, int wave, int lane) const {
        const uint32_t soff = (uint32_t)k0 * kstep;
        if (!KM && k0 + BK > K) {  // K tail of a row-major operand (k-major tails fall to the range check)
            asm volatile("; K-tail tile" ::: "memory");  // keeps this a branch: the hot path must not pay for the selects
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int jj = 0; jj < PPW; ++jj) {
                    const int j = wave * PPW + jj;
                    const int row = j * 8 + (lane >> 3);
                    const int c = (lane & 7) ^ ((row >> 1) & 7);
                    blds16(rsrc, k0 + c * 8 < K ? voff[sub * PPW + jj] : OOB, soff, stage + sub * TILE_BYTES + j * 1024);
                }
        } else {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int jj = 0; jj < PPW; ++jj)
                    blds16(rsrc, voff[sub * PPW + jj], soff, stage + sub * TILE_BYTES + (wave * PPW + jj) * 1024);
        }
    }
};

// The transpose read is issued as inline asm: hipcc (ROCm 7.2) treats the ds_read_tr16 builtin as a possible LDS
// STORE and puts an s_waitcnt vmcnt(0) in front of it whenever LDS-DMA is in flight -- i.e. it would drain the
// prefetch of tile t+1 before the first fragment of tile t is read.  In asm form the compiler does not track the
// read either, so its completion is awaited by hand: lgkm_wait_tied<N>() = s_waitcnt lgkmcnt(N) with the fragment
// registers tied to it (a register-only MFMA cannot be hoisted above it).  LDS returns in order and the counter
// also counts the compiler's own reads, so waiting for "<= N younger asm reads" can only over-wait.
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(size_t)(lds_void_t*)p; }
template <int OFF>
__device__ __forceinline__ bf16x4_t ds_read_tr16(uint32_t addr) {
    bf16x4_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void lgkm_wait_tied(bf16x8_t& a, bf16x8_t& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lgkm_wait_tied(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
// 4 fragments + the two B operands they will be multiplied with: places the wait AFTER the arithmetic that produced x, y
template <int N>
__device__ __forceinline__ void lgkm_wait_tied(bf16x8_t& a, bf16x8_t& b, bf16x8_t& c, bf16x8_t& d, bf16x8_t& x,
                                               bf16x8_t& y) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(x), "+v"(y) : "n"(N) : "memory");
}
// per-lane byte offset into a k-major tile for the 32 operand rows [rowbase, rowbase+32): k row 8 (G>>1) + (si>>2),
// 4 rows starting at rowbase + 16 (G&1) + 4 (si&3)
__device__ __forceinline__ int km_lane_off(int rowbase, int lane) {
    const int G = lane >> 4, si = lane & 15;
    const int kk = 8 * (G >> 1) + (si >> 2);
    const int mbyte = (rowbase + 16 * (G & 1) + 4 * (si & 3)) * 2;
    return kk * 256 + (mbyte ^ ((kk & 3) << 6));
}
// 8 consecutive k (k-step S of 16) for row `rowbase + (lane&31)`: two transpose reads 4 k-rows apart.
template <int S>
__device__ __forceinline__ bf16x8_t frag_kmajor(uint32_t lane_addr) {
    const bf16x4_t lo = ds_read_tr16<S * 4096>(lane_addr);
    const bf16x4_t hi = ds_read_tr16<S * 4096 + 1024>(lane_addr);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- block shapes --------------------------------------------------------------------------------------------
// WM = wave rows of the block (each wave 64x64, two wave columns):
//   WM = 2: 128x128 tile, 256 threads, 2 LDS stages of 32 KiB, two blocks per CU.
//   WM = 4: 256x128 tile, 512 threads, THREE stages of 48 KiB (A = two 128-row sub-tiles, then B), one block per CU:
//           the LDS-DMA of tile t+2 is in flight while tile t is multiplied (counted s_waitcnt vmcnt, raw s_barrier),
//           so the ~2 us load latency is covered by two tiles of MFMA work instead of one.
template <int WM>
struct BlockShape {
    static constexpr int THREADS = WM * 128;
    static constexpr int ROWS = WM * 64;
    static constexpr int NSUB = WM / 2;                        // 128-row A sub-tiles
    static constexpr int PPW = 8 / WM;                         // DMA pieces per wave per 16-KiB tile
    static constexpr int STAGE_BYTES = (NSUB + 1) * TILE_BYTES;
    static constexpr int NSTAGE = WM == 2 ? 2 : 3;
    static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
    static constexpr int DMA_PER_TILE = (NSUB + 1) * PPW;      // per wave
};
// wait until this wave's DMA of the current tile has landed (the next tile's may stay in flight), then barrier
template <int KEEP>
__device__ __forceinline__ void wait_dma_and_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
    __builtin_amdgcn_s_barrier();
}

// the two 32-row fragments (i = 0, 1) of a wave's 64 operand rows for k-step S
template <bool KM, int S>
__device__ __forceinline__ void load_frags(const char* tile, const uint32_t (&km_addr)[2], int rowbase, int lane,
                                           bf16x8_t (&f)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if constexpr (KM)
            f[i] = frag_kmajor<S>(km_addr[i]);
        else
            f[i] = frag_rowmajor(tile, rowbase + i * 32 + (lane & 31), S, lane >> 5);
    }
}
__device__ __forceinline__ void mma_step(const bf16x8_t (&af)[2], const bf16x8_t (&bfr)[2], f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
}
template <bool AKM, bool BKM, int N>
__device__ __forceinline__ void frags_ready(bf16x8_t (&af)[2], bf16x8_t (&bfr)[2]) {
    if constexpr (AKM && BKM)
        lgkm_wait_tied<N>(af[0], af[1], bfr[0], bfr[1]);
    else if constexpr (AKM)
        lgkm_wait_tied<N>(af[0], af[1]);
    else if constexpr (BKM)
        lgkm_wait_tied<N>(bfr[0], bfr[1]);
}

// One 64-deep K tile: acc[i][j] (+)= B-frag(j) x A-frag(i)  (operands swapped: a lane gets 4 consecutive n).
// Fragments of k-step s+1 are requested before the MFMAs of step s.
template <bool AKM, bool BKM>
__device__ __forceinline__ void mma_tile(const char* At, const char* Bt, f32x16 (&acc)[2][2], int wm, int wn, int lane) {
    constexpr int NASM = (AKM ? 4 : 0) + (BKM ? 4 : 0);  // asm reads per k-step
    uint32_t ka[2] = {0, 0}, kb[2] = {0, 0};
    if constexpr (AKM) {
        ka[0] = lds_addr(At) + km_lane_off(wm * 64, lane), ka[1] = lds_addr(At) + km_lane_off(wm * 64 + 32, lane);
    }
    if constexpr (BKM) {
        kb[0] = lds_addr(Bt) + km_lane_off(wn * 64, lane), kb[1] = lds_addr(Bt) + km_lane_off(wn * 64 + 32, lane);
    }
    bf16x8_t a0[2], b0[2], a1[2], b1[2];
    load_frags<AKM, 0>(At, ka, wm * 64, lane, a0), load_frags<BKM, 0>(Bt, kb, wn * 64, lane, b0);
    load_frags<AKM, 1>(At, ka, wm * 64, lane, a1), load_frags<BKM, 1>(Bt, kb, wn * 64, lane, b1);
    frags_ready<AKM, BKM, NASM>(a0, b0);
    mma_step(a0, b0, acc);
    load_frags<AKM, 2>(At, ka, wm * 64, lane, a0), load_frags<BKM, 2>(Bt, kb, wn * 64, lane, b0);
    frags_ready<AKM, BKM, NASM>(a1, b1);
    mma_step(a1, b1, acc);
    load_frags<AKM, 3>(At, ka, wm * 64, lane, a1), load_frags<BKM, 3>(Bt, kb, wn * 64, lane, b1);
    frags_ready<AKM, BKM, NASM>(a0, b0);
    mma_step(a0, b0, acc);
    frags_ready<AKM, BKM, 0>(a1, b1);
    mma_step(a1, b1, acc);
}

// ---- ping-pong K loop for the 256x128 / 3-stage block (option gemm_sched = 1) -----------------------------------
// The 8 waves form two groups (waves 0-3 and 4-7: one wave of each per SIMD) that run one barrier apart.  A K tile is
// two phases of two k-steps; in a phase a wave first requests its 8 fragments (and issues its share of the DMA of tile
// t+2), then, between two barriers, runs its 8 MFMAs at raised priority.  Because the groups are staggered, on every
// SIMD one wave is in its MFMA cluster while the other is in its load part: the matrix pipe no longer idles while a
// block waits at the per-tile barrier of the plain loop (PMC: 46 % of wave time in s_waitcnt there).
//   ordering rules (LDS-DMA is only ordered by the ISSUING wave's vmcnt + a barrier the reader has passed):
//   * tile t+2 is requested at the START of tile t: the other group is at most one barrier behind, i.e. past the last
//     ds_read of tile t-1 (whose buffer is being refilled) -- its reads completed before its previous MFMA cluster;
//   * every wave waits for ITS pieces of tile t+1 before barrier 3 of tile t (vmcnt(6): tile t+2's six may fly); the
//     leading group first reads tile t+1 after its barrier 4, which the trailing group reaches after that wait.
template <bool AKM, bool BKM, int H>
__device__ __forceinline__ void pp_load_half(const char* At, const char* Bt, const uint32_t (&ka)[2], const uint32_t (&kb)[2],
                                             int wm, int wn, int lane, bf16x8_t (&a)[2][2], bf16x8_t (&b)[2][2]) {
    load_frags<AKM, 2 * H>(At, ka, wm * 64, lane, a[0]), load_frags<BKM, 2 * H>(Bt, kb, wn * 64, lane, b[0]);
    load_frags<AKM, 2 * H + 1>(At, ka, wm * 64, lane, a[1]), load_frags<BKM, 2 * H + 1>(Bt, kb, wn * 64, lane, b[1]);
}
template <bool AKM, bool BKM, class Mid>
__device__ __forceinline__ void pp_compute_half(bf16x8_t (&a)[2][2], bf16x8_t (&b)[2][2], f32x16 (&acc)[2][2], Mid mid) {
    constexpr int NASM = (AKM ? 4 : 0) + (BKM ? 4 : 0);
    __builtin_amdgcn_s_setprio(1);
    frags_ready<AKM, BKM, NASM>(a[0], b[0]);
    mma_step(a[0], b[0], acc);
    mid();  // DMA requests issued from inside the MFMA cluster ride in the matrix pipe's shadow
    frags_ready<AKM, BKM, 0>(a[1], b[1]);
    mma_step(a[1], b[1], acc);
    __builtin_amdgcn_s_setprio(0);
}
// issueA(t, stage) / issueB(t, stage): request this wave's pieces of the A / B part of tile t into `stage`.  The requests of tile t+2
// are issued from inside the MFMA clusters (after the first four MFMAs of each), which shortens the load part to the eight fragment
// reads (issued in the load parts instead: +0.5 % step time, profiles/r01_ab_gemm_sched*.log).  The refilled buffer is tile t-1's,
// which the other group finished reading before its previous cluster, and the wait before barrier 3 lets exactly the already-issued
// pieces of tile t+2 stay in flight.
template <bool AKM, bool BKM, class IssueA, class IssueB>
__device__ __forceinline__ void k_loop_pingpong(char* smem, int nt, int wave, int lane, int wm, int wn, f32x16 (&acc)[2][2],
                                                IssueA issueA, IssueB issueB) {
    using S = BlockShape<4>;
    constexpr int PIECES_A = S::NSUB * S::PPW;  // this wave's A pieces of a tile; the B pieces are PPW
    char* b0 = smem;
    char* b1 = smem + S::STAGE_BYTES;
    char* b2 = smem + 2 * S::STAGE_BYTES;
    if (nt <= 0) return;
    issueA(0, smem), issueB(0, smem);  // the first two K tiles of the output tile (stages 0 and 1)
    if (nt > 1) {
        issueA(1, smem + S::STAGE_BYTES), issueB(1, smem + S::STAGE_BYTES);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::DMA_PER_TILE) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                 // tile 0 is visible to everyone
    if (wave >= 4) __builtin_amdgcn_s_barrier();  // the trailing group starts one barrier late
    for (int t = 0; t < nt; ++t) {
        const char* At = b0 + (wm >> 1) * TILE_BYTES;
        const char* Bt = b0 + S::NSUB * TILE_BYTES;
        uint32_t ka[2] = {0, 0}, kb[2] = {0, 0};
        if constexpr (AKM) {
            ka[0] = lds_addr(At) + km_lane_off((wm & 1) * 64, lane), ka[1] = lds_addr(At) + km_lane_off((wm & 1) * 64 + 32, lane);
        }
        if constexpr (BKM) {
            kb[0] = lds_addr(Bt) + km_lane_off(wn * 64, lane), kb[1] = lds_addr(Bt) + km_lane_off(wn * 64 + 32, lane);
        }
        const bool more = t + 2 < nt;
        bf16x8_t a[2][2], b[2][2];
        // ---- phase A: k-steps 0, 1
        pp_load_half<AKM, BKM, 0>(At, Bt, ka, kb, wm & 1, wn, lane, a, b);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // 1
        pp_compute_half<AKM, BKM>(a, b, acc, [&]() {
            if (more) issueA(t + 2, b2);
        });
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // 2
        // ---- phase B: k-steps 2, 3
        pp_load_half<AKM, BKM, 1>(At, Bt, ka, kb, wm & 1, wn, lane, a, b);
        if (more)  // own pieces of tile t+1 landed; the A pieces of tile t+2 may stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES_A) : "memory");
        else if (t + 1 < nt)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // 3
        pp_compute_half<AKM, BKM>(a, b, acc, [&]() {
            if (more) issueB(t + 2, b2);
        });
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // 4
        char* tmp = b0;
        b0 = b1, b1 = b2, b2 = tmp;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();  // the leading group waits for the trailing one
}

// ---- loader-wave K loop (block = 8 MFMA waves + 1 loader wave) -------------------------------------------------------------
// Measured on the ping-pong loop above (round 3's per-block time stamps, profiles/r03_gemm_timeline_*.log): a K tile takes 0.9 us where its 128 MFMAs need 0.55 us, because
// every wave interleaves its MFMAs with its share of the LDS-DMA requests and a vector-memory instruction costs its wave 60-185
// issue cycles on this chip (MI355X_MICROARCH.md: "LDS-DMA piece issue cost").  Here the eight MFMA waves issue NO vector-memory
// instruction inside the K loop; loader waves request the 48 one-KiB pieces of a K tile (a third after each of the first three
// barriers of the tile, so that a refilled stage is never touched before both wave groups have passed their last read of it),
// wait for their own requests with counted vmcnt and publish them through the barriers all waves of the block share.
template <bool KM, int NL = 4>
struct LoaderStage {
    static constexpr int PG = 16 / NL;  // this loader wave's pieces of a 16-piece (128-row) sub-tile
    rsrc_t rsrc;
    long ld;
    int rows, r0;
    // Per-lane byte offsets of this loader wave's pieces (PG w .. PG w + PG - 1 of each 128-row sub-tile), computed ONCE per output tile: in
    // the K loop a piece is then `s_mov m0` + the load with an SGPR K offset.  Measured (tools/gemm_kloop_anatomy.py): the loop is paced
    // by the loader waves' instruction stream -- a dozen extra VALU/SALU per piece there cost the whole block 20 % of its K loop.
    uint32_t voff[2][PG];
    uint32_t kstep;  // bytes per unit of k
    __device__ __forceinline__ void init(const bf16_t* base, long ld_, int rows_, int K, int r0_) {
        ld = ld_, rows = rows_, r0 = r0_;
        rsrc = KM ? make_rsrc(base, (uint32_t)(((long)(K - 1) * ld + rows) * 2)) : make_rsrc(base, (uint32_t)(((long)(rows - 1) * ld + K) * 2));
    }
    __device__ __forceinline__ void init_offsets(int nsub, int w, int lane) {
        kstep = KM ? (uint32_t)(ld * 2) : 2u;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int jj = 0; jj < PG; ++jj) {
                const int j = w * PG + jj;
                uint32_t v = OOB;
                if (sub < nsub) {
                    if constexpr (KM) {
                        const int kr = j * 4 + (lane >> 4);
                        const int c = (lane & 15) ^ ((kr & 3) << 2);
                        const int r = r0 + sub * 128 + c * 8;
                        if (r < rows) v = (uint32_t)(((long)kr * ld + r) * 2);
                    } else {
                        const int row = j * 8 + (lane >> 3);
                        const int c = (lane & 7) ^ ((row >> 1) & 7);
                        const int gr = r0 + sub * 128 + row;
                        if (gr < rows) v = (uint32_t)(((long)gr * ld + c * 8) * 2);
                    }
                }
                voff[sub][jj] = v;
            }
    }
    // piece PG w + jj of sub-tile `sub` of a K tile that lies entirely below K (row-major) / any K tile (k-major: rows k >= K fall to the range check)
    __device__ __forceinline__ void piece_fast(int k0, char* tile, int sub, int jj, int w) const {
        blds16(rsrc, voff[sub][jj], (uint32_t)k0 * kstep, tile + (w * PG + jj) * 1024);
    }
    // piece j (0..15) of 128-row sub-tile `sub` of K tile [k0, k0 + 64) -> tile + j KiB
    __device__ __forceinline__ void piece(int k0, int K, char* tile, int sub, int j, int lane) const {
        uint32_t v, soff;
        if constexpr (KM) {
            const int kr = j * 4 + (lane >> 4);
            const int c = (lane & 15) ^ ((kr & 3) << 2);
            const int r = r0 + sub * 128 + c * 8;
            v = r < rows ? (uint32_t)(((long)kr * ld + r) * 2) : OOB;  // k >= K: beyond the descriptor (zero-filled)
            soff = (uint32_t)((long)k0 * ld * 2);
        } else {
            const int row = j * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            const int gr = r0 + sub * 128 + row;
            v = (gr < rows && k0 + c * 8 < K) ? (uint32_t)(((long)gr * ld + c * 8) * 2) : OOB;
            soff = (uint32_t)k0 * 2;
        }
        blds16(rsrc, v, soff, tile + j * 1024);
    }
};
// One vector-memory instruction costs its WAVE ~60 issue cycles, so one loader wave tops out near 40 GB/s (measured: K tile 1.14 us
// with a single loader against 0.9 us with every wave loading).  NLOAD = 4 loader waves, one per SIMD: loader w requests pieces
// 4 w .. 4 w + 3 of each 16-piece group (A sub-tile 0, A sub-tile 1, B), i.e. 12 of the 48 pieces of a K tile, and waits for them.
constexpr int NLOAD = 4;
template <bool AKM, bool BKM, int NL>
__device__ __forceinline__ void loader_issue_group(const LoaderStage<AKM, NL>& sa, const LoaderStage<BKM, NL>& sb, int k0, int K, char* stage, int g,
                                                   int w, int lane) {
    constexpr int PG = 16 / NL;
    if (g < 2) {
        if (AKM || k0 + BK <= K) {
#pragma unroll
            for (int jj = 0; jj < PG; ++jj) sa.piece_fast(k0, stage + g * TILE_BYTES, g, jj, w);
        } else {  // the K tail of a row-major operand: per-lane column check
            asm volatile("; K-tail tile" ::: "memory");
#pragma unroll
            for (int jj = 0; jj < PG; ++jj) sa.piece(k0, K, stage + g * TILE_BYTES, g, w * PG + jj, lane);
        }
    } else {
        if (BKM || k0 + BK <= K) {
#pragma unroll
            for (int jj = 0; jj < PG; ++jj) sb.piece_fast(k0, stage + 2 * TILE_BYTES, 0, jj, w);
        } else {
            asm volatile("; K-tail tile" ::: "memory");
#pragma unroll
            for (int jj = 0; jj < PG; ++jj) sb.piece(k0, K, stage + 2 * TILE_BYTES, 0, w * PG + jj, lane);
        }
    }
}
template <bool AKM, bool BKM, int NL>
__device__ __forceinline__ void loader_prologue(const LoaderStage<AKM, NL>& sa, const LoaderStage<BKM, NL>& sb, char* smem, int kt0, int nt, int K,
                                                int w, int lane) {
    using S = BlockShape<4>;
    if (nt <= 0) return;
#pragma unroll
    for (int g = 0; g < 3; ++g) loader_issue_group<AKM, BKM, NL>(sa, sb, kt0 * BK, K, smem, g, w, lane);
    if (nt > 1) {
#pragma unroll
        for (int g = 0; g < 3; ++g) loader_issue_group<AKM, BKM, NL>(sa, sb, (kt0 + 1) * BK, K, smem + S::STAGE_BYTES, g, w, lane);
    }
}
// a loader wave's K loop: the prologue (its pieces of K tiles 0 and 1) has been requested.  (The loop is paced by this instruction
// stream: a dozen extra SALU / VALU per piece here cost the block 20 % of its K loop, profiles/r04_gemm_kloop_anatomy*.log.)
template <bool AKM, bool BKM, int NL>
__device__ __forceinline__ void k_loop_loader(const LoaderStage<AKM, NL>& sa, const LoaderStage<BKM, NL>& sb, char* smem, int kt0, int nt, int K,
                                              int w, int lane) {
    using S = BlockShape<4>;
    constexpr int PER_TILE = 48 / NL;  // this wave's pieces of a K tile
    char* b0 = smem;
    char* b1 = smem + S::STAGE_BYTES;
    char* b2 = smem + 2 * S::STAGE_BYTES;
    if (nt <= 0) return;
    if (nt > 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");  // own pieces of K tile 0 have landed (tile 1's may fly)
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // tile 0 is visible to everyone
    for (int t = 0; t < nt; ++t) {
        const bool more = t + 2 < nt;
        const int k0 = (kt0 + t + 2) * BK;
        __builtin_amdgcn_s_barrier();  // 1: the trailing group has finished its last read of tile t-1 (whose stage b2 is refilled)
        if (more) loader_issue_group<AKM, BKM, NL>(sa, sb, k0, K, b2, 0, w, lane);
        __builtin_amdgcn_s_barrier();  // 2
        if (more) loader_issue_group<AKM, BKM, NL>(sa, sb, k0, K, b2, 1, w, lane);
        __builtin_amdgcn_s_barrier();  // 3
        if (more) loader_issue_group<AKM, BKM, NL>(sa, sb, k0, K, b2, 2, w, lane);
        if (t + 1 < nt) {  // tile t+1 has landed before the barrier after which the leading group reads it
            if (more)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // 4
        char* tmp = b0;
        b0 = b1, b1 = b2, b2 = tmp;
    }
    __builtin_amdgcn_s_barrier();  // the closing barrier of the leading group
}
// the MFMA waves' K loop: the schedule of k_loop_pingpong without any vector-memory instruction or vmcnt wait
template <bool AKM, bool BKM>
__device__ __forceinline__ void k_loop_consumer(char* smem, int nt, int wave, int lane, int wm, int wn, f32x16 (&acc)[2][2]) {
    using S = BlockShape<4>;
    char* b0 = smem;
    char* b1 = smem + S::STAGE_BYTES;
    char* b2 = smem + 2 * S::STAGE_BYTES;
    if (nt <= 0) return;
    __builtin_amdgcn_s_barrier();                 // tile 0 is visible to everyone
    if (wave >= 4) __builtin_amdgcn_s_barrier();  // the trailing group starts one barrier late
    for (int t = 0; t < nt; ++t) {
        const char* At = b0 + (wm >> 1) * TILE_BYTES;
        const char* Bt = b0 + S::NSUB * TILE_BYTES;
        uint32_t ka[2] = {0, 0}, kb[2] = {0, 0};
        if constexpr (AKM) {
            ka[0] = lds_addr(At) + km_lane_off((wm & 1) * 64, lane), ka[1] = lds_addr(At) + km_lane_off((wm & 1) * 64 + 32, lane);
        }
        if constexpr (BKM) {
            kb[0] = lds_addr(Bt) + km_lane_off(wn * 64, lane), kb[1] = lds_addr(Bt) + km_lane_off(wn * 64 + 32, lane);
        }
        bf16x8_t a[2][2], b[2][2];
        pp_load_half<AKM, BKM, 0>(At, Bt, ka, kb, wm & 1, wn, lane, a, b);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // 1
        pp_compute_half<AKM, BKM>(a, b, acc, []() {});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // 2
        pp_load_half<AKM, BKM, 1>(At, Bt, ka, kb, wm & 1, wn, lane, a, b);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // 3
        pp_compute_half<AKM, BKM>(a, b, acc, []() {});
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();  // 4
        char* tmp = b0;
        b0 = b1, b1 = b2, b2 = tmp;
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();  // the leading group waits for the trailing one
}

// ---- epilogue staging: one 32-row slab of every wave's accumulators -> LDS [64][SLAB_PITCH] fp32, so that the
// global reads/writes of the epilogue are row-contiguous (512 B per row) instead of 16-B pieces at a row stride.
constexpr int SLAB_PITCH = 132;  // floats; +4 keeps the 8-lane ds_write_b128 groups on distinct banks
__device__ __forceinline__ void slab_write(const f32x16 (&acc)[2][2], int i, float* slab, int wm, int wn, int lane) {
    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(slab + (wm * 32 + frow) * SLAB_PITCH + wn * 64 + j * 32 + 8 * q + 4 * fh) =
                make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
}
// piece k (0..7) of thread tid (of NT): slab row r, tile-local row m_local (for slab i), tile-local column c
template <int NT = 256>
__device__ __forceinline__ void slab_piece(int tid, int k, int i, int& r, int& m_local, int& c) {
    const int idx = tid + NT * k;
    r = idx >> 5;
    c = (idx & 31) * 4;
    m_local = (r >> 5) * 64 + i * 32 + (r & 31);
}

}  // namespace mmvid_core
