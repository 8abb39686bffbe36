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
// Out-of-range chunks (row >= rows, k >= K) read a 16-byte zero block instead: the DMA cannot be predicated per
// lane without leaving stale LDS bytes.
#pragma once
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

// ---- row-major operand -------------------------------------------------------------------------------------
__device__ __forceinline__ int rm_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Issue the 4 DMA pieces this wave owns of a row-major tile: rows [r0, r0+128) x k [k0, k0+64) of base[rows][ld].
__device__ __forceinline__ void stage_rowmajor(const bf16_t* base, long ld, int rows, int K, int r0, int k0, char* tile,
                                               int wave, int lane) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = wave * 4 + jj;
        const int row = j * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int gr = r0 + row, k = k0 + c * 8;
        const void* src = (gr < rows && k < K) ? (const void*)(base + (long)gr * ld + k) : (const void*)g_zero16;
        glds16(src, tile + j * 1024);
    }
}
__device__ __forceinline__ bf16x8_t frag_rowmajor(const char* tile, int row, int s, int fh) {
    return *reinterpret_cast<const bf16x8_t*>(tile + rm_off(row, 2 * s + fh));
}

// ---- k-major operand ---------------------------------------------------------------------------------------
// tile k [k0, k0+64) x rows [r0, r0+128) of base[K][ld]
__device__ __forceinline__ void stage_kmajor(const bf16_t* base, long ld, int rows, int K, int r0, int k0, char* tile,
                                             int wave, int lane) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = wave * 4 + jj;
        const int kr = j * 4 + (lane >> 4);
        const int c = (lane & 15) ^ ((kr & 3) << 2);  // logical 16-B chunk (8 rows) that lands in slot lane&15
        const int k = k0 + kr, r = r0 + c * 8;
        const void* src = (k < K && r < rows) ? (const void*)(base + (long)k * ld + r) : (const void*)g_zero16;
        glds16(src, tile + j * 1024);
    }
}
// 8 consecutive k (k-step s of 16, half fh) for row `rowbase + (lane&31)`: two transpose reads 4 k-rows apart.
__device__ __forceinline__ bf16x8_t frag_kmajor(const char* tile, int rowbase, int s, int lane) {
    const int G = lane >> 4, si = lane & 15;
    const int kk = 16 * s + 8 * (G >> 1) + (si >> 2);
    const int mbyte = (rowbase + 16 * (G & 1) + 4 * (si & 3)) * 2;
    const char* p = tile + kk * 256 + (mbyte ^ ((kk & 3) << 6));
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)p);
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_t*)(p + 1024));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool KM>
__device__ __forceinline__ void stage(const bf16_t* base, long ld, int rows, int K, int r0, int k0, char* tile, int wave,
                                      int lane) {
    if constexpr (KM)
        stage_kmajor(base, ld, rows, K, r0, k0, tile, wave, lane);
    else
        stage_rowmajor(base, ld, rows, K, r0, k0, tile, wave, lane);
}
template <bool KM>
__device__ __forceinline__ bf16x8_t frag(const char* tile, int rowbase, int s, int lane) {
    if constexpr (KM)
        return frag_kmajor(tile, rowbase, s, lane);
    else
        return frag_rowmajor(tile, rowbase + (lane & 31), s, lane >> 5);
}

// One 64-deep K tile: acc[i][j] (+)= B-frag(j) x A-frag(i)  (operands swapped: a lane gets 4 consecutive n).
template <bool AKM, bool BKM>
__device__ __forceinline__ void mma_tile(const char* At, const char* Bt, f32x16 (&acc)[2][2], int wm, int wn, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        bf16x8_t af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = frag<AKM>(At, wm * 64 + i * 32, s, lane);
            bfr[i] = frag<BKM>(Bt, wn * 64 + i * 32, s, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
}

// =====================================================================================================
// 4-stage variant with a 32-deep K tile ("P4"): 4 x (8 KiB A + 8 KiB B) = 64 KiB, three tiles of LDS-DMA in flight
// across the per-tile barrier (raw s_barrier + COUNTED s_waitcnt vmcnt, guide T3/T4) instead of one tile and a
// vmcnt(0) drain.  Row-major tile: [128 rows][32 k] (64-B rows), chunk ^= (row>>2)&3; k-major tile: [32 k][128 rows].
constexpr int BK4 = 32;
constexpr int TILE4_BYTES = BM * BK4 * 2;  // 8 KiB
constexpr int STAGE4_BYTES = 2 * TILE4_BYTES;

__device__ __forceinline__ int rm4_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <bool KM>
__device__ __forceinline__ void stage4(const bf16_t* base, long ld, int rows, int K, int r0, int k0, char* tile, int wave,
                                       int lane) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int j = wave * 2 + jj;  // 8 DMA pieces of 1 KiB per tile, 2 per wave
        const void* src;
        if constexpr (KM) {
            const int kr = j * 4 + (lane >> 4);
            const int c = (lane & 15) ^ ((kr & 3) << 2);
            const int k = k0 + kr, r = r0 + c * 8;
            src = (k < K && r < rows) ? (const void*)(base + (long)k * ld + r) : (const void*)g_zero16;
        } else {
            const int row = j * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            const int gr = r0 + row, k = k0 + c * 8;
            src = (gr < rows && k < K) ? (const void*)(base + (long)gr * ld + k) : (const void*)g_zero16;
        }
        glds16(src, tile + j * 1024);
    }
}
template <bool KM>
__device__ __forceinline__ bf16x8_t frag4(const char* tile, int rowbase, int s, int lane) {
    if constexpr (KM) {
        return frag_kmajor(tile, rowbase, s, lane);  // same [k][128 rows] image, s in {0,1}
    } else {
        const int row = rowbase + (lane & 31);
        return *reinterpret_cast<const bf16x8_t*>(tile + rm4_off(row, 2 * s + (lane >> 5)));
    }
}
template <bool AKM, bool BKM>
__device__ __forceinline__ void mma_tile4(const char* At, const char* Bt, f32x16 (&acc)[2][2], int wm, int wn, int lane) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        bf16x8_t af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = frag4<AKM>(At, wm * 64 + i * 32, s, lane);
            bfr[i] = frag4<BKM>(Bt, wn * 64 + i * 32, s, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
}
// wait until at most `tiles_in_flight` later tiles (4 DMA pieces each per wave) are outstanding, then barrier
__device__ __forceinline__ void wait_tiles_and_barrier(int tiles_in_flight) {
    if (tiles_in_flight >= 2)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (tiles_in_flight == 1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
// piece k (0..7) of thread tid: slab row r, tile-local row m_local (for slab i), tile-local column c
__device__ __forceinline__ void slab_piece(int tid, int k, int i, int& r, int& m_local, int& c) {
    const int idx = tid + 256 * k;
    r = idx >> 5;
    c = (idx & 31) * 4;
    m_local = (r >> 5) * 64 + i * 32 + (r & 31);
}

}  // namespace mmvid_core
