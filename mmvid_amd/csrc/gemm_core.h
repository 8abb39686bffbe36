// Shared tile machinery of the bf16 MFMA kernels (gemm.hip, conv.hip): 128x128x64 block tile, LDS tiles of
// 128 rows x 64 bf16 with 16-B chunks XOR-swizzled by ((row>>1)&7), 2x2 waves each 2x2 v_mfma_f32_32x32x16_bf16.
#pragma once
#include "common.h"

namespace mmvid_core {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile


__device__ __forceinline__ int lds_off(int row, int chunk) {  // byte offset inside a tile
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// Row-major operand: tile rows [r0, r0+128) x k [k0, k0+64).  Thread t owns chunk c = t&7 of rows
// (t>>3) + 32*i, i = 0..3.
struct RowMajorStage {
    uint4 v[4];
    __device__ __forceinline__ void load(const bf16_t* base, long ld, int rows, int K, int r0, int k0, int tid) {
        const int c = tid & 7;
        const int k = k0 + c * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + (tid >> 3) + 32 * i;
            if (r < rows && k < K)
                v[i] = *reinterpret_cast<const uint4*>(base + (long)r * ld + k);
            else
                v[i] = make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(char* tile, int tid) const {
        const int c = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 3) + 32 * i;
            *reinterpret_cast<uint4*>(tile + lds_off(r, c)) = v[i];
        }
    }
};

// k-major operand stored [K][rows]: tile k [k0,k0+64) x rows [r0,r0+128).  Thread t owns the 4(k) x 8(row)
// block  half = t&1 (k sub-block), kc = (t>>1)&7 (k chunk), nb = t>>4 (row block of 8).
struct KMajorStage {
    uint4 v[4];
    __device__ __forceinline__ void load(const bf16_t* base, long ld, int rows, int K, int r0, int k0, int tid) {
        const int half = tid & 1, kc = (tid >> 1) & 7, nb = tid >> 4;
        const int r = r0 + nb * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + kc * 8 + half * 4 + q;
            if (k < K && r < rows)
                v[q] = *reinterpret_cast<const uint4*>(base + (long)k * ld + r);
            else
                v[q] = make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void store(char* tile, int tid) const {
        const int half = tid & 1, kc = (tid >> 1) & 7, nb = tid >> 4;
        const uint32_t w[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w},
                                  {v[1].x, v[1].y, v[1].z, v[1].w},
                                  {v[2].x, v[2].y, v[2].z, v[2].w},
                                  {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
        for (int p = 0; p < 4; ++p) {  // row pair (2p, 2p+1) of the 8-row block
            uint2 ev, od;
            ev.x = (w[0][p] & 0xffffu) | (w[1][p] << 16);
            ev.y = (w[2][p] & 0xffffu) | (w[3][p] << 16);
            od.x = (w[0][p] >> 16) | (w[1][p] & 0xffff0000u);
            od.y = (w[2][p] >> 16) | (w[3][p] & 0xffff0000u);
            const int re = nb * 8 + 2 * p;
            *reinterpret_cast<uint2*>(tile + lds_off(re, kc) + half * 8) = ev;
            *reinterpret_cast<uint2*>(tile + lds_off(re + 1, kc) + half * 8) = od;
        }
    }
};

template <bool KM>
struct StageSel {
    using type = RowMajorStage;
};
template <>
struct StageSel<true> {
    using type = KMajorStage;
};


// One 64-deep K tile: acc[i][j] (+)= B-frag(j) x A-frag(i)  (operands swapped: lane gets 4 consecutive n).
__device__ __forceinline__ void mma_tile(const char* At, const char* Bt, f32x16 (&acc)[2][2], int wm, int wn, int lane) {
    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int chunk = 2 * s + fh;
        bf16x8_t af[2], bfr[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            af[i] = *reinterpret_cast<const bf16x8_t*>(At + lds_off(wm * 64 + i * 32 + frow, chunk));
            bfr[i] = *reinterpret_cast<const bf16x8_t*>(Bt + lds_off(wn * 64 + i * 32 + frow, chunk));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
}

}  // namespace mmvid_core
