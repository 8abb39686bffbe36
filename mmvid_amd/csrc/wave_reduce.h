// Cross-lane reductions for the latency-bound decode kernels (csrc/decode.hip, csrc/decode_persistent.hip).
#pragma once
#include "common.h"

// ---- reduce-scatter of a short list of per-lane partial sums over the 64 lanes of a wave.  A gemv wave ends with 2 * NB partial dot
// products per lane (2 output features x NB rows); summing each with a full butterfly costs 6 cross-lane steps per value.  Here the
// list is HALVED per lane bit instead: at bit 5 (v_permlane32_swap) a lane hands one half of its list to its partner and keeps the
// sum of the other half, then bit 4 (v_permlane16_swap), bit 3 (row_mirror), bit 2 (row_half_mirror) -- N - 1 exchanges for N values
// -- and the remaining lane bits are a butterfly on the ONE value left.  Value j of the list ends, complete, in lanes
// [j << (6 - LOGN), (j + 1) << (6 - LOGN)).
template <int BIT>
__device__ __forceinline__ float xchg_add(float x, float y, int lane) {  // lanes with BIT clear: x over the pair; set: y over the pair
    if constexpr (BIT == 5) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (BIT == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        const bool up = (lane >> BIT) & 1;
        const float keep = up ? y : x, send = up ? x : y;
        return keep + dpp_mov<BIT == 3 ? 0x140 : (BIT == 2 ? 0x141 : (BIT == 1 ? 0x4E : 0xB1))>(send);
    }
}
template <int BIT>
__device__ __forceinline__ float bfly_add(float v) {  // all-reduce step over lane bit BIT
    if constexpr (BIT == 5) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else if constexpr (BIT == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    } else {
        return v + dpp_mov<BIT == 3 ? 0x140 : (BIT == 2 ? 0x141 : (BIT == 1 ? 0x4E : 0xB1))>(v);
    }
}
template <int LOGN>
__device__ __forceinline__ float reduce_scatter(float (&v)[1 << LOGN], int lane) {
    if constexpr (LOGN >= 1) {
#pragma unroll
        for (int i = 0; i < (1 << LOGN) / 2; ++i) v[i] = xchg_add<5>(v[i], v[i + (1 << LOGN) / 2], lane);
    }
    if constexpr (LOGN >= 2) {
#pragma unroll
        for (int i = 0; i < (1 << LOGN) / 4; ++i) v[i] = xchg_add<4>(v[i], v[i + (1 << LOGN) / 4], lane);
    }
    if constexpr (LOGN >= 3) {
#pragma unroll
        for (int i = 0; i < (1 << LOGN) / 8; ++i) v[i] = xchg_add<3>(v[i], v[i + (1 << LOGN) / 8], lane);
    }
    if constexpr (LOGN >= 4) v[0] = xchg_add<2>(v[0], v[1], lane);
    float r = v[0];
    if constexpr (LOGN < 1) r = bfly_add<5>(r);
    if constexpr (LOGN < 2) r = bfly_add<4>(r);
    if constexpr (LOGN < 3) r = bfly_add<3>(r);
    if constexpr (LOGN < 4) r = bfly_add<2>(r);
    r = bfly_add<1>(r);
    return bfly_add<0>(r);
}

// sum over the 8 lanes of a group (lane bits 0-2), result in all of them
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    return v + dpp_mov<0x141>(v);
}
// sum over lane bits 3, 4, 5 (the lanes that share bits 0-2), result in all of them
__device__ __forceinline__ float over_groups_sum(float v) {
    v += dpp_mov<0x128>(v);  // row_ror:8
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const uint32_t w = __float_as_uint(v);
    const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
