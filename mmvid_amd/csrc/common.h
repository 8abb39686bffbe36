// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the MMVID hot path.
// wave = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define MMVID_OK 0
#define MMVID_ERR_ARG 1
#define MMVID_ERR_HIP 2

// ---- error reporting (thread-local message, read through mmvid_last_error()) ----
extern "C" const char* mmvid_last_error();
void mmvid_set_error(const char* fmt, ...);

#define MMVID_REQUIRE(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            mmvid_set_error(__VA_ARGS__); \
            return MMVID_ERR_ARG;         \
        }                                 \
    } while (0)

#define MMVID_LAUNCH_CHECK(name)                                               \
    do {                                                                       \
        hipError_t e__ = hipGetLastError();                                    \
        if (e__ != hipSuccess) {                                               \
            mmvid_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return MMVID_ERR_HIP;                                              \
        }                                                                      \
    } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, same as torch): gfx950 converts in hardware (v_cvt_pk_bf16_f32) ----
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, f) & 0xffffu); }
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// The same reductions without the LDS crossbar: __shfl_xor is ds_bpermute_b32 (an LDS-pipe round trip of ~100+ cycles per step, six
// dependent steps); here four DPP steps inside a row of 16 lanes (quad xor 1, xor 2, row_half_mirror, row_mirror) and the gfx950
// row / half-wave swaps (v_permlane16_swap, v_permlane32_swap) -- all VALU, ~8 cycles a step.  Every lane ends with the total.  The
// summation ORDER differs from wave_sum's, so kernels whose results are pinned bit-for-bit keep the old form; the latency-bound decode
// kernels (csrc/decode.hip: a gemv spent 3.8 us of 11 in LayerNorm reductions, tools/decode_gemv_timeline.py) use this one.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_fast(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad of the 8-lane group
    v += dpp_mov<0x140>(v);  // row_mirror: the other 8-lane group of the row
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // rows 0|1 and 2|3
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const uint32_t w = __float_as_uint(v);
    const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);  // the two half-waves
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float wave_max_fast(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const uint32_t w = __float_as_uint(v);
    const auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

// Guard against reading MFMA results too early.  v_mfma_f32_32x32x16_bf16 needs 12 wait states before a VALU may
// read its destination; hipcc (ROCm 7.2) counts them along the fall-through path only, so a conditional branch
// right after an MFMA chain can reach the first reader after ~7 (observed: intermittent stale accumulators in the
// attention kernels under load).  Tying 16 nops to the accumulator makes every path safe.
__device__ __forceinline__ void mfma_settle(f32x16& acc) { asm volatile("s_nop 15" : "+v"(acc)); }
__device__ __forceinline__ void mfma_settle(f32x4& acc) { asm volatile("s_nop 15" : "+v"(acc)); }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// Run-time options: set by environment variable at first use or through mmvid_set_option().
//   graphs   MMVID_GRAPHS   1 = library-level hipGraph replay of the long launch sequences (tower passes, VQGAN plans); default 0 -- the
//                           training engine captures the whole step instead (mmvid_amd/engine.py)
// Rounds 1-5 carried up to 25 A/B knobs here (block shapes, K-loop schedules, epilogue forms, loader / storer / fat waves, split-K
// reductions, attention tail splits, packed softmax arithmetic, ...).  Each was decided by a whole-step measurement and the losing code
// was deleted at the end of round 5; the logs under profiles/ (r01_ab_*, r03_ab_*, r04_gemm_*_experiment.log, r05_ab_whole_step_*) are
// the record, and the comments next to the surviving code say what was measured against it.
enum { MMVID_OPT_GRAPHS = 0, MMVID_OPT_COUNT = 1 };
int mmvid_option(int which);  // errors.hip
