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
// Tuning knobs (A/B testing): set by environment variable at first use or at run time through mmvid_set_option().
//   gemm_tile      MMVID_GEMM_TILE      0 = block shape by grid fill (default), 128 / 256 = forced
//   tower_streams  MMVID_TOWER_STREAMS  1 = tower backward on one stream (default), 2 = weight-gradient side stream
//   graphs         MMVID_GRAPHS         1 = library-level hipGraph replay of the long launch sequences (default 0)
//   gemm_sched     MMVID_GEMM_SCHED     K loop of the 256x128 block: 0 = one barrier per tile; 1 = two-group ping-pong with
//                                       the DMA requests in the load parts (-2 % step time); 2 = ping-pong with the DMA
//                                       requests inside the MFMA clusters (default; a further -0.5 %, tools/ab_graph.py)
//   fuse_colsum    MMVID_FUSE_COLSUM    1 (default) = c_fc's bias gradient from the epilogue of the GEMM that produces d_pre, the in-projection's from the
//                                       registers of the attention backward (unrounded fp32 sums); 0 = both as column sums of the bf16 tensors the
//                                       weight gradients are computed from (mmvid_colsum_bf16)
//   ln_bwd_blocks  MMVID_LN_BWD_BLOCKS  grid cap of the LayerNorm backward (default 512: its dw/db atomics scale with the
//                                       grid -- measured on the whole step 2048: +0.6 ms, 1024: +0.11 ms, 256: +0.25 ms)
//   strip_sched    MMVID_STRIP_SCHED    strip convolution: 0 = LDS-DMA requests right after the tile barrier, 1 = spread over the
//                                       first three k-steps of the tile, 2 = two-group ping-pong (default; whole-step A/B with
//                                       tools/ab_graph.py: 18.59 / 18.53 / 18.49 ms for 0 / 1 / 2)
//   gemm_wshape    MMVID_GEMM_WSHAPE    forward / dX GEMMs on the 4-wave 256x128 shape with two blocks per CU (gemm.hip): 0 = off
//                                       (default: measured 5-25 % slower, profiles/r02_gemm_anatomy.log), 1 = when the grid has
//                                       >= 200 tiles, 2 = always
//   attn_occ       MMVID_ATTN_OCC       attention kernels compiled for one more block per CU than their register use gives (110 / 135 / 211
//                                       registers = 4 / 3 / 2 blocks per CU by default; every kernel is launched as its <2> instance):
//                                       bit 0 forward (<5>), bit 1 dQ (<4>), bit 2 dK/dV (<3>); registers capped, a few spilled --
//                                       measured 1.1-1.6x SLOWER (profiles/r03_attention_microbench_occupancy_variants.log), default 0
//   gemm_persist   MMVID_GEMM_PERSIST   1 (default) = GEMMs with more tiles than CUs run 256 persistent blocks that walk the tiles
//                                       (whole-step A/B: 17.82 -> 17.77 ms; the block turnover is paid once per launch)
//   gemm_debug     MMVID_GEMM_DEBUG     measurement only (tools/bench_gemm.py anatomy): 1 = the GEMM epilogue skips its global
//                                       stores, 2 = the K loop is skipped (results are wrong in both)
//   gemm_epi       MMVID_GEMM_EPI       epilogue of the 256x128 GEMM blocks: 0 = accumulators staged through an LDS slab (row-contiguous
//                                       512-B stores), 1 = stored straight from the MFMA registers through buffer descriptors
//                                       (16 B per lane, 8 stores per 128-B line) with the NEXT tile's first two K tiles requested
//                                       before the stores, so a persistent block never waits for its own writes; 2 = as 1, and
//                                       persistent bf16-output GEMMs defer a tile's stores into the next tile's K loop
//   dh_bf16        MMVID_DH_BF16        1 (default) = the tower backward keeps d(LayerNorm output) in bf16 between the dX GEMM and the
//                                       LayerNorm backward (as every other GEMM operand gradient already is); 0 = fp32
//   gemm_loader    MMVID_GEMM_LOADER    1 (default) = 256x128 GEMM blocks with a register-direct epilogue run 8 MFMA waves + 1 LOADER wave
//                                       that issues every LDS-DMA request (the MFMA waves issue no vector-memory instruction in
//                                       the K loop); 0 = every wave requests its own share between its MFMAs (round 2)
//   gemm_groupn    MMVID_GEMM_GROUPN    1 (default) = persistent GEMM blocks walk the output tiles in column GROUPS sized so that one XCD round's
//                                       B tiles + A panels fit its L2: fabric-side reads of the c_fc GEMM 179 -> 113 MB, qkv 109 -> 83 MB
//                                       (PMC, profiles/r03_pmc_fetch_column_groups.txt); whole step 16.57 -> 16.52 ms
//   gemm_loader    MMVID_GEMM_LOADER    1 (default) = 256x128 GEMM blocks with a register-direct epilogue run 8 MFMA waves + 1 LOADER wave
//                                       that issues every LDS-DMA request (the MFMA waves issue no vector-memory instruction in
//                                       the K loop); 0 = every wave requests its own share between its MFMAs (round 2)
//   attn_res       MMVID_ATTN_RES       0 (default) = streaming attention kernels; 1 = "resident" forms (one 8-wave block per (batch, head), K/V
//                                       or Q/dO staged into LDS once, L <= 608; bit-identical; measured 16 % SLOWER: 2 waves per SIMD cannot
//                                       hide a wave's ~1.7-us per-tile dependency chain); 2 = the forward kernel with 16 waves (equal to 0)
//   gemm_fused_reduce MMVID_GEMM_FUSED_REDUCE 0 (default) = split-K slabs of the weight-gradient GEMM added by splitk_reduce_kernel; 1 = by the last
//                                       block of each output tile inside the GEMM (bit-identical; measured +5 ms per step: the device-scope
//                                       release writes back the XCD's whole L2)
//   dw_grouped     MMVID_DW_GROUPED     1 (default) = the tower backward keeps every layer's dY tensors in the saved arena and computes the weight
//                                       gradients of all layers and all four Linear shapes in ONE launch after the layer loop, and
//                                       the LayerNorm parameter-gradient reductions in one launch (tower.hip; captured step 16.36 ->
//                                       15.4 ms); 0 = per-layer split-K launches + reduces (rounds 1-2); measurement only: 2 = one
//                                       launch per shape (+0.14 ms), 3 = grouped weights, per-LayerNorm reductions (+0.09 ms).
//                                       Must not change between a forward and its backward (the arena's slice size follows it)
//   gemm_loaders   MMVID_GEMM_LOADERS   loader waves of the loader-wave GEMM block: 4 (default), 8 = sixteen-wave blocks (bf16-output form only)
//   dw_order       MMVID_DW_ORDER       1 (default) = the grouped weight-gradient launch walks the tiles of an output whose X operand is the wider
//                                       one (c_proj: 768 x 3072) column-major, so that operand is streamed once; 0 = always row-major (round 3)
//   gn_fused       MMVID_GN_FUSED       1 (default) = GroupNorm of maps of <= 256 pixels without fused statistics runs as ONE launch per call (statistics,
//                                       finalisation, apply: one block per image; bit-identical to the three launches); 0 = three launches
//   attn_tail      MMVID_ATTN_TAIL      1 (default) = the attention backward cuts the blocks of its last, partly filled round into parts over disjoint
//                                       query ranges (a workspace + a small combine launch; csrc/attn.hip TailSplit); 0 = whole blocks only
//   gemm_fat       MMVID_GEMM_FAT       1 = the loader-wave GEMM block runs FOUR MFMA waves of 128 x 64 (0.75 KiB of LDS fragment reads per MFMA
//                                       instead of 1: the K loop is bound by LDS bandwidth, csrc/gemm_core.h k_loop_consumer_fat); 0 (default) = eight of 64 x 64
//   ln_fast        MMVID_LN_FAST        1 (default) = LayerNorm backward of E = 512 / 768 rows on the software-pipelined kernel (next row's operands in
//                                       flight while a row is reduced; bit-identical); 0 = the generic kernel (rounds 1-4)
//   attn_pk        MMVID_ATTN_PK        attention forward / dQ softmax arithmetic: 1 = packed fp32 fma / add (v_pk_*), 0 = single-lane instructions
enum { MMVID_OPT_GEMM_TILE = 0, MMVID_OPT_TOWER_STREAMS = 1, MMVID_OPT_GRAPHS = 2, MMVID_OPT_LN_BWD_BLOCKS = 3, MMVID_OPT_GEMM_SCHED = 4, MMVID_OPT_FUSE_COLSUM = 5, MMVID_OPT_STRIP_SCHED = 6, MMVID_OPT_GEMM_DEBUG = 7, MMVID_OPT_GEMM_WSHAPE = 8, MMVID_OPT_ATTN_OCC = 9, MMVID_OPT_GEMM_PERSIST = 10, MMVID_OPT_GEMM_EPI = 11, MMVID_OPT_DH_BF16 = 12, MMVID_OPT_GEMM_LOADER = 13, MMVID_OPT_GEMM_GROUPN = 14, MMVID_OPT_ATTN_RES = 15, MMVID_OPT_GEMM_FUSED_REDUCE = 16, MMVID_OPT_DW_GROUPED = 17, MMVID_OPT_GEMM_LOADERS = 18, MMVID_OPT_DW_ORDER = 19, MMVID_OPT_GN_FUSED = 20, MMVID_OPT_ATTN_TAIL = 21, MMVID_OPT_GEMM_FAT = 22, MMVID_OPT_LN_FAST = 23, MMVID_OPT_ATTN_PK = 24, MMVID_OPT_COUNT = 25 };
int mmvid_option(int which);  // errors.hip
static inline int mmvid_tile_override() { return mmvid_option(MMVID_OPT_GEMM_TILE); }
