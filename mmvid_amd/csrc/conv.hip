// VQGAN convolutions as implicit GEMM on NHWC bf16, for gfx950.
//   M = N*Hout*Wout output pixels, N_gemm = Cout, K = taps*Cin with k = (ky*3+kx)*Cin + ci.
// The A tile is gathered straight from the NHWC activation by LDS-DMA (each 16-B chunk = 8 input channels of
// one tap of one pixel: contiguous; padding taps read a zero block); the weight [Cout][taps][Cin] is the
// row-major B operand.  Same block shapes (128x128 2-stage / 256x128 3-stage ring), swizzled LDS and 32x32x16 MFMA
// core as gemm.hip.
//   mode 0: 3x3 s1 p1                     (model.py:102-115 ResnetBlock convs, conv_in/out)
//   mode 1: 3x3 s2, zero pad right/bottom (model.py:77-81 Downsample)
//   mode 2: nearest x2 upsample + 3x3 p1  (model.py:56-62 Upsample; the upsampled image is never materialised)
//   mode 3: 1x1                           (nin_shortcut, AttnBlock q/k/v/proj_out, quant_conv, post_quant_conv)
// Epilogue: + bias, + residual (ResnetBlock/AttnBlock skip), optional (clamp(x,-1,1)+1)/2 (vae.py:55).
// Roofline: bf16 MFMA; algorithmic FLOPs = 2 * M * Cout * taps * Cin.
#include "../../include/mmvid_hip.h"
#include "gemm_core.h"
#include "wave_reduce.h"
#include "prof.h"

namespace {
using namespace mmvid_core;

struct ConvParams {
    const bf16_t* x;
    const bf16_t* w;
    int N, Hin, Win, Cin, cin_log2, Hout, Wout, Cout, mode, taps;
    long M;
    int K;         // reduction length the K loop runs over = terms * K1
    int K1;        // taps * Cin
    long x_plane;  // split operator (terms == 3): elements between the hi and lo planes of x; term 1 reads the lo plane
    const float* bias;
    const bf16_t* res_bf16;
    const float* res_f32;
    int clamp01;
    bf16_t* out_bf16;
    float* out_f32;
    float* gn_partial;  // GroupNorm partial sums of the OUTPUT: [N][Hout*Wout/128][32 groups][sum, sumsq], or null
    int splitk;         // > 1: blockIdx.z owns a K range; raw accumulators go to `partial`, conv_splitk_reduce_kernel finishes
    float* partial;     // [splitk][M][Cout] fp32
};

// A-operand gather by LDS-DMA: this lane owns, in each of its wave's 4 DMA pieces, LDS slot (lane&7) of tile row
// (wave*4+jj)*8 + (lane>>3); the logical k chunk that belongs in that slot follows the row swizzle of gemm_core.h.
//
// FAST form (Cin % 64 == 0, modes 0/1/3 -- every encoder layer but conv_in): a 64-deep K tile lies inside ONE tap,
// so the tap is wave-uniform.  Per piece the lane keeps the byte offset of its centre pixel (+ channel chunk) and a
// 9-bit mask of the taps that fall inside the image; a tile is then: move the buffer descriptor's base by the tap
// displacement (SALU), and per piece one mask test -> offset or the out-of-range marker (zero fill by the range
// check) -> buffer_load ... lds.  No per-tile address arithmetic in VGPRs.
template <bool FAST>
struct ConvAStage;

template <>
struct ConvAStage<true> {
    uint32_t voff[4], mask[4];
    __device__ __forceinline__ void init(const ConvParams& p, long m0, int wave, int lane) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int row = (wave * 4 + jj) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            const long m = m0 + row;
            voff[jj] = 0, mask[jj] = 0;
            if (m < p.M) {
                const long hw = (long)p.Hout * p.Wout;
                const long n = m / hw;
                const int rem = (int)(m - n * hw);
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                const int cy = p.mode == 1 ? 2 * oy : oy, cx = p.mode == 1 ? 2 * ox : ox;
                voff[jj] = (uint32_t)(((((n * p.Hin + cy) * p.Win + cx) << p.cin_log2) + chunk * 8) * 2);
                uint32_t rb, cb;  // valid ky / kx bits
                if (p.mode == 0) {
                    rb = (oy >= 1 ? 1u : 0u) | 2u | (oy + 1 < p.Hin ? 4u : 0u);
                    cb = (ox >= 1 ? 1u : 0u) | 2u | (ox + 1 < p.Win ? 4u : 0u);
                } else if (p.mode == 1) {
                    rb = 3u | (2 * oy + 2 < p.Hin ? 4u : 0u);
                    cb = 3u | (2 * ox + 2 < p.Win ? 4u : 0u);
                } else {
                    rb = 1u, cb = 1u;
                }
                mask[jj] = ((rb & 1u) ? cb : 0u) | ((rb & 2u) ? cb << 3 : 0u) | ((rb & 4u) ? cb << 6 : 0u);
            }
        }
    }
    __device__ __forceinline__ void issue(const ConvParams& p, int k0, char* tile, int wave) const {
        // split operator: K = [x_hi.w_hi | x_lo.w_hi | x_hi.w_lo]; K1 % 64 == 0 here, so a K tile lies inside one term
        const int term = (k0 >= p.K1 ? 1 : 0) + (k0 >= 2 * p.K1 ? 1 : 0);  // wave-uniform; 0 for the plain operator
        k0 -= term * p.K1;
        const int tap = k0 >> p.cin_log2, ci0 = k0 & (p.Cin - 1);
        const int ky = tap / 3, kx = tap - ky * 3;
        const int disp = p.mode == 0 ? (ky - 1) * p.Win + (kx - 1) : (p.mode == 1 ? ky * p.Win + kx : 0);
        const long byte_disp = (((long)disp << p.cin_log2) + ci0 + (term == 1 ? p.x_plane : 0)) * 2;
        const rsrc_t rsrc = make_rsrc(reinterpret_cast<const char*>(p.x) + byte_disp, 0x7fffffffu);
        const uint32_t bit = 1u << tap;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            blds16(rsrc, (mask[jj] & bit) ? voff[jj] : OOB, 0, tile + (wave * 4 + jj) * 1024);
    }
};

template <>
struct ConvAStage<false> {
    int oy[4], ox[4], chunk[4];
    long nbase[4];  // n*Hin*Win, or -1 when the output pixel is out of range
    __device__ __forceinline__ void init(const ConvParams& p, long m0, int wave, int lane) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int row = (wave * 4 + jj) * 8 + (lane >> 3);
            chunk[jj] = (lane & 7) ^ ((row >> 1) & 7);
            const long m = m0 + row;
            if (m < p.M) {
                const long hw = (long)p.Hout * p.Wout;
                const long n = m / hw;
                const int rem = (int)(m - n * hw);
                oy[jj] = rem / p.Wout;
                ox[jj] = rem - oy[jj] * p.Wout;
                nbase[jj] = n * p.Hin * p.Win;
            } else {
                nbase[jj] = -1, oy[jj] = ox[jj] = 0;
            }
        }
    }
    __device__ __forceinline__ void issue(const ConvParams& p, int k0, char* tile, int wave) const {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            int k = k0 + chunk[jj] * 8;
            bool ok = k < p.K && nbase[jj] >= 0;
            const int term = (k >= p.K1 ? 1 : 0) + (k >= 2 * p.K1 ? 1 : 0);  // split operator: K1 % 8 == 0, a chunk lies in one term
            k -= term * p.K1;
            const bf16_t* xb = p.x + (term == 1 ? p.x_plane : 0);
            const int tap = k >> p.cin_log2;
            const int ci = k & (p.Cin - 1);
            const int ky = (p.mode == 3) ? 0 : tap / 3;
            const int kx = (p.mode == 3) ? 0 : tap - ky * 3;
            int iy, ix;
            if (p.mode == 0) {
                iy = oy[jj] + ky - 1, ix = ox[jj] + kx - 1;
                ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            } else if (p.mode == 1) {
                iy = 2 * oy[jj] + ky, ix = 2 * ox[jj] + kx;
                ok = ok && iy < p.Hin && ix < p.Win;
            } else if (p.mode == 2) {
                const int uy = oy[jj] + ky - 1, ux = ox[jj] + kx - 1;
                ok = ok && uy >= 0 && uy < 2 * p.Hin && ux >= 0 && ux < 2 * p.Win;
                iy = uy >> 1, ix = ux >> 1;
            } else {
                iy = oy[jj], ix = ox[jj];
            }
            const void* src = ok ? (const void*)(xb + ((nbase[jj] + (long)iy * p.Win + ix) << p.cin_log2) + ci)
                                 : (const void*)g_zero16;
            glds16(src, tile + (wave * 4 + jj) * 1024);
        }
    }
};

template <int WM, bool FAST, int PP>
__global__ __launch_bounds__(WM * 128, WM == 2 ? 2 : 1) void conv_igemm_kernel(ConvParams p) {
    using S = BlockShape<WM>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int wg = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int bn0 = (wg % gridDim.x) * BN;
    const long bm0 = (long)(wg / gridDim.x) * S::ROWS;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    ConvAStage<FAST> sa;  // 4 pieces (32 output pixels) per wave: 4 waves cover 128 rows, 8 waves 256
    sa.init(p, bm0, wave, lane);
    OperandStage<false, 1, S::PPW> sb;  // weights [Cout][K] row-major
    sb.init(p.w, p.K, p.Cout, p.K, bn0, wave, lane);
    const int ktiles = (p.K + BK - 1) / BK, per = (ktiles + p.splitk - 1) / p.splitk;
    const int kt0 = (int)blockIdx.z * per;
    int nt = (kt0 + per > ktiles ? ktiles : kt0 + per) - kt0;  // this split's K tiles (all of them when splitk == 1)
    if (nt < 0) nt = 0;                                            // a trailing split may be empty: it writes zeros
    auto stage_tile = [&](int t, char* buf) {
        sa.issue(p, (kt0 + t) * BK, buf, wave);
        sb.issue((kt0 + t) * BK, p.K, buf + S::NSUB * TILE_BYTES, wave, lane);
    };
    auto compute_tile = [&](const char* buf) {
        mma_tile<false, false>(buf + (wm >> 1) * TILE_BYTES, buf + S::NSUB * TILE_BYTES, acc, wm & 1, wn, lane);
    };
    if constexpr (S::NSTAGE == 2) {
        if (nt > 0) stage_tile(0, smem);
        for (int t = 0; t < nt; ++t) {
            char* cur = smem + (t & 1) * S::STAGE_BYTES;
            char* nxt = smem + ((t + 1) & 1) * S::STAGE_BYTES;
            dma_publish_barrier();
            if (t + 1 < nt) stage_tile(t + 1, nxt);
            compute_tile(cur);
        }
    } else if constexpr (PP != 0) {  // two-group ping-pong (gemm_core.h)
        k_loop_pingpong<false, false>(
            smem, nt, wave, lane, wm, wn, acc, [&](int t, char* buf) { sa.issue(p, (kt0 + t) * BK, buf, wave); },
            [&](int t, char* buf) { sb.issue((kt0 + t) * BK, p.K, buf + S::NSUB * TILE_BYTES, wave, lane); });
    } else {  // 3-stage ring with counted vmcnt (see gemm.hip)
        char* b0 = smem;
        char* b1 = smem + S::STAGE_BYTES;
        char* b2 = smem + 2 * S::STAGE_BYTES;
        if (nt > 0) stage_tile(0, b0);
        if (nt > 1) stage_tile(1, b1);
        for (int t = 0; t < nt; ++t) {
            if (t + 1 < nt)
                wait_dma_and_barrier<S::DMA_PER_TILE>();
            else
                wait_dma_and_barrier<0>();
            if (t + 2 < nt) stage_tile(t + 2, b2);
            compute_tile(b0);
            char* tmp = b0;
            b0 = b1, b1 = b2, b2 = tmp;
        }
    }
    // epilogue through LDS (row-contiguous global traffic; see gemm.hip).  Round 6: a thread finishes EIGHT channels of a row (two float4
    // of the slab): 16-byte bf16 stores / residual loads, half the vector-memory instructions (as conv_strip.hip)
    mfma_settle(acc[0][0]), mfma_settle(acc[0][1]), mfma_settle(acc[1][0]), mfma_settle(acc[1][1]);
    float* slab = reinterpret_cast<float*>(smem);
    const int n = bn0 + 8 * (tid & 15);
    const bool n_ok = n < p.Cout;  // (Cout % 8 == 0)
    float4 bias4[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
        bias4[e] = (p.bias && n_ok) ? *reinterpret_cast<const float4*>(p.bias + n + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int HALVES = WM / 2;  // 128-row blocks of the tile = GroupNorm partial blocks
    float gs[HALVES][8], gq[HALVES][8];
#pragma unroll
    for (int hf = 0; hf < HALVES; ++hf)
#pragma unroll
        for (int e = 0; e < 8; ++e) gs[hf][e] = gq[hf][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
        slab_write(acc, i, slab, wm, wn, lane);
        __syncthreads();
        float4 v4[4][2], rf[4][2];
        uint4 rb[4];
        long mrow[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // piece k: slab row r = (tid + THREADS k) / 16, channels [8 (tid & 15), + 8)
            const int r = (tid + S::THREADS * k) >> 4;
            mrow[k] = bm0 + (r >> 5) * 64 + i * 32 + (r & 31);
            const bool ok = n_ok && mrow[k] < p.M;
            const long o = mrow[k] * p.Cout + n;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v4[k][e] = *reinterpret_cast<const float4*>(slab + r * SLAB_PITCH + 8 * (tid & 15) + 4 * e);
                rf[k][e] = (p.res_f32 && ok) ? *reinterpret_cast<const float4*>(p.res_f32 + o + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            rb[k] = (p.res_bf16 && ok) ? *reinterpret_cast<const uint4*>(p.res_bf16 + o) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!n_ok || mrow[k] >= p.M) continue;
            if (p.partial) {  // split-K: the raw partial sums; bias, residual and stores happen in the fixed-order reduce
                float* dst = p.partial + ((long)blockIdx.z * p.M + mrow[k]) * p.Cout + n;
                *reinterpret_cast<float4*>(dst) = v4[k][0];
                *reinterpret_cast<float4*>(dst + 4) = v4[k][1];
                continue;
            }
            const uint32_t rbw[4] = {rb[k].x, rb[k].y, rb[k].z, rb[k].w};
            float v[8];
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v[4 * e] = v4[k][e].x + bias4[e].x + rf[k][e].x + bf_lo(rbw[2 * e]);
                v[4 * e + 1] = v4[k][e].y + bias4[e].y + rf[k][e].y + bf_hi(rbw[2 * e]);
                v[4 * e + 2] = v4[k][e].z + bias4[e].z + rf[k][e].z + bf_lo(rbw[2 * e + 1]);
                v[4 * e + 3] = v4[k][e].w + bias4[e].w + rf[k][e].w + bf_hi(rbw[2 * e + 1]);
            }
            if (p.clamp01) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (fminf(fmaxf(v[e], -1.f), 1.f) + 1.f) * 0.5f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
            const long o = mrow[k] * p.Cout + n;
            if (p.out_f32) {
                *reinterpret_cast<float4*>(p.out_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.out_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (p.out_bf16) *reinterpret_cast<uint4*>(p.out_bf16 + o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            if (p.gn_partial) {  // statistics of the values the GroupNorm will read (bf16-rounded unless fp32 is stored)
                const int hf = (WM == 4) ? (k >> 1) : 0;  // pieces 0, 1 -> rows 0..127, 2, 3 -> 128..255
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float u = p.out_f32 ? v[e] : ((e & 1) ? bf_hi(pk[e >> 1]) : bf_lo(pk[e >> 1]));
                    gs[hf][e] += u, gq[hf][e] += u * u;
                }
            }
        }
    }
    // ---- fused GroupNorm statistics (model.py:38-42 reads this tensor next): fixed-order reductions, no atomics.
    // thread -> LDS [half][row group][channel][2] -> per channel over row groups -> per group over its channels ->
    // partial[n][128-row block][group][2]; groupnorm finalisation adds the blocks in order.
    if (p.gn_partial) {
        constexpr int RG = S::THREADS / 16;
        float* red = reinterpret_cast<float*>(smem);             // [HALVES][RG][128][2]
        float* chs = red + HALVES * RG * 256;                     // [HALVES][128][2]
        __syncthreads();
        const int rg = tid >> 4, c8 = (tid & 15) * 8;
#pragma unroll
        for (int hf = 0; hf < HALVES; ++hf)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<float2*>(red + ((hf * RG + rg) * 128 + c8 + e) * 2) = make_float2(gs[hf][e], gq[hf][e]);
        __syncthreads();
        for (int idx = tid; idx < HALVES * 256; idx += S::THREADS) {
            const int hf = idx >> 8, cw = idx & 255;
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < RG; ++r) a += red[(hf * RG + r) * 256 + cw];
            chs[idx] = a;
        }
        __syncthreads();
        const int cpg = p.Cout >> 5, gpt = 128 / cpg;  // channels per group, groups per 128-channel tile
        const long hw = (long)p.Hout * p.Wout;
        const int nblk = (int)(hw >> 7);
        for (int idx = tid; idx < HALVES * gpt * 2; idx += S::THREADS) {
            const int which = idx & 1, g = (idx >> 1) % gpt, hf = idx / (2 * gpt);
            const long mh = bm0 + hf * 128;
            if (mh >= p.M) continue;
            float a = 0.f;
            for (int e = 0; e < cpg; ++e) a += chs[hf * 256 + (g * cpg + e) * 2 + which];
            const long img = mh / hw;
            const int blk = (int)((mh - img * hw) >> 7);
            p.gn_partial[((img * nblk + blk) * 32 + bn0 / cpg + g) * 2 + which] = a;
        }
    }
}

// out = bias + residual + sum_s partial[s]  (s in order: deterministic), both output precisions; 4 channels per thread
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ partial, int splitk, long M, int Cout,
                                                                 const float* __restrict__ bias,
                                                                 const bf16_t* __restrict__ res_bf16,
                                                                 const float* __restrict__ res_f32, int clamp01,
                                                                 bf16_t* __restrict__ out_bf16, float* __restrict__ out_f32) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const long mn = M * Cout;
    if (i >= mn) return;
    const int n = (int)(i % Cout);
    float4 a = *reinterpret_cast<const float4*>(partial + i);
    for (int s = 1; s < splitk; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(partial + (long)s * mn + i);
        a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
    }
    if (res_f32) {
        const float4 r = *reinterpret_cast<const float4*>(res_f32 + i);
        a.x += r.x, a.y += r.y, a.z += r.z, a.w += r.w;
    }
    if (res_bf16) {
        const uint2 r = *reinterpret_cast<const uint2*>(res_bf16 + i);
        a.x += bf_lo(r.x), a.y += bf_hi(r.x), a.z += bf_lo(r.y), a.w += bf_hi(r.y);
    }
    if (clamp01) {
        a.x = (fminf(fmaxf(a.x, -1.f), 1.f) + 1.f) * 0.5f, a.y = (fminf(fmaxf(a.y, -1.f), 1.f) + 1.f) * 0.5f;
        a.z = (fminf(fmaxf(a.z, -1.f), 1.f) + 1.f) * 0.5f, a.w = (fminf(fmaxf(a.w, -1.f), 1.f) + 1.f) * 0.5f;
    }
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + i) = a;
    if (out_bf16) *reinterpret_cast<uint2*>(out_bf16 + i) = make_uint2(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w));
}

// img NCHW f32 [N,3,H,W] in [0,1] -> NHWC bf16 [N,H,W,8] holding 2x-1 (vae.py:41), channels 3..7 = 0
__global__ __launch_bounds__(256) void image_to_nhwc8_kernel(const float* __restrict__ img, long npix, long hw,
                                                             bf16_t* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const long n = i / hw, p = i - n * hw;
    const float* s = img + n * 3 * hw + p;
    const float r = 2.f * s[0] - 1.f, g = 2.f * s[hw] - 1.f, b = 2.f * s[2 * hw] - 1.f;
    *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(pack_bf2(r, g), pack_bf2(b, 0.f), 0u, 0u);
}

// the same as a bf16 PAIR for the split operator: planes [2][N,H,W,8], hi = bf16(v), lo = bf16(v - hi)
__global__ __launch_bounds__(256) void image_to_nhwc8_split_kernel(const float* __restrict__ img, long npix, long hw,
                                                                   bf16_t* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    const long n = i / hw, p = i - n * hw;
    const float* s = img + n * 3 * hw + p;
    const float v[3] = {2.f * s[0] - 1.f, 2.f * s[hw] - 1.f, 2.f * s[2 * hw] - 1.f};
    const uint32_t h01 = pack_bf2(v[0], v[1]), h2 = pack_bf2(v[2], 0.f);
    *reinterpret_cast<uint4*>(out + i * 8) = make_uint4(h01, h2, 0u, 0u);
    *reinterpret_cast<uint4*>(out + (npix + i) * 8) =
        make_uint4(pack_bf2(v[0] - bf_lo(h01), v[1] - bf_hi(h01)), pack_bf2(v[2] - bf_lo(h2), 0.f), 0u, 0u);
}

// fp32 -> bf16 pair: hi = bf16(v), lo = bf16(v - hi); 8 elements per thread; planes [2][n]
__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ x, long n8, bf16_t* __restrict__ planes) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const float4 a = reinterpret_cast<const float4*>(x + i * 8)[0], b = reinterpret_cast<const float4*>(x + i * 8)[1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        l[e] = pack_bf2(v[2 * e] - bf_lo(h[e]), v[2 * e + 1] - bf_hi(h[e]));
    }
    *reinterpret_cast<uint4*>(planes + i * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(planes + (n8 + i) * 8) = make_uint4(l[0], l[1], l[2], l[3]);
}

// NHWC f32 -> NCHW f32 (first Cuse channels); small tensors only (decoder output, z)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, long total, long hw, int C,
                                                           int Cuse, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // over N*Cuse*hw (output order)
    if (i >= total) return;
    const long n = i / (Cuse * hw);
    const long r = i - n * Cuse * hw;
    const int c = (int)(r / hw);
    const long p = r - (long)c * hw;
    out[i] = x[(n * hw + p) * C + c];
}

// row softmax: P[r, :] = softmax(S[r, :] * scale) -> bf16 ; one wave per row, cols % 4 == 0
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, long rows, int cols,
                                                           float scale, bf16_t* __restrict__ p) {
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    const float4* src = reinterpret_cast<const float4*>(s + r * cols);
    float mx = -INFINITY;
    for (int c = lane; c < (cols >> 2); c += 64) {
        const float4 v = src[c];
        mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    mx = wave_max(mx) * scale;
    float sum = 0.f;
    for (int c = lane; c < (cols >> 2); c += 64) {
        const float4 v = src[c];
        sum += (__expf(v.x * scale - mx) + __expf(v.y * scale - mx)) + (__expf(v.z * scale - mx) + __expf(v.w * scale - mx));
    }
    const float inv = 1.0f / wave_sum(sum);
    uint2* dst = reinterpret_cast<uint2*>(p + r * cols);
    for (int c = lane; c < (cols >> 2); c += 64) {
        const float4 v = src[c];
        dst[c] = make_uint2(pack_bf2(__expf(v.x * scale - mx) * inv, __expf(v.y * scale - mx) * inv),
                            pack_bf2(__expf(v.z * scale - mx) * inv, __expf(v.w * scale - mx) * inv));
    }
}

template <int WM, bool FAST, int PP = 0>
void launch_conv(const ConvParams& p, hipStream_t stream) {
    using S = BlockShape<WM>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<WM, FAST, PP>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  S::LDS_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<WM, FAST, PP>), dim3(cdiv(p.Cout, BN), cdiv(p.M, S::ROWS), p.splitk), dim3(S::THREADS),
                       S::LDS_BYTES, stream, p);
}


// ---- conv_in (model.py:382-386: Conv2d(3, 128, 3, padding 1) on the 128x128 frame; round 6) ------------------------------------------------
// The per-tap kernel above spends this layer in its epilogue: K = 72 is two K tiles per 256x128 output tile, 27 tiles per CU, each with an
// LDS-staged epilogue of ~7 us -- 189 us for 226 MB of bf16 stores, four times the HBM time.  This kernel is built
// around the stores instead: a block = two image rows (256 pixels) x 128 channels; the 4 x 130 pixel input strip (8 channels = 16 B per
// pixel, zero border) is staged once in LDS; wave w owns channels [32 w, 32 w + 32) -- its 5 weight fragments (two taps of 8 channels per
// v_mfma_f32_32x32x16_bf16 k step) live in registers, the pixel operand is read from the strip at (row + ky, x + kx); results leave the
// registers directly: after a half-wave exchange a lane holds 8 consecutive channels of its pixel = one 16-byte store (2 per 32x32
// tile), or four float4 stores for fp32 outputs; GroupNorm partial sums per 128-pixel row come from the same registers (a lane's four
// accumulators of a quad are exactly one group of 128 / 32 = 4 channels), reduced over the 32 pixel lanes in a fixed order.
// TERMS = 3: the bf16-pair operator (x_hi w_hi + x_lo w_hi + x_hi w_lo: three times the k steps, both planes staged).
template <int TERMS, bool F32OUT>
__global__ __launch_bounds__(256, 2) void conv_in_kernel(const bf16_t* __restrict__ x, long x_plane, const bf16_t* __restrict__ w,
                                                         const float* __restrict__ bias, int H, bf16_t* __restrict__ out_bf16,
                                                         float* __restrict__ out_f32, float* __restrict__ gn_partial) {
    constexpr int W = 128, PW = W + 2, ROWS = 4, PLANES = TERMS > 1 ? 2 : 1, NK = 5 * TERMS;
    __shared__ uint4 strip[PLANES][ROWS * PW];
    const int tid = threadIdx.x, lane = tid & 63, ct = tid >> 6, l32 = lane & 31, kh = lane >> 5;
    const int n = blockIdx.x / (H >> 1), y0 = (blockIdx.x - n * (H >> 1)) * 2;
    for (int i = tid; i < PLANES * ROWS * PW; i += 256) {
        const int pl = i / (ROWS * PW), rc = i - pl * (ROWS * PW), r = rc / PW, c = rc - r * PW;
        const int y = y0 - 1 + r, xx = c - 1;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (y >= 0 && y < H && xx >= 0 && xx < W) v = *reinterpret_cast<const uint4*>(x + pl * x_plane + (((long)n * H + y) * W + xx) * 8);
        strip[pl][rc] = v;
    }
    bf16x8_t wf[NK];
    int boff[NK];  // strip offset of this lane's tap in k step ks (pixel (0, 0) of a tile), plane folded in
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int term = ks / 5, tap = 2 * (ks % 5) + kh;
        const int tc = tap < 9 ? tap : 8;
        wf[ks] = tap < 9 ? *reinterpret_cast<const bf16x8_t*>(w + (((long)(ct * 32 + l32) * TERMS + term) * 9 + tap) * 8) : bf16x8_t{};
        boff[ks] = (term == 1 ? ROWS * PW : 0) + (tc / 3) * PW + (tc % 3) + l32;
    }
    float4 b4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = bias ? *reinterpret_cast<const float4*>(bias + ct * 32 + 8 * j + 4 * kh) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const uint4* sp = &strip[0][0];
    float gs[2][4], gq[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) gs[r][j] = gq[r][j] = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int row = t >> 2, x0 = (t & 3) * 32;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < NK; ++ks) {
            const uint4 u = sp[boff[ks] + row * PW + x0];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], __builtin_bit_cast(bf16x8_t, u), acc, 0, 0, 0);
        }
        mfma_settle(acc);
        const long pix = ((long)n * H + y0 + row) * W + x0 + l32;
        uint32_t pk[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[4] = {acc[4 * j] + b4[j].x, acc[4 * j + 1] + b4[j].y, acc[4 * j + 2] + b4[j].z, acc[4 * j + 3] + b4[j].w};
            if constexpr (F32OUT) {
                *reinterpret_cast<float4*>(out_f32 + pix * 128 + ct * 32 + 8 * j + 4 * kh) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                pk[j][0] = pack_bf2(v[0], v[1]), pk[j][1] = pack_bf2(v[2], v[3]);
                v[0] = bf_lo(pk[j][0]), v[1] = bf_hi(pk[j][0]), v[2] = bf_lo(pk[j][1]), v[3] = bf_hi(pk[j][1]);  // what the GroupNorm reads
            }
            gs[row][j] += (v[0] + v[1]) + (v[2] + v[3]);
            gq[row][j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        if constexpr (!F32OUT) {
            // quads (j, j + 1): the lower half-wave ends with channels [8 j, 8 j + 8) of its pixel, the upper with [8 (j + 1), 8 (j + 1) + 8)
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const auto a = __builtin_amdgcn_permlane32_swap(pk[j][0], pk[j + 1][0], false, false);
                const auto b = __builtin_amdgcn_permlane32_swap(pk[j][1], pk[j + 1][1], false, false);
                *reinterpret_cast<uint4*>(out_bf16 + pix * 128 + ct * 32 + 8 * (j + kh)) = make_uint4(a[0], b[0], a[1], b[1]);
            }
        }
    }
    if (gn_partial) {  // [img][128-pixel block = image row][group][2]; group = 8 ct + 2 j + kh
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = gs[r][j], q = gq[r][j];
                a = bfly_add<0>(a), q = bfly_add<0>(q);
                a = bfly_add<1>(a), q = bfly_add<1>(q);
                a = bfly_add<2>(a), q = bfly_add<2>(q);
                a = bfly_add<3>(a), q = bfly_add<3>(q);
                a = bfly_add<4>(a), q = bfly_add<4>(q);
                if (l32 == 0)
                    *reinterpret_cast<float2*>(gn_partial + ((((long)n * H + y0 + r) * 32) + ct * 8 + 2 * j + kh) * 2) = make_float2(a, q);
            }
    }
}

bool conv_in_shape(int mode, int Hin, int Win, int Cin, int Cout, const void* rb, const void* rf, int clamp01, int splitk, const void* o16,
                   const void* o32) {
    return mode == 0 && Cin == 8 && Cout == 128 && Win == 128 && Hin % 2 == 0 && Hin >= 2 && !rb && !rf && !clamp01 && splitk == 1 &&
           ((o16 != nullptr) != (o32 != nullptr));
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace

extern "C" int mmvid_conv2d_nhwc(int mode, const void* x, int N, int Hin, int Win, int Cin, const void* w,
                                 const float* bias, int Cout, const void* residual_bf16, const float* residual_f32,
                                 int clamp01, void* out_bf16, float* out_f32, float* gn_partial, void* stream) {
    return mmvid_conv2d_nhwc_splitk(mode, x, N, Hin, Win, Cin, w, bias, Cout, residual_bf16, residual_f32, clamp01, out_bf16,
                                    out_f32, gn_partial, 1, nullptr, stream);
}

// splitk > 1: the reduction (taps x input channels) is cut into `splitk` ranges computed by separate blocks into
// workspace [splitk][M][Cout] fp32, then added in order by conv_splitk_reduce_kernel together with bias / residual.  For the
// deep, small-map layers (8x8 maps: M = 64 N rows, K = 4,608) whose 128x128 tiles would fill a fraction of the chip.
static int conv2d_launch(int terms, int mode, const void* x, int N, int Hin, int Win, int Cin, const void* w, const float* bias,
                         int Cout, const void* residual_bf16, const float* residual_f32, int clamp01, void* out_bf16, float* out_f32,
                         float* gn_partial, int splitk, float* workspace, void* stream);

extern "C" int mmvid_conv2d_nhwc_splitk(int mode, const void* x, int N, int Hin, int Win, int Cin, const void* w,
                                        const float* bias, int Cout, const void* residual_bf16, const float* residual_f32,
                                        int clamp01, void* out_bf16, float* out_f32, float* gn_partial, int splitk,
                                        float* workspace, void* stream) {
    return conv2d_launch(1, mode, x, N, Hin, Win, Cin, w, bias, Cout, residual_bf16, residual_f32, clamp01, out_bf16, out_f32,
                         gn_partial, splitk, workspace, stream);
}

// The "split" operator (vae.strict = 'split'): activations and weights are bf16 PAIRS, x = x_hi + x_lo, w = w_hi + w_lo (16
// mantissa bits each), and the convolution is the three products x_hi.w_hi + x_lo.w_hi + x_hi.w_lo accumulated in fp32 inside
// ONE K loop of three times the length (the dropped x_lo.w_lo term is 2^-18 relative): ~1e-5 of the fp32 result on the bf16
// matrix pipe.  x_planes: [2][N,Hin,Win,Cin] (hi plane, lo plane); w3: [Cout][3][taps][Cin] = (w_hi | w_hi | w_lo).
extern "C" int mmvid_conv2d_nhwc_split3(int mode, const void* x_planes, int N, int Hin, int Win, int Cin, const void* w3,
                                        const float* bias, int Cout, const float* residual_f32, int clamp01, float* out_f32,
                                        float* gn_partial, int splitk, float* workspace, void* stream) {
    return conv2d_launch(3, mode, x_planes, N, Hin, Win, Cin, w3, bias, Cout, nullptr, residual_f32, clamp01, nullptr, out_f32, gn_partial,
                         splitk, workspace, stream);
}

static int conv2d_launch(int terms, int mode, const void* x, int N, int Hin, int Win, int Cin, const void* w, const float* bias,
                         int Cout, const void* residual_bf16, const float* residual_f32, int clamp01, void* out_bf16, float* out_f32,
                         float* gn_partial, int splitk, float* workspace, void* stream) {
    MMVID_REQUIRE(x && w && (out_bf16 || out_f32), "conv2d_nhwc: null pointer");
    MMVID_REQUIRE(splitk >= 1 && splitk <= 16 && (splitk == 1 || (workspace && !gn_partial && Cout % 4 == 0)),
                  "conv2d_nhwc: split-K (%d) needs a workspace and cannot emit GroupNorm statistics", splitk);
    MMVID_REQUIRE(mode >= 0 && mode <= 3, "conv2d_nhwc: mode %d", mode);
    const int l2 = ilog2_exact(Cin);
    MMVID_REQUIRE(l2 >= 3, "conv2d_nhwc: Cin=%d must be a power of two >= 8", Cin);
    MMVID_REQUIRE(Cout % 8 == 0, "conv2d_nhwc: Cout=%d must be a multiple of 8", Cout);
    ConvParams p;
    p.x = (const bf16_t*)x, p.w = (const bf16_t*)w;
    p.N = N, p.Hin = Hin, p.Win = Win, p.Cin = Cin, p.cin_log2 = l2, p.Cout = Cout, p.mode = mode;
    p.taps = mode == 3 ? 1 : 9;
    if (mode == 1) {
        p.Hout = Hin / 2, p.Wout = Win / 2;  // pad (0,1,0,1) then 3x3 stride 2: floor((H+1-3)/2)+1 = H/2 for even H
        MMVID_REQUIRE(Hin % 2 == 0 && Win % 2 == 0, "conv2d_nhwc: downsample needs even H, W");
    } else if (mode == 2) {
        p.Hout = 2 * Hin, p.Wout = 2 * Win;
    } else {
        p.Hout = Hin, p.Wout = Win;
    }
    p.M = (long)N * p.Hout * p.Wout;
    p.K1 = p.taps * Cin, p.K = terms * p.K1;
    p.x_plane = terms > 1 ? (long)N * Hin * Win * Cin : 0;
    p.bias = bias, p.res_bf16 = (const bf16_t*)residual_bf16, p.res_f32 = residual_f32, p.clamp01 = clamp01;
    p.out_bf16 = (bf16_t*)out_bf16, p.out_f32 = out_f32, p.gn_partial = gn_partial;
    p.splitk = splitk, p.partial = splitk > 1 ? workspace : nullptr;
    if (gn_partial)
        MMVID_REQUIRE(((long)p.Hout * p.Wout) % 128 == 0 && Cout % 128 == 0,
                      "conv2d_nhwc: fused GroupNorm statistics need Hout*Wout %% 128 == 0 and Cout %% 128 == 0");
    if (p.M == 0) return MMVID_OK;
    MMVID_REQUIRE((long)N * Hin * Win * Cin * 2 * (terms > 1 ? 2 : 1) < (1ll << 31) && (long)Cout * p.K * 2 < (1ll << 31),
                  "conv2d_nhwc: input or weight of 2 GiB or more (32-bit buffer offsets)");
    MmvidProfScope prof(PROF_CONV, 2.0 * (double)p.M * Cout * p.K, (hipStream_t)stream);  // executed MFMA work (3x for split)
    if (conv_in_shape(mode, Hin, Win, Cin, Cout, residual_bf16, residual_f32, clamp01, splitk, out_bf16, out_f32)) {
        // the full-size encoder's first layer (geometry only, like every kernel choice of the encoder): conv_in_kernel
        const dim3 grid(N * (Hin / 2));
        hipStream_t st = (hipStream_t)stream;
        if (terms == 3)
            hipLaunchKernelGGL((conv_in_kernel<3, true>), grid, dim3(256), 0, st, p.x, p.x_plane, p.w, bias, Hin, (bf16_t*)nullptr, out_f32, gn_partial);
        else if (out_f32)
            hipLaunchKernelGGL((conv_in_kernel<1, true>), grid, dim3(256), 0, st, p.x, 0l, p.w, bias, Hin, (bf16_t*)nullptr, out_f32, gn_partial);
        else
            hipLaunchKernelGGL((conv_in_kernel<1, false>), grid, dim3(256), 0, st, p.x, 0l, p.w, bias, Hin, (bf16_t*)out_bf16, (float*)nullptr, gn_partial);
        MMVID_LAUNCH_CHECK("conv2d_nhwc (conv_in)");
        return MMVID_OK;
    }
    bool big = false;
    // with fused GroupNorm statistics the block shape must not depend on the batch size: the order in which a
    // 128-pixel block's partial sums are formed differs between the shapes, and a frame's tokens must not depend on
    // which other frames share its batch (tests/test_models_gpu.py::test_vqgan_roundtrip_full_size).  It is therefore
    // chosen from the layer's own geometry.
    if (gn_partial) {
        big = (long)p.Hout * p.Wout >= 4096;
    } else if (splitk == 1 && (long)cdiv(p.M, 256) * cdiv(Cout, BN) >= 200) {
        big = true;
    }
    const bool fast = Cin % 64 == 0 && mode != 2;
    if (big && fast)
        launch_conv<4, true, 1>(p, (hipStream_t)stream);  // two-group ping-pong K loop
    else if (big)
        launch_conv<4, false>(p, (hipStream_t)stream);
    else if (fast)
        launch_conv<2, true>(p, (hipStream_t)stream);
    else
        launch_conv<2, false>(p, (hipStream_t)stream);
    if (splitk > 1)
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(cdiv(p.M * Cout / 4, 256)), dim3(256), 0, (hipStream_t)stream, workspace,
                           splitk, p.M, Cout, bias, (const bf16_t*)residual_bf16, residual_f32, clamp01, (bf16_t*)out_bf16, out_f32);
    MMVID_LAUNCH_CHECK("conv2d_nhwc");
    return MMVID_OK;
}

extern "C" int mmvid_image_to_nhwc8(const float* img, int N, int H, int W, void* out_bf16, void* stream) {
    MMVID_REQUIRE(img && out_bf16, "image_to_nhwc8: null pointer");
    const long npix = (long)N * H * W;
    if (npix == 0) return MMVID_OK;
    hipLaunchKernelGGL(image_to_nhwc8_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, img, npix,
                       (long)H * W, (bf16_t*)out_bf16);
    MMVID_LAUNCH_CHECK("image_to_nhwc8");
    return MMVID_OK;
}

extern "C" int mmvid_image_to_nhwc8_split(const float* img, int N, int H, int W, void* planes_bf16, void* stream) {
    MMVID_REQUIRE(img && planes_bf16, "image_to_nhwc8_split: null pointer");
    const long npix = (long)N * H * W;
    if (npix == 0) return MMVID_OK;
    hipLaunchKernelGGL(image_to_nhwc8_split_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, (hipStream_t)stream, img, npix,
                       (long)H * W, (bf16_t*)planes_bf16);
    MMVID_LAUNCH_CHECK("image_to_nhwc8_split");
    return MMVID_OK;
}

extern "C" int mmvid_split_f32_bf16x2(const float* x, int64_t n, void* planes_bf16, void* stream) {
    MMVID_REQUIRE(x && planes_bf16 && n % 8 == 0, "split_f32_bf16x2: null pointer or n=%lld not a multiple of 8", (long long)n);
    if (n == 0) return MMVID_OK;
    hipLaunchKernelGGL(split_f32_kernel, dim3(cdiv(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)(n / 8),
                       (bf16_t*)planes_bf16);
    MMVID_LAUNCH_CHECK("split_f32_bf16x2");
    return MMVID_OK;
}

extern "C" int mmvid_nhwc_to_nchw_f32(const float* x, int N, int H, int W, int C, int Cuse, float* out, void* stream) {
    MMVID_REQUIRE(x && out && Cuse <= C, "nhwc_to_nchw_f32: bad arguments");
    const long total = (long)N * Cuse * H * W;
    if (total == 0) return MMVID_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, total,
                       (long)H * W, C, Cuse, out);
    MMVID_LAUNCH_CHECK("nhwc_to_nchw_f32");
    return MMVID_OK;
}

// AttnBlock core (model.py:188-201): w = softmax(q k^T * C^-0.5) over keys; o = w v.   q,k,v,o: [N, HW, C] bf16.
// scores_scratch: N*HW*HW fp32 followed by N*HW*HW bf16.
extern "C" int mmvid_spatial_attention(const void* q, const void* k, const void* v, int N, int HW, int C, float scale,
                                       float* scores_scratch, void* out_bf16, void* stream) {
    return mmvid_spatial_attention_ld(q, k, v, C, N, HW, C, scale, scores_scratch, out_bf16, stream);
}

// q, k, v rows `ld` elements apart (ld = 3C: the three of them are column blocks of ONE fused 1x1 convolution's output)
extern "C" int mmvid_spatial_attention_ld(const void* q, const void* k, const void* v, int64_t ld, int N, int HW, int C, float scale,
                                          float* scores_scratch, void* out_bf16, void* stream) {
    MMVID_REQUIRE(q && k && v && scores_scratch && out_bf16, "spatial_attention: null pointer");
    MMVID_REQUIRE(HW % 8 == 0 && C % 8 == 0 && ld % 8 == 0 && ld >= C, "spatial_attention: HW=%d, C=%d, ld=%lld must be multiples of 8", HW,
                  C, (long long)ld);
    const long hw2 = (long)HW * HW;
    void* P = (void*)(scores_scratch + (long)N * hw2);
    int rc = mmvid_gemm_bf16(0, 0, HW, HW, C, q, ld, k, ld, N, (long)HW * ld, (long)HW * ld, hw2, 1, 1.0f, nullptr, nullptr, 0,
                             nullptr, nullptr, 0, 0, 0, scores_scratch, nullptr, HW, nullptr, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv((long)N * HW, 4)), dim3(256), 0, (hipStream_t)stream,
                       scores_scratch, (long)N * HW, HW, scale, (bf16_t*)P);
    MMVID_LAUNCH_CHECK("spatial_attention.softmax");
    // o[q][c] = sum_key P[q][key] v[key][c] : A = P row-major [HW, HW], B = v k-major [HW(red)][C]
    return mmvid_gemm_bf16(0, 1, HW, C, HW, P, HW, v, ld, N, hw2, (long)HW * ld, (long)HW * C, 1, 1.0f, nullptr, nullptr, 0,
                           nullptr, nullptr, 0, 0, 0, nullptr, out_bf16, C, nullptr, stream);
}
