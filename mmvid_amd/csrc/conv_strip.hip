// 3x3 / stride-1 / pad-1 convolution on NHWC bf16 for gfx950, "strip" form -- the VQGAN ResnetBlock convolutions
// (taming/modules/diffusionmodules/model.py:102-115) at the resolutions that dominate the encoder (128^2, 64^2, 32^2).
//
// Why a second kernel.  conv_igemm_kernel (conv.hip) treats the convolution as a GEMM whose A tile is gathered per TAP:
// every input pixel goes through the global->LDS path nine times, 48 one-KiB LDS-DMA pieces per 128 MFMAs, and the PMC
// passes showed that path -- not HBM, not the matrix pipe -- bounding the kernel.  Here a K tile is
// (kernel row ky, 32 input channels): the A operand is a STRIP of the input -- the tile's image rows shifted by ky-1, with
// one zero pixel of padding left and right -- staged ONCE and read three times (kx = 0,1,2 are the same strip, one pixel
// apart), and the block covers 512 output pixels x 128 output channels, so the weights are amortised over twice the
// pixels:
//                         conv_igemm 256x128      conv_strip 512x128
//   LDS-DMA pieces / MFMA        0.375                  0.15
//   ds_read_b128   / MFMA        1.0                    0.75
//   barriers       / MFMA        1/32 (4 per 128)       1/384
// 8 waves, each 64 pixels x 128 channels (2 x 4 v_mfma_f32_32x32x16_bf16 tiles, 128 accumulator VGPRs); two LDS stages of
// 72 KiB (A strip 40 KiB + B 32 KiB); operands reach LDS by buffer_load ... lds with the swizzle on the source address
// (gemm_core.h), zero padding and ragged edges by the descriptor's range check.  Epilogue as conv.hip: bias, residual
// (fp32 / bf16), fp32 and/or bf16 stores, GroupNorm partial sums of the output (per 64-pixel block, fixed order).
// Roofline: bf16 MFMA; algorithmic FLOPs = 2 * M * Cout * 9 * Cin.
#include "../../include/mmvid_hip.h"
#include "gemm_core.h"
#include "prof.h"

namespace {
using namespace mmvid_core;

constexpr int ST_M = 512, ST_N = 128;
constexpr int A_STAGE = 40 * 1024, B_STAGE = 32 * 1024, STAGE = A_STAGE + B_STAGE;
constexpr int PA = 5, PB = 4;  // LDS-DMA pieces per wave and K tile (A strip: 40, B: 32)
constexpr int ST_LDS = 2 * STAGE;

struct StripParams {
    const bf16_t* x;
    const bf16_t* w;  // [Cout][terms][9][Cin]
    int N, H, W, Cin, Cout, w_log2;
    int terms;        // 1, or 3 for the split operator (conv.hip): K = x_hi.w_hi | x_lo.w_hi | x_hi.w_lo
    long x_plane;     // elements between the hi and lo planes of x
    long M;
    const float* bias;
    const bf16_t* res_bf16;
    const float* res_f32;
    bf16_t* out_bf16;
    float* out_f32;
    float* gn_partial;  // [N][H*W/64][32][2] or null
    bf16_t* out_planes;  // split operator: the result as a bf16 pair (hi plane, lo plane `out_plane` elements further) or null
    long out_plane;
};

// K loop: two-group ping-pong -- the block's halves alternate between a load part and a 16-MFMA cluster, one barrier apart.  (Rounds
// 1-2 also carried a one-barrier-per-tile loop with the LDS-DMA requests right behind the barrier, and one with the requests spread
// over the first three k-steps: 18.59 / 18.53 / 18.49 ms per step, profiles/r01_ab_gemm_sched*.log; removed in round 5.)
template <bool F16>  // F16: the operands are IEEE half (v_mfma_f32_32x32x16_f16: same rate, same layouts; 11 significand bits against 8)
__global__ __launch_bounds__(512, 1) void conv_strip_kernel(StripParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk_n = p.Cout / ST_N;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int cout0 = (wg % nblk_n) * ST_N;
    const long m0 = (long)(wg / nblk_n) * ST_M;
    const int W = p.W, H = p.H, Cin = p.Cin, WP = W + 2;
    const int R = ST_M >> p.w_log2;            // image rows per tile
    const long grow0 = m0 >> p.w_log2;         // first global image-row index (over all images)
    const long grows = (long)p.N * H;

    // ---- LDS-DMA descriptors of this lane: A strip (centre row, channel chunk 0) and B
    uint32_t avoff[PA], amask[PA];
#pragma unroll
    for (int jj = 0; jj < PA; ++jj) {
        const int slot = (wave * PA + jj) * 64 + lane;
        const int sp = slot >> 2, pc = slot & 3;
        const int c = pc ^ ((sp >> 2) & 3);
        const int rr = sp / WP, xx = sp - rr * WP - 1;
        const long g = grow0 + rr;
        avoff[jj] = 0, amask[jj] = 0;
        if (rr < R && g < grows && xx >= 0 && xx < W) {
            const int y = (int)(g % H);
            avoff[jj] = (uint32_t)(((g * W + xx) * Cin + c * 8) * 2);
            amask[jj] = (y >= 1 ? 1u : 0u) | 2u | (y + 1 < H ? 4u : 0u);
        }
    }
    uint32_t bvoff[PB];
#pragma unroll
    for (int jj = 0; jj < PB; ++jj) {
        const int row = (wave * PB + jj) * 4 + (lane >> 4), ps = lane & 15;
        const int q = ps ^ (row & 15);
        bvoff[jj] = (q < 12 && cout0 + row < p.Cout)
                        ? (uint32_t)((((long)(cout0 + row) * 9 * p.terms + (q >> 2)) * Cin + (q & 3) * 8) * 2)
                        : OOB;
    }
    const rsrc_t brsrc = make_rsrc(p.w, (uint32_t)((long)p.Cout * 9 * p.terms * Cin * 2));
    const int chunks = Cin >> 5, nt = 3 * chunks * p.terms;
    // tile t = (term, kernel row ky, channel chunk ch), term outermost: A displacement and B offset in bytes (wave-uniform)
    auto a_disp = [&](int term, int ky, int ch) { return ((long)(ky - 1) * W * Cin + ch * 32 + (term == 1 ? p.x_plane : 0)) * 2; };
    auto b_soff = [&](int term, int ky, int ch) { return (uint32_t)((((term * 3 + ky) * 3) * Cin + ch * 32) * 2); };
    auto stage_tile = [&](int t, char* buf) {
        const int term = t / (3 * chunks), tt = t - term * 3 * chunks;
        const int ky = tt / chunks, ch = tt - ky * chunks;
        const rsrc_t arsrc = make_rsrc(reinterpret_cast<const char*>(p.x) + a_disp(term, ky, ch), 0x7fffffffu);
        const uint32_t bit = 1u << ky;
#pragma unroll
        for (int jj = 0; jj < PA; ++jj)
            blds16(arsrc, (amask[jj] & bit) ? avoff[jj] : OOB, 0, buf + (wave * PA + jj) * 1024);
        const uint32_t bsoff = b_soff(term, ky, ch);
#pragma unroll
        for (int jj = 0; jj < PB; ++jj) blds16(brsrc, bvoff[jj], bsoff, buf + A_STAGE + (wave * PB + jj) * 1024);
    };

    // ---- fragment addresses (byte offsets inside a stage)
    const int h8 = lane >> 5;
    uint32_t aaddr[2][3], baddr[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave * 64 + i * 32 + (lane & 31);
        const int rr = r >> p.w_log2, x = r & (W - 1);
        const int sp = rr * WP + x;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int spk = sp + kx;
            aaddr[i][kx] = (uint32_t)(spk * 64 + ((h8 ^ ((spk >> 2) & 3)) << 4));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = j * 32 + (lane & 31);
        baddr[j] = (uint32_t)(A_STAGE + row * 256 + ((h8 ^ (row & 15)) << 4));
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // Fragments of k-step q = kx*2 + s (compile-time) of the stage at LDS byte offset `st` (a multiple of 8 KiB, so the XOR
    // on address bits 4..7 commutes with adding it).
    auto load_frags = [&](uint32_t st, int q, bf16x8_t (&fa)[2], bf16x8_t (&fb)[4]) {
        const int kx = q >> 1, sx = q & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(smem + ((st + aaddr[i][kx]) ^ (uint32_t)(sx << 5)));
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(smem + ((st + baddr[j]) ^ (uint32_t)((kx << 6) | (sx << 5))));
    };
    auto mma8 = [&](const bf16x8_t (&fa)[2], const bf16x8_t (&fb)[4]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (F16)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, fb[j]), __builtin_bit_cast(f16x8_t, fa[i]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
            }
    };
    {
        // ---- two-group ping-pong (MI355X_MICROARCH.md "Two waves per SIMD"): waves 0-3 and 4-7 -- one of each per SIMD -- run one
        // barrier apart, so that on every SIMD one wave is in its MFMA cluster while the other requests operands.  A phase is one
        // kx (two k-steps): LOAD part = 12 fragment reads (+ this wave's LDS-DMA requests of the next tile), COMPUTE part = 16
        // MFMAs at raised priority, a barrier after each.
        //   WAR (the next tile overwrites the buffer tile t-1 was read from): every wave drains its fragment reads (lgkmcnt(0))
        //   BEFORE the barrier that ends its load part; a wave issuing DMA in L(t,0) has passed the barrier the trailing group
        //   passed into C(t-1,2), i.e. after that group's last read of tile t-1 completed.
        //   RAW: A pieces are requested in L(t,0), B pieces in L(t,1); every wave waits for its own pieces (vmcnt(0)) at the end
        //   of L(t,2) and then passes a barrier the other group must also pass before its first read of tile t+1.
        stage_tile(0, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wave >= 4) __builtin_amdgcn_s_barrier();  // the trailing group starts one barrier late
        int ky_n = 0, ch_n = 0, term_n = 0;
        for (int t = 0; t < nt; ++t) {
            const uint32_t st = (uint32_t)(t & 1) * STAGE;
            char* nxt = smem + ((t + 1) & 1) * STAGE;
            if (++ch_n == chunks) {
                ch_n = 0;
                if (++ky_n == 3) ky_n = 0, ++term_n;
            }
            const bool more = t + 1 < nt;
            const rsrc_t arsrc = make_rsrc(reinterpret_cast<const char*>(p.x) + (more ? a_disp(term_n, ky_n, ch_n) : 0), 0x7fffffffu);
            const uint32_t bit = 1u << ky_n;
            const uint32_t bsoff = b_soff(term_n, ky_n, ch_n);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                bf16x8_t fa[2][2], fb[2][4];
                load_frags(st, 2 * kx, fa[0], fb[0]);
                load_frags(st, 2 * kx + 1, fa[1], fb[1]);
                if (kx == 0 && more) {
#pragma unroll
                    for (int jj = 0; jj < PA; ++jj) blds16(arsrc, (amask[jj] & bit) ? avoff[jj] : OOB, 0, nxt + (wave * PA + jj) * 1024);
                }
                if (kx == 1 && more) {
#pragma unroll
                    for (int jj = 0; jj < PB; ++jj) blds16(brsrc, bvoff[jj], bsoff, nxt + A_STAGE + (wave * PB + jj) * 1024);
                }
                // fragments complete before the barrier (WAR rule above); the registers are tied so no MFMA moves above the wait
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]),
                               "+v"(fb[0][3]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3])
                             :
                             : "memory");
                if (kx == 2 && more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_s_setprio(1);
                mma8(fa[0], fb[0]);
                mma8(fa[1], fb[1]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        if (wave < 4) __builtin_amdgcn_s_barrier();  // the leading group waits for the trailing one
    }
    // ---- epilogue: per wave a private [32][132] fp32 slab, two passes (i = 0, 1); global traffic is row-contiguous
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_settle(acc[i][j]);
    __syncthreads();  // every wave is done reading the stages
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * SLAB_PITCH);
    // Round 6: a lane finishes EIGHT channels of a row (two float4 of the slab) instead of four: 16-byte bf16 stores and residual loads,
    // half the vector-memory instructions of the epilogue (a store costs the CU ~30 ns of issue time whatever it moves: 256 of them per
    // block were ~8 us of a 47-us block).  Lane = (row quarter rq, channel octet c8): rows rq, rq + 4, ..., rq + 28 of the pass.
    const int c8 = lane & 15, rq = lane >> 4, n = cout0 + 8 * c8;
    const bool n_ok = n < p.Cout;
    float4 bias4[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
        bias4[e] = (p.bias && n_ok) ? *reinterpret_cast<const float4*>(p.bias + n + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};  // per channel quad of the octet
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(slab + (lane & 31) * SLAB_PITCH + j * 32 + 8 * q + 4 * h8) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS write -> read
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // 4 rows at a time: every load of a batch is issued before any is consumed
        float4 v4[4][2], rf[4][2];
        uint4 rb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = rq + 4 * (half * 4 + k);
            const long m = m0 + wave * 64 + i * 32 + r;
            const bool ok = n_ok && m < p.M;
            const long o = m * p.Cout + n;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v4[k][e] = *reinterpret_cast<const float4*>(slab + r * SLAB_PITCH + 8 * c8 + 4 * e);
                rf[k][e] = (p.res_f32 && ok) ? *reinterpret_cast<const float4*>(p.res_f32 + o + 4 * e) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            rb[k] = (p.res_bf16 && ok) ? *reinterpret_cast<const uint4*>(p.res_bf16 + o) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = rq + 4 * (half * 4 + k);
            const long m = m0 + wave * 64 + i * 32 + r;
            if (!n_ok || m >= p.M) continue;
            const uint32_t rbw[4] = {rb[k].x, rb[k].y, rb[k].z, rb[k].w};
            float v[8];
            uint32_t pk[4], lo[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v[4 * e] = v4[k][e].x + bias4[e].x + rf[k][e].x + bf_lo(rbw[2 * e]);
                v[4 * e + 1] = v4[k][e].y + bias4[e].y + rf[k][e].y + bf_hi(rbw[2 * e]);
                v[4 * e + 2] = v4[k][e].z + bias4[e].z + rf[k][e].z + bf_lo(rbw[2 * e + 1]);
                v[4 * e + 3] = v4[k][e].w + bias4[e].w + rf[k][e].w + bf_hi(rbw[2 * e + 1]);
                pk[2 * e] = pack_bf2(v[4 * e], v[4 * e + 1]), pk[2 * e + 1] = pack_bf2(v[4 * e + 2], v[4 * e + 3]);
            }
            const long o = m * p.Cout + n;
            if (p.out_f32) {
                *reinterpret_cast<float4*>(p.out_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(p.out_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (p.out_bf16) *reinterpret_cast<uint4*>(p.out_bf16 + o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            if (p.out_planes) {  // hi = bf16(v), lo = bf16(v - hi): what mmvid_split_f32_bf16x2 would make of the fp32 result
#pragma unroll
                for (int e = 0; e < 4; ++e) lo[e] = pack_bf2(v[2 * e] - bf_lo(pk[e]), v[2 * e + 1] - bf_hi(pk[e]));
                *reinterpret_cast<uint4*>(p.out_planes + o) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                *reinterpret_cast<uint4*>(p.out_planes + p.out_plane + o) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
            }
            if (p.gn_partial) {  // statistics of the values the GroupNorm will read (bf16-rounded unless fp32 is stored)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float u[4] = {p.out_f32 ? v[4 * e] : bf_lo(pk[2 * e]), p.out_f32 ? v[4 * e + 1] : bf_hi(pk[2 * e]),
                                        p.out_f32 ? v[4 * e + 2] : bf_lo(pk[2 * e + 1]), p.out_f32 ? v[4 * e + 3] : bf_hi(pk[2 * e + 1])};
#pragma unroll
                    for (int c = 0; c < 4; ++c) gs[e] += u[c], gq[e] += u[c] * u[c];
                }
            }
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // slab reads done before the next pass overwrites it
    }
    // ---- GroupNorm partial sums of this wave's 64 pixels (one image: H*W % 64 == 0), fixed order: the lane's rows -> the four row
    // quarters -> the lanes of a group.  cpg = Cout / 32 channels per group: 4 (a channel quad each), 8 (the lane's octet) or a
    // multiple of 8 (cpg / 8 adjacent lanes).
    if (p.gn_partial) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            gs[e] += __shfl_xor(gs[e], 16, 64), gq[e] += __shfl_xor(gq[e], 16, 64);
            gs[e] += __shfl_xor(gs[e], 32, 64), gq[e] += __shfl_xor(gq[e], 32, 64);
        }
        const int cpg = p.Cout >> 5;
        const long mw = m0 + wave * 64;
        const long hw = (long)H * W;
        const long img = mw / hw;
        const int blk = (int)((mw - img * hw) >> 6);
        float* dst = p.gn_partial + ((img * (hw >> 6) + blk) * 32) * 2;
        if (cpg == 4) {
            if (rq == 0 && n_ok && mw < p.M) {
                *reinterpret_cast<float2*>(dst + (n / 4) * 2) = make_float2(gs[0], gq[0]);
                *reinterpret_cast<float2*>(dst + (n / 4 + 1) * 2) = make_float2(gs[1], gq[1]);
            }
        } else {
            float a = gs[0] + gs[1], q = gq[0] + gq[1];
            const int lpg = cpg >> 3;  // lanes per group: 1, 2, ...
            for (int o = 1; o < lpg; o <<= 1) a += __shfl_xor(a, o, 64), q += __shfl_xor(q, o, 64);
            if (rq == 0 && (c8 & (lpg - 1)) == 0 && n_ok && mw < p.M) *reinterpret_cast<float2*>(dst + (n / cpg) * 2) = make_float2(a, q);
        }
    }
}

}  // namespace

// 1 when mmvid_conv3x3_strip_nhwc takes this geometry (the caller falls back to mmvid_conv2d_nhwc otherwise)
// The choice must NOT depend on the batch size: the two kernels add in a different order, and a frame's tokens may not depend
// on which other frames share its batch (tests/test_models_gpu.py::test_vqgan_roundtrip_full_size; the VID negative reuses the
// target's tokens).  So it is made from the layer's own geometry: 32x32 maps and larger, where a training batch fills the chip
// with 512-pixel tiles.
extern "C" int mmvid_conv3x3_strip_supported(int H, int W, int Cin, int Cout) {
    if (H <= 0 || W < 8 || W > 128 || (W & (W - 1)) != 0) return 0;
    if (Cin < 32 || (Cin & (Cin - 1)) != 0 || Cout % 128 != 0) return 0;
    if (((long)H * W) % 64 != 0) return 0;
    return (long)H * W >= 1024 ? 1 : 0;
}

static int strip_launch(int terms, const void* x, int N, int H, int W, int Cin, const void* w, const float* bias, int Cout,
                        const void* residual_bf16, const float* residual_f32, void* out_bf16, float* out_f32, float* gn_partial64,
                        void* stream, bool f16, void* out_planes = nullptr);

extern "C" int mmvid_conv3x3_strip_nhwc(const void* x, int N, int H, int W, int Cin, const void* w, const float* bias, int Cout,
                                        const void* residual_bf16, const float* residual_f32, void* out_bf16, float* out_f32,
                                        float* gn_partial64, void* stream) {
    return strip_launch(1, x, N, H, W, Cin, w, bias, Cout, residual_bf16, residual_f32, out_bf16, out_f32, gn_partial64, stream, false);
}

// the split operator of conv.hip (mmvid_conv2d_nhwc_split3) in strip form: x_planes [2][N,H,W,Cin], w3 [Cout][3][9][Cin]
extern "C" int mmvid_conv3x3_strip_nhwc_split3(const void* x_planes, int N, int H, int W, int Cin, const void* w3, const float* bias,
                                               int Cout, const float* residual_f32, float* out_f32, float* gn_partial64, void* out_planes,
                                               void* stream) {
    return strip_launch(3, x_planes, N, H, W, Cin, w3, bias, Cout, nullptr, residual_f32, nullptr, out_f32, gn_partial64, stream, false, out_planes);
}

// one product of IEEE-half operands, fp32 in / out around it: x fp16 [N,H,W,Cin], w fp16 [Cout][9][Cin] (the exact-index mode's 128x128 and
// 64x64 levels, mmvid_amd/vae.py strict = 'mixed': 2^-12 per operand against the pair's 2^-17 and plain bf16's 2^-9)
extern "C" int mmvid_conv3x3_strip_nhwc_f16(const void* x_f16, int N, int H, int W, int Cin, const void* w_f16, const float* bias, int Cout,
                                            const float* residual_f32, float* out_f32, float* gn_partial64, void* out_planes, void* stream) {
    return strip_launch(1, x_f16, N, H, W, Cin, w_f16, bias, Cout, nullptr, residual_f32, nullptr, out_f32, gn_partial64, stream, true, out_planes);
}

static int strip_launch(int terms, const void* x, int N, int H, int W, int Cin, const void* w, const float* bias, int Cout,
                        const void* residual_bf16, const float* residual_f32, void* out_bf16, float* out_f32, float* gn_partial64,
                        void* stream, bool f16, void* out_planes) {
    MMVID_REQUIRE(x && w && (out_bf16 || out_f32 || out_planes), "conv3x3_strip: null pointer");
    MMVID_REQUIRE(W >= 8 && W <= 128 && (W & (W - 1)) == 0 && Cin >= 32 && (Cin & (Cin - 1)) == 0 && Cout % 128 == 0 &&
                      ((long)H * W) % 64 == 0,
                  "conv3x3_strip: unsupported geometry H=%d W=%d Cin=%d Cout=%d", H, W, Cin, Cout);
    MMVID_REQUIRE((long)N * H * W * Cin * 2 * (terms > 1 ? 2 : 1) < (1ll << 31) && (long)Cout * 9 * terms * Cin * 2 < (1ll << 31),
                  "conv3x3_strip: input or weight of 2 GiB or more (32-bit buffer offsets)");
    StripParams p;
    p.x = (const bf16_t*)x, p.w = (const bf16_t*)w, p.N = N, p.H = H, p.W = W, p.Cin = Cin, p.Cout = Cout;
    p.terms = terms, p.x_plane = terms > 1 ? (long)N * H * W * Cin : 0;
    p.w_log2 = 0;
    while ((1 << p.w_log2) < W) ++p.w_log2;
    p.M = (long)N * H * W;
    p.bias = bias, p.res_bf16 = (const bf16_t*)residual_bf16, p.res_f32 = residual_f32;
    p.out_bf16 = (bf16_t*)out_bf16, p.out_f32 = out_f32, p.gn_partial = gn_partial64;
    p.out_planes = (bf16_t*)out_planes, p.out_plane = (long)N * H * W * Cout;
    MMVID_REQUIRE(!(out_planes && gn_partial64 && !out_f32), "conv3x3_strip: GroupNorm statistics of a planes-only result are not defined");
    if (p.M == 0) return MMVID_OK;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)conv_strip_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
        (void)hipFuncSetAttribute((const void*)conv_strip_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, ST_LDS);
        attr = true;
    }
    MmvidProfScope prof(PROF_CONV, 2.0 * (double)p.M * Cout * 9 * Cin * terms, (hipStream_t)stream);
    const int blocks = cdiv(p.M, ST_M) * (Cout / ST_N);
    if (f16)
        hipLaunchKernelGGL(conv_strip_kernel<true>, dim3(blocks), dim3(512), ST_LDS, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(conv_strip_kernel<false>, dim3(blocks), dim3(512), ST_LDS, (hipStream_t)stream, p);
    MMVID_LAUNCH_CHECK("conv3x3_strip");
    return MMVID_OK;
}
