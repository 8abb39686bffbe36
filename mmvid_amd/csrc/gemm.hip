// bf16 MFMA GEMM with fused epilogues for the transformer tower (SURVEY K2, K4-K7 and their
// backward GEMMs).  One kernel template, three operand layouts:
//
//   C[m][n] = sum_k A(m,k) * B(n,k)
//     A row-major  [M][K] (lda)  or k-major [K][M]   (AKM)
//     B row-major  [N][K] (ldb)  or k-major [K][N]   (BKM)
//
//   forward  Y = X W^T          : A = X [M,K] row-major, B = W [N,K] row-major      (AKM=0,BKM=0)
//   dX = dY W                   : A = dY [M,N'] row-major, B = W [N'(red)][K(out)] k-major (AKM=0,BKM=1)
//   dW = dY^T X                 : A = dY [M(red)][N(out)] k-major, B = X [M(red)][K(out)] k-major (AKM=1,BKM=1)
//
// so neither weights nor activations ever need a transposed copy in HBM: k-major tiles are copied as they
// are and transposed by the LDS transpose read (ds_read_b64_tr_b16) when fragments are formed (gemm_core.h).
//
// Tiling for gfx950 (gemm_core.h): every wave computes 64x64 as 2x2 v_mfma_f32_32x32x16_bf16 tiles (fp32
// accumulate); a block is 128x128 (4 waves, two LDS stages, two blocks per CU) or 256x128 (8 waves, three stages,
// two-group ping-pong K loop), chosen by grid fill.  LDS tiles are rows of 64 bf16 (128 B) with 16-B chunks
// XOR-swizzled by ((row>>1)&7): conflict-free for the ds_read_b128 lane groups of the 32x32 fragment read.  Tiles are
// filled by LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds; ragged edges zero-filled by the range
// check).  MFMA operands are swapped (a = B-frag, b = A-frag) so that each lane ends up with 4 consecutive n for one
// m: 16-B epilogue loads/stores.  Roofline: bf16 MFMA (2.5 PFLOP/s dense); algorithmic FLOPs = 2*M*N*K.
#include "../../include/mmvid_hip.h"
#include "gemm_core.h"
#include "prof.h"

namespace {
using namespace mmvid_core;


constexpr int GROUP_MAX = 48;  // outputs of one grouped launch (their pointers travel in the kernel arguments)
constexpr int KIND_MAX = 4;    // shapes of one grouped launch (mmvid_gemm_bf16_dw_multi: the four Linear weights of a ResidualAttentionBlock)
struct GroupKind {             // one shape of a multi-shape grouped launch: `groups` products dW[M][N] = A_g^T B_g
    const unsigned short* A;   // (bf16) group g at A + g * strideA, k-major [K][lda]
    const unsigned short* B;
    long strideA, strideB, lda, ldb;
    int M, N, tiles_n, tiles_m;
    int first;  // first tile (in the launch's linear tile order) of this kind
    int out0;   // its outputs are out_list[out0 + g]
    int colmajor;  // walk a group's tiles column-major (rows fastest): set when the B operand (X) is the wider one, so that the tiles
                   // sharing one of ITS panels are neighbours in the XCD's stretch and the big operand is streamed once
};

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    int M, N, K;
    long lda, ldb;
    long strideA, strideB, strideC;  // batch strides in elements (0 = shared)
    int splitk;                       // >1: K split over blockIdx.z % splitk (partials to `partial`, or fp32 atomics)
    // epilogue
    const float* bias;      // [N] or null
    const float* residual;  // [M][ldr] fp32 or null  (added after activation)
    long ldr;
    const bf16_t* dact_pre;  // [M][ldp] bf16: multiply by QuickGELU'(pre) (backward) or null
    bf16_t* save_pre;        // [M][ldp] bf16: store pre-activation (forward) or null
    long ldp;
    int act;         // 0 none, 1 QuickGELU
    int accumulate;  // out_f32 += result (non-atomic) when splitk == 1
    float alpha;     // scale applied to the accumulator first
    float* out_f32;  // [M][ldc] or null
    bf16_t* out_bf16;
    long ldc;
    float* partial;  // split-K workspace [splitk][M][N] (plain stores, reduced by splitk_reduce_kernel) or null
    float* colsum;   // [N] += column sums of the stored result (the bias gradient when the result is a dY), or null
    int tiles_n, tiles_m;  // > 0: persistent blocks walk this tile grid (more tiles than CUs); 0: one block per tile
    int group_n;     // persistent blocks: tiles are walked column-GROUP-major (groups of group_n column tiles), see launch_shape
};
// Second kernel argument of the grouped launch (mmvid_gemm_bf16_dw_multi): grid.x = all tiles of all kinds and groups; a block finds
// its (kind, group, tile) here and patches its private copy of GemmParams (A, B, M, N, lda, ldb, ldc, out_f32).  The outputs are
// separate allocations (the layers' weight gradients): their pointers travel in the kernel arguments; null = skipped (frozen weight).
struct GroupTable {
    int n_kinds;
    GroupKind kinds[KIND_MAX];
    float* out_list[GROUP_MAX];
};
// sigmoid(1.702 x) on the hardware exp2 / rcp (1 ulp each; the result is rounded to bf16 right after): an IEEE division here is
// ~10 VALU instructions per element, and the epilogue of the c_fc GEMM evaluates 32,768 of them per tile with no MFMA to hide under
__device__ __forceinline__ float sigmoid1702(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x));
}
__device__ __forceinline__ float quick_gelu(float x) { return x * sigmoid1702(x); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
    float s = sigmoid1702(x);
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

// ---- epilogue through LDS, shared by both block shapes.  NI = 32-row fragments per wave (2: 64-row wave tiles, 4: 128-row),
// WAVE_ROWS = rows of a wave tile; the block's wave grid is (THREADS/128) x 2.  Pass i stages fragment i of every wave.
__device__ __forceinline__ void slab_write_row(const f32x16 (&acc_i)[2], float* slab, int wm, int wn, int lane) {
    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(slab + (wm * 32 + frow) * SLAB_PITCH + wn * 64 + j * 32 + 8 * q + 4 * fh) =
                make_float4(acc_i[j][4 * q], acc_i[j][4 * q + 1], acc_i[j][4 * q + 2], acc_i[j][4 * q + 3]);
}
template <int NT, int WAVE_ROWS>
__device__ __forceinline__ void slab_piece_wr(int tid, int k, int i, int& r, int& m_local, int& c) {
    const int idx = tid + NT * k;
    r = idx >> 5;
    c = (idx & 31) * 4;
    m_local = (r >> 5) * WAVE_ROWS + i * 32 + (r & 31);
}
template <int THREADS, int WAVE_ROWS, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, char* smem, f32x16 (&acc)[NI][2], int bm0, int bn0, int batch,
                                              int ks, int tid, int wm, int wn, int lane) {
    // ---- epilogue through LDS: every global read/write below is row-contiguous (16 B per lane, 512 B per row).
    // Thread t owns column n = bn0 + 4*(t&31) of rows (t>>5)+8k of each 64-row slab; all its reads are issued
    // before any is consumed.
#pragma unroll
    for (int i = 0; i < NI; ++i) mfma_settle(acc[i][0]), mfma_settle(acc[i][1]);
    float* slab = reinterpret_cast<float*>(smem);
    const long cb = (long)batch * p.strideC;
    const int n = bn0 + 4 * (tid & 31);
    const bool n_ok = n < p.N;
    const bool use_bias = p.bias && (p.splitk == 1 || ks == 0);
    const float4 bias4 = (use_bias && n_ok) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* addbase = p.residual ? p.residual + cb
                                      : ((p.accumulate && !p.partial && p.splitk == 1) ? p.out_f32 + cb : nullptr);
    const long ldadd = p.residual ? p.ldr : p.ldc;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};  // this thread's share of the column sums (p.colsum)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        __syncthreads();
        slab_write_row(acc[i], slab, wm, wn, lane);
        __syncthreads();
        float4 v4[8], add4[8];
        uint2 pre2[8];
        int mrow[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int r, ml, c;
            slab_piece_wr<THREADS, WAVE_ROWS>(tid, k, i, r, ml, c);
            mrow[k] = bm0 + ml;
            const bool ok = n_ok && mrow[k] < p.M;
            v4[k] = *reinterpret_cast<const float4*>(slab + r * SLAB_PITCH + c);
            add4[k] = (addbase && ok) ? *reinterpret_cast<const float4*>(addbase + (long)mrow[k] * ldadd + n)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            pre2[k] = (p.dact_pre && ok) ? *reinterpret_cast<const uint2*>(p.dact_pre + cb + (long)mrow[k] * p.ldp + n)
                                         : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int m = mrow[k];
            if (!n_ok || m >= p.M) continue;
            float v[4] = {v4[k].x * p.alpha + bias4.x, v4[k].y * p.alpha + bias4.y, v4[k].z * p.alpha + bias4.z,
                          v4[k].w * p.alpha + bias4.w};
            if (p.partial) {
                *reinterpret_cast<float4*>(p.partial + ((long)ks * p.M + m) * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
                continue;
            }
            if (p.splitk > 1) {
                float* o = p.out_f32 + cb + (long)m * p.ldc + n;
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(o + e, v[e]);
                continue;
            }
            if (p.save_pre)
                *reinterpret_cast<uint2*>(p.save_pre + cb + (long)m * p.ldp + n) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            if (p.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
            }
            if (p.dact_pre) {
                v[0] *= quick_gelu_grad(bf_lo(pre2[k].x));
                v[1] *= quick_gelu_grad(bf_hi(pre2[k].x));
                v[2] *= quick_gelu_grad(bf_lo(pre2[k].y));
                v[3] *= quick_gelu_grad(bf_hi(pre2[k].y));
            }
            v[0] += add4[k].x, v[1] += add4[k].y, v[2] += add4[k].z, v[3] += add4[k].w;
            if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + cb + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
            if (p.out_bf16)
                *reinterpret_cast<uint2*>(p.out_bf16 + cb + (long)m * p.ldc + n) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            cs[0] += v[0], cs[1] += v[1], cs[2] += v[2], cs[3] += v[3];
        }
    }
    if (p.colsum) {  // block-level column sums -> one atomic per column (as mmvid_colsum_bf16 does per 256 rows)
        constexpr int RG = THREADS / 32;
        float* red = reinterpret_cast<float*>(smem);  // [RG][128]
        __syncthreads();
        *reinterpret_cast<float4*>(red + (tid >> 5) * 128 + 4 * (tid & 31)) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        __syncthreads();
        if (tid < 128 && bn0 + tid < p.N) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < RG; ++r) a += red[r * 128 + tid];
            unsafeAtomicAdd(p.colsum + bn0 + tid, a);
        }
    }
}


// ---- register-direct epilogue (256x128 blocks).  A lane of the swapped-operand 32x32x16 MFMA holds, for
// fragment (i, j) and q = 0..3, four consecutive columns n = bn0 + wn*64 + j*32 + 8q + 4*(lane>>5) of row m = bm0 + wm*64 + i*32 +
// (lane&31): one 16-B (fp32) / 8-B (bf16) piece, so a store instruction writes 32 B / 16 B into each of 32 rows and the eight
// (j, q) stores of a fragment row fill one 128-B line.  Poorly coalesced per instruction -- but it needs no LDS slab and no
// barrier, so (a) the next tile's first two K tiles are requested BEFORE the stores and (b) the waves fall straight into the next
// K loop while the writes drain: the LDS-staged epilogue costs ~10 us per block round at M = 10,422 (three dependent phases per
// 32-row slab behind two barriers, then the first-tile latency of the next tile behind the in-order vmcnt of the stores:
// profiles/r02_gemm_anatomy.log).  Every global access goes through a buffer descriptor with out-of-range lanes pointed at the
// OOB marker: the instruction count per wave is then a compile-time constant (no exec-masked branches skipping stores), which
// keeps the MFMA waves' instruction stream free of divergence.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
struct DirectEpi {
    rsrc_t r_add, r_pre_in, r_pre_out, r_f32, r_bf16;
    bool has_add, has_dact, has_save, has_f32, has_bf16;
    int nst;  // vector-memory STORES per wave and tile (16 per output tensor)
    __device__ __forceinline__ void init(const GemmParams& p, int batch) {
        const long cb = (long)batch * p.strideC;
        const float* addbase = p.residual ? p.residual + cb : ((p.accumulate && p.out_f32) ? p.out_f32 + cb : nullptr);
        const long ldadd = p.residual ? p.ldr : p.ldc;
        has_add = addbase != nullptr, has_dact = p.dact_pre != nullptr, has_save = p.save_pre != nullptr;
        has_f32 = p.out_f32 != nullptr, has_bf16 = p.out_bf16 != nullptr;
        const void* any = p.out_f32 ? (const void*)p.out_f32 : (const void*)p.out_bf16;
        r_add = make_rsrc(has_add ? (const void*)addbase : any, has_add ? (uint32_t)(((long)(p.M - 1) * ldadd + p.N) * 4) : 0u);
        r_pre_in = make_rsrc(has_dact ? (const void*)(p.dact_pre + cb) : any, has_dact ? (uint32_t)(((long)(p.M - 1) * p.ldp + p.N) * 2) : 0u);
        r_pre_out = make_rsrc(has_save ? (const void*)(p.save_pre + cb) : any, has_save ? (uint32_t)(((long)(p.M - 1) * p.ldp + p.N) * 2) : 0u);
        r_f32 = make_rsrc(has_f32 ? (const void*)(p.out_f32 + cb) : any, has_f32 ? (uint32_t)(((long)(p.M - 1) * p.ldc + p.N) * 4) : 0u);
        r_bf16 = make_rsrc(has_bf16 ? (const void*)(p.out_bf16 + cb) : any, has_bf16 ? (uint32_t)(((long)(p.M - 1) * p.ldc + p.N) * 2) : 0u);
        nst = 16 * ((has_save ? 1 : 0) + (has_f32 ? 1 : 0) + (has_bf16 ? 1 : 0));
    }
};
// ---- packed bf16 form (bf16 outputs, N % 128 == 0): the finished tile is turned into its PACKED bf16 results (bias / activation
// applied; 32 dwords per output tensor and lane) and stored with eight 16-B stores per tensor.  (Round 3 also issued these stores a
// few at a time from inside the next tile's K loop of the loader-less kernel; with the loader waves the MFMA waves' stores drain on
// their own and that form was removed.)
template <int NOUT>
struct Pending {
    u32x4_t v[NOUT][2][2][2];   // [tensor][i][j][k]: 16 B = 8 consecutive bf16 columns of one row (after the half-wave exchange)
    uint32_t row[NOUT][2];      // byte offset of (row m_i, this lane's first column of j = k = 0), or OOB
};
// the 8 * NOUT stores of a tile, one store per tensor in (i, j, k) order
template <int NOUT>
__device__ __forceinline__ void pending_flush(const DirectEpi& d, const Pending<NOUT>& pd) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int o = 0; o < NOUT; ++o)
                    __builtin_amdgcn_raw_buffer_store_b128(pd.v[o][i][j][k], (NOUT == 2 && o == 0) ? d.r_pre_out : d.r_bf16,
                                                           pd.row[o][i] + (j * 64 + k * 32), 0, 0);
}
// the half-wave exchange of MI355X_MICROARCH / cdna_hip_programming.md T21: a lane holds columns 8q + 4 fh .. +3 of its row for q =
// 0..3; v_permlane32_swap on (q = 2k, q = 2k+1) leaves lanes 0-31 with columns 16k .. 16k+7 and lanes 32-63 with 16k+8 .. 16k+15:
// one 16-B store instead of two 8-B ones (a store costs its issue slot, not its bytes)
__device__ __forceinline__ u32x4_t widen_pair(u32x2_t lo, u32x2_t hi) {
    const auto a = __builtin_amdgcn_permlane32_swap(lo.x, hi.x, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(lo.y, hi.y, false, false);
    return u32x4_t{a[0], b[0], a[1], b[1]};
}
// the inverse exchange: a 16-B piece (8 consecutive columns) loaded per lane -> the lane's own 4 columns of q = 2k and q = 2k+1
__device__ __forceinline__ void narrow_pair(u32x4_t t, u32x2_t& lo, u32x2_t& hi) {
    const auto a = __builtin_amdgcn_permlane32_swap(t.x, t.z, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(t.y, t.w, false, false);
    lo = u32x2_t{a[0], b[0]}, hi = u32x2_t{a[1], b[1]};
}
// dact operand of a tile (the saved pre-activation, bf16): 8 16-B loads per lane, requested at the START of the tile so that
// they land under the K loop
struct PreRegs {
    u32x4_t t[2][2][2];  // [i][j][k]
};
__device__ __forceinline__ void pre_load(const GemmParams& p, const DirectEpi& d, int bm0, int bn0, int wm, int wn, int lane, PreRegs& pr) {
    const int frow = lane & 31, fh = lane >> 5;
    const int n0 = bn0 + wn * 64 + 8 * fh;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = bm0 + wm * 64 + i * 32 + frow;
        const uint32_t row = m < p.M ? (uint32_t)(((long)m * p.ldp + n0) * 2) : OOB;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) pr.t[i][j][k] = __builtin_amdgcn_raw_buffer_load_b128(d.r_pre_in, row + (j * 64 + k * 32), 0, 0);
    }
}
// column sums of a wave's 64 x 64 result: cs[c] (c = 16 j + 4 q + e, already summed over the lane's two rows) is reduced over the
// 32 lanes of each half by a halving exchange -- 31 cross-lane moves instead of 160 -- after which lane l holds column l of its
// half; one atomic per lane
__device__ __forceinline__ void colsum_wave(float (&cs)[32], float* colsum, int bn0, int wn, int lane, int N) {
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
    const int c = lane & 31, fh = lane >> 5;
    const int n = bn0 + wn * 64 + (c >> 4) * 32 + ((c >> 2) & 3) * 8 + 4 * fh + (c & 3);
    if (n < N) unsafeAtomicAdd(colsum + n, cs[0]);
}
// this lane's 32 bias values of a tile ([j][q] float4), read from the LDS copy BEFORE the next tile's LDS-DMA is requested (an LDS
// read behind in-flight LDS-DMA makes hipcc drain vmcnt)
struct BiasRegs {
    float4 b[2][4];
};
__device__ __forceinline__ void bias_load(const float* bias_lds, int bn0, int wn, int lane, BiasRegs& br) {
    const int fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            br.b[j][q] = bias_lds ? *reinterpret_cast<const float4*>(bias_lds + bn0 + wn * 64 + j * 32 + 8 * q + 4 * fh) : make_float4(0.f, 0.f, 0.f, 0.f);
}
// accumulators -> packed results (bias, activation; out tensor 0 = save_pre when NOUT == 2, last = out_bf16)
template <int NOUT>
__device__ __forceinline__ void pending_fill(const GemmParams& p, const BiasRegs& br, f32x16 (&acc)[2][2], Pending<NOUT>& pd,
                                             int bm0, int bn0, int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) mfma_settle(acc[i][0]), mfma_settle(acc[i][1]);
    const int frow = lane & 31, fh = lane >> 5;
    const int n0 = bn0 + wn * 64 + 8 * fh;  // after the exchange: lanes 32-63 own the second 8 columns of every 16
    const bool scaled = p.alpha != 1.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = bm0 + wm * 64 + i * 32 + frow;
        const bool ok = m < p.M;
        if constexpr (NOUT == 2) pd.row[0][i] = ok ? (uint32_t)(((long)m * p.ldp + n0) * 2) : OOB;
        pd.row[NOUT - 1][i] = ok ? (uint32_t)(((long)m * p.ldc + n0) * 2) : OOB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x2_t pre[4], out[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = scaled ? acc[i][j][4 * q + e] * p.alpha : acc[i][j][4 * q + e];
                v[0] += br.b[j][q].x, v[1] += br.b[j][q].y, v[2] += br.b[j][q].z, v[3] += br.b[j][q].w;
                pre[q] = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                }
                out[q] = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if constexpr (NOUT == 2) pd.v[0][i][j][k] = widen_pair(pre[2 * k], pre[2 * k + 1]);
                pd.v[NOUT - 1][i][j][k] = widen_pair(out[2 * k], out[2 * k + 1]);
            }
        }
    }
}

// accumulators * QuickGELU'(pre) -> packed bf16 (+ column sums of what is stored): the d_pre GEMM of the tower backward
__device__ __forceinline__ void pending_fill_dact(const GemmParams& p, const PreRegs& pr, f32x16 (&acc)[2][2], Pending<1>& pd, int bm0,
                                                  int bn0, int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) mfma_settle(acc[i][0]), mfma_settle(acc[i][1]);
    const int frow = lane & 31, fh = lane >> 5;
    const int n0 = bn0 + wn * 64 + 8 * fh;
    float cs[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) cs[c] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = bm0 + wm * 64 + i * 32 + frow;
        const bool live = m < p.M;
        pd.row[0][i] = live ? (uint32_t)(((long)m * p.ldc + n0) * 2) : OOB;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x2_t out[4];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                u32x2_t pre[2];
                narrow_pair(pr.t[i][j][k], pre[0], pre[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = 2 * k + h;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * p.alpha;
                    v[0] *= quick_gelu_grad(bf_lo(pre[h].x)), v[1] *= quick_gelu_grad(bf_hi(pre[h].x));
                    v[2] *= quick_gelu_grad(bf_lo(pre[h].y)), v[3] *= quick_gelu_grad(bf_hi(pre[h].y));
                    if (live) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) cs[16 * j + 4 * q + e] += v[e];
                    }
                    out[q] = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) pd.v[0][i][j][k] = widen_pair(out[2 * k], out[2 * k + 1]);
        }
    }
    if (p.colsum) colsum_wave(cs, p.colsum, bn0, wn, lane, p.N);
}

// bias_lds: the whole bias vector [N] staged in LDS once per block (or null: no bias).  Returns nothing; d.nst stores were issued.
__device__ __forceinline__ void gemm_epilogue_direct(const GemmParams& p, const DirectEpi& d, const float* bias_lds,
                                                     f32x16 (&acc)[2][2], int bm0, int bn0, int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) mfma_settle(acc[i][0]), mfma_settle(acc[i][1]);
    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = bm0 + wm * 64 + i * 32 + frow;
        const bool m_ok = m < p.M;
        f32x4 add4[2][4];
        u32x2_t pre2[2][4];
        uint32_t eoff[2][4];  // element offsets m * ld + n for ld = ldc; OOB when out of range
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = bn0 + wn * 64 + j * 32 + 8 * q + 4 * fh;
                const bool ok = m_ok && n < p.N;
                eoff[j][q] = ok ? (uint32_t)n : OOB;
                if (d.has_add) {
                    const long ldadd = p.residual ? p.ldr : p.ldc;
                    const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(d.r_add, ok ? (uint32_t)(((long)m * ldadd + n) * 4) : OOB, 0, 0);
                    add4[j][q] = __builtin_bit_cast(f32x4, t);
                }
                if (d.has_dact) pre2[j][q] = __builtin_amdgcn_raw_buffer_load_b64(d.r_pre_in, ok ? (uint32_t)(((long)m * p.ldp + n) * 2) : OOB, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = bn0 + wn * 64 + j * 32 + 8 * q + 4 * fh;
                const bool ok = eoff[j][q] != OOB;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e] * p.alpha;
                if (bias_lds) {
                    const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + (n < p.N ? n : 0));
                    v[0] += b4.x, v[1] += b4.y, v[2] += b4.z, v[3] += b4.w;
                }
                if (d.has_save) {
                    const u32x2_t w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(w, d.r_pre_out, ok ? (uint32_t)(((long)m * p.ldp + n) * 2) : OOB, 0, 0);
                }
                if (p.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                }
                if (d.has_dact) {
                    v[0] *= quick_gelu_grad(bf_lo(pre2[j][q].x));
                    v[1] *= quick_gelu_grad(bf_hi(pre2[j][q].x));
                    v[2] *= quick_gelu_grad(bf_lo(pre2[j][q].y));
                    v[3] *= quick_gelu_grad(bf_hi(pre2[j][q].y));
                }
                if (d.has_add) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += add4[j][q][e];
                }
                if (d.has_f32) {
                    const f32x4 o = {v[0], v[1], v[2], v[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o), d.r_f32,
                                                           ok ? (uint32_t)(((long)m * p.ldc + n) * 4) : OOB, 0, 0);
                }
                if (d.has_bf16) {
                    const u32x2_t w = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(w, d.r_bf16, ok ? (uint32_t)(((long)m * p.ldc + n) * 2) : OOB, 0, 0);
                }
            }
    }
}

// The LDS-epilogue kernel: 128x128 blocks (WM = 2: small grids, two LDS stages, two blocks per CU) and the 256x128 ping-pong block
// for what the loader-wave kernel below does not take (split-K with atomics, column sums of a general result, N > 4096).
template <bool AKM, bool BKM, int WM>
__global__ __launch_bounds__(WM * 128, WM == 2 ? 2 : 1) void gemm_bf16_kernel(GemmParams p) {
    using S = BlockShape<WM>;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [NSTAGE][A sub-tiles | B tile]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: LDS bases stay in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    // p.tiles_n > 0: PERSISTENT blocks -- gridDim.x blocks walk the p.tiles_n x p.tiles_m output tiles in XCD-aware order, so a CU
    // pays the block turnover once per launch instead of once per tile
    const int ntiles = p.tiles_n > 0 ? p.tiles_n * p.tiles_m : 1;
    const int tile_step = p.tiles_n > 0 ? (int)gridDim.x : 1;
    const int batch = blockIdx.z / p.splitk, ks = blockIdx.z % p.splitk;
  for (int tile = p.tiles_n > 0 ? (int)blockIdx.x : 0; tile < ntiles; tile += tile_step) {
    const int gx = p.tiles_n > 0 ? p.tiles_n : (int)gridDim.x;
    const int wg = p.tiles_n > 0 ? xcd_remap(tile, ntiles) : xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int bn0 = (wg % gx) * BN, bm0 = (wg / gx) * S::ROWS;
    const bf16_t* A = p.A + (long)batch * p.strideA;
    const bf16_t* B = p.B + (long)batch * p.strideB;

    // K range of this split (multiples of BK)
    const int ktiles_total = (p.K + BK - 1) / BK;
    const int per = (ktiles_total + p.splitk - 1) / p.splitk;
    const int kt0 = ks * per;
    int kt1 = kt0 + per;
    if (kt1 > ktiles_total) kt1 = ktiles_total;
    const int nt = kt1 - kt0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    OperandStage<AKM, S::NSUB, S::PPW> sa;
    OperandStage<BKM, 1, S::PPW> sb;
    sa.init(A, p.lda, p.M, p.K, bm0, wave, lane);
    sb.init(B, p.ldb, p.N, p.K, bn0, wave, lane);
    if constexpr (S::NSTAGE == 2) {
        // 2 stages, one barrier per K tile: the barrier (behind an explicit vmcnt(0)) makes tile t
        // visible to every wave and proves everyone is done reading the buffer tile t+1 overwrites.
        auto stage_tile = [&](int t, char* buf) {
            const int k0 = (kt0 + t) * BK;
            sa.issue(k0, p.K, buf, wave, lane);
            sb.issue(k0, p.K, buf + S::NSUB * TILE_BYTES, wave, lane);
        };
        if (nt > 0) stage_tile(0, smem);
        for (int t = 0; t < nt; ++t) {
            char* cur = smem + (t & 1) * S::STAGE_BYTES;
            char* nxt = smem + ((t + 1) & 1) * S::STAGE_BYTES;
            dma_publish_barrier();
            if (t + 1 < nt) stage_tile(t + 1, nxt);
            mma_tile<AKM, BKM>(cur + (wm >> 1) * TILE_BYTES, cur + S::NSUB * TILE_BYTES, acc, wm & 1, wn, lane);
        }
    } else {  // three stages, two-group ping-pong (gemm_core.h)
        k_loop_pingpong<AKM, BKM>(
            smem, nt, wave, lane, wm, wn, acc, [&](int t, char* buf) { sa.issue((kt0 + t) * BK, p.K, buf, wave, lane); },
            [&](int t, char* buf) { sb.issue((kt0 + t) * BK, p.K, buf + S::NSUB * TILE_BYTES, wave, lane); });
    }

    gemm_epilogue<S::THREADS, 64, 2>(p, smem, acc, bm0, bn0, batch, ks, tid, wm, wn, lane);
    if (tile + tile_step < ntiles) __syncthreads();  // the slab has been read: the next tile's DMA may overwrite the stages
  }
}

// ================================================================================================================
// Loader-wave form of the 256x128 block: 8 MFMA waves + NLOAD loader waves (gemm_core.h, k_loop_loader / k_loop_consumer),
// register-direct epilogues only (EPI 1: general, 2 / 3: packed bf16 with one / two outputs, 4: packed bf16 x QuickGELU'(pre) + column
// sums).  The loader requests the next output tile's first two K tiles while the MFMA waves are in their epilogue, so a persistent
// block streams operands continuously; the MFMA waves never wait on vmcnt (their epilogue stores drain on their own).
// (Measured and removed in round 5, logs under profiles/: four 128x64 "fat" MFMA waves (r04_gemm_fat_wave_experiment.log: 0.93-1.02x
//  per shape), sixteen waves with eight loaders, storer waves (r04_gemm_storer_wave_experiment.log), split-K slabs reduced by the last
//  block of a tile (a device-scope release = a write-back of the XCD's L2 per block: +105 us per weight-gradient GEMM), staggered
//  starts of the persistent blocks (r05_gemm_layer_calls_stagger_sweep_rejected.log).)
template <bool AKM, bool BKM, int EPI, bool GROUPED>
__device__ __forceinline__ void gemm_lw_body(GemmParams p, const GroupTable* gt) {
    using S = BlockShape<4>;
    constexpr int NMW = 8, NL = NLOAD;  // MFMA waves, loader waves
    int multi_bm0 = 0, multi_bn0 = 0;
    if constexpr (GROUPED) {  // this block's (kind, group, tile): every XCD walks a contiguous stretch of the (kind, group, row, column) order
        const int wg = xcd_remap(blockIdx.x, gridDim.x);
        GroupKind kd = gt->kinds[0];  // (static indices + selects: a dynamically indexed copy would live in scratch memory)
#pragma unroll
        for (int i = 1; i < KIND_MAX; ++i)
            if (wg >= gt->kinds[i].first) kd = gt->kinds[i];  // (unused kinds carry first = INT_MAX)
        const int rem = wg - kd.first, per = kd.tiles_n * kd.tiles_m;
        const int g = rem / per, t = rem - g * per;
        float* const out = gt->out_list[kd.out0 + g];
        if (out == nullptr) return;  // (block-uniform, before any barrier)
        int tm, tn;
        if (kd.colmajor)
            tn = t / kd.tiles_m, tm = t - tn * kd.tiles_m;
        else
            tm = t / kd.tiles_n, tn = t - tm * kd.tiles_n;
        multi_bm0 = tm * S::ROWS, multi_bn0 = tn * BN;
        p.A = kd.A + (long)g * kd.strideA, p.B = kd.B + (long)g * kd.strideB;
        p.strideA = p.strideB = p.strideC = 0;
        p.M = kd.M, p.N = kd.N, p.lda = kd.lda, p.ldb = kd.ldb, p.ldc = kd.N;
        p.out_f32 = out;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];  // three stages, then the bias vector
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave & 7) >> 1, wn = wave & 1;
    const int ntiles = p.tiles_n > 0 ? p.tiles_n * p.tiles_m : 1;
    const int tile_step = p.tiles_n > 0 ? (int)gridDim.x : 1;
    // blockIdx.z = batch entry, or (split-K: batch 1) the K range whose partial product goes to slab z of the workspace
    const int ks = p.splitk > 1 ? (int)blockIdx.z : 0;
    const int batch = blockIdx.z;  // (split-K: strideA = strideB = 0, strideC = M * N: slab ks)
    const int gx = p.tiles_n > 0 ? p.tiles_n : (int)gridDim.x;
    const bf16_t* A = p.A + (long)batch * p.strideA;
    const bf16_t* B = p.B + (long)batch * p.strideB;
    const int ktiles_total = (p.K + BK - 1) / BK;
    const int per = (ktiles_total + p.splitk - 1) / p.splitk;
    const int kt0 = ks * per;
    const int nt = kt0 + per > ktiles_total ? (ktiles_total - kt0 > 0 ? ktiles_total - kt0 : 0) : per;
    float* bias_lds = nullptr;
    if (p.bias) {
        bias_lds = reinterpret_cast<float*>(smem + S::LDS_BYTES);
        for (int e = tid; e < p.N; e += 64 * NMW + 64 * NL) bias_lds[e] = p.bias[e];
        __syncthreads();
    }
    auto tile_origin = [&](int tile, int& bm0, int& bn0) {
        if constexpr (GROUPED) {
            bm0 = multi_bm0, bn0 = multi_bn0;
            return;
        }
        const int wg = p.tiles_n > 0 ? xcd_remap(tile, ntiles) : xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
        int tn = wg % gx, tm = wg / gx;
        if (p.tiles_n > 0 && p.group_n > 0) {  // column-group-major: group g = column tiles [g * group_n, ...), rows inside it, columns fastest
            const int per_group = p.tiles_m * p.group_n;
            const int g = wg / per_group, rem = wg - g * per_group;
            const int c0 = g * p.group_n;
            const int width = p.tiles_n - c0 < p.group_n ? p.tiles_n - c0 : p.group_n;
            tm = rem / width, tn = c0 + rem - tm * width;
        }
        bn0 = tn * BN, bm0 = tm * S::ROWS;
    };
    const int first = p.tiles_n > 0 ? (int)blockIdx.x : 0;
    if (wave >= NMW) {  // ---------------------------------------------------------------- the loader waves
        const int w = wave - NMW;
        LoaderStage<AKM, NL> sa;
        LoaderStage<BKM, NL> sb;
        bool have = false;
        for (int tile = first; tile < ntiles; tile += tile_step) {
            int bm0, bn0;
            if (!have) {
                tile_origin(tile, bm0, bn0);
                sa.init(A, p.lda, p.M, p.K, bm0), sb.init(B, p.ldb, p.N, p.K, bn0), sa.init_offsets(2, w, lane), sb.init_offsets(1, w, lane);
                loader_prologue<AKM, BKM, NL>(sa, sb, smem, kt0, nt, p.K, w, lane);
            }
            k_loop_loader<AKM, BKM, NL>(sa, sb, smem, kt0, nt, p.K, w, lane);
            have = false;
            if (tile + tile_step < ntiles) {  // every stage is free: stream the next output tile's first K tiles during the epilogue
                tile_origin(tile + tile_step, bm0, bn0);
                sa.init(A, p.lda, p.M, p.K, bm0), sb.init(B, p.ldb, p.N, p.K, bn0), sa.init_offsets(2, w, lane), sb.init_offsets(1, w, lane);
                loader_prologue<AKM, BKM, NL>(sa, sb, smem, kt0, nt, p.K, w, lane);
                have = true;
            }
        }
        return;
    }
    DirectEpi de;
    de.init(p, batch);
    constexpr int NOUT = EPI == 3 ? 2 : 1;
    [[maybe_unused]] Pending<NOUT> pend;
    // ------------------------------------------------------------------------------------- the eight MFMA waves
    for (int tile = first; tile < ntiles; tile += tile_step) {
        int bm0, bn0;
        tile_origin(tile, bm0, bn0);
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        [[maybe_unused]] PreRegs pre_in;
        if constexpr (EPI == 4) pre_load(p, de, bm0, bn0, wm, wn, lane, pre_in);
        k_loop_consumer<AKM, BKM>(smem, nt, wave, lane, wm, wn, acc);
        if constexpr (EPI == 4) {
            pending_fill_dact(p, pre_in, acc, pend, bm0, bn0, wm, wn, lane);
            pending_flush<1>(de, pend);
        } else if constexpr (EPI >= 2) {
            BiasRegs br;
            bias_load(bias_lds, bn0, wn, lane, br);
            pending_fill<NOUT>(p, br, acc, pend, bm0, bn0, wm, wn, lane);
            pending_flush<NOUT>(de, pend);
        } else {
            gemm_epilogue_direct(p, de, bias_lds, acc, bm0, bn0, wm, wn, lane);
        }
    }
}

template <bool AKM, bool BKM, int EPI>
__global__ __launch_bounds__(512 + 64 * mmvid_core::NLOAD, 1) void gemm_bf16_lw_kernel(GemmParams p) {
    gemm_lw_body<AKM, BKM, EPI, false>(p, nullptr);
}
// the grouped weight-gradient launch: both operands k-major, general register-direct epilogue (fp32 += result)
__global__ __launch_bounds__(512 + 64 * mmvid_core::NLOAD, 1) void gemm_bf16_lw_grouped_kernel(GemmParams p, GroupTable gt) {
    gemm_lw_body<true, true, 1, true>(p, &gt);
}

// out[i] = (accumulate ? out[i] : 0) + sum_s partial[s][i]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splitk, long mn,
                                                            float* __restrict__ out, int accumulate) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= mn) return;
    float4 a = accumulate ? *reinterpret_cast<const float4*>(out + i) : make_float4(0, 0, 0, 0);
    for (int s = 0; s < splitk; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(partial + (long)s * mn + i);
        a.x += v.x, a.y += v.y, a.z += v.z, a.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = a;
}

// Block shape by grid fill: the 256x128 / 3-stage kernel runs one block per CU, so it needs ~a full wave of blocks.
bool use_big_tile(long M, long N, long zdim) {
    const long blocks = (long)cdiv(M, 256) * cdiv(N, BN) * zdim;
    return M >= 256 && blocks >= 200;
}

constexpr int BIAS_LDS_BYTES = 16384;  // the bias vector above the stages (N <= 4096)
// register-direct epilogue: everything elementwise; not split-K, no column sums (those reduce across rows), N fits the bias slab,
// outputs addressable through 32-bit buffer offsets
bool direct_epilogue_ok(const GemmParams& p) {
    if (p.splitk != 1 || p.partial || p.colsum || p.N > BIAS_LDS_BYTES / 4) return false;
    const long rows = p.M - 1;
    const long ld = p.ldc > p.ldp ? p.ldc : p.ldp;
    const long ldr = p.residual ? p.ldr : 0;
    return (rows * (ld > ldr ? ld : ldr) + p.N) * 4 < (1ll << 31);
}

template <bool AKM, bool BKM, int WM>
void launch_shape(const GemmParams& p, int batch, hipStream_t stream) {
    using S = BlockShape<WM>;
    dim3 grid(cdiv(p.N, BN), cdiv(p.M, S::ROWS), batch * p.splitk);
    GemmParams q = p;
    q.tiles_n = q.tiles_m = 0;
    q.group_n = 0;
    if (WM == 4 && (long)grid.x * grid.y > 256) {  // more than one tile per CU: persistent blocks
        q.tiles_n = (int)grid.x, q.tiles_m = (int)grid.y;
        grid = dim3(256, 1, grid.z);
        // Several rounds per CU: an XCD's 32 blocks then meet the same B (weight) column tiles again in every round, and with all
        // column tiles in play (qkv: 3.5 MB, c_fc: 4.7 MB of W next to the A panels) they do not survive in its 4-MB L2 -- PMC r02:
        // 34 % L2 misses, 2.4x the algorithmic reads.  Walking the tiles in column GROUPS keeps one group's B tiles resident:
        // working set of one round of an XCD (32 tiles) for a group width c: c B tiles + ceil(32 / c) A panels; take the width
        // that minimises it, when that beats walking all column tiles and fits the 4-MB L2
        const double b_tile = 128.0 * p.K * 2, a_panel = 256.0 * p.K * 2;
        auto wset = [&](int c) { return c * b_tile + cdiv(32, c) * a_panel; };
        int best = q.tiles_n;
        for (int c = 1; c < q.tiles_n; ++c)
            if (wset(c) < wset(best)) best = c;
        if (best < q.tiles_n && wset(best) <= 4.0e6) {
            const int ngroups = cdiv(q.tiles_n, best);
            q.group_n = cdiv(q.tiles_n, ngroups);  // equal-width groups
        }
    }
    if constexpr (WM == 4) {
        // the d_pre GEMM (dX of c_proj): packed bf16 result, QuickGELU' of the saved pre-activation, column sums = c_fc's bias gradient
        const bool dact_packed = p.dact_pre && p.out_bf16 && !p.out_f32 && !p.residual && !p.accumulate && !p.save_pre && !p.bias &&
                                 p.act == 0 && p.N % 128 == 0 && batch == 1 && p.ldp == p.ldc;
        GemmParams pc = p;
        if (dact_packed) pc.colsum = nullptr;  // (direct_epilogue_ok refuses column sums: the dact form reduces them in registers)
        // split-K through a workspace (the dW GEMMs): slab ks = "batch entry" ks of an fp32 result without bias / accumulate
        const bool slabs = p.splitk > 1 && p.partial && batch == 1 && !p.bias && !p.residual && !p.dact_pre && !p.save_pre && !p.colsum &&
                           p.act == 0 && !p.out_bf16;
        if (slabs) {
            pc.out_f32 = p.partial, pc.partial = nullptr, pc.accumulate = 0, pc.ldc = p.N, pc.strideC = (long)p.M * p.N, pc.strideA = 0,
            pc.strideB = 0, pc.splitk = 1;  // (for the eligibility test; the kernel gets splitk back below)
        }
        if (direct_epilogue_ok((dact_packed || slabs) ? pc : p)) {
            if (slabs) q.out_f32 = pc.out_f32, q.partial = nullptr, q.accumulate = 0, q.ldc = pc.ldc, q.strideC = pc.strideC, q.strideA = 0, q.strideB = 0;
            const bool packed = p.out_bf16 && !p.out_f32 && !p.residual && !p.dact_pre && !p.accumulate && p.N % 128 == 0 && batch == 1;
            const int epi = dact_packed ? 4 : (packed ? (p.save_pre ? 3 : 2) : 1);
            const size_t lds = S::LDS_BYTES + BIAS_LDS_BYTES;
            static bool attr[5] = {false, false, false, false, false};
            auto go = [&](auto kern) {
                if (!attr[epi]) {
                    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                    attr[epi] = true;
                }
                hipLaunchKernelGGL(kern, grid, dim3(512 + 64 * NLOAD), lds, stream, q);
            };
            if (epi == 4)
                go(gemm_bf16_lw_kernel<AKM, BKM, 4>);
            else if (epi == 3)
                go(gemm_bf16_lw_kernel<AKM, BKM, 3>);
            else if (epi == 2)
                go(gemm_bf16_lw_kernel<AKM, BKM, 2>);
            else
                go(gemm_bf16_lw_kernel<AKM, BKM, 1>);
            return;
        }
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_kernel<AKM, BKM, WM>, hipFuncAttributeMaxDynamicSharedMemorySize, S::LDS_BYTES);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<AKM, BKM, WM>), grid, dim3(S::THREADS), S::LDS_BYTES, stream, q);
}

template <bool AKM, bool BKM>
int launch(const GemmParams& p, int batch, hipStream_t stream) {
    MmvidProfScope prof(AKM ? PROF_GEMM_TN : (BKM ? PROF_GEMM_NN : PROF_GEMM_NT), 2.0 * p.M * p.N * (double)p.K * batch, stream);
    if (use_big_tile(p.M, p.N, (long)batch * p.splitk))
        launch_shape<AKM, BKM, 4>(p, batch, stream);
    else
        launch_shape<AKM, BKM, 2>(p, batch, stream);
    return 0;
}

}  // namespace

// See include/mmvid_hip.h for the contract.
extern "C" int mmvid_gemm_bf16(int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int64_t lda,
                               const void* B, int64_t ldb, int batch, int64_t strideA, int64_t strideB,
                               int64_t strideC, int splitk, float alpha, const float* bias, const float* residual,
                               int64_t ldr, const void* dact_pre, void* save_pre, int64_t ldp, int act,
                               int accumulate, float* out_f32, void* out_bf16, int64_t ldc, float* out_colsum,
                               void* stream) {
    MMVID_REQUIRE(A && B && (out_f32 || out_bf16), "gemm_bf16: null pointer");
    MMVID_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "gemm_bf16: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
    MMVID_REQUIRE(N % 8 == 0 && ldc % 4 == 0, "gemm_bf16: N (%d) must be a multiple of 8 and ldc of 4", N);
    MMVID_REQUIRE(lda % 8 == 0 && ldb % 8 == 0, "gemm_bf16: lda/ldb must be multiples of 8 elements (16-B rows)");
    if (!a_kmajor) MMVID_REQUIRE(K % 8 == 0, "gemm_bf16: K (%d) must be a multiple of 8 for a row-major A", K);
    if (!b_kmajor) MMVID_REQUIRE(K % 8 == 0, "gemm_bf16: K (%d) must be a multiple of 8 for a row-major B", K);
    if (a_kmajor) MMVID_REQUIRE(M % 8 == 0, "gemm_bf16: M (%d) must be a multiple of 8 for a k-major A", M);
    MMVID_REQUIRE(!(a_kmajor && !b_kmajor), "gemm_bf16: layout (A k-major, B row-major) is not used on this path");
    MMVID_REQUIRE(splitk >= 1, "gemm_bf16: splitk must be >= 1");
    {  // operands are addressed with 32-bit byte offsets through buffer descriptors
        const int64_t ea = a_kmajor ? (int64_t)K * lda : (int64_t)M * lda, eb = b_kmajor ? (int64_t)K * ldb : (int64_t)N * ldb;
        MMVID_REQUIRE(ea * 2 < (1ll << 31) && eb * 2 < (1ll << 31), "gemm_bf16: an operand of 2 GiB or more per batch entry");
    }
    if (splitk > 1)
        MMVID_REQUIRE(out_f32 && !out_bf16 && !act && !dact_pre && !save_pre && !residual && !out_colsum,
                      "gemm_bf16: split-K supports only fp32 atomic accumulation (+bias)");
    MMVID_REQUIRE(!out_colsum || batch == 1, "gemm_bf16: out_colsum needs batch == 1");
    if (dact_pre || save_pre) MMVID_REQUIRE(ldp % 4 == 0, "gemm_bf16: ldp must be a multiple of 4");
    if (residual) MMVID_REQUIRE(ldr % 4 == 0 && !accumulate, "gemm_bf16: ldr must be a multiple of 4; residual and accumulate are exclusive");
    GemmParams p;
    p.A = (const bf16_t*)A, p.B = (const bf16_t*)B;
    p.M = M, p.N = N, p.K = K, p.lda = lda, p.ldb = ldb;
    p.strideA = strideA, p.strideB = strideB, p.strideC = strideC, p.splitk = splitk;
    p.bias = bias, p.residual = residual, p.ldr = ldr;
    p.dact_pre = (const bf16_t*)dact_pre, p.save_pre = (bf16_t*)save_pre, p.ldp = ldp;
    p.act = act, p.accumulate = accumulate, p.alpha = alpha;
    p.out_f32 = out_f32, p.out_bf16 = (bf16_t*)out_bf16, p.ldc = ldc;
    p.partial = nullptr, p.colsum = out_colsum;
    p.tiles_n = p.tiles_m = 0, p.group_n = 0;
    hipStream_t s = (hipStream_t)stream;
    if (!a_kmajor && !b_kmajor)
        launch<false, false>(p, batch, s);
    else if (!a_kmajor && b_kmajor)
        launch<false, true>(p, batch, s);
    else
        launch<true, true>(p, batch, s);
    MMVID_LAUNCH_CHECK("gemm_bf16");
    return MMVID_OK;
}

// Split factor for the dW GEMM: its output [N][K] has few tiles and the reduction (tokens) is long.  One wave of
// 256x128 blocks (one per CU); at least 6 K tiles per split; the workspace holds up to 16 partial copies.
extern "C" int mmvid_gemm_dw_pick_splitk(int64_t M, int N, int K) {
    const long tiles = (long)cdiv(N, 256) * cdiv(K, BN);
    const long ktiles = cdiv(M, BK);
    long sk = tiles > 0 ? 256 / tiles : 1;
    if (sk > ktiles / 6) sk = ktiles / 6;
    if (sk > 16) sk = 16;
    if (sk < 1) sk = 1;
    return (int)sk;
}

// dW[N][K] (+)= dY^T X reduced over M tokens: both operands k-major.  Split-K goes through `workspace`
// ([splitk][N][K] fp32) and a fixed-order reduction: deterministic, no atomics.  dY [M][ldy>=N], X [M][ldx>=K].
extern "C" int mmvid_gemm_bf16_dw(int64_t M, int N, int K, const void* dY, int64_t ldy, const void* X, int64_t ldx,
                                  int splitk, float* workspace, float* dW, int accumulate, void* stream) {
    MMVID_REQUIRE(dY && X && dW && M > 0 && N > 0 && K > 0, "gemm_bf16_dw: bad arguments");
    MMVID_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0, "gemm_bf16_dw: N, K, ldy, ldx must be multiples of 8");
    MMVID_REQUIRE(splitk >= 1 && (splitk == 1 || workspace), "gemm_bf16_dw: split-K needs a workspace");
    MMVID_REQUIRE(M * ldy * 2 < (1ll << 31) && M * ldx * 2 < (1ll << 31), "gemm_bf16_dw: an operand of 2 GiB or more");
    GemmParams p;
    p.A = (const bf16_t*)dY, p.B = (const bf16_t*)X;
    p.M = N, p.N = K, p.K = (int)M, p.lda = ldy, p.ldb = ldx;
    p.strideA = p.strideB = p.strideC = 0, p.splitk = splitk;
    p.bias = nullptr, p.residual = nullptr, p.ldr = 0, p.dact_pre = nullptr, p.save_pre = nullptr, p.ldp = 0;
    p.act = 0, p.accumulate = accumulate, p.alpha = 1.0f;
    p.out_f32 = dW, p.out_bf16 = nullptr, p.ldc = K;
    p.partial = splitk > 1 ? workspace : nullptr;
    p.colsum = nullptr;
    p.tiles_n = p.tiles_m = 0, p.group_n = 0;
    hipStream_t s = (hipStream_t)stream;
    launch<true, true>(p, 1, s);
    if (splitk > 1) {  // the slabs are added in slab order: deterministic
        const long mn = (long)N * K;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(mn / 4, 256)), dim3(256), 0, s, workspace, splitk, mn, dW, accumulate);
    }
    MMVID_LAUNCH_CHECK("gemm_bf16_dw");
    return MMVID_OK;
}

// dW_{k,g}[N_k][K_k] (+)= dY_{k,g}^T X_{k,g} for every kind k < nkinds (a shape) and group g < groups (a layer) in ONE launch: no
// split-K, no workspace, every block reduces over all M tokens in fp32 (deterministic).  The grid is the list of all output tiles;
// XCD x walks a contiguous stretch of it (tiles of one layer share their operand panels in that XCD's L2).  MI355X-first: with
// 288 GB the backward can keep every layer's dY, and the weight gradients of ALL layers then fill the chip without split-K slabs
// (tools/bench_dw_grouped.py; captured step 16.36 -> 15.56 ms, profiles/r03_ab_whole_step_dw_grouped.log).
extern "C" int mmvid_gemm_bf16_dw_multi(int64_t M, int nkinds, const mmvid_dw_kind_t* kinds, int groups, int accumulate, void* stream) {
    MMVID_REQUIRE(kinds && M > 0 && nkinds > 0 && nkinds <= KIND_MAX && groups > 0, "gemm_bf16_dw_multi: bad arguments");
    using S = BlockShape<4>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_lw_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)(S::LDS_BYTES + BIAS_LDS_BYTES));
        attr = true;
    }
    for (int k = 0; k < nkinds; ++k) {
        const mmvid_dw_kind_t& kd = kinds[k];
        MMVID_REQUIRE(kd.dY && kd.X && kd.dW_list && kd.N > 0 && kd.K > 0, "gemm_bf16_dw_multi: kind %d: null pointer / bad size", k);
        MMVID_REQUIRE(kd.N % 8 == 0 && kd.K % 8 == 0 && kd.ldy % 8 == 0 && kd.ldx % 8 == 0 && kd.strideY % 8 == 0 && kd.strideX % 8 == 0,
                      "gemm_bf16_dw_multi: kind %d: N, K, ldy, ldx and the group strides must be multiples of 8", k);
        MMVID_REQUIRE(M * kd.ldy * 2 < (1ll << 31) && M * kd.ldx * 2 < (1ll << 31), "gemm_bf16_dw_multi: an operand of 2 GiB or more per group");
        MMVID_REQUIRE((int64_t)kd.N * kd.K < (1ll << 29), "gemm_bf16_dw_multi: an output of 2 GiB or more");
    }
    const int per_launch = GROUP_MAX / nkinds;  // groups whose output pointers fit one launch
    for (int g0 = 0; g0 < groups; g0 += per_launch) {
        const int n = groups - g0 < per_launch ? groups - g0 : per_launch;
        GroupTable gt;
        gt.n_kinds = nkinds;
        int tiles = 0;
        bool any = false;
        double flops = 0;
        for (int g = 0; g < GROUP_MAX; ++g) gt.out_list[g] = nullptr;
        for (int k = 0; k < KIND_MAX; ++k) {
            GroupKind& o = gt.kinds[k];
            if (k >= nkinds) {
                o = gt.kinds[0];
                o.first = 0x7fffffff;
                continue;
            }
            const mmvid_dw_kind_t& kd = kinds[k];
            o.A = (const unsigned short*)kd.dY + (int64_t)g0 * kd.strideY, o.B = (const unsigned short*)kd.X + (int64_t)g0 * kd.strideX;
            o.strideA = kd.strideY, o.strideB = kd.strideX, o.lda = kd.ldy, o.ldb = kd.ldx;
            o.M = kd.N, o.N = kd.K, o.tiles_n = cdiv(kd.K, BN), o.tiles_m = cdiv(kd.N, S::ROWS);
            o.first = tiles, o.out0 = k * n;
            // PMC (profiles/r03_pmc_fetch_size.csv, FETCH_SIZE doubled per the gfx950 rule, calibrated in r04_pmc_fetch_calibration.txt):
            // the round-3 launch fetched 6.6 GB for 3.1 GB of operands -- row-major tile order re-reads the 64-MB activation of the
            // c_proj weight gradient (a 768 x 3072 output: 3 tile rows x 24 tile columns) once per tile row
            // once per tile row; column-major order for the kinds whose X operand is the wider one: 5.55 GB (r04)
            o.colmajor = kd.K > kd.N ? 1 : 0;
            tiles += o.tiles_n * o.tiles_m * n;
            for (int g = 0; g < n; ++g) {
                gt.out_list[o.out0 + g] = kd.dW_list[g0 + g];
                if (kd.dW_list[g0 + g]) any = true, flops += 2.0 * M * kd.N * (double)kd.K;
            }
        }
        if (!any) continue;
        GemmParams p;
        p.A = gt.kinds[0].A, p.B = gt.kinds[0].B;
        p.M = gt.kinds[0].M, p.N = gt.kinds[0].N, p.K = (int)M, p.lda = gt.kinds[0].lda, p.ldb = gt.kinds[0].ldb;
        p.strideA = p.strideB = p.strideC = 0, p.splitk = 1;
        p.bias = nullptr, p.residual = nullptr, p.ldr = 0, p.dact_pre = nullptr, p.save_pre = nullptr, p.ldp = 0;
        p.act = 0, p.accumulate = accumulate, p.alpha = 1.0f;
        p.out_f32 = nullptr, p.out_bf16 = nullptr, p.ldc = gt.kinds[0].N;
        p.partial = nullptr, p.colsum = nullptr;
        p.tiles_n = p.tiles_m = 0, p.group_n = 0;
        MmvidProfScope prof(PROF_GEMM_TN, flops, (hipStream_t)stream);
        hipLaunchKernelGGL(gemm_bf16_lw_grouped_kernel, dim3(tiles), dim3(512 + 64 * NLOAD), S::LDS_BYTES + BIAS_LDS_BYTES, (hipStream_t)stream, p, gt);
    }
    MMVID_LAUNCH_CHECK("gemm_bf16_dw_multi");
    return MMVID_OK;
}

// one kind: dW_list[g][N][K] (+)= dY_g^T X_g
extern "C" int mmvid_gemm_bf16_dw_grouped(int64_t M, int N, int K, const void* dY, int64_t ldy, int64_t strideY, const void* X,
                                          int64_t ldx, int64_t strideX, int groups, float* const* dW_list, int accumulate,
                                          void* stream) {
    mmvid_dw_kind_t kd;
    kd.N = N, kd.K = K, kd.dY = dY, kd.ldy = ldy, kd.strideY = strideY, kd.X = X, kd.ldx = ldx, kd.strideX = strideX, kd.dW_list = dW_list;
    return mmvid_gemm_bf16_dw_multi(M, 1, &kd, groups, accumulate, stream);
}

// How full the chip is when `tiles` output tiles of 256x128 go out as one launch of whole-token-reduction blocks: tiles over whole
// rounds of the 256 CUs.  The tower backward groups its weight gradients when this is >= 0.7 and otherwise keeps the per-layer
// split-K launches (a few small matrices would leave most of the chip idle for a whole token reduction).
extern "C" double mmvid_gemm_dw_multi_fill(int nkinds, const mmvid_dw_kind_t* kinds, int groups) {
    long tiles = 0;
    for (int k = 0; k < nkinds; ++k) {
        long live = 0;
        for (int g = 0; g < groups; ++g) live += kinds[k].dW_list == nullptr || kinds[k].dW_list[g] != nullptr;
        tiles += (long)cdiv(kinds[k].N, 256) * cdiv(kinds[k].K, BN) * live;
    }
    const long rounds = cdiv(tiles, 256);
    return tiles > 0 ? (double)tiles / (double)(rounds * 256) : 0.0;
}
