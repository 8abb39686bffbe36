/* mmvid_hip.h -- C-ABI of libmmvid_hip.so: the MI355X (gfx950) kernels behind the MMVID video-token hot path.
 *
 * The reference (snap-research/MMVID) has no FFI: its hot path is stock PyTorch ops called from
 * mmvid_pytorch/{dalle_bert.py, dalle_artv.py, vae.py, transformers/clip_model.py} and taming/modules/.
 * Each entry point below replaces the PyTorch op(s) at the cited reference lines.  Conventions:
 *   - plain C, device pointers + sizes, caller-allocated outputs and workspaces, no ownership transfer;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work;
 *   - return 0 on success; otherwise an error code, message via mmvid_last_error() (per host thread);
 *   - bf16 buffers are raw uint16 bit patterns (`void*` here), fp32 accumulation everywhere;
 *   - "ld*" are leading dimensions in ELEMENTS.
 * The Python binding is mmvid_amd/_lib.py (ctypes); INTEGRATION.md shows the reference-side call sites.
 */
#ifndef MMVID_HIP_H
#define MMVID_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* mmvid_last_error(void);
int mmvid_abi_version(void);
int mmvid_device_count(void);

/* ---- VQ codebook lookup: taming/modules/vqvae/quantize.py:306-311 (VectorQuantizer2.forward) ---------------
 * ee[j] = sum_k e[j][k]^2 (fmaf chain, k ascending).  idx[r] = first argmin_j (zz_r + ee_j) - 2 z_r.e_j with the
 * operation order of oracle/vq_argmin.c (bit-exact).  dim must be 256, n a multiple of 32.  dmin may be NULL. */
int mmvid_vq_sqnorm(const float* codebook, int n, int dim, float* ee, void* stream);
int mmvid_vq_argmin_l2(const float* z, const float* codebook, const float* ee, int64_t rows, int n, int dim,
                       int64_t* idx, float* dmin, void* stream);
/* mmvid_pytorch/vae.py:50 quantize.embedding(img_seq) (+ the NHWC layout the conv kernels use) and every
 * nn.Embedding lookup on the path.  out_f32 / out_bf16: either may be NULL. */
int mmvid_gather_rows(const float* table, int64_t table_rows, const int64_t* idx, int64_t rows, int dim,
                      float* out_f32, void* out_bf16, void* stream);

/* ---- bf16 MFMA GEMM with fused epilogue: nn.Linear / MultiheadAttention projections,
 * clip_model.py:208-213,222 and dalle_bert.py:414-417 (+ their autograd backward GEMMs).
 *   C[m][n] = alpha * sum_k A(m,k) B(n,k)   A row-major [M][K] or k-major [K][M]; B row-major [N][K] or k-major [K][N]
 *   then: +bias[n]; save_pre<-bf16; act (1 = QuickGELU, clip_model.py:196-198); *QuickGELU'(dact_pre);
 *         +residual[m][n]; (+= out_f32 if accumulate); store out_f32 and/or out_bf16.
 *   splitk > 1: K is split over blocks, fp32 atomicAdd into out_f32 (which must hold the base value).  * out_colsum (optional, batch 1, no split-K): [N] += column sums of the stored result -- the bias gradient of the
 * Linear whose output gradient this GEMM produces (fused instead of a separate pass over the result). */
int mmvid_gemm_bf16(int a_kmajor, int b_kmajor, int M, int N, int K, const void* A, int64_t lda, const void* B,
                    int64_t ldb, int batch, int64_t strideA, int64_t strideB, int64_t strideC, int splitk,
                    float alpha, const float* bias, const float* residual, int64_t ldr, const void* dact_pre,
                    void* save_pre, int64_t ldp, int act, int accumulate, float* out_f32, void* out_bf16,
                    int64_t ldc, float* out_colsum,
                    void* stream);

/* Weight gradient dW[N][K] (+)= dY^T X over M tokens (autograd of nn.Linear); split-K through `workspace`
 * ([splitk][N][K] fp32) with a fixed-order reduction: deterministic. */
int mmvid_gemm_bf16_dw(int64_t M, int N, int K, const void* dY, int64_t ldy, const void* X, int64_t ldx, int splitk,
                       float* workspace, float* dW, int accumulate, void* stream);
/* The split factor this library would choose for that GEMM on MI355X (1..16): size `workspace` with it. */
int mmvid_gemm_dw_pick_splitk(int64_t M, int N, int K);

/* ---- LayerNorm: clip_model.py:188-193 (fp32 statistics, eps 1e-5) and the nn.LayerNorm of the heads. */
int mmvid_layernorm_fwd(const float* x, int64_t ldx, int64_t rows, int E, const float* w, const float* b, float eps,
                        void* y_bf16, float* y_f32, int64_t ldy, float* mean, float* rstd, void* stream);
/* dx (+)= LN backward of dy; optional bf16 copy of the resulting dx (same lddx); dw/db accumulated with atomics
 * (may be NULL); dx_colsum (may be NULL) += column sums of the resulting dx = the bias gradient of the Linear that
 * produced the residual branch this gradient flows into next. */
int mmvid_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                        const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                        int add_into_dx, void* dx_bf16, float* dw, float* db, float* dx_colsum, void* stream);
/* GroupNorm(32, eps) [+ swish] on NHWC: taming/modules/diffusionmodules/model.py:38-42, 33-35.
 * stats_scratch: fp32 [N * (2*C + 64 * ceil(hw / 128))] = the per-channel affine [N][C][2], then partial sums
 * [N][blocks][32][2].  partial_blocks = 0: the statistics pass runs here; = hw/128: the producing convolution
 * already wrote the partial sums (mmvid_conv2d_nhwc gn_partial).  Deterministic: fixed-order reductions, no atomics. */
int mmvid_groupnorm_swish_nhwc(const void* x, int x_is_bf16, int N, int64_t hw, int C, const float* w,
                               const float* b, float eps, int swish, float* stats_scratch, int partial_blocks,
                               void* y_bf16, float* y_f32, void* stream);

/* ---- attention core, head_dim 64: clip_model.py:217-222 with the masks of clip_model.py:561-578.
 * qkv: token-major [B*L, ld] bf16 with Q at column 0, K at E, V at 2E (nn.MultiheadAttention packing).
 * K/V (and Q/dO in the backward) tiles are read from this layout directly; no transposed copies are needed.
 * mask_mode 0 none | 1 causal | 2 rows (r0: columns < c0 masked, r1: columns < c1 masked; use -1 for unused).
 * lse2[b][h][q] = log2-domain log-sum-exp, consumed by the backward.  delta: fp32 [B,H,L] scratch. */
int mmvid_attention_fwd(const void* qkv, int64_t ld, int B, int L, int H, int E, float scale, int mask_mode, int r0,
                        int c0, int r1, int c1, void* out, int64_t ldo, float* lse2, void* stream);
int mmvid_attention_bwd(const void* qkv, int64_t ld, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                        const float* lse2, float* delta, int B, int L, int H, int E, float scale, int mask_mode,
                        int r0, int c0, int r1, int c1, void* dqkv, int64_t ldg, void* stream);

/* ---- sequence assembly + losses: dalle_bert.py:899-973,1030-1040; dalle_artv.py:441-491,526-539. */
int mmvid_assemble_sequence(const float* const* tables, const int64_t* table_rows, int ntables, const int64_t* ids,
                            const int32_t* seg, const float* pos, int64_t B, int L, int E, float* out, void* stream);
int mmvid_assemble_sequence_bwd(float* const* grad_tables, const int64_t* table_rows, int ntables,
                                const int64_t* ids, const int32_t* seg, const float* dx, int64_t B, int L, int E,
                                float* dpos, int accumulate_dpos, void* stream);
int mmvid_cross_entropy_fwd(const float* logits, int64_t ldl, const int64_t* target, const uint8_t* select,
                            int64_t rows, int V, float* lse, float* loss_sum, void* stream);
int mmvid_cross_entropy_bwd(const float* logits, int64_t ldl, const int64_t* target, const uint8_t* select,
                            const float* lse, const float* gscale, int64_t rows, int V, void* dlogits_bf16,
                            int64_t ldd, void* stream);
int mmvid_colsum_bf16(const void* dy, int64_t ld, int64_t M, int N, float* db, void* stream);

/* ---- optimiser: train.py:322-325 (clip_grad_norm_ 1.0 + Adam), utils_train.py:167-172. */
int mmvid_grad_sqnorm(const float* g, int64_t n, float* out_accum, void* stream);
/* step_dev (optional): device fp32 scalar holding the step count t; when given it replaces `step` in the bias
 * corrections, so that a captured training step can be replayed with an advancing step count. */
int mmvid_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, const float* step_dev,
                    float max_norm, const float* sqnorm, float grad_scale, void* stream);
int mmvid_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* ---- whole CLIP tower (12 x ResidualAttentionBlock), layer loop in native code:
 * clip_model.py:580-584 -> 230-247 -> 224-227.  x is [B*L, E] fp32, batch-first. */
typedef struct {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    const void* in_w;  /* [3E, E] bf16 */
    const float* in_b; /* [3E] */
    const void* out_w; /* [E, E] */
    const float* out_b;
    const void* fc_w; /* [F, E] */
    const float* fc_b;
    const void* pj_w; /* [E, F] */
    const float* pj_b;
    /* fp32 gradient accumulators (NULL for inference) */
    float *g_ln1_w, *g_ln1_b, *g_ln2_w, *g_ln2_b, *g_in_w, *g_in_b, *g_out_w, *g_out_b, *g_fc_w, *g_fc_b, *g_pj_w,
        *g_pj_b;
} mmvid_tower_layer_t;

typedef struct {
    int B, L, E, H, F, layers;
    int mask_mode, r0, c0, r1, c1;
    float ln_eps;
} mmvid_tower_cfg_t;

/* Bytes of the saved-activation arena (training) and of the scratch arena (both modes). */
int mmvid_tower_workspace(const mmvid_tower_cfg_t* cfg, int64_t* saved_bytes, int64_t* scratch_bytes);
/* saved == NULL: inference (activations are not kept).  x_out may alias x_in only when saved == NULL. */
int mmvid_tower_forward(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                        float* x_out, void* saved, void* scratch, void* stream);
/* g: dL/dx_out on entry, dL/dx_in on exit (in place, fp32 [B*L, E]). */
int mmvid_tower_backward(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, float* g,
                         const void* saved, void* scratch, void* stream);


/* ---- incremental (KV-cache) decoding of the causal tower: what dalle_artv.py:236-304 recomputes from scratch for
 * every sampled token (SURVEY next-row N1).  kv_cache: [layers][B][Lmax][2E] bf16, a row = K then V of one position.
 * prefill = the causal forward over the prompt (cfg->L positions) that also fills the cache; decode = one new
 * position per sequence (x_in / x_out [B, E] fp32) at index *pos_dev (device scalar: the call can be captured and
 * replayed for every position) or `pos` when pos_dev is NULL.  Needs cfg->mask_mode == 1; scratch as for the forward. */
int mmvid_tower_prefill(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                        float* x_out, void* kv_cache, int Lmax, void* scratch, void* stream);
int mmvid_tower_decode(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in, float* x_out,
                       void* kv_cache, int Lmax, const int32_t* pos_dev, int pos, void* scratch, void* stream);
/* building blocks: append K|V rows of qkv [B*L, ldq] at positions pos..pos+L-1, and one-query attention over the
 * cached positions 0..pos (head_dim 64, Lmax <= 4096). */
int mmvid_kv_store(const void* qkv, int64_t ldq, int B, int L, int E, const int32_t* pos_dev, int pos0, int Lmax,
                   void* cache, void* stream);
int mmvid_attention_decode(const void* qkv, int64_t ldq, const void* cache, int B, int Lmax, int H, int E,
                           const int32_t* pos_dev, int pos0, float scale, void* out, int64_t ldo, void* stream);

/* ---- VQGAN convolutions on NHWC bf16: taming/modules/diffusionmodules/model.py:56-62,77-81,102-128,159-205,
 * taming/models/vqgan.py:41-43.  mode 0: 3x3 stride 1 pad 1 | 1: 3x3 stride 2, zero pad right/bottom (Downsample)
 * | 2: nearest x2 upsample fused with 3x3 pad 1 (Upsample) | 3: 1x1.
 * x [N, Hin, Win, Cin] bf16 (Cin % 8 == 0), w [Cout][kh][kw][Cin] bf16, bias fp32 [Cout];
 * out = conv + bias (+ residual bf16/f32 NHWC) [-> (clamp(.,-1,1)+1)/2 when clamp01, vae.py:55]
 *     -> bf16 and/or fp32 NHWC [N, Hout, Wout, Cout].
 * gn_partial (optional; needs Hout*Wout % 128 == 0 and Cout % 128 == 0): GroupNorm(32) partial sums of the output,
 * [N][Hout*Wout/128][32][sum, sumsq] -- the partial-sum area of mmvid_groupnorm_swish_nhwc's stats_scratch. */
int mmvid_conv2d_nhwc(int mode, const void* x, int N, int Hin, int Win, int Cin, const void* w, const float* bias,
                      int Cout, const void* residual_bf16, const float* residual_f32, int clamp01, void* out_bf16,
                      float* out_f32, float* gn_partial, void* stream);
/* img NCHW fp32 [N,3,H,W] in [0,1] -> NHWC bf16 [N,H,W,8] of 2x-1 (vae.py:41), channels 3..7 zero. */
int mmvid_image_to_nhwc8(const float* img, int N, int H, int W, void* out_bf16, void* stream);
/* NHWC fp32 [N,H,W,C] -> NCHW fp32 (first Cuse channels). */
int mmvid_nhwc_to_nchw_f32(const float* x, int N, int H, int W, int C, int Cuse, float* out, void* stream);
/* single-head spatial attention of AttnBlock (model.py:180-205): q,k,v NHWC bf16 [N, HW, C] -> o bf16. */
int mmvid_spatial_attention(const void* q, const void* k, const void* v, int N, int HW, int C, float scale,
                            float* scores_scratch, void* out_bf16, void* stream);

/* ---- native op-list executor for the VQGAN encoder / decoder (model.py:439-466, 551-582; vae.py:38-56): the host
 * plans the op sequence once per input shape, every call is then one host->native transition.  Offsets are bytes
 * into `arena` (-1 = unused); w/b/ext_* are device pointers. */
enum {
    MMVID_VQOP_IMG2NHWC8 = 0, /* ext_in img [N,3,H,W] f32 -> out_bf16 [N,H,W,8]                                  */
    MMVID_VQOP_CONV = 1,      /* in0 x [N,H,W,C] bf16, w, b, Cout, mode; in1 residual (flags&1: f32); flags&2 clamp01;
                                 flags&4: write GroupNorm partial sums of the output into `scratch` (a GN stats area) */
    MMVID_VQOP_GROUPNORM = 2, /* in0 [N,H,W,C] (flags&1: f32), w, b, eps, mode = swish, scratch = stats;
                                 flags&2: the partial sums in `scratch` were written by the producing CONV          */
    MMVID_VQOP_CAST = 3,      /* in0 f32 -> out_bf16, N*H*W*C elements                                           */
    MMVID_VQOP_SPATIAL_ATTN = 4, /* in0,in1,in2 = q,k,v [N,H*W,C] bf16, eps = scale, scratch                        */
    MMVID_VQOP_VQ_ARGMIN = 5, /* in0 z [N*H*W, C] f32, w = codebook [Cout, C], b = ee -> ext_out int64            */
    MMVID_VQOP_GATHER = 6,    /* ext_in idx int64 [N*H*W], w = table [Cout, C] f32 -> out_bf16 [N,H,W,C]           */
    MMVID_VQOP_NHWC2NCHW = 7  /* in0 [N,H,W,C] f32 -> ext_out [N,Cout,H,W] f32                                    */
};
typedef struct {
    int32_t op, mode;
    int32_t N, H, W, C;
    int32_t Cout, flags;
    int64_t in0, in1, in2;
    int64_t out_bf16, out_f32, scratch;
    const void* w;
    const float* b;
    const void* ext_in;
    void* ext_out;
    float eps;
    int32_t pad;
} mmvid_vqgan_op_t;
int mmvid_vqgan_run(const mmvid_vqgan_op_t* ops, int nops, void* arena, void* stream);

/* hardware probe (tools/gpu_probe.py): which = 0 -> ds_read_b64_tr_b16 lane layout. */
int mmvid_probe(int which, const void* in, void* out, void* stream);

/* ---- optional HIP-event timing of the MFMA kernel families on their launch stream (bench.py roofline line).
 * classes: 0 gemm A.B^T (forward) | 1 gemm dX | 2 gemm dW | 3 conv implicit GEMM | 4 attention fwd | 5 attention bwd.
 * Every `stride`-th launch of a class is timed.  prof_end: ms / flops summed over the sampled launches, their count,
 * and the total launch count per class. */
int mmvid_prof_begin(int stride);
int mmvid_prof_enable(int on); /* pause / resume recording without resetting */
int mmvid_prof_end(double* ms, int64_t* sampled, double* flops, int64_t* launches_total, int nclass);

/* ---- hipGraph replay of the long launch sequences (mmvid_vqgan_run, mmvid_tower_forward / _backward): a
 * sequence seen twice with identical arguments (shapes, device pointers, stream) is captured once and replayed
 * afterwards.  Opt-in (option "graphs" = 1); bypassed while the profiler above is recording.
 * counts[0..2] = sequences run directly / captured / replayed since the library was loaded. */
int mmvid_graph_stats(int64_t* counts);

/* ---- tuning knobs for A/B measurements (tools/ab_graph.py); each is also read from an environment variable at first
 * use.  "gemm_tile" (MMVID_GEMM_TILE): 0 = GEMM / conv block shape by grid fill, 128 | 256 = forced.
 * "tower_streams" (MMVID_TOWER_STREAMS): 1 = tower backward on the caller's stream (default), 2 = weight-gradient
 * chains on an internal side stream.  "graphs" (MMVID_GRAPHS): 1 = library-level graph replay (default 0). */
int mmvid_set_option(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
