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

/* Measurement only: [512 blocks][8] time stamps (100-MHz wall clock) of the next mmvid_gemv_rows launches (csrc/decode.hip;
 * tools/bench_decode_step.py); NULL switches it off. */
int mmvid_decode_trace(void* dev_buf);
/* and for the persistent decode step (csrc/decode_persistent.hip): [4 blocks][12 layers][16] stamps (tools/decode_persistent_timeline.py). */
int mmvid_decode_persistent_trace(void* dev_buf);

/* Weight gradient dW[N][K] (+)= dY^T X over M tokens (autograd of nn.Linear); split-K through `workspace`
 * ([splitk][N][K] fp32) with a fixed-order reduction: deterministic. */
int mmvid_gemm_bf16_dw(int64_t M, int N, int K, const void* dY, int64_t ldy, const void* X, int64_t ldx, int splitk,
                       float* workspace, float* dW, int accumulate, void* stream);
/* The split factor this library would choose for that GEMM on MI355X (1..16): size `workspace` with it. */
int mmvid_gemm_dw_pick_splitk(int64_t M, int N, int K);
/* The same weight gradients for `groups` layers x `nkinds` Linear shapes in ONE launch, without split-K (the autograd of the twelve
 * ResidualAttentionBlocks' in_proj / out_proj / c_fc / c_proj weights, clip_model.py:196-227, taken after the layer loop):
 * dW_list[g][N][K] (+)= dY_g^T X_g with dY_g = dY + g * strideY ([M][ldy]) and X_g = X + g * strideX ([M][ldx]), strides in
 * elements; dW_list is a HOST array of `groups` device pointers (copied into the launch; a null entry is skipped).  Every block
 * reduces over all M tokens in fp32: deterministic, no workspace.  nkinds <= 4; any number of groups (several launches beyond
 * 48 / nkinds).  mmvid_gemm_bf16_dw_grouped is the one-kind form.  mmvid_gemm_dw_multi_fill = output tiles / (whole rounds of
 * the 256 CUs): the share of the chip such a launch keeps busy (the tower groups its weight gradients when it is >= 0.7). */
typedef struct {
    int N, K;              /* dW is [N][K] */
    const void* dY;        /* bf16 [groups][M][ldy >= N] */
    int64_t ldy, strideY;
    const void* X;         /* bf16 [groups][M][ldx >= K] */
    int64_t ldx, strideX;
    float* const* dW_list; /* host array [groups] of device pointers */
} mmvid_dw_kind_t;
int mmvid_gemm_bf16_dw_multi(int64_t M, int nkinds, const mmvid_dw_kind_t* kinds, int groups, int accumulate, void* stream);
int mmvid_gemm_bf16_dw_grouped(int64_t M, int N, int K, const void* dY, int64_t ldy, int64_t strideY, const void* X, int64_t ldx,
                               int64_t strideX, int groups, float* const* dW_list, int accumulate, void* stream);
double mmvid_gemm_dw_multi_fill(int nkinds, const mmvid_dw_kind_t* kinds, int groups);

/* ---- LayerNorm: clip_model.py:188-193 (fp32 statistics, eps 1e-5) and the nn.LayerNorm of the heads. */
int mmvid_layernorm_fwd(const float* x, int64_t ldx, int64_t rows, int E, const float* w, const float* b, float eps,
                        void* y_bf16, float* y_f32, int64_t ldy, float* mean, float* rstd, void* stream);
/* dx (+)= LN backward of dy; optional bf16 copy of the resulting dx (same lddx); dw/db accumulated with atomics
 * (may be NULL); dx_colsum (may be NULL) += column sums of the resulting dx = the bias gradient of the Linear that
 * produced the residual branch this gradient flows into next. */
int mmvid_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                        const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                        int add_into_dx, void* dx_bf16, float* dw, float* db, float* dx_colsum, void* stream);
/* The same with a caller-provided fp32 workspace (workspace_floats >= 3 * E * 64; 3 * E * 1024 is what the tower uses):
 * dw / db / dx_colsum are then reduced in two stages in a fixed order -- deterministic, no atomics, larger grid. */
int mmvid_layernorm_bwd_ws(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                           const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx,
                           int add_into_dx, void* dx_bf16, float* dw, float* db, float* dx_colsum, float* workspace,
                           int64_t workspace_floats, void* stream);
/* The same with the incoming gradient in bf16 (dy_is_bf16 != 0): the tower's dX GEMMs write d(LN output) as packed bf16 (half
 * the store instructions and bytes of an fp32 result) and this kernel reads half. */
int mmvid_layernorm_bwd_ex(const void* dy, int dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                           const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx, int add_into_dx,
                           void* dx_bf16, float* dw, float* db, float* dx_colsum, float* workspace, int64_t workspace_floats,
                           void* stream);
/* The two stages apart, for a caller that runs several LayerNorm backwards and reduces their parameter gradients together (the
 * tower backward: ln_1 / ln_2 of clip_model.py:188-193, 224-227 for all layers): _partial computes dx and leaves the weight / bias /
 * column-sum gradients as *blocks_out partial rows [blocks][3][E] in `workspace` (>= 64 * 3 * E floats, this call's own);
 * _reduce_multi adds the rows of n such calls into their targets (null targets skipped) in one launch, in the fixed order of the
 * single-call reduction: deterministic, bit-identical to mmvid_layernorm_bwd_ex with a workspace. */
typedef struct {
    const float* partial; /* [blocks][3][E] */
    float *dw, *db, *dx_colsum;
} mmvid_ln_reduce_t;
int mmvid_layernorm_bwd_partial(const void* dy, int dy_is_bf16, int64_t lddy, const float* x, int64_t ldx, const float* mean,
                                const float* rstd, const float* w, int64_t rows, int E, float* dx, int64_t lddx, int add_into_dx,
                                void* dx_bf16, int want_dw, int want_db, int want_colsum, float* workspace,
                                int64_t workspace_floats, int* blocks_out, void* stream);
int mmvid_layernorm_bwd_reduce_multi(int n, const mmvid_ln_reduce_t* items, int blocks, int E, void* stream);
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
/* The same, and dbias[3E] (optional) += column sums of dqkv: the bias gradient of the packed in-projection (nn.MultiheadAttention
 * in_proj_bias), taken from the registers that hold dq / dk / dv right before they are stored (fp32 atomics, as the separate
 * column-sum pass it replaces). */
int mmvid_attention_bwd_bias(const void* qkv, int64_t ld, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                             const float* lse2, float* delta, int B, int L, int H, int E, float scale, int mask_mode, int r0,
                             int c0, int r1, int c1, void* dqkv, int64_t ldg, float* dbias, void* stream);
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
/* Dense positional table of a sequence (dalle_bert.py:903-973; axial_positional_embedding, summed mode) in one launch, and
 * its backward in one launch.  A segment fills table rows [dst0, dst0 + rows): naxes == 0 -> rows src0.. of w[0] ([*, E]);
 * naxes 2 / 3 -> row i gets w[0][i0] + w[1][i1] (+ w[2][i2]) with (i0, i1, i2) the row-major index of i in d[0] x d[1] (x d[2]).
 * Rows no segment covers are zero.  Backward: gw[a] += d(table) summed over the rows that read it (fixed order); null gw skips. */
typedef struct {
    const float* w[3];
    float* gw[3];
    int32_t dst0, rows, naxes, src0;
    int32_t d[3];
    int32_t pad;
} mmvid_pos_segment_t;
int mmvid_pos_table_fwd(const mmvid_pos_segment_t* segs, int nseg, int L, int E, float* out, void* stream);
int mmvid_pos_table_bwd(const mmvid_pos_segment_t* segs, int nseg, int E, const float* g, void* stream);
/* out[0] = wa a[0] + wb b[0] + wc c[0] on device scalars (train.py:320: the weighted sum of the three losses; null = term
 * absent), and its backward ga / gb / gc = w * g[0]. */
int mmvid_lincomb3(const float* a, const float* b, const float* c, float wa, float wb, float wc, float* out, void* stream);
int mmvid_scale3(const float* g, float wa, float wb, float wc, float* ga, float* gb, float* gc, void* stream);
/* ---- row-wise exchange of a sparse table gradient under data parallelism (the reference all-reduces every gradient densely,
 * train.py:28-35; the text-embedding table's gradient has at most B * text_seq_len non-zero rows per rank).  rows_pack: ids [n <= 4096]
 * -> uid [n] ascending, repeats blanked to -1, rows [n][E] = W[uid] (zeros where blanked): one rank's fixed-shape message.
 * rows_merge: W[ids[i]] += rows[i] for ids[i] >= 0; ids unique within a message (no atomics): call once per peer, in rank order. */
int mmvid_rows_pack(const float* W, int64_t V, int E, const int64_t* ids, int n, int64_t* uid, float* rows, void* stream);
int mmvid_rows_merge(float* W, int64_t V, int E, const int64_t* ids, const float* rows, int n, void* stream);

/* Kernels never fault on a bad index: an embedding id outside its table reads row 0, a cross-entropy target outside [0, V)
 * counts as class 0 -- and both are COUNTED on the device (the reference's nn.Embedding / F.cross_entropy raise a device-side
 * assert instead).  counts[0] = bad embedding ids, counts[1] = bad CE targets, [2..3] reserved; reset != 0 clears them.
 * Synchronises the device: call between steps, never during stream capture. */
int mmvid_device_faults(int64_t* counts, int reset);

/* ---- optimiser: train.py:322-325 (clip_grad_norm_ 1.0 + Adam), utils_train.py:167-172. */
int mmvid_grad_sqnorm(const float* g, int64_t n, float* out_accum, void* stream);
/* step_dev (optional): device fp32 scalar holding the step count t; when given it replaces `step` in the bias
 * corrections, so that a captured training step can be replayed with an advancing step count. */
int mmvid_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                    float beta1, float beta2, float eps, float weight_decay, int step, const float* step_dev,
                    float max_norm, const float* sqnorm, float grad_scale, void* stream);
/* deterministic form of mmvid_grad_sqnorm (fixed-order reduction through `partials`, fp32 [2048]); out_accum[0] += sum g^2 */
int mmvid_grad_sqnorm_det(const float* g, int64_t n, float* partials, float* out_accum, void* stream);
/* mmvid_adam_step with the learning rate read from the device scalar lr_dev when it is not NULL (see mmvid_lr_schedule). */
int mmvid_adam_step_lr(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                       const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int step,
                       const float* step_dev, float max_norm, const float* sqnorm, float grad_scale, void* stream);
/* utils_train.py:373-385 (deepspeed WarmupLR, restated) evaluated on the device from the optimiser-step counter:
 * kind 0 constant lr_max | 1 warm-up log schedule stepped every `every` iterations (train.py:373-374). */
/* Adam / gradient norm with one LAZY table inside the flat buffers: elements [table_lo, table_lo + table_rows * rowlen) are rows
 * of an embedding table, row_flags[r] != 0 marks the rows that have EVER received a gradient.  Unflagged rows have g = m = v = 0:
 * Adam without weight decay leaves them unchanged and they add nothing to the norm, so both kernels skip them -- exact, and 30 B
 * per skipped parameter less traffic (BERT's text embedding is 30 % of its parameters; a step touches <= 3*B*64 of 49,472 rows).
 * row_flags NULL = the plain kernels. */
int mmvid_grad_sqnorm_rows(const float* g, int64_t n, float* partials, float* out_accum, const uint8_t* row_flags,
                           int64_t table_lo, int64_t table_rows, int rowlen, void* stream);
int mmvid_adam_step_rows(float* p, const float* g, float* m, float* v, void* shadow_bf16, int64_t n, float lr,
                         const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int step,
                         const float* step_dev, float max_norm, const float* sqnorm, float grad_scale,
                         const uint8_t* row_flags, int64_t table_lo, int64_t table_rows, int rowlen, void* stream);
int mmvid_lr_schedule(const float* step_dev, int kind, float lr_min, float lr_max, int warmup_steps, int every,
                      float* lr_out, void* stream);
int mmvid_counter_add(float* counter, float value, void* stream);
int mmvid_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);

/* ---- stochastic front-end of a BERT training step on the device (csrc/frontend.hip).  Every decision is a
 * function of (key, step, sample, purpose) through Philox4x32-10.  step_dev is the front-end STATE: four 32-bit device words
 * {forward-call counter (fp32), seed low word, seed high word, reserved}; key = seed argument XOR the state's seed words, so
 * a captured step follows a seed kept on the device (train.py:87 seeds every rank with seed + rank).  step_dev may be NULL
 * (step 0, key = seed argument).
 * MSM masks, dalle_bert.py:992-1029: strategy_prob[4] = Bernoulli(p ~ U(bern_lo, bern_hi)) | fully masked | RandomErasing
 * box hidden | only the box visible; pc_prob: frame preservation (1022-1026).  mask1 [B, T*f*f] (1 = token visible),
 * not_fully_masked [B]; strategy_out (optional) [B] the strategy drawn (tests). */
int mmvid_msm_masks(uint64_t seed, const float* step_dev, int B, int T, int f, const float* strategy_prob, float bern_lo,
                    float bern_hi, float pc_prob, uint8_t* mask1, float* not_fully_masked, int32_t* strategy_out,
                    void* stream);
/* The same kernel with the decisions supplied instead of drawn (parity tests against a reference run, tests/golden/
 * frontend.npz): decisions [B, 72] int32 = {strategy 1..4, has_box, box i, j, h, w, 2 reserved, keep_frame[64]}; bernoulli
 * [B, T*f*f] uint8 = the Bernoulli field of strategy 1 (may be NULL when no sample uses it). */
int mmvid_msm_masks_inject(const int32_t* decisions, const uint8_t* bernoulli, int B, int T, int f, uint8_t* mask1,
                           float* not_fully_masked, void* stream);
/* VID negative, dalle_bert.py:204-238 (+93-202): x, out [B,T,C,H,W] fp32 in [0,1]; strategy_prob[4] = frame of another
 * sample | frame shuffle | colour shift | affine warp (affine_grid + bilinear grid_sample, reflection padding).
 * params_scratch: B * mmvid_warp_params_bytes() bytes; draw_params = 0 applies the parameters already in it (tests). */
int mmvid_warp_params_bytes(void);
int mmvid_vid_warp(uint64_t seed, const float* step_dev, const float* x, int B, int T, int C, int H, int W,
                   const float* strategy_prob, void* params_scratch, int draw_params, float* out, void* stream);
/* The same negative without re-encoding what did not change: the VQGAN tokenises frames independently and the warp
 * changes the pixels of at most one frame per sample, so only that frame (new_frames [B,C,H,W]) goes through the
 * encoder again; mmvid_vid_warp_tokens then assembles the negative's tokens [B, T*n] from the target's tokens
 * [B, T*n], the new frames' tokens [B, n] and the drawn parameters (frame permutation, frame of another sample,
 * replaced frame).  Bit-identical to tokenising mmvid_vid_warp's output. */
int mmvid_vid_warp_new_frames(uint64_t seed, const float* step_dev, const float* x, int B, int T, int C, int H, int W,
                              const float* strategy_prob, void* params_scratch, int draw_params, float* new_frames,
                              void* stream);
int mmvid_vid_warp_tokens(const int64_t* target_tok, const int64_t* new_frame_tok, const void* params, int B, int T, int n,
                          int64_t* out, void* stream);
/* visual-token erasing on token maps tok [B, Tv, f, f] int64, in place.  erase_codebook_face (dalle_bert.py:796-848):
 * one of `nchoice` (<= 4) alternatives is drawn per call from the cumulative probabilities; modes[i] 0 = untouched,
 * 1 = keep only boxes[i] = (r0, r1, c0, c1), 2 = erase the box; frame0_full leaves frame 0 untouched. */
int mmvid_erase_tokens_choice(uint64_t seed, const float* step_dev, int nchoice, const float* cumprob, const int32_t* modes,
                              const int32_t* boxes, int frame0_full, int B, int Tv, int f, int64_t value, int64_t* tok,
                              void* stream);
/* random_erase_codebook (dalle_bert.py:779-794): per sample torchvision-style RandomErasing(p, scale, ratio) box set to
 * `value` on every frame, or (erase_half) the lower half of every frame. */
int mmvid_random_erase_tokens(uint64_t seed, const float* step_dev, int B, int Tv, int f, float p, float scale_lo,
                              float scale_hi, float ratio_lo, float ratio_hi, int erase_half, int64_t value, int64_t* tok,
                              void* stream);

/* visual_aug_mode 'motion_color' (dalle_bert.py:140-158, 940-943; dalle_artv.py:460-463): with probability p (one draw per
 * call) every sample's frames first_frame.. of x [B,Tv,C,H,W] fp32 get a per-sample colour shift U(-0.5,0.5) on all channels
 * or one of them, clamped to [0,1]; in place.  params_out (optional) [B,3] = {gate, shift, channel choice 0..3}. */
int mmvid_visual_color_jitter(uint64_t seed, const float* step_dev, float* x, int B, int Tv, int C, int H, int W, float p,
                              int first_frame, float* params_out, void* stream);

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
/* g: dL/dx_out on entry, dL/dx_in on exit (in place, fp32 [B*L, E]).  Each layer's slice of the saved arena ends in room for that
 * layer's four dY tensors (bf16): the backward writes them there (the forward part of the arena is only read) and computes the weight
 * gradients of all `cfg->layers` layers of this call in one launch after the layer loop (mmvid_gemm_bf16_dw_multi). */
int mmvid_tower_backward(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, float* g,
                         void* saved, void* scratch, void* stream);


/* ---- incremental (KV-cache) decoding of the causal tower: what dalle_artv.py:236-304 recomputes from scratch for
 * every sampled token (SURVEY next-row N1).  kv_cache: [layers][B][Lmax][2E] bf16, a row = K then V of one position.
 * prefill = the causal forward over the prompt (cfg->L positions) that also fills the cache; decode = one new
 * position per sequence (x_in / x_out [B, E] fp32) at index *pos_dev (device scalar: the call can be captured and
 * replayed for every position) or `pos` when pos_dev is NULL.  Needs cfg->mask_mode == 1; scratch as for the forward. */
int mmvid_tower_prefill(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                        float* x_out, void* kv_cache, int Lmax, void* scratch, void* stream);
int mmvid_tower_decode(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in, float* x_out,
                       void* kv_cache, int Lmax, const int32_t* pos_dev, int pos, void* scratch, void* stream);
/* The same step as five matrix-vector launches per layer (weights streamed once, rows in LDS, 256 CUs busy) instead of
 * the M = B corner of the training GEMM: ~10x less time per token.  B <= 64 (widths 512 / 768; other towers: B <= 16).  scratch: B * (7E + F) floats.
 * mmvid_gemv_rows is the building block (y = act(LN?(x) W^T + b) (+ residual), x / y fp32 [NB, *], W bf16 [N, K]);
 * mmvid_decode_embed writes the embedding row of the token just sampled (table[tok] + pos_rows[*pos_dev + pos_off]). */
int mmvid_tower_decode_fused(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                             float* x_out, void* kv_cache, int Lmax, const int32_t* pos_dev, int pos, void* scratch,
                             void* stream);
/* ... for cfg->B consecutive sequences of a cache of cache_batch sequences (kv_cache = the first of them in layer 0; x_in / x_out = their
 * rows): how batches above 64 run, as slices of 64.  Round 6: with 3..64 sequences the four linear layers run on the matrix pipe
 * (csrc/decode.hip::gemv16_mfma_kernel: 16 rows = one v_mfma_f32_16x16x32_bf16 row block, up to four of them per wave against the same
 * weight fragments -- the weights are streamed once per pass --, K split over the block's eight waves; the attention output and the
 * activation travel as bf16); 1-2 sequences keep the vector-ALU
 * kernels (the form the persistent step falls back to).  advance_pos != 0: *pos_dev += 1 when the step is done (by the last layer's
 * last launch: no separate launch per token). */
int mmvid_tower_decode_fused_slice(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in,
                                   float* x_out, void* kv_cache, int Lmax, int cache_batch, int32_t* pos_dev, int pos,
                                   int advance_pos, void* scratch, void* stream);
/* The same step as ONE launch: 256 co-resident blocks walk the 60 phases and hand values to each other as tagged 8-byte words that
 * the consumers poll (csrc/decode_persistent.hip) -- no launch boundary, no barrier.  _supported: the 768 / 3072 / 12-head causal tower,
 * <= 12 layers, B <= 2, Lmax <= 4096, a device with >= 256 CUs and nothing else running beside the step.
 * workspace: _workspace_bytes(B) bytes, ZERO before the first call, then owned by the session (it carries the step counter the tags are
 * made from; word 1 becomes non-zero if a poll ever timed out, i.e. the blocks were not resident together).  advance_pos != 0: the
 * step also increments *pos_dev when it is done (a launch less per token for the sampler). */
int mmvid_tower_decode_persistent_supported(const mmvid_tower_cfg_t* cfg, int Lmax);
int64_t mmvid_tower_decode_persistent_workspace_bytes(int B);
int mmvid_tower_decode_persistent(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const float* x_in, float* x_out,
                                  void* kv_cache, int Lmax, int32_t* pos_dev, int pos, int advance_pos, void* workspace, void* stream);
/* The sampler's whole token as ONE launch: embedding row of the token drawn last (table[tok] + pos_rows[*pos_dev + pos_off], dalle_artv.py:484-491)
 * -> the persistent tower step -> LN + the image block of to_logits -> the draw of the next token (csrc/sample.hip's rule on the pre-drawn
 * variates E [draws][B][V], draw number *pos_dev + 1 - e_pos0) -> *pos_dev += 1.  tok [B] is read at the start and overwritten at the end;
 * record (optional) [B][record_ld]: record[b][*pos_dev - record_pos0] = the token embedded.  V = 1,024 or 2,048. */
typedef struct {
    int64_t* tok;
    const float* table;
    int64_t table_rows;
    const float* pos_rows;
    int32_t pos_off;
    int32_t record_pos0;
    int64_t* record;
    int64_t record_ld;
    const float *lnf_w, *lnf_b;
    const void* head_w; /* bf16 [V][E] */
    const float* head_b;
    const float* E;
    int64_t e_step_stride;
    int64_t tok_offset;
    float* logits_out; /* [B][V] or NULL */
    int32_t V, e_pos0;
    float lnf_eps, temperature;
} mmvid_decode_token_t;
int mmvid_artv_token_step_persistent(const mmvid_tower_cfg_t* cfg, const mmvid_tower_layer_t* layers, const mmvid_decode_token_t* t,
                                     float* x_out, void* kv_cache, int Lmax, int32_t* pos_dev, void* workspace, void* stream);
int mmvid_gemv_rows(const float* x, int64_t ldx, int NB, int K, const float* ln_w, const float* ln_b, float eps, const void* W,
                    const float* bias, int N, int act, const float* residual, int64_t ldr, int round_in, int round_out,
                    float* out, int64_t ldo, void* stream);
int mmvid_decode_embed(const int64_t* tok, const float* table, int64_t table_rows, const float* pos_rows,
                       const int32_t* pos_dev, int pos_off, int B, int E, float* x, void* stream);
/* ... that also files the token: record[b][*pos_dev - record_pos0] = tok[b] (int64 [B][record_ld]; null = mmvid_decode_embed) */
int mmvid_decode_embed_record(const int64_t* tok, const float* table, int64_t table_rows, const float* pos_rows,
                              const int32_t* pos_dev, int pos_off, int B, int E, float* x, int64_t* record, int64_t record_ld,
                              int record_pos0, void* stream);
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
 * [N][Hout*Wout/128][32][sum, sumsq] -- the partial-sum area of mmvid_groupnorm_swish_nhwc's stats_scratch.
 * The full-size encoder's first layer (model.py:382-386: mode 0, Cin = 8 (3 stored as 8), Cout = 128, Win = 128, even Hin, no residual,
 * one output precision) runs a kernel of its own built around its stores (csrc/conv.hip::conv_in_kernel, round 6); the choice is
 * made from the layer's geometry only, like every kernel choice of the encoder (a frame's tokens may not depend on its batch). */
int mmvid_conv2d_nhwc(int mode, const void* x, int N, int Hin, int Win, int Cin, const void* w, const float* bias,
                      int Cout, const void* residual_bf16, const float* residual_f32, int clamp01, void* out_bf16,
                      float* out_f32, float* gn_partial, void* stream);
/* The same convolution with the reduction (taps x input channels) cut into `splitk` ranges computed by separate blocks and
 * added in a fixed order (deterministic): for deep layers on small maps whose output tiles alone cannot fill the chip.
 * workspace: fp32 [splitk][N*Hout*Wout][Cout]; gn_partial must be null when splitk > 1. */
int mmvid_conv2d_nhwc_splitk(int mode, const void* x, int N, int Hin, int Win, int Cin, const void* w, const float* bias,
                             int Cout, const void* residual_bf16, const float* residual_f32, int clamp01, void* out_bf16,
                             float* out_f32, float* gn_partial, int splitk, float* workspace, void* stream);
/* The ResnetBlock convolutions (mode 0: 3x3, stride 1, pad 1; model.py:102-115) at 32x32 and above in "strip" form
 * (csrc/conv_strip.hip): a K tile is (kernel row, 32 input channels), the input strip is staged once for the three kx
 * taps, the block is 512 pixels x 128 channels.  Same contract as mmvid_conv2d_nhwc(mode 0) except that the GroupNorm
 * partial sums are per 64-pixel block: gn_partial64 [N][H*W/64][32][2].  mmvid_conv3x3_strip_supported: geometry test
 * (W a power of two in 8..128, Cin a power of two >= 32, Cout % 128 == 0, H*W >= 1024); independent of N by design. */
int mmvid_conv3x3_strip_supported(int H, int W, int Cin, int Cout);
int mmvid_conv3x3_strip_nhwc(const void* x, int N, int H, int W, int Cin, const void* w, const float* bias, int Cout,
                             const void* residual_bf16, const float* residual_f32, void* out_bf16, float* out_f32,
                             float* gn_partial64, void* stream);
/* img NCHW fp32 [N,3,H,W] in [0,1] -> NHWC bf16 [N,H,W,8] of 2x-1 (vae.py:41), channels 3..7 zero. */
int mmvid_image_to_nhwc8(const float* img, int N, int H, int W, void* out_bf16, void* stream);
/* NHWC fp32 [N,H,W,C] -> NCHW fp32 (first Cuse channels). */
int mmvid_nhwc_to_nchw_f32(const float* x, int N, int H, int W, int C, int Cuse, float* out, void* stream);
/* single-head spatial attention of AttnBlock (model.py:180-205): q,k,v NHWC bf16 [N, HW, C] -> o bf16. */
int mmvid_spatial_attention(const void* q, const void* k, const void* v, int N, int HW, int C, float scale,
                            float* scores_scratch, void* out_bf16, void* stream);
/* The same with q, k, v rows `ld` elements apart: ld = 3C when they are the column blocks of one fused q|k|v 1x1 convolution
 * (model.py:159-178 computes the three with separate convs of the same input). */
int mmvid_spatial_attention_ld(const void* q, const void* k, const void* v, int64_t ld, int N, int HW, int C, float scale,
                               float* scores_scratch, void* out_bf16, void* stream);

/* ---- device-side samplers (csrc/sample.hip): BERT mask-predict, dalle_bert.py:514-714, and the ART-V token draw,
 * dalle_artv.py:61-67,274-281.  Randomness enters as tensors of Exp(1) variates E (what torch.multinomial draws
 * internally): tok = first argmin_c E[c] / P[c] with P = exp(x - max x), x = logits / logit_div (+ temperature *
 * Gumbel(noise_u) when noise_u != NULL, dalle_bert.py:527-538); y (optional) = softmax probability of the drawn token. */
int mmvid_sample_race(const float* logits, int64_t ld, const float* E, const float* noise_u, float temperature,
                      float logit_div, int64_t R, int V, int64_t tok_offset, int64_t* tok, float* y, void* stream);
/* ... with the variates of draw number (*step_dev - step0) of a pre-drawn block E [draws][R][V] (e_step_stride = R * V): the whole
 * sampling loop's variates are drawn once, outside the captured per-token step.  step_dev needs y == NULL, noise_u == NULL, R <= 1024. */
int mmvid_sample_race_at(const float* logits, int64_t ld, const float* E, const int32_t* step_dev, int step0, int64_t e_step_stride,
                         const float* noise_u, float temperature, float logit_div, int64_t R, int V, int64_t tok_offset, int64_t* tok,
                         float* y, void* stream);
/* keep-mask of a refinement step (dalle_bert.py:646-668): of the positions with preserve == 0, the k with the smallest
 * E / Y stay visible (k outside [1, #non-zero weights] -> 1, the reference's except branch); preserved positions always
 * stay.  Y [b, TS], E [b, Bm, TS], mask1 out [b, Bm, TS] (1 = keep). */
int mmvid_mp_select_keep(const float* Y, const float* E, const uint8_t* preserve, int b, int Bm, int TS, int k,
                         uint8_t* mask1, void* stream);
/* tower input of the b*Bm candidates: control_emb [b, csl, E] broadcast per candidate, then image_emb[id] + tpos with
 * id = mask1 ? (mask1[.] ? I_tok : mask_id) : I_tok.  out [b*Bm, csl+TS, E]. */
int mmvid_mp_build_input(const float* control_emb, const float* image_emb, int64_t table_rows, const float* tpos,
                         const int64_t* I_tok, const uint8_t* mask1, int b, int Bm, int csl, int TS, int E, int64_t mask_id,
                         float* out, void* stream);
/* per video: S_j = (sigmoid(rel_j) + sigmoid(vid_j)) / 2, jmax = first argmax, the sequential where-chain over
 * candidates 0..jmax (dalle_bert.py:675-692), dynamic stop after 5 steps without a better score (701-707).
 * State (updated in place, only where active[i]): Y, I_tok [b, TS], Imax [b, TS], Smax [b], tmax [b], active [b]. */
int mmvid_mp_update(const uint8_t* mask1, const float* Ynew, const int64_t* Inew, const float* rel_logit,
                    const float* vid_logit, int b, int Bm, int TS, int t, int dynamic, float* Y, int64_t* I_tok,
                    int64_t* Imax, float* Smax, int32_t* tmax, uint8_t* active, float* S_out, int32_t* jmax_out,
                    void* stream);
/* to_logits_rel / to_logits_vid (LayerNorm + Linear(dim,1), dalle_bert.py:418-425) on R gathered rows x[rows[r]] and
 * binary_cross_entropy_with_logits (1067-1084, 1107-1123): loss = sum_r row_weight[r] * bce(z_r, label_r) / den with
 * den = den_from ? max(1, sum den_from[0..nden)) : den_const.  label == NULL: logits only (sampling).  The backward
 * accumulates (+=) dw [E], db [1], dln_w, dln_b [E] and adds each row's input gradient into dx[rows[r]]. */
int mmvid_head_bce_fwd(const float* x, int64_t ldx, const int64_t* rows, int R, int E, const float* ln_w, const float* ln_b,
                       float eps, const float* w, const float* b, const float* label, const float* row_weight,
                       const float* den_from, int nden, float den_const, float* z, float* mean, float* rstd, float* loss,
                       void* stream);
int mmvid_head_bce_bwd(const float* x, int64_t ldx, const int64_t* rows, int R, int E, const float* ln_w, const float* ln_b,
                       const float* w, const float* z, const float* mean, const float* rstd, const float* label,
                       const float* row_weight, const float* den_from, int nden, float den_const, const float* gloss,
                       float* dx, int64_t lddx, float* dw, float* db, float* dln_w, float* dln_b, void* stream);
/* every token id of a BERT training step in one launch (dalle_bert.py:903-973, 1030-1035, 1057, 1094-1100):
 * ids [(1+rel+vid)*B, L] = MSM | REL negative (control of sample (b+B/2)%B, or text_neg) | VID negative (target_warp);
 * select_full [B*L] / target_full [B*L] = rows and labels of the MSM cross entropy; select_count = #selected. */
int mmvid_bert_build_ids(const int64_t* text, const int64_t* text_neg, const int64_t* visual_tok, const int64_t* target,
                         const int64_t* target_warp, const uint8_t* mask1, int B, int Ttxt, int Nvis, int TS,
                         int64_t pad_base, int64_t mask_id, int has_rel, int has_vid, int64_t* ids, uint8_t* select_full,
                         int64_t* target_full, float* select_count, void* stream);

/* ---- strict-parity (fp32-accurate) VQGAN operators: `VQGanVAE1024.strict = True` (csrc/strict.hip).  Same reference
 * lines as the bf16 entry points above, but every product and accumulate is fp32: convolutions and matmuls are
 * k-ordered fmaf chains on the f32-input matrix instruction, GroupNorm statistics are fp64, exp is expf.
 * Used where token indices must equal the reference's exactly (north star: "token-index bit-exact"). */
/* C[m][n] = alpha * sum_k A[m][k] B(n,k) (+bias[n]) (+residual[m][n]); B row-major [N][ldb] or k-major [K][ldb]. */
int mmvid_gemm_f32(int b_kmajor, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                   int batch, int64_t strideA, int64_t strideB, int64_t strideC, float alpha, const float* bias,
                   const float* residual, float* C, int64_t ldc, void* stream);
/* modes as mmvid_conv2d_nhwc; x [N,Hin,Win,Cin] fp32 (Cin a power of two >= 4), w [Cout][taps][Cin] fp32. */
int mmvid_conv2d_nhwc_f32(int mode, const float* x, int N, int Hin, int Win, int Cin, const float* w,
                          const float* bias, int Cout, const float* residual, int clamp01, float* out, void* stream);
int mmvid_image_to_nhwc4_f32(const float* img, int N, int H, int W, float* out, void* stream);
/* stats_scratch: fp32 [N][C][2] (mean, rstd per channel). */
int mmvid_groupnorm_swish_nhwc_f32(const float* x, int N, int64_t hw, int C, const float* w, const float* b, float eps,
                                   int swish, float* stats_scratch, float* y, void* stream);
/* scratch: 2*N*HW*HW floats. */
int mmvid_spatial_attention_f32(const float* q, const float* k, const float* v, int N, int HW, int C, float scale,
                                float* scratch, float* out, void* stream);

/* ---- the "split" VQGAN operators: `VQGanVAE1024.strict = 'split'` (csrc/conv.hip, conv_strip.hip, norm.hip).  The middle
 * path between the bf16 operator and the fp32 one: every activation and weight that enters a convolution is a bf16 PAIR
 * (x = x_hi + x_lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits) and a convolution is the three products
 * x_hi.w_hi + x_lo.w_hi + x_hi.w_lo accumulated in fp32 inside ONE K loop of three times the length on the bf16 matrix pipe
 * (the dropped x_lo.w_lo term is 2^-18 relative).  Residual stream, GroupNorm (fp64 finalisation, expf swish), spatial
 * attention (mmvid_spatial_attention_f32) and the VQ argmin stay fp32.  Same reference lines as the bf16 entry points.
 * x_planes: [2][N,Hin,Win,Cin] bf16 (hi plane, lo plane); w3: [Cout][3][taps][Cin] bf16 = (w_hi | w_hi | w_lo). */
int mmvid_conv2d_nhwc_split3(int mode, const void* x_planes, int N, int Hin, int Win, int Cin, const void* w3, const float* bias,
                             int Cout, const float* residual_f32, int clamp01, float* out_f32, float* gn_partial, int splitk,
                             float* workspace, void* stream);
/* out_planes (optional; with it out_f32 may be null): the result also / only as a bf16 pair [2][N,H,W,Cout] -- what
 * mmvid_split_f32_bf16x2 would make of out_f32 -- for a result whose only reader is another pair-operator convolution (the last
 * convolution of an encoder level in front of its Downsample: no fp32 store and no cast pass, 8 bytes per element less traffic). */
int mmvid_conv3x3_strip_nhwc_split3(const void* x_planes, int N, int H, int W, int Cin, const void* w3, const float* bias,
                                    int Cout, const float* residual_f32, float* out_f32, float* gn_partial64, void* out_planes,
                                    void* stream);
/* fp32 [n] -> bf16 pair planes [2][n] (n % 8 == 0). */
int mmvid_split_f32_bf16x2(const float* x, int64_t n, void* planes_bf16, void* stream);
/* img NCHW fp32 [N,3,H,W] in [0,1] -> bf16 pair planes [2][N,H,W,8] of 2x-1 (vae.py:41). */
int mmvid_image_to_nhwc8_split(const float* img, int N, int H, int W, void* planes_bf16, void* stream);
/* GroupNorm(32) [+ swish] (model.py:38-42): x fp32 NHWC -> bf16 pair planes [2][N,hw,C].  partial_blocks as in
 * mmvid_groupnorm_swish_nhwc (0: a statistics pass runs here).  stats_scratch: fp32 [N*(2*C + 64*ceil(hw/64))]. */
int mmvid_groupnorm_swish_nhwc_split(const float* x, int N, int64_t hw, int C, const float* w, const float* b, float eps,
                                     int swish, float* stats_scratch, int partial_blocks, void* planes_bf16, void* stream);
/* ---- the fp16 layers of the exact-index mode `vae.strict = 'mixed'` (mmvid_amd/vae.py; the same reference lines).  The per-layer
 * sweep (profiles/r05_exact_index_layer_sensitivity_sweep.log) shows where the pair operator's 2^-17 per operand is needed: one layer
 * in plain bf16 (2^-9) raises the z error 50-100x, one layer with IEEE-half operands (2^-12; the matrix pipe's f16 rate equals its
 * bf16 rate) 6-18x.  The 3x3 residual-block convolutions of the 128x128, 64x64 and 32x32 levels (82 % of the encoder's multiply-adds)
 * run as ONE product of fp16 operands with fp32 accumulation and everything else stays the pair operator: the reference's top-2
 * distance gap stays above 8x the error of that gap on every golden frame (the acceptance test of the split mode) at 1.35x instead of
 * 3x the plain bf16 work.
 * x_f16 [N,H,W,Cin] IEEE half, w_f16 [Cout][9][Cin] IEEE half; residual / output fp32; gn_partial64 as in the strip kernel. */
int mmvid_conv3x3_strip_nhwc_f16(const void* x_f16, int N, int H, int W, int Cin, const void* w_f16, const float* bias, int Cout,
                                 const float* residual_f32, float* out_f32, float* gn_partial64, void* out_planes, void* stream);
/* mmvid_groupnorm_swish_nhwc_split with the result stored as one plane of IEEE-half values [N,hw,C] */
int mmvid_groupnorm_swish_nhwc_f16out(const float* x, int N, int64_t hw, int C, const float* w, const float* b, float eps,
                                      int swish, float* stats_scratch, int partial_blocks, void* y_f16, void* stream);

/* ---- native op-list executor for the VQGAN encoder / decoder (model.py:439-466, 551-582; vae.py:38-56): the host
 * plans the op sequence once per input shape, every call is then one host->native transition.  Offsets are bytes
 * into `arena` (-1 = unused); w/b/ext_* are device pointers. */
enum {
    MMVID_VQOP_IMG2NHWC8 = 0, /* ext_in img [N,3,H,W] f32 -> out_bf16 [N,H,W,8]                                  */
    MMVID_VQOP_CONV = 1,      /* in0 x [N,H,W,C] bf16, w, b, Cout, mode; in1 residual (flags&1: f32); flags&2 clamp01;
                                 flags&4: write GroupNorm partial sums of the output into `scratch` (a GN stats area);
                                 flags&8: the strip kernel (mmvid_conv3x3_strip_nhwc; partial sums per 64 pixels);
                                 flags&32: split-K by 4 through the fp32 workspace at `scratch` (mmvid_conv2d_nhwc_splitk)   */
    MMVID_VQOP_GROUPNORM = 2, /* in0 [N,H,W,C] (flags&1: f32), w, b, eps, mode = swish, scratch = stats;
                                 flags&2: the partial sums in `scratch` were written by the producing CONV (flags&8: per
                                 64-pixel block instead of per 128)                                                  */
    MMVID_VQOP_CAST = 3,      /* in0 f32 -> out_bf16, N*H*W*C elements                                           */
    MMVID_VQOP_SPATIAL_ATTN = 4, /* in0,in1,in2 = q,k,v [N,H*W,C] bf16, eps = scale, scratch                        */
    MMVID_VQOP_VQ_ARGMIN = 5, /* in0 z [N*H*W, C] f32, w = codebook [Cout, C], b = ee -> ext_out int64            */
    MMVID_VQOP_GATHER = 6,    /* ext_in idx int64 [N*H*W], w = table [Cout, C] f32 -> out_bf16 [N,H,W,C]           */
    MMVID_VQOP_NHWC2NCHW = 7, /* in0 [N,H,W,C] f32 -> ext_out [N,Cout,H,W] f32                                    */
    MMVID_VQOP_EXT_CAST = 8   /* ext_in f32 [N*H*W*C] -> out_bf16 (or out_f32 copy when strict): decode_train's z     */
};
/* flags & 16 (IMG2NHWC8, CONV, GROUPNORM, SPATIAL_ATTN, GATHER, EXT_CAST): the strict fp32 operator; every arena
 * tensor of such a plan is fp32 (out_f32 / in* are fp32), w is fp32 [Cout][taps][Cin]. */
#define MMVID_VQFLAG_STRICT 16
/* flags & 64 (IMG2NHWC8, CONV, GROUPNORM, CAST): the split operator.  IMG2NHWC8 / GROUPNORM / CAST write bf16 pair planes at
 * out_bf16 (in0 fp32); CONV reads pair planes at in0, w = w3, residual in1 fp32, writes out_f32 (flags&2 clamp01, flags&8 strip
 * form, flags&32 split-K by 4 through `scratch`, flags&4 GroupNorm partial sums of the output into the stats area `scratch`);
 * GROUPNORM flags&2 / flags&8 as for the bf16 operator (partial sums written by the producing CONV). */
#define MMVID_VQFLAG_SPLIT 64
/* flags & 128, together with SPLIT: GROUPNORM writes ONE fp16 plane at out_bf16 (mmvid_groupnorm_swish_nhwc_f16out); CONV (strip form,
 * flags&8) reads such a plane at in0 with w = fp16 [Cout][9][Cin] (mmvid_conv3x3_strip_nhwc_f16).  A split CONV in strip form with
 * out_bf16 >= 0 also / only (out_f32 < 0) writes its result as pair planes there. */
#define MMVID_VQFLAG_F16 128
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

/* hardware probe (tools/gpu_probe.py): which = 0 -> ds_read_b64_tr_b16 lane layout | 1 -> LDS-DMA through a buffer descriptor |
 * 2 -> store-pattern bandwidth | 3 -> the DPP / permlane wave reductions (in float[64] -> out float[3][64]) | 4 -> LDS read patterns |
 * 5 -> register-only MFMA chains: the matrix pipe's sustained (power-limited) ceiling, in = HOST int32[3] {iterations, blocks, operand mode} |
 * 6 -> a read stream of known size (LDS-DMA or global_load) for calibrating FETCH_SIZE, in = HOST int64[2] {bytes, mode}, out = the buffer. */
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

/* ---- run-time options; each is also read from an environment variable at first use.  "graphs" (MMVID_GRAPHS): 1 = library-level
 * graph replay (default 0).  Unknown names return MMVID_ERR_ARG. */
int mmvid_set_option(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
