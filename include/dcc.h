/*
 * dcc.h — C ABI of libdcc_hip.so, the MI355X (gfx950) kernel library under the DeepCORO_CLIP
 * video-text contrastive training path.
 *
 * The reference (HeartWise-AI/DeepCORO_CLIP) has no native boundary: its hot path is stock
 * PyTorch ops called from Python.  Each entry point below replaces the ATen op(s) cited next to
 * it; the Python modules in deepcoro_clip_amd/ (same class names and signatures as the reference's
 * models/, utils/loss/, runners/) are the only callers, via ctypes (deepcoro_clip_amd/_lib.py parses
 * THIS header to build the argtypes, so header and binding cannot drift).
 *
 * Conventions
 *   - every function returns 0 (DCC_OK) or a negative error code; no exceptions cross the ABI
 *   - pointers are device pointers unless noted; the library never allocates or frees caller memory
 *   - `stream` is a hipStream_t; all work is stream-ordered and re-entrant, no global state
 *   - dtype codes: 0 = fp32, 1 = bf16.  Accumulation / statistics are always fp32
 *   - strides and leading dimensions are in ELEMENTS; innermost dimension is contiguous
 *   - "accumulated into" outputs are fp32 and must be zeroed (or hold a running sum) by the caller
 */
#ifndef DCC_H
#define DCC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCC_OK 0
#define DCC_ERR_ARG (-1)
#define DCC_ERR_LAUNCH (-2)
#define DCC_ERR_UNSUPPORTED (-3)
#define DCC_F32 0
#define DCC_BF16 1
#define DCC_F16 2 /* dcc_cast only (f32 <-> f16): wire format of SiglipLossDDP's gather, utils/loss/losses.py:243-248 */

/* GEMM epilogue flags */
#define DCC_GEMM_BIAS 1
#define DCC_GEMM_GELU 2
#define DCC_GEMM_SAVE_PRE 4
#define DCC_GEMM_RESID 8
#define DCC_GEMM_DGELU 16
#define DCC_GEMM_ACCUM 32
#define DCC_GEMM_COLSUM_A 64 /* colsum[m] += sum_k A(m,k): bias gradient fused into the k-major weight-gradient GEMM */
#define DCC_OPT_CHUNK 4096
#define DCC_DWCONV_WS_BLOCKS 1024 /* upper bound of dcc_dwconv_bwd_weight's grid (its `workspace` argument is unused since r03) */

int dcc_version(void);

/* ---- MFMA GEMM (csrc/gemm.hip) --------------------------------------------------------------
 * C[m][n] = alpha * sum_k A(m,k) B(n,k) (+bias[n]) (GELU | * GELU'(aux)) (+R[m][n])
 * A(m,k) = a_km ? A[k*lda+m] : A[m*lda+k];  B(n,k) = b_km ? B[k*ldb+n] : B[n*ldb+k].
 * Replaces nn.Linear forward/backward everywhere on the path: MViT qkv/project/mlp (torchvision
 * mvit.py via models/video_encoder.py:232-253,285,376), proj heads (models/video_encoder.py:175-180,
 * models/text_encoder.py:63-68), BERT dense layers (models/text_encoder.py:131), nn.MultiheadAttention
 * in/out projections (models/attention_pool.py:45-50, models/video_aggregator.py:16-21), the patch-embed
 * conv as im2col GEMM (models/video_encoder.py:237) and the similarity matmuls of the losses
 * (utils/loss/losses.py:50, utils/loss/contrastive.py:150,263).
 * batch > 1 walks sA/sB/sC; splitk > 1 or DCC_GEMM_ACCUM atomically accumulates into fp32 C. */
int dcc_gemm(const void* A, const void* B, void* C, const float* bias, const void* R, void* aux, float* colsum,
             int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldaux, int M, int N, int K, int batch,
             int64_t sA, int64_t sB, int64_t sC, int splitk, float alpha, int flags, int a_km, int b_km,
             int in_dtype, int out_dtype, void* stream);

/* ---- fp8 GEMMs (csrc/gemm_ring.hip F8 variant, csrc/elementwise.hip quantisers): BASELINE configs[4] -----------------
 * C[m][n] = alpha * (*scale_a) * (*scale_b) * sum_k A8(m,k) B8(n,k) with the epilogues of dcc_gemm, bf16 output, computed
 * by the block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 (unit block scales; per-tensor scales in alpha).  A8 [M,K] and
 * B8 [N,K] are k-contiguous fp8 bytes (fmt 0 = OCP e4m3, 1 = e5m2), K % 16 == 0, K > 128.  Replaces the same torch
 * Linear layers as dcc_gemm when the encoder runs its forward / data-gradient GEMMs in fp8. */
int dcc_gemm_fp8(const void* A8, const void* B8, void* C, const float* bias, const void* R, void* aux,
                 const float* scale_a, const float* scale_b, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr,
                 int64_t ldaux, int M, int N, int K, float alpha, int flags, int fmt_a, int fmt_b, void* stream);
/* amax[0] = max(amax[0], max |x|) over a [rows, cols] view (row pitch ldx); amax zero-initialised by the caller. */
int dcc_absmax(const void* x, int64_t rows, int64_t cols, int64_t ldx, int dtype, float* amax, void* stream);
/* out8 = fp8(x * FMAX / *amax) (transpose != 0: out8[c][r]); scale_out[0] = *amax / FMAX. */
int dcc_fp8_quantize(const void* x, int64_t rows, int64_t cols, int64_t ldx, int dtype, const float* amax, void* out8,
                     int64_t ldo, int transpose, int fmt, float* scale_out, void* stream);

/* ---- flash attention (csrc/attention.hip) ---------------------------------------------------
 * O = softmax(scale * Qf Kf^T + kbias) V (+ Qres[q][0:DV] for q >= res_start); LSE[B,H,Nq] kept.
 * Operands are [B, N, H, D] views; strides = {q_b,q_n,q_h, k_b,k_n,k_h, v_b,v_n,v_h, o_b,o_n,o_h}.
 * Replaces q@k^T*scale, _add_rel_pos, softmax, attn@v and the residual-pool add of torchvision's
 * MultiscaleAttention / the RoPE-patched forward (models/video_encoder.py:348-378), BERT SDPA
 * (models/text_encoder.py:131) and nn.MultiheadAttention's core (models/attention_pool.py:86-91,
 * models/video_aggregator.py:40-45).  (DQK,DV) in {(64,64),(96,96),(128,96),(160,96),(128,128)}.
 * drop_p > 0: attention-probability dropout, O = (P * keep / (1 - drop_p)) V with keep a counter hash of
 * (drop_seed, *drop_step (nullable device counter), b, h, q, k) — nn.MultiheadAttention(dropout=) in
 * models/attention_pool.py:45-50 and models/video_aggregator.py:16-21, HF BERT attention_probs_dropout_prob
 * (models/text_encoder.py:52); head shapes (64,64) and (128,128) only.  The backward regenerates the mask from the
 * same three values. */
int dcc_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, const float* kbias,
                 const void* Qres, int res_start, const int64_t* strides, int B, int H, int Nq, int Nk, int DQK,
                 int DV, float scale, float drop_p, uint64_t drop_seed, const int64_t* drop_step, int dtype,
                 void* stream);
/* strides = 21 entries: q,k,v,o,dq,dk,dv x (b,n,h).  Dv: workspace [B,H,Nq] fp32.
 * dK/dV element type: input dtype when qsplit == 1, else fp32 pre-zeroed (atomic accumulation).
 * With Qres != NULL the residual shortcut's gradient (dO for q >= res_start) is added into dQ[:, 0:DV]. */
int dcc_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                 float* Dv, const float* kbias, const void* Qres, int res_start, void* dQ, void* dK, void* dV,
                 const int64_t* strides, int B, int H, int Nq, int Nk, int DQK, int DV, float scale, int qsplit,
                 float drop_p, uint64_t drop_seed, const int64_t* drop_step, int dtype, void* stream);

/* ---- relative-position bias as contraction (csrc/relpos.hip) -------------------------------
 * torchvision mvit._add_rel_pos (call site models/video_encoder.py:354-364).  The inner products q . R are one
 * dcc_gemm against the three tables stacked ([L, D]); these kernels move its entries into / out of the
 * augmentation columns of the attention operands.
 * dims = {B,H,D,A,L,qt,qh,qw,kt,kh,kw,Lh,Lw,Lt,has_cls}; strides = {b,token,head} of the [B,N,H,D+A] operand;
 * P / dP: dense [B*N*H, L]; idx_a: int32 [q_a][k_a] table row of (coord, key coord); inv_a: int32 [q_a][L_a]. */
int dcc_relpos_gather(void* Qf, const void* P, const int* idx_h, const int* idx_w, const int* idx_t, const int* dims,
                      const int64_t* strides, float scale, int dtype, void* stream);
int dcc_relpos_scatter(const void* dQf, void* dP, const int* inv_h, const int* inv_w, const int* inv_t,
                       const int* dims, const int64_t* strides, float scale, int dtype, void* stream);
int dcc_relpos_kaug(void* Kf, const int* dims, const int64_t* strides, int dtype, void* stream);
/* Forward on the matrix cores (bf16, D == 96, A in {32, 64}, L <= 256): per 64-row tile P = q Rcat^T lands in LDS and the
 * augmentation columns are picked there; bit-identical to dcc_gemm + dcc_relpos_gather. */
int dcc_relpos_aug_fwd_mfma(void* Qf, const void* Rcat, const int* idx_h, const int* idx_w, const int* idx_t, const int* dims,
                            const int64_t* strides, float scale, int dtype, void* stream);
/* Fused backward on the matrix cores (bf16, D == 96, A in {32, 64}, L <= 256): per 64-row tile the dP the scatter kernel would
 * write is built in LDS and consumed there: dq [B*N*H][D] = dP Rcat + dQf[:, :D], dRcat [L][D] fp32 += dP^T q — i.e.
 * dcc_relpos_scatter + the two dcc_gemm calls without the [rows, L] tensor.  ws: fp32 scratch [max_blocks * L * D] (per-block
 * partial table gradients, no initialisation needed).  DCC_ERR_UNSUPPORTED outside that domain. */
int dcc_relpos_aug_bwd(const void* dQf, const void* Qf, void* dq, const void* Rcat, float* dRcat, float* ws, int max_blocks,
                       const int* idx_h, const int* idx_w, const int* idx_t, const int* dims, const int64_t* strides,
                       float scale, int dtype, void* stream);
/* dcc_relpos_aug_bwd + the dcc_layernorm_bwd autograd runs next on its result (Pool.norm_act of the pooled queries,
 * torchvision MultiscaleAttention.pool_q as called from models/video_encoder.py:232-253) in one kernel: conv [B*Nq*H][D] bf16 is
 * the LayerNorm's input, mean / rstd [B*Nq*H] its saved statistics, gamma [D] its weight; dconv [B*Nq*H][D] is stored instead of
 * dq (dq itself is rounded to bf16 before use, exactly as the two-kernel path stores it), dgamma / dbeta [D] fp32 are
 * accumulated into.  ws: fp32 scratch [max_blocks * (L + 2) * D]. */
int dcc_relpos_aug_bwd_ln(const void* dQf, const void* Qf, void* dconv, const void* Rcat, float* dRcat, float* ws, int max_blocks,
                          const int* idx_h, const int* idx_w, const int* idx_t, const int* dims, const int64_t* strides,
                          float scale, const void* conv, const float* mean, const float* rstd, const float* gamma, float* dgamma,
                          float* dbeta, int dtype, void* stream);

/* ---- fused MLP forward of an MViT block (csrc/mlp_fused.hip): y = [x +] fc2(gelu(fc1(LayerNorm(x)))), hidden = 4 C --------
 * torchvision MultiscaleBlock's `x = x + mlp(norm2(x))` (reached through models/video_encoder.py:232-253) in ONE launch: the
 * [M, 4C] hidden tensor is produced and consumed in registers and only streams out for the backward.  bf16, C in {96, 192, 384}.
 *   packed   = dcc_mlp_pack_weights(w1 [4C, C], w2 [C, 4C], both row-major bf16): the two weights in MFMA-fragment order,
 *              dcc_mlp_packed_bytes(C) bytes (0: C outside the domain); re-pack whenever the weights change
 *   y [M, C] (row pitch ldy) = (residual ? x : 0) + b2 + W2 gelu(bf16(W1 LN(x) + b1)); LN statistics in fp32
 *   saved for the backward (each nullable): xn [M, C] = LayerNorm(x), mean / rstd [M] fp32, pre [M, 4C] = bf16(fc1 + b1),
 *   act [M, 4C] = gelu(pre) (act != NULL needs pre != NULL).
 * DCC_ERR_UNSUPPORTED outside the domain: use dcc_layernorm_fwd + dcc_gemm (the two-GEMM path). */
int dcc_mlp_packed_bytes(int C);
int dcc_mlp_pack_weights(const void* w1, const void* w2, void* packed, int C, void* stream);
int dcc_ln_mlp_fwd(const void* x, int64_t ldx, const float* gamma, const float* beta, float eps, const void* packed,
                   const float* b1, const float* b2, void* y, int64_t ldy, void* xn, float* mean, float* rstd, void* pre,
                   void* act, int64_t M, int C, int residual, void* stream);

/* ---- LayerNorm (csrc/norm.hip): nn.LayerNorm in MViT blocks / Pool.norm_act / BERT / aggregator /
 * attention pool (models/video_aggregator.py:15,24,99; models/attention_pool.py:53,94). */
int dcc_layernorm_fwd(const void* x, int64_t ldx, void* y, int64_t ldy, const float* gamma, const float* beta,
                      float* mean, float* rstd, int64_t rows, int C, float eps, int dtype, void* stream);
/* dres (nullable, dx's type and leading dimension): gradient reaching x through its OTHER use in a transformer block
 * (the residual / skip branch), added into dx here instead of by a separate autograd add kernel.
 * workspace (nullable): DCC_LN_WS_FLOATS fp32 of scratch owned by one stream (no initialisation needed).  With it every
 * block stores its dgamma / dbeta partial and a second small kernel folds them (no atomics, grid up to 2048 blocks);
 * without it every block adds straight into dgamma / dbeta with atomics (grid capped at 512 blocks). */
#define DCC_LN_WS_FLOATS (1 << 20)
int dcc_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const float* mean, const float* rstd,
                      const float* gamma, void* dx, int64_t lddx, float* dgamma, float* dbeta, const void* dres,
                      float* workspace, int64_t rows, int C, int dtype, int grad_dtype, void* stream);

/* ---- convolution-shaped ops (csrc/conv.hip) ------------------------------------------------
 * dcc_im2col3d: operand of the conv_proj GEMM; video is NDHWC fp32 as produced by the permute+view of
 *   models/video_encoder.py:572-574.  Column order (kt, kh, kw, cin), zero padded to Kp.
 * dcc_pool_*: torchvision mvit.Pool (depthwise Conv3d k=3 on q/k/v heads, MaxPool3d on the skip path);
 *   dims = {B,H,C,T,Hs,Ws,kt,kh,kw,st,sh,sw,pt,ph,pw,has_cls}; strides = {x_b,x_n,x_h,y_b,y_n,y_h};
 *   mode 0 = depthwise conv (w [C][taps] fp32), mode 1 = max pool (argmax uint8 [B,No,H,C]). */
int dcc_im2col3d(const float* x, void* col, int B, int T, int H, int W, int Cin, int KT, int KH, int KW, int st,
                 int sh, int sw, int pt, int ph, int pw, int Kp, int dtype, void* stream);
/* Same rows from raw uint8 frames: (pixel - mean[c]) / std[c] (utils/video.py:278-288) is applied inside the gather, so
 * the clip crosses PCIe as bytes and no normalised fp32 copy is ever written. */
int dcc_im2col3d_u8(const uint8_t* x, const float* mean, const float* stdv, void* col, int B, int T, int H, int W, int Cin,
                    int KT, int KH, int KW, int st, int sh, int sw, int pt, int ph, int pw, int Kp, int dtype,
                    void* stream);
int dcc_pool_fwd(const void* x, void* y, const float* w, uint8_t* argmax, const int* dims, const int64_t* strides,
                 int mode, int dtype, void* stream);
/* mvit.Pool of a q / k / v branch in ONE launch: depthwise Conv3d k=3 + the LayerNorm(head_dim) that follows it
 * (torchvision MultiscaleAttention pool_q / pool_k / pool_v = Pool(conv, norm), called from models/video_encoder.py:232-253).
 * bf16, head_dim 96, spatial stride 1/2/4/8.  y = conv output (saved for the LayerNorm backward, y-strides of `strides`),
 * o = LayerNorm(y) rows in (b, n, h) order with row stride ldo (elements), mean / rstd fp32 [B*No*H].
 * aug = 32 / 64 (k pool): the one-hot rel-pos key augmentation (dcc_relpos_kaug) is written to columns 96 .. 96 + aug of the o rows too.
 * DCC_ERR_UNSUPPORTED outside that domain: use dcc_pool_fwd + dcc_layernorm_fwd. */
int dcc_pool_ln_fwd(const void* x, void* y, void* o, int64_t ldo, const float* w, const float* gamma, const float* beta,
                    float* mean, float* rstd, const int* dims, const int64_t* strides, float eps, int aug, int dtype, void* stream);
/* Two pools of identical geometry and strides (a block's k and v) in ONE launch of the same kernel; ptrs0 / ptrs1 =
 * {x, y, o, w, gamma, beta, mean, rstd} of each branch, ldo / aug per branch. */
int dcc_pool_ln_fwd2(const void* const* ptrs0, const void* const* ptrs1, int64_t ldo0, int64_t ldo1, int aug0, int aug1,
                     const int* dims, const int64_t* strides, float eps, int dtype, void* stream);
int dcc_pool_bwd_data(const void* dy, void* dx, const float* w, const uint8_t* argmax, const int* dims,
                      const int64_t* strides, int mode, int dtype, int grad_dtype, void* stream);
int dcc_dwconv_bwd_weight(const void* dy, const void* x, float* dw, float* workspace, const int* dims,
                          const int64_t* strides, int dtype, int grad_dtype, void* stream);

/* ---- streaming helpers (csrc/elementwise.hip) ---------------------------------------------- */
int dcc_cast(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);
int dcc_cast2d(const void* src, int64_t lds_, void* dst, int64_t ldd, int64_t rows, int cols, int src_dtype,
               int dst_dtype, void* stream);
/* dst[c][r] = bf16(src[r][c]): transposed bf16 shadow of an fp32 weight [rows, cols] (lds_ / ldd: row pitches in elements).  The
 * data-gradient product dX = dY W of the early MViT blocks runs on the streaming GEMM, which takes its weight as [N][K] rows. */
int dcc_cast_transpose(const float* src, int64_t lds_, void* dst, int64_t ldd, int rows, int cols, void* stream);
int dcc_axpby(void* out, float a, const void* x, float b, const void* y, int64_t n, int dtype, void* stream);
int dcc_dgelu_mul(const void* dy, const void* pre, void* out, int64_t n, int dtype, void* stream);
/* uint8 frames -> (x - mean[c]) / std[c] fp32, channel innermost (v2.Normalize of utils/video.py:278-288 moved to the GPU:
 * the host uploads 1 byte per sample).  n = number of samples (multiple of 4), mean/std device fp32 [C]. */
int dcc_u8_normalize(const uint8_t* src, float* dst, int64_t n, int C, const float* mean, const float* stdv, void* stream);
/* Decoded uint8 frames -> the clip tensor the trunk reads, on the GPU (csrc/resize.hip; replaces the host fp32 pipeline of
 * utils/video.py:246-288: frame index select / last-frame padding, v2.Resize((s, s), antialias=True) = F.interpolate(bilinear,
 * antialias, align_corners=False), grayscale -> 3 channels, v2.Normalize).  src: uint8 [T_in, H_in, W_in, C_in] (C_in 1 | 3);
 * frame_idx: device int32 [T_out] source frame per output frame (null = identity); dst: [T_out, H_out, W_out, 3] fp32
 * ((v - mean[c]) / stdv[c] when mean / stdv are given) or, with out_u8 = 1, uint8 rounded half-to-even and clamped (feeds
 * dcc_im2col3d_u8, which normalises inside the patch gather).  antialias = 0: plain bilinear. */
int dcc_resize_frames_u8(const uint8_t* src, const int* frame_idx, void* dst, const float* mean, const float* stdv, int T_in,
                         int H_in, int W_in, int C_in, int T_out, int H_out, int W_out, int antialias, int out_u8,
                         void* stream);
/* torchvision.ops.StochasticDepth(p, "row") of MultiscaleBlock (torchvision mvit.py via models/video_encoder.py:93):
 * out = resid + x * scale[sample] with scale = bernoulli(1-p)/(1-p) per clip; resid may be null (backward). */
int dcc_scale_rows(const void* x, const float* scale, const void* resid, void* out, int64_t samples,
                   int64_t per_sample, int dtype, void* stream);
int dcc_colsum(const void* x, int64_t ldx, float* out, int64_t M, int N, int dtype, void* stream);
int dcc_broadcast_row(const float* src, void* dst, int64_t ld, int64_t rows, int C, int dtype, void* stream);
/* mean over tokens (models/video_encoder.py:603) */
int dcc_token_mean(const void* in, void* out, int B, int N, int D, int backward, int dtype, void* stream);
/* BERT embeddings (HF BertEmbeddings via models/text_encoder.py:131) */
int dcc_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type, void* out,
                  int64_t rows, int S, int D, int vocab, int dtype, void* stream);
int dcc_embed_bwd(const int64_t* ids, const void* dy, float* dword, float* dpos, float* dtype_, int64_t rows, int S,
                  int D, int vocab, int dtype, void* stream);
/* Rope3D.forward (models/rope_3d.py:186-252); tables [N,D] fp32; strides = {x_b,x_n,x_h,y_b,y_n,y_h} */
int dcc_rope(const void* x, void* y, const float* cos_t, const float* sin_t, int B, int N, int H, int D,
             const int64_t* strides, int backward, int dtype, void* stream);
/* EnhancedVideoAggregator learned-query softmax pooling (models/video_aggregator.py:131-159), fp32 */
int dcc_qpool_fwd(const float* x, const float* q, float* out, float* w, int B, int N, int D, void* stream);
int dcc_qpool_bwd(const float* x, const float* q, const float* w, const float* dout, float* dx, float* dq, int B,
                  int N, int D, void* stream);
/* torch.nan_to_num guards (runners/video_constrative_learning_runner.py:1189,1236-1238) */
int dcc_nan_to_num(void* x, int64_t n, float nanv, float posv, float negv, int* count, int dtype, void* stream);
/* nn.Dropout: keep mask = hash(seed [+ *step * golden ratio], index); `step` (nullable, device int64) lets a captured launch
 * draw a new mask per replay; the same (seed, *step) regenerates the mask for the backward. */
int dcc_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, const int64_t* step, int dtype, void* stream);
int dcc_scale_by_dev(float* x, int64_t n, const float* coef, void* stream);

/* ReLU of nn.TransformerEncoderLayer (models/attention_pool.py:139-147): gate NULL -> y = max(x, 0); else y = x * [gate > 0]. */
int dcc_relu(const void* x, const void* gate, void* y, int64_t n, int dtype, void* stream);

/* ---- multi-instance pooling (csrc/mil.hip): token-level consumers of the encoder, models/multi_instance_linear_probing.py
 * :494-538 (gated attention features tanh(Vx) * sigmoid(Ux), masked softmax over instances, weighted instance sums, masked
 * mean / max) and the patch pooling of utils/loss/multitask_loss.py:281.  mask: uint8 [B,N], non-zero = valid, nullable. */
int dcc_gate_fwd(const void* v, const void* u, void* out, int64_t n, int dtype, void* stream);
int dcc_gate_bwd(const void* d, const void* v, const void* u, void* dv, void* du, int64_t n, int dtype, void* stream);
int dcc_masked_softmax_fwd(const float* s, const uint8_t* mask, float* w, int B, int N, void* stream);
int dcc_masked_softmax_bwd(const float* w, const float* dw, float* ds, int B, int N, void* stream);
int dcc_wsum_fwd(const float* w, const void* x, void* out, int B, int N, int D, int dtype, void* stream);
int dcc_wsum_bwd(const float* w, const void* x, const void* dout, void* dx, float* dw, int B, int N, int D, int dtype,
                 void* stream);
int dcc_masked_reduce_fwd(const void* x, const uint8_t* mask, void* out, int* arg, int B, int N, int D, int mode, int dtype,
                          void* stream);
int dcc_masked_reduce_bwd(const void* dout, const uint8_t* mask, const int* arg, void* dx, int B, int N, int D, int mode,
                          int dtype, void* stream);
/* View-angle embeddings of the MIL head (models/multi_instance_linear_probing.py:142-145 `nn.Embedding(num_view_classes + 1,
 * D)`, :265-269 `x = x + view_embedding(view_ids)` broadcast over the L patch tokens of a video): out[r][l][:] = x[r][l][:] +
 * table[ids[r]][:], rows r = (study, video); the backward accumulates dtable[ids[r]][:] += sum_l dy[r][l][:] (fp32 atomics into
 * a zeroed table gradient).  ids: int64, range-checked by the caller. */
int dcc_view_embed_add(const float* x, const int64_t* ids, const float* table, float* out, int64_t rows, int L, int D,
                       void* stream);
int dcc_view_embed_bwd(const float* dy, const int64_t* ids, float* dtable, int64_t rows, int L, int D, void* stream);

/* ---- contrastive losses (csrc/loss.hip); scal = fp32[8]
 * {0 loss, 1 dlog_temp, 2 dbias, 3 mean diag log-softmax, 4 mean diag cosine, 5 sum ||v||, 6 sum ||t||, 7 -} */
int dcc_l2norm_fwd(const float* x, int64_t ldx, float* y, float* norms, int R, int D, float eps, float* norm_sum,
                   void* stream);
int dcc_l2norm_bwd(const float* dy, const float* y, const float* norms, float* dx, int64_t lddx, int R, int D,
                   float eps, void* stream);
int dcc_logits_fwd(const float* sim, float* z, int B, int M, int mode, const float* log_temp, int clamp_temp,
                   const float* bias, const float* col_margin, void* stream);
int dcc_logits_bwd(const float* sim, const float* dz, float* dsim, int B, int M, int mode, const float* log_temp,
                   int clamp_temp, const float* bias, float* scal, void* stream);
int dcc_lse(const float* z, float* lse_rows, float* lse_cols, int B, int M, void* stream);
int dcc_ce_diag(const float* z, const float* lse_r, const float* lse_c, float* dz, int B, float smooth, float* scal,
                const float* sim, void* stream);
int dcc_rowcol_sums(const float* w, float* rs, float* cs, int B, int M, void* stream);
int dcc_weighted_ce(const float* z, const float* pos, const float* lse_r, const float* lse_c, const float* rs,
                    const float* cs, float* dz, int B, int M, float eps, float* scal, void* stream);
int dcc_bce(const float* z, const float* y, const float* pw, float pos_weight, float neg_weight, float thr,
            float denom_host, const float* denom_dev, float* dz, int64_t n, float* scal, void* stream);
/* Multi-positive alignment diagnostics of the runner's inline SigLIP branch (runner.py:1298-1313): out fp32[8]
 * (accumulated into) = {sum_rows weighted log-prob (weights t*w), #valid rows, the same with weights t, #valid rows,
 * sum of cosines over the positives, #positives, sum |w|}.  weights may be NULL. */
int dcc_mp_align(const float* z, const float* sim, const float* targets, const float* weights, int B, int M, float* out,
                 void* stream);
/* out = max(a * b, 0), b nullable (clamped positive weights, utils/loss/multi_positive_infonce.py:22-26,52). */
int dcc_mul_clamp0(const float* a, const float* b, float* out, int64_t n, void* stream);
/* out = clamp(x, lo, hi) (BCE targets: pos_mask.clamp(0, 1), utils/loss/contrastive.py:284). */
int dcc_clamp(const float* x, float* out, int64_t n, float lo, float hi, void* stream);
/* Auto-balanced positive weights: out[i][:] = max((T - n_i) / n_i, 1), n_i = max(sum_j counts[i][j], 1)
 * (utils/loss/contrastive.py:291-296, utils/loss/siglip_pairwise.py:161-166). */
int dcc_auto_balance_weights(const float* counts, float* out, int B, int T, void* stream);
/* Entropy regulariser on logits z [B,M] (utils/loss/contrastive.py:19-68): scal[0] += weight * relu(thr - mean_i H_i),
 * dz (nullable) += its gradient; ent fp32[4] in = {0, +inf, 0, 0}, out = {sum H, min H, max H, -}. */
int dcc_entropy_reg(const float* z, float* dz, int B, int M, float weight, float thr, float* ent, float* scal,
                    void* stream);
/* Symmetric multi-positive InfoNCE on logits z [N,M] with weights w (utils/loss/multi_positive_infonce.py:9-110):
 * wsum / msum / imp = row and column sums of the clamped weights, of the mask, of the importance base (imp nullable);
 * ws = zeroed fp32[2(N+M)]; scal[0] += loss; dz (nullable) = dL/dz. */
int dcc_mp_infonce(const float* z, const float* w, const float* lse_r, const float* lse_c, const float* wsum_r,
                   const float* wsum_c, const float* msum_r, const float* msum_c, const float* imp_r,
                   const float* imp_c, float* ws, float* dz, int N, int M, int mean, float imp_eps, float* scal,
                   void* stream);
/* out = targets * weights unless every weight is zero, then targets (runner.py:1276-1278); nz: zeroed fp32[1]. */
int dcc_mp_pos_weights(const float* targets, const float* weights, float* out, int64_t n, float* nz, void* stream);

/* ---- step tail (csrc/optim.hip): clip_grad_norm_ + AdamW (runner.py:1398-1421).
 * tab = device array of DccTensorDesc {float* p,g,m,v; void* shadow; int64 n; float lr,beta1,beta2,eps,
 * weight_decay,bc1,bc2; int flags}; work = device int2[nblocks] (tensor, chunk of DCC_OPT_CHUNK). */
int dcc_grad_sumsq(const void* tab, const void* work, int nblocks, double* out, void* stream);
int dcc_clip_coef(const double* sumsq, float max_norm, float* coef, float* norm_out, void* stream);
int dcc_scale_grads(const void* tab, const void* work, int nblocks, const float* coef, void* stream);
int dcc_adam_step(const void* tab, const void* work, int nblocks, const float* grad_scale, void* stream);

/* ---- epoch-end retrieval ranks (csrc/misc.hip): position of each ground-truth candidate in the stable descending order
 * of its similarity row — all of utils/retrieval_metrics.py:65-324 (Recall@k, MRR, MAP, NDCG@k, median rank) follows from
 * these integers.  gt int32 [Nq, G], -1 = empty; ranks int32 [Nq, G], 1-based, 0 for empty slots. */
int dcc_retrieval_ranks(const float* sim, int64_t ld, const int* gt, int* ranks, int Nq, int Nc, int G, void* stream);

/* ---- inference() top-k (csrc/misc.hip): torch.topk(sim, k, dim=1) of runners/video_constrative_learning_runner.py:1731-1758.
 * idx int64 [rows, k], val fp32 [rows, k]: the first k entries of a stable descending sort of each row (NaN first). */
int dcc_topk_rows(const float* sim, int64_t ld, int rows, int M, int k, int64_t* idx, float* val, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCC_H */
