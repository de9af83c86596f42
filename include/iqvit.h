/* iqvit.h -- C ABI of the MI355X-native (gfx950) training path for the ViT / raw-IQ
 * modulation classifiers of aliftffd/ViT-vs-Raw-IQ.
 *
 * The reference has no FFI layer: its boundary is the Python nn.Module surface
 * (SURVEY.md section 8b).  This header is the boundary the MI355X build puts underneath
 * that surface: plain pointers and sizes, no torch types.  Every pointer is DEVICE memory
 * (HBM) unless marked host.  All activations are bf16 (uint16 storage), row-major,
 * tokens-major [B*S, D]; parameters, gradients, statistics, logits and losses are fp32.
 * Every launch goes to the hipStream_t passed as `stream` (iq_stream_t == hipStream_t);
 * no call synchronises, allocates or frees, so a call sequence can be captured in a hipGraph.
 * Return value: 0 ok; IQ_ERR_* otherwise (host mirrors map them to Python exceptions).
 *
 * Citations are to /root/reference/Transformer_Thesis/ (V/ = ViT/, R/ = transformer_rawIQ/).
 */
#ifndef IQVIT_H
#define IQVIT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* iq_stream_t; /* hipStream_t */

enum { IQ_STATUS_OK = 0, IQ_STATUS_ARG = 1, IQ_STATUS_UNSUPPORTED = 2, IQ_STATUS_LAUNCH = 3 };

/* Dropout site: Philox4x32-7 (7 rounds: csrc/common.h) keyed by seed, counter (element_index/8, site, step).
 * p == 0 disables.  Masks are regenerated in backward from the same triple, never stored. */
typedef struct iq_dropout {
  uint64_t seed;
  uint32_t step;
  uint32_t site;
  float p;
  const uint32_t* step_dev; /* optional DEVICE u32 overriding `step` (bumped on device under hipGraph replay) */
} iq_dropout_t;

/* ---------------------------------------------------------------------------------------
 * LayerNorm.  Replaces LayerNorm.forward, V/models/layers/layers_norm.py:11-19 (eps 1e-12,
 * biased variance) and its autograd backward; also nn.LayerNorm (eps 1e-5) of the rawIQ head.
 * z,x,dx,dz,dy: bf16 [M,D]; gamma,beta,dgamma,dbeta: fp32 [D]; mean,rstd: fp32 [M].
 * iq_ln_bwd optionally emits dy = dropout(dz) for the site that preceded the residual add
 * (V/models/blocks/encoder_layer.py:24-25,32-33).  ws: iq_ln_bwd_ws_bytes(D) scratch. */
int iq_ln_supported(int D);
int iq_ln_fwd(const void* z, const float* gamma, const float* beta, void* x, float* mean, float* rstd, int M, int D,
              float eps, iq_stream_t stream);
size_t iq_ln_bwd_ws_bytes(int D);
int iq_ln_bwd(const void* dx, const void* z, const float* mean, const float* rstd, const float* gamma, void* dz,
              void* dy, const iq_dropout_t* drop, float* dgamma, float* dbeta, float* ws, int accumulate, int M,
              int D, iq_stream_t stream);
/* iq_ln_bwd with dgamma == dbeta == NULL leaves the per-block partial sums in ws as
 * [iq_ln_bwd_partial_rows(M, D)][2*D] fp32 (dgamma partials | dbeta partials per row) for a later fused, fixed-order
 * reduction: pass them as iq_reduce_seg_t entries to iq_gemm_bf16_wgrad_grouped. */
int iq_ln_bwd_partial_rows(int M, int D);

/* ---------------------------------------------------------------------------------------
 * bf16 MFMA GEMM, C[M,N] = epilogue(A[M,K] * B[N,K]^T), fp32 accumulate.
 * Replaces every nn.Linear on the path: w_q/w_k/w_v/w_concat
 * (V/models/layers/multi_head_attention.py:18,28), linear1/linear2
 * (V/models/layers/position_wise_feed_forward.py:13-16), the non-overlapping Conv2d/Conv1d
 * of the embeddings (V/models/embedding/patch_embedding.py:9-15,
 * R/models/embedding/patch_embedding.py:29-43) and, with B = W^T shadows, their dgrads.
 * Epilogue order: +bias, relu, +pe (row remap), dropout, *gate, +residual.
 *   tok>0 : embedding mode, output row = (m/tok)*seq + m%tok + cls_off and pe[(m%tok+cls_off),:]
 *           is added (V/models/encoder.py:42-47).
 *   gate  : v *= (gate[m,n] > 0) ? gate_scale : 0   (ReLU+dropout backward from the saved hidden).
 * K%8==0, N%8==0, lda/ldb/ldc/ldr/ldg in elements and %8==0.
 * The kernel is chosen from the shape; results do not depend on the choice beyond fp32 summation order:
 *   N%256==0, K%64==0, K>=256 and >= 512 tiles of 256x256 (ViT-Base at its batch): persistent 256x256 tiles
 *   (gemm_big.hip, 0.85-1.19 PFLOP/s); otherwise 128x{128,64} tiles (gemm_nt.hip); K%32!=0: register-staged fallback. */
typedef struct iq_epilogue {
  const float* bias;    /* [N] or NULL */
  int relu;
  const float* pe;      /* [seq, N] fp32 or NULL (embedding mode) */
  int tok, seq, cls_off;
  iq_dropout_t drop;    /* indexed by OUTPUT element (row_out*N + n) */
  const void* gate;     /* bf16 [M, ldg] or NULL */
  int ldg;
  float gate_scale;
  const void* residual; /* bf16 [M_out, ldr] or NULL */
  int ldr;
} iq_epilogue_t;
int iq_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                    const iq_epilogue_t* epi, iq_stream_t stream);

/* GEMM + the post-norm tail of an encoder sub-layer in ONE launch (whole-row tiles, D in {128,192,256}, K % 32 == 0, K >= 64):
 *   Z = dropout(A[M,K] * W[D,K]^T + bias) + residual        bf16 [M,D]   (kept for backward)
 *   X = gamma * (Z - mean) * rstd + beta                     bf16 [M,D];  mean, rstd fp32 [M] (kept for backward)
 * Replaces `x = norm1(dropout1(attention(x)) + x)` / `x = norm2(dropout2(ffn(x)) + x)`,
 * V/models/blocks/encoder_layer.py:24-25,32-33 with LayerNorm.forward (V/models/layers/layers_norm.py:11-19), i.e. an
 * iq_gemm_bf16_nt (bias, dropout, residual) followed by iq_ln_fwd: same Z bit for bit, same statistics to fp32
 * summation order (taken from the bf16-rounded Z, two-pass).  All pointers 16-byte aligned. */
int iq_gemm_ln_supported(int D, int K);
int iq_gemm_bf16_ln(const void* A, int lda, const void* W, int ldw, const float* bias, const void* residual, int ldr,
                    const iq_dropout_t* drop, const float* gamma, const float* beta, float eps, void* Z, void* X,
                    float* mean, float* rstd, int M, int D, int K, iq_stream_t stream);

/* Data-gradient GEMM + the LayerNorm BACKWARD that consumes its result, in ONE launch (whole-row tiles, D in {128,192},
 * K % 32 == 0, K >= 64):   dX = A[M,K] * Wt[D,K]^T + residual   (fp32, not stored), then exactly iq_ln_bwd on it:
 *   dz = rstd * (g - mean_D(g) - xhat * mean_D(g * xhat)),  g = dX * gamma,  xhat = (z - mean) * rstd      bf16 [M,D]
 *   dy = dropout_mask(dz) * scale (only when drop->p > 0)                                                   bf16 [M,D]
 *   partial: iq_gemm_lnbwd_partial_rows(M) rows of [2*D] fp32 (dgamma | dbeta partial sums), to be reduced like
 *   iq_ln_bwd's (iq_reduce_seg_t).
 * The autograd backward of `x = norm(dropout(f(x)) + x)`, V/models/blocks/encoder_layer.py:24-25,32-33, behind the FFN1 /
 * QKV data-gradient GEMM that feeds it.  All pointers 16-byte aligned. */
int iq_gemm_lnbwd_supported(int D, int K);
int iq_gemm_lnbwd_partial_rows(int M);
int iq_gemm_bf16_lnbwd(const void* A, int lda, const void* Wt, int ldw, const void* residual, int ldr, const void* z,
                       const float* mean, const float* rstd, const float* gamma, const iq_dropout_t* drop, void* dz,
                       void* dy, float* partial, int M, int D, int K, iq_stream_t stream);

/* The feed-forward sub-layer + norm2 in one launch (rows are owned by waves, 32 at a time; M = frames * S rows, D in {128, 192},
 * F % 64 == 0, the LDS weight ring + F floats within 160 KB: iq_ffn_chain_supported):
 *   H = dropout1(relu(X1 * W1^T + b1))            bf16 [M, F]  (written once; the weight gradients read it)
 *   Z = dropout2(H * W2^T + b2) + X1              bf16 [M, D]  (kept for backward)
 *   X = gamma * (Z - mean) * rstd + beta          bf16 [M, D];  mean, rstd fp32 [M]
 * = PositionwiseFeedForward.forward (V/models/layers/position_wise_feed_forward.py:12-17) + `x = norm2(dropout2(ffn(x)) + x)`
 * (V/models/blocks/encoder_layer.py:30-33).  The hidden tile goes from the first product's epilogue into the second product in
 * registers, never re-read from HBM.  H and Z equal iq_gemm_bf16_nt (bias, relu, drop1) followed by iq_gemm_bf16_ln up to fp32
 * summation order inside one MFMA (bf16 rounding ties only); dropout indices as there (output element row*N + n).
 * W1 [F, D], W2 [D, F] bf16 row-major; all pointers 16-byte aligned.
 * iq_attn_out_ffn_chain_fwd: the same launch with the attention output projection + dropout + residual + norm1 in front
 * (`x = norm1(dropout1(attention_out) + x)`, encoder_layer.py:24-28 -- the whole layer after the attention core):
 *   Z1 = dropout0(A * Wo^T + bo) + R              bf16 [M, D]  (kept for backward)
 *   X1 = gamma1 * (Z1 - mean1) * rstd1 + beta1    bf16 [M, D] (written for backward; consumed from registers);  mean1, rstd1 [M]
 * then as above on X1.  A [M, D] attention output, Wo [D, D], R [M, D] the layer input.  Z1 equals iq_gemm_bf16_ln's up to bf16
 * rounding ties (summation order inside one MFMA); the rest equals iq_ffn_chain_fwd on the X1 written, bit for bit.
 * Wq / bq / Yq (all three or none): Yq[M, 3D] = X * Wq[3D, D]^T + bq behind norm2 -- the NEXT layer's packed q,k,v projection
 * (multi_head_attention.py:17-19) of this layer's output, from registers. */
/* iq_ffn_chain_bwd: the data path of the same sub-layer's backward in one launch (replaces the gate data-gradient GEMM and
 * iq_gemm_bf16_lnbwd; autograd of position_wise_feed_forward.py:12-17 and of the norm1 feeding it, encoder_layer.py:24-25):
 *   gH = (H > 0) ? (dO * W2t^T) * gate_scale : 0         bf16 [M,F]   (written once: the W1 / W2 weight gradients read it)
 *        "H > 0" (ReLU and dropout1 of the forward pass at once) comes as ONE BIT per hidden unit in `gate_bits`,
 *        iq_ffn_chain_gate_bytes(M, F) bytes written by iq_ffn_chain_fwd (its `gate_bits` argument, NULL = not wanted) in the
 *        backward kernel's own wave / chunk / lane order (an opaque buffer between the two calls)
 *   dX1 = gH * W1t^T + residual (rounded to bf16), then exactly iq_ln_bwd on it with z1 / mean / rstd / gamma:
 *   dz bf16 [M,D], dy = dropout_mask(dz) * scale (only when drop->p > 0), partial: iq_ffn_chain_bwd_partial_rows(M) rows of
 *   [2*D] fp32 (dgamma | dbeta partial sums, one row per workgroup) for the fused fixed-order reduction (iq_reduce_seg_t).
 * W2t [F, D] and W1t [D, F] are the TRANSPOSED weights (bf16 row-major), M = frames * S.
 * Wot / dA (both or neither): dA[M, D] = dy * Wot[D, D]^T behind it (dz where there is no dropout) -- the data gradient of the
 * attention output projection (Wot = Wo transposed; autograd of multi_head_attention.py:28), i.e. what iq_attn_bwd starts
 * from, in the same launch: equals iq_gemm_bf16_nt(dy, Wot) up to bf16 rounding ties.
 * iq_qkv_dgrad_ffn_chain_bwd: the same launch (Wot / dA required) with the launch that would produce its dO in front -- exactly
 * iq_gemm_bf16_lnbwd(gQKV [M,3D], Wqkv_t [D,3D], residual0, z2, mean2, rstd2, gamma2, drop2) -> dz2, dy2, partial2
 * (iq_ffn_chain_bwd_partial_rows(M) rows): the data gradient of the packed q,k,v projection of the layer ABOVE
 * (multi_head_attention.py:17-19) + the residual path, and this layer's norm2 backward (encoder_layer.py:30-33).  dy2 (dz2 where
 * there is no dropout; same probability at both sites) is the second stage's dO, taken from the CU's LDS instead of HBM;
 * `residual` is normally dz2 itself.  Results equal the two launches up to bf16 rounding ties. */
int iq_ffn_chain_supported(int S, int D, int F);
int iq_ffn_chain_bwd_partial_rows(int M);
size_t iq_ffn_chain_gate_bytes(int M, int F);
int iq_ffn_chain_bwd(const void* dO, const void* W2t, const void* gate_bits, float gate_scale, void* gH, const void* W1t,
                     const void* residual, const void* z1, const float* mean, const float* rstd, const float* gamma,
                     const iq_dropout_t* drop, void* dz, void* dy, float* partial, const void* Wot, void* dA, int frames, int S,
                     int D, int F, iq_stream_t stream);
int iq_qkv_dgrad_ffn_chain_bwd(const void* gQKV, const void* Wqkv_t, const void* residual0, const void* z2, const float* mean2,
                               const float* rstd2, const float* gamma2, const iq_dropout_t* drop2, void* dz2, void* dy2, float* partial2,
                               const void* W2t, const void* gate_bits, float gate_scale, void* gH, const void* W1t, const void* residual,
                               const void* z1, const float* mean, const float* rstd, const float* gamma, const iq_dropout_t* drop, void* dz,
                               void* dy, float* partial, const void* Wot, void* dA, int frames, int S, int D, int F, iq_stream_t stream);
int iq_ffn_chain_fwd(const void* X1, const void* W1, const float* b1, const iq_dropout_t* drop1, void* H, const void* W2,
                     const float* b2, const iq_dropout_t* drop2, const float* gamma, const float* beta, float eps, void* Z,
                     void* X, float* mean, float* rstd, void* gate_bits, int frames, int S, int D, int F, iq_stream_t stream);
int iq_attn_out_ffn_chain_fwd(const void* A, const void* Wo, const float* bo, const iq_dropout_t* drop0, const void* R,
                              const float* gamma1, const float* beta1, void* Z1, void* X1, float* mean1, float* rstd1,
                              const void* W1, const float* b1, const iq_dropout_t* drop1, void* H, const void* W2, const float* b2,
                              const iq_dropout_t* drop2, const float* gamma2, const float* beta2, float eps, void* Z2, void* X,
                              float* mean2, float* rstd2, void* gate_bits, const void* Wq, const float* bq, void* Yq, int frames,
                              int S, int D, int F, iq_stream_t stream);

/* Weight gradient: dW[N,K] (+)= dY[M,N]^T * X[M,K]; dbias[N] (+)= colsum(dY) (NULL to skip).
 * Split over M into slabs in `ws` (iq_wgrad_ws_bytes), reduced deterministically (no atomics). */
size_t iq_wgrad_ws_bytes(int M, int N, int K);
int iq_gemm_bf16_wgrad(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dbias, int M, int N, int K,
                       float* ws, size_t ws_bytes, int accumulate, iq_stream_t stream);

/* Several weight gradients that share M (the four Linear layers of one encoder layer: what torch.autograd runs as
 * separate addmm nodes behind V/models/blocks/encoder_layer.py:16-36) in ONE launch + ONE slab reduce.  At most 4
 * problems are fused; larger groups / large N*K run one at a time through the same workspace.
 * Fast path (gemm_wgrad_big.hip: LDS-shared 256-row tiles, a problem whose N is the model width computed transposed):
 * M%64==0, M>=4096, dY / X 128-byte aligned with ldy%64==0 and ldx%64==0, and one column tile of 128, 192 or 256
 * dividing the shorter side of every problem; anything else runs the wave-private / shared-tile kernels of
 * gemm_wgrad.hip.  Either way the slabs are summed in a fixed order: results are bit-reproducible run to run. */
typedef struct iq_wgrad_problem {
  const void* dY; /* bf16 [M, ldy] */
  int ldy;
  const void* X;  /* bf16 [M, ldx] */
  int ldx;
  float* dW;      /* fp32 [N, K], 16-byte aligned */
  float* dbias;   /* fp32 [N] or NULL */
  int N, K;
} iq_wgrad_problem_t;
/* Extra fixed-order column reductions that ride on the group's slab-reduce launch (the LayerNorm gamma/beta partials
 * of the same encoder layer): out[0..n) (+)= sum over `rows` rows of partials[row * row_stride + 0..n). */
typedef struct iq_reduce_seg {
  const float* partials;
  int rows;
  int64_t row_stride; /* floats */
  float* out;
  int64_t n;
} iq_reduce_seg_t;
/* max_workgroups: 0 = fill the GPU once (768 workgroups); a smaller budget leaves CU slots free for kernels of
 * another stream (the model's backward can overlap a layer's weight gradients with the next layer's dX chain).
 * extra / nextra: up to 4 iq_reduce_seg_t (or NULL / 0). */
size_t iq_wgrad_grouped_ws_bytes(const iq_wgrad_problem_t* probs, int nprob, int M, int max_workgroups);
int iq_gemm_bf16_wgrad_grouped(const iq_wgrad_problem_t* probs, int nprob, int M, float* ws, size_t ws_bytes,
                               int accumulate, int max_workgroups, const iq_reduce_seg_t* extra, int nextra,
                               iq_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Scaled-dot-product attention core, softmax(Q K^T / sqrt(dh)) V per (frame, head), no mask,
 * no dropout.  Replaces ScaleDotProductAttention.forward
 * (V/models/layers/scale_dot_product_attention.py:23-39) plus MultiHeadAttention.split/concat
 * (V/models/layers/multi_head_attention.py:34-47): reads the packed projection output
 * qkv[B*S, 3*D] (q | k | v, head h at columns h*dh) and writes out[B*S, D] already "concatenated".
 * The S x S scores never reach HBM; lse[B,H,S] (fp32, natural log) is kept for backward.
 * dh in {16,32,64}; S <= 4096 (backward keeps the staged side in LDS in chunks when one image does not fit:
 * embedding_type='conv1d', S = 1025, R/models/encoder.py:34-41). */
int iq_attn_supported(int S, int dh);
int iq_attn_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int dh, iq_stream_t stream);
int iq_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int B, int S, int H,
                int dh, iq_stream_t stream);
/* The optional mask branch, V/models/layers/scale_dot_product_attention.py:30-31 (`score.masked_fill(mask == 0, -10000)`
 * after the 1/sqrt(dh) scaling; no reference caller passes a mask).  mask: uint8 (B, 1 | H, S, S), 0 = masked;
 * mask_hstride = S*S when the mask has a head dimension, 0 when it is shared by all heads.  mask NULL = the calls above.
 * Backward passes no gradient through masked positions (masked_fill). */
int iq_attn_fwd_masked(const void* qkv, void* out, float* lse, const uint8_t* mask, long mask_hstride, int B, int S,
                       int H, int dh, iq_stream_t stream);
int iq_attn_bwd_masked(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                       const uint8_t* mask, long mask_hstride, int B, int S, int H, int dh, iq_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Embedding front end.  iq_patchify turns the fp32 input frame batch into the bf16 GEMM
 * operand [B*tok, Kpad] (patch element order = the conv weight's (c, py, px) / (c, j) order,
 * zero padded to Kpad); kind 0: src (B,C,H,W), patch p  (V/models/embedding/patch_embedding.py:11-15)
 *               kind 1: src (B,C,L),   kernel = stride = k (R/models/embedding/patch_embedding.py:47-60).
 * iq_cls_rows writes row 0 of every frame: dropout(cls + pe[0]) (V/models/encoder.py:42-47).
 * iq_embed_bwd_gather compacts d(x0) rows (dropping cls rows, re-applying the dropout mask)
 * into [B*tok, D] for the embedding wgrad and reduces d(cls) = sum_b d(x0)[b,0,:]. */
/* Input pipeline on the device: raw[n_frames, len, 2] fp32 I/Q frames -> per-channel z-score
 * (stats = {i_mean, i_std, q_mean, q_std}) -> out[n_frames, 2, take]: the first `take` I samples then the first `take`
 * Q samples of every frame.  take = len is both reference layouts: viewed as (1, 32, 64) it is the ViT "image" of
 * V/dataloader/dataset.py:210-224, as (2, len) the transpose of R/dataloader/dataset.py:214-222.  Bit-identical to the
 * CPU preprocessing (IEEE subtract and divide). */
int iq_frames_preprocess(const float* raw, float* out, int n_frames, int len, int take, const float* stats,
                         iq_stream_t stream);
int iq_patchify(const float* src, void* patches, int kind, int B, int C, int H, int W, int p, int Kpad,
                iq_stream_t stream);
int iq_cls_rows(const float* cls, const float* pe, void* x0, int B, int S, int D, const iq_dropout_t* drop,
                iq_stream_t stream);
int iq_embed_bwd_gather(const void* dx0, void* demb, float* dcls, int B, int S, int tok, int D, int has_cls,
                        const iq_dropout_t* drop, int accumulate, iq_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Classification head + loss.
 * iq_head_fwd: feat = x[:,0,:] (pool=0) or mean over tokens (pool=1); optional LayerNorm (eps 1e-5,
 * R/models/transformer_rawIQ.py:67-70,88-96); logits = feat*W^T + b (V/models/amc_transformer.py:29-30).
 * featn [B,D] fp32 (the pooled feature, normalised but pre-affine when LN is on) and hstat [B,2]
 * (mean, rstd) are kept for backward.
 * iq_ce_fwd_bwd: CrossEntropyLoss(label_smoothing) mean over `denom` frames (global batch under DDP); a label
 * outside [0, K) makes that frame's loss and gradient NaN (torch asserts on the device),
 * V/training/train.py:405; writes loss_sum (sum over local frames of per-frame loss), n_correct
 * (argmax==label, V/training/train.py:205-207) and dlogits.  Pass dlogits NULL to skip the gradient.
 * iq_head_bwd: gradients of W,b,(ln gamma,beta) and d(x_L) (bf16 [B*S,D], zero outside the pooled rows). */
int iq_head_fwd(const void* x, const float* ln_g, const float* ln_b, const float* W, const float* b, float* featn,
                float* hstat, float* logits, int B, int S, int D, int K, int pool, iq_stream_t stream);
int iq_ce_fwd_bwd(const float* logits, const int64_t* labels, int B, int K, float smoothing, float denom,
                  float* loss_sum, int32_t* n_correct, float* dlogits, iq_stream_t stream);
int iq_head_bwd(const float* dlogits, const float* featn, const float* hstat, const float* ln_g, const float* ln_b,
                const float* W, float* dW, float* db, float* dln_g, float* dln_b, void* dx, int B, int S, int D, int K,
                int pool, int accumulate, iq_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer on the flat fp32 parameter / gradient buffers.
 * iq_gradnorm_sq: out[0] = sum(g^2)  (clip_grad_norm_, V/training/train.py:199).
 * iq_adamw_step: clip scale min(1, max_norm/(sqrt(gnorm_sq)+1e-6)) folded into AdamW with decoupled
 * weight decay on every element (V/training/train.py:407-412); also writes the bf16 shadow.
 * max_norm <= 0 or gnorm_sq NULL disables clipping.  grad_scale multiplies g first (1/world_size). */
size_t iq_gradnorm_ws_bytes(size_t n);
int iq_gradnorm_sq(const float* g, size_t n, float grad_scale, float* ws, float* out, iq_stream_t stream);
int iq_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, const float* gnorm_sq, float max_norm,
                  float grad_scale, const float* dyn, iq_stream_t stream);
/* dyn: optional DEVICE float[2] = {lr, step}; when non-NULL it overrides lr/step (graph replay).
 * iq_counter_add: *ctr_u32 += inc_u32 and/or *ctr_f32 += inc_f32 (either pointer may be NULL). */
int iq_counter_add(uint32_t* ctr_u32, uint32_t inc_u32, float* ctr_f32, float inc_f32, iq_stream_t stream);
int iq_cast_bf16(const float* src, void* dst, size_t n, iq_stream_t stream);
/* dst[c, r] (ld = rows_pad) = bf16(src[r, c]); src fp32 [rows, cols] */
int iq_transpose_cast_bf16(const float* src, void* dst, int rows, int cols, iq_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Whole-model plan: the native runtime underneath AMCTransformer.forward / loss.backward().
 * Mirrors the constructors V/models/amc_transformer.py:9 and R/models/transformer_rawIQ.py:14-26. */
typedef struct iq_model_cfg {
  int kind;            /* 0 = ViT (2-D patches), 1 = raw-IQ (1-D sequence) */
  int in_channels;
  int img_h, img_w, patch;            /* kind 0 */
  int seq_length, conv_k, use_cls;    /* kind 1: conv_k = segment_size, or 1 for embedding_type 'conv1d' */
  int num_classes, d_model, n_head, n_layers, ffn_hidden;
  float drop_prob;
} iq_model_cfg_t;

typedef struct iq_model iq_model_t;

int iq_model_create(const iq_model_cfg_t* cfg, iq_model_t** out);
void iq_model_destroy(iq_model_t* m);
const char* iq_model_last_error(const iq_model_t* m);
int iq_model_tokens(const iq_model_t* m);   /* S, cls included */
/* flat fp32 parameter buffer layout: entries in reference state_dict naming */
size_t iq_model_param_floats(const iq_model_t* m);
int iq_model_param_entries(const iq_model_t* m);
int iq_model_param_entry(const iq_model_t* m, int i, char* name, int name_cap, size_t* offset, int* ndim, int* dims);
size_t iq_model_shadow_bytes(const iq_model_t* m);
size_t iq_model_workspace_bytes(const iq_model_t* m, int batch, int training);
/* bind device buffers (caller-owned, must outlive use): params/grads flat fp32, pe fp32 [S,D], shadow bytes */
int iq_model_bind(iq_model_t* m, float* params, float* grads, const float* pe, void* shadow);
/* Optional: a caller-owned persistent DEVICE u32 that holds the dropout step.  When bound, iq_model_forward with
 * step == 0xFFFFFFFF increments it on the device (hipGraph replays then draw fresh masks), any other step value is
 * written into it; backward regenerates masks from it.  Unbound, a slot of the workspace is used (uninitialised
 * after every workspace reallocation: bind a counter for reproducible masks under graph replay). */
int iq_model_bind_step_counter(iq_model_t* m, uint32_t* counter);
int iq_model_refresh_shadow(iq_model_t* m, iq_stream_t stream);
/* same, minus the flat fp32->bf16 mirror (iq_adamw_step has just written it) */
int iq_model_refresh_transposed(iq_model_t* m, iq_stream_t stream);
/* forward: src fp32 (B,C,H,W)/(B,C,L); enc_out fp32 [B,S,D] or NULL; logits fp32 [B,K] or NULL */
int iq_model_forward(iq_model_t* m, const float* src, int batch, void* workspace, size_t ws_bytes, int training,
                     uint64_t seed, uint32_t step, float* enc_out, float* logits, iq_stream_t stream);
/* backward of the last training forward in `workspace`: dlogits fp32 [B,K] and/or denc fp32 [B,S,D].
 * Gradients are WRITTEN (accumulate=0) or added into the bound flat grad buffer.
 * layer_hi/layer_lo select a slice of the chain for comm overlap: stage ids run
 * n_layers+1 (head) ... 1 (layer 0) ... 0 (embedding); call with (n_layers+1, 0) for everything. */
int iq_model_backward(iq_model_t* m, const float* dlogits, const float* denc, int batch, void* workspace,
                      size_t ws_bytes, int accumulate, int stage_hi, int stage_lo, iq_stream_t stream);
/* Gradient exchange: SURVEY.md 8(b) lists iq_comm_{init, allreduce_bucket, finalize} among the entry points.  They are
 * deliberately NOT part of this library: BASELINE.json's north_star keeps the host in PyTorch-ROCm, whose
 * torch.distributed (backend "nccl" = RCCL over xGMI) already owns communicators, streams and the process group.  The C
 * ABI ends at "this contiguous range of the flat gradient is complete on the stream" (iq_model_backward by stages +
 * iq_model_grad_range below); vit-vs-raw-iq_amd/trainer.py issues one asynchronous all-reduce per range. */
/* flat-gradient range [*off, *off+*len) written by stages [stage_lo, stage_hi] (DDP buckets) */
int iq_model_grad_range(const iq_model_t* m, int stage_hi, int stage_lo, size_t* off, size_t* len);

/* ---------------------------------------------------------------------------------------
 * Measurement aid (bench.py roofline leg): when enabled, every entry point above brackets its
 * launches with a HIP event pair on the launch stream.  iq_prof_collect synchronises and returns, per
 * kernel family, the summed elapsed milliseconds and the number of bracketed calls.
 * Families: 0 gemm_nt, 1 wgrad (+slab reduce), 2 attn_fwd, 3 attn_bwd, 4 ln_fwd, 5 ln_bwd,
 *           6 misc (embedding, head, loss, casts), 7 optimizer (gradnorm, adamw). */
#define IQ_PROF_FAMILIES 8
int iq_prof_enable(int on);
int iq_prof_collect(double* ms, long long* count);
/* Per-KERNEL sums of everything iq_prof_collect has gathered since the last reset: one text line per kernel name (the
 * name rocprofv3 --kernel-trace --stats prints, without its namespace):
 *   name \t family \t launches \t total ms \t total algorithmic bytes \t total flops \n
 * (algorithmic bytes / flops of a launch are computed at its launch site from the problem sizes: operands read once,
 * results written once).  Returns the length of the full text; writes at most cap-1 bytes + a terminating 0. */
size_t iq_prof_kernels(char* out, size_t cap, int reset);

#ifdef __cplusplus
}
#endif
#endif /* IQVIT_H */
