/*
 * gvd_hip.h — C-ABI of libgvd_hip.so: the MI355X (gfx950) kernels of the GVD decode/train hot path.
 *
 * The reference (facebookresearch/grounded-video-description) has no FFI: its hot path is a chain of
 * implicit ATen ops behind the Python module API misc.AttModel.TopDownModel.  This
 * library is what a maintainer binds (ctypes) to replace those ATen op chains.
 * Every entry point cites the reference lines whose arithmetic it replaces.
 *
 * Conventions
 *   - plain C: device pointers + sizes + a hipStream_t; no torch types.  All floating point is fp32.
 *   - the caller owns all memory (inputs, outputs, workspaces); the library never allocates or frees
 *     device memory and keeps no state between calls: every function is re-entrant and may be called
 *     concurrently from several host threads on different streams/devices (nn.DataParallel, main.py:655).
 *   - return value: 0 on success, a hipError_t (> 0) when a launch failed, GVD_EINVAL (-1) when the
 *     shapes/alignment are outside what the kernels support.  No exceptions cross the ABI.
 *   - launches are asynchronous on `stream`; nothing synchronises with the host.
 *   - row-major, innermost dimension contiguous; `ld*` are leading dimensions in ELEMENTS.
 *   - masks are uint8 (0/1) exactly as the reference's dataloader produces them (dataloader_anet.py:336-354).
 */
#ifndef GVD_HIP_H
#define GVD_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* gvd_stream_t; /* == hipStream_t */

#define GVD_EINVAL (-1)
#define GVD_MIN_VALUE (-1e8f) /* AttModel.py:31,66; model.py:71 */

/* library / build identification (also lets tests prove the HIP library, not a fallback, is loaded).
 * GVD_ABI_VERSION changes whenever a struct layout or signature below changes; the Python binding refuses a
 * library whose gvd_abi_version() differs from the version it was written against (hip.ABI_VERSION). */
#define GVD_ABI_VERSION 20
const char* gvd_version(void);
int gvd_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Kernel timing: pairs of HIP events recorded on the launch stream around the dominant kernel, so a
 * benchmark can read that kernel's launch durations live over its timed region (bench.py `roofline`).
 * ------------------------------------------------------------------------------------------- */
typedef struct gvd_prof gvd_prof;
gvd_prof* gvd_prof_create(int max_pairs);
void gvd_prof_destroy(gvd_prof* p);
void gvd_prof_reset(gvd_prof* p);
/* call after synchronising the stream: sum of elapsed ms over the recorded pairs, and their count */
int gvd_prof_read(gvd_prof* p, float* total_ms, int* count);
/* per pair: ms[i] and (tags may be NULL) the 8 int64 words the launcher attached to pair i - for the pipelined GEMM:
 * {M, N, K, batch, a_kstrided, w_kstrided, has device row count, has addend}.  Returns the pairs written (<= max_pairs) or a
 * negative error.  Call after synchronising the stream. */
int gvd_prof_read_pairs(gvd_prof* p, float* ms, int64_t* tags, int max_pairs);

/* ---------------------------------------------------------------------------------------------
 * Dense projections (MFMA fp32, v_mfma_f32_32x32x2_f32: exact fp32 fma chains)
 * ------------------------------------------------------------------------------------------- */

/* One K-segment of a "NT" product: A[M,K] (lda) times W[N,K]^T (ldw).  K % 32 == 0, 16-B aligned. */
typedef struct {
  const float* A; int64_t lda; int64_t a_batch_stride;
  const float* W; int64_t ldw; int64_t w_batch_stride;
  int K;
} gvd_gemm_seg;

/* C[b][M,N] = act( sum_s A_s[b] W_s[b]^T + nbias[N] + nbias2[N] + mbias[M] + rowbias[b][M,N] ), then
 * optional masked fill with GVD_MIN_VALUE where mask[b][m*mask_ldm + n] != 0.
 * Replaces nn.Linear / torch.matmul call sites:
 *   fc7 `ctx2pool_grd` + ReLU   model.py:158-161,312   (M=B*R, N=2048, K=2048, act=1)
 *   `ctx2pool` / `ctx2att`      model.py:391,405       (N=512, K=1024)
 *   `h2att`                     AttModel.py:39,77      (M=B, N=512(+512 stacked), K=1024)
 *   `logit`                     model.py:464,615       (N=V, K=1024)
 *   `_grounder` dot branch      model.py:262-278       (batched over B: A=vis words, W=g_pool[b])
 */
typedef struct {
  gvd_gemm_seg seg[3]; int nseg;
  const float* nbias;            /* [N] or NULL */
  const float* nbias2;           /* [N] or NULL */
  const float* mbias;            /* [M] or NULL; mbias_batch_stride elements between batches */
  int64_t mbias_batch_stride;
  const float* rowbias;          /* [M,N] (ld = rowbias_ld) or NULL */
  int64_t rowbias_ld; int64_t rowbias_batch_stride;
  const uint8_t* mask;           /* or NULL */
  int64_t mask_ldm; int64_t mask_batch_stride;   /* mask_ldm may be 0 (broadcast over rows) */
  float* C; int64_t ldc; int64_t c_batch_stride;
  int M, N, batch;
  int act;                       /* 0 = identity, 1 = ReLU */
  const int* m_dev;              /* optional: device int holding the live row count (<= M); rows / tiles past it are skipped */
  const int* a_row_map;          /* optional [M] (single segment, plain operands, >= 256-tile products): output row m reads
                                    row a_row_map[m] of A (a_src_rows rows, a_src_rows * lda * 4 < 2^32) - a row gather
                                    fused into the operand loads (fc7 over the compacted proposal set) */
  int64_t a_src_rows;
  /* K-strided operands (the backward products of nn.Linear; single segment, large shapes only):
   *   w_kstrided: W is given as [K, N] (ldw >= N, N % 4 == 0)            dX[M,K'] = dY[M,N'] W[N',K']
   *   a_kstrided (with w_kstrided): A is given as [K, M] (lda >= M)       dW[N',K'] = dY[M',N']^T X[M',K']
   * a batch over contraction chunks (a/w_batch_stride = chunk rows * ld) gives a deterministic split-K: the caller sums
   * the partial C slabs. */
  int a_kstrided, w_kstrided;
  /* Two-level batch (batch_inner > 1): batch index b = outer * batch_inner + inner; operand / output bases are
   * outer * (a|w|c)_batch_stride + inner * (a|w|c)_inner_stride (e.g. outer = sample, inner = attention head living at a
   * column offset of a packed [B,R,heads*HP] tensor).  mbias / rowbias / mask must be NULL then.  0 or 1 = flat batch. */
  int batch_inner;
  int64_t a_inner_stride, w_inner_stride, c_inner_stride;
} gvd_gemm_args;

int gvd_gemm_nt_f32(const gvd_gemm_args* args, gvd_stream_t stream);

/* Measurement hook (bench.py `roofline_mfma`): while `prof` is non-NULL every launch of the pipelined fp32-MFMA GEMM kernel
 * (csrc/gemm_pipe.hip) is bracketed by an event pair of `prof` on its stream, and 2 x rows x N x K x batch flops - rows read
 * from the launch's device-side row count where it has one - are added to the device double *dev_flops (may be NULL), and
 * launch i's live row count is written to dev_rows[i] (i < n_rows; may be NULL) for bench.py's per-shape table.
 * Process-global, not thread-safe, off by default; pass NULL to disarm.  No reference counterpart (instrumentation). */
int gvd_gemm_prof_set(gvd_prof* prof, double* dev_flops, int* dev_rows, int n_rows);

/* nn.LSTMCell forward (AttModel.py:121,123,139,160): gates = sum_s X_s W_s^T + b_ih + b_hh (+ rowbias),
 * gate order i,f,g,o; c' = sig(f) c + sig(i) tanh(g); h' = sig(o) tanh(c').  The gate GEMM and the
 * pointwise epilogue are one kernel.  `seg[s].W` points at the column block of weight_ih / weight_hh
 * that multiplies X_s (all [4H, *] row-major), so concatenated inputs ([fc|xt], [att+att2|h_att]) never
 * need to be materialised.  h_prev may alias nothing written here (h_out/c_out must be distinct buffers).
 * gates_out (optional, [B,4H]): post-activation i,f,g,o kept for the backward pass. */
typedef struct {
  gvd_gemm_seg seg[3]; int nseg;
  const float* b_ih; const float* b_hh;       /* [4H] each, or NULL */
  const float* rowbias; int64_t rowbias_ld;   /* [B,4H] precomputed constant part (e.g. fc W_ih[:, :H]^T), or NULL */
  const float* c_prev; int64_t ldc_prev;      /* [B,H] */
  float* h_out; int64_t ldh;                  /* [B,H] */
  float* c_out; int64_t ldc_out;              /* [B,H] */
  float* gates_out; int64_t ldg;              /* [B,4H] or NULL */
  int B, H;
} gvd_lstm_args;

int gvd_lstm_cell_fwd(const gvd_lstm_args* args, gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Additive visual attention (the HBM-bound north-star kernel)
 * ------------------------------------------------------------------------------------------- */

/* One attention "side": Attention2 over regions (AttModel.py:71-108) or Attention over frames
 * (AttModel.py:33-53; masks NULL).
 *   e[n] = w . tanh(p_feats[b,n,:] + q[b,:]) + *alpha_bias ; e[att_mask] = -1e8
 *          (score_mode GVD_SCORE_ADD: `region_attn_mode` 'mix', and always the frame side;
 *           GVD_SCORE_MUL 'mix_mul': w . tanh(p_feats * q) + *alpha_bias, AttModel.py:82-83;
 *           GVD_SCORE_DOT 'dp': p_feats . q - no alpha_net, w / alpha_bias NULL, AttModel.py:92-95)
 *   alpha = softmax_n(e) ; ctx[b,:] = sum_n alpha[n] feats[b,n,:]
 *   logits_out[b,n] = e[n] with [pnt_mask] = -1e8          (= `att2_weight`, pre-softmax)
 */
typedef struct {
  const float* feats;    /* [B,N,H]  (att_feats / pool_feats / conv_feats) */
  const float* p_feats;  /* [B,N,A]  (projected: p_pool_feats / p_conv_feats) */
  const float* q; int64_t ldq;            /* [B,A] = h2att(h) incl. bias */
  const float* w;                          /* [A]  alpha_net.weight */
  const float* alpha_bias;                 /* device scalar alpha_net.bias */
  const uint8_t* att_mask; int64_t ld_att_mask;   /* [B,N] or NULL */
  const uint8_t* pnt_mask; int64_t ld_pnt_mask;   /* [B,N] or NULL */
  float* logits_out; int64_t ld_logits;           /* [B,N] or NULL */
  float* scores_out; int64_t ld_scores;           /* [B,N] or NULL: e[n] before the pnt_mask fill (kept for backward) */
  const int* row_map;    /* optional [B,N] (the per-row attention kernel only): row n of sample b is row row_map[b*N+n] of
                            the FLAT arrays feats [rows,H] / p_feats [rows,A] - the compacted preamble's features are
                            consumed in place, no dense [B,N,.] copy (csrc/compact.hip); NULL = dense layout */
  int N;
  int group;   /* 0/1: feats/p_feats have one entry per row b.  K>1: rows b share entry b/K (the K beams of a
                  sample attend over ONE copy of its features: feats/p_feats are [B/K,N,*]) */
  int score_mode;   /* GVD_SCORE_ADD (0) / GVD_SCORE_MUL / GVD_SCORE_DOT: the score function above (region side only; the
                       temporal side of a launch must be GVD_SCORE_ADD) */
} gvd_attn_side;
#define GVD_SCORE_ADD 0
#define GVD_SCORE_MUL 1
#define GVD_SCORE_DOT 2

/* y[i] = the kernels' tanh (csrc/gvd_common.h: tanh_fast = 1 - 2 / (1 + 2^(2 log2(e) x)) on v_exp_f32 / v_rcp_f32,
 * absolute error <= 2.5e-7) - exported so that the error bound the score kernels rely on is measured on the device
 * (tests/test_gpu_kernels.py::test_tanh_fast_error_bound).  Replaces torch.tanh inside Attention.forward /
 * Attention2.forward (AttModel.py:45, 90). */
int gvd_tanh_fast_f32(const float* x, float* y, int64_t n, gvd_stream_t stream);

/* Both attentions of one decoder step in one pass over HBM.
 * out_sum[b,:] = ctx_region + ctx_temporal (the `att+att2` input of the language LSTM, AttModel.py:148);
 * ctx_region / ctx_temporal (optional) receive the two contexts separately.  `temporal` may be NULL.
 * Workspace: gvd_attn_workspace_bytes(B, Nr, Nt, H) bytes, 16-B aligned. */
size_t gvd_attn_workspace_bytes(int B, int n_region, int n_temporal, int H);
int gvd_attn_fwd(const gvd_attn_side* region, const gvd_attn_side* temporal, int B, int A, int H,
                 float* out_sum, int64_t ld_out, float* ctx_region, float* ctx_temporal,
                 void* workspace, gvd_stream_t stream);
/* same, recording one event pair around the streaming (partial) kernel when prof != NULL */
int gvd_attn_fwd_prof(const gvd_attn_side* region, const gvd_attn_side* temporal, int B, int A, int H,
                      float* out_sum, int64_t ld_out, float* ctx_region, float* ctx_temporal,
                      void* workspace, gvd_prof* prof, gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused row kernels of the per-segment preamble (inference path)
 * ------------------------------------------------------------------------------------------- */

/* out[row,:] = gamma * (s - mean(s)) / (std_unbiased(s) + eps) + beta with s = x[row,:] + y[row,:] (y may be NULL):
 * ResidualBlock + the encoder's custom LayerNorm (transformer.py:66-88).  D must be 1024. */
/* rows_dev (here and below): optional device int with the live row count (<= rows), for the compacted preamble. */
int gvd_add_layernorm_unbiased(const float* x, const float* y, const float* gamma, const float* beta, float* out,
                               int64_t rows, const int* rows_dev, int D, float eps, gvd_stream_t stream);

/* Backward of gvd_add_layernorm_unbiased: ds[row,:] = d loss / d (x + y)[row,:] (the gradient of both addends), and
 * partials [gvd_add_layernorm_unbiased_bwd_parts(rows), 2, D]: per-workgroup sums of dgamma (first D) and dbeta (second D),
 * to be added over the first axis by the caller (ordered: reproducible).  Statistics are recomputed from x (+ y). */
int gvd_add_layernorm_unbiased_bwd_parts(int64_t rows);
int gvd_add_layernorm_unbiased_bwd(const float* x, const float* y, const float* dout, const float* gamma, float* ds,
                                   float* partials, int64_t rows, int D, float eps, gvd_stream_t stream);

/* Training form of the ResidualBlock (transformer.py:79-88: `layernorm(x + dropout(layer(x)))`): the branch dropout is
 * applied while y streams through the row kernel, s = x + y * keep / (1 - p_drop), keep ~ Bernoulli(1 - p_drop) from
 * Philox4x32-10 (one 128-bit block per 4 consecutive elements of the contiguous [rows, D] tensor, key = seed) - the mask
 * gvd_dropout_rows draws for the same seed.  The backward regenerates the mask, recomputes the statistics from
 * (x, y, seed) and writes the two addends' gradients: ds (x) and dy = ds * keep / (1 - p_drop) (the branch).
 * 0 < p_drop < 1 (without dropout: the functions above). */
int gvd_add_layernorm_unbiased_drop(const float* x, const float* y, const float* gamma, const float* beta, float* out,
                                    int64_t rows, int D, float eps, float p_drop, uint64_t seed, gvd_stream_t stream);
int gvd_add_layernorm_unbiased_drop_bwd(const float* x, const float* y, const float* dout, const float* gamma, float* ds,
                                        float* dy, float* partials, int64_t rows, int D, float eps, float p_drop,
                                        uint64_t seed, gvd_stream_t stream);

/* Training path of the encoder's self-attention core (transformer.py:90-117: softmax(Q K^T / sqrt d) -> dropout -> @ V)
 * over the padded training layout: qkv f32 [B, Rp, ld >= 3 * n_heads * head_pad] = packed q | k | v with every head in its
 * own zero-padded head_pad-column slot (head_pad = 176), Rp % 32 == 0, rows >= R of a sample are padding.
 *
 * forward (flash-style, csrc/flash_attn_pad.hip): o f32 [B, Rp, ldo >= n_heads * head_pad] (rows < R written, pad columns
 *   zero), lse f32 [B * n_heads, Rp] = log2-domain logsumexp of every query's scaled + biased scores.  key_bias (nullable)
 *   f32 [B, Rp] is added to the scaled scores of a key for every query of the sample (log n: the key stands for n identical
 *   keys; -inf: no such key; the compacted training layout, train_compact.py).  Dropout of probability p_drop on the
 *   attention weights from a counter-based hash of (seed, map row, key) (csrc/enc_dropout.h); no [B, heads, R, R] map is
 *   written.
 * backward maps (csrc/enc_attn_bwd.hip): delta f32 [B * n_heads, Rp] <- rowsum(dO * O), then per (sample, head) the two
 *   products S = Q K^T and dY = dO V^T in one kernel whose epilogue recomputes P = exp2(c S + bias - lse), re-evaluates the
 *   keep mask and writes Pd = P * keep / (1 - p) and dS = scale * P * (dY * keep / (1 - p) - delta), both
 *   f32 [B * n_heads, Rp, Rp] with rows / columns >= R zero - the operands of dV = Pd^T dO, dQ = dS K, dK = dS^T Q
 *   (gvd_gemm_nt_f32 with K-strided operands).  B * n_heads <= 65535.
 * sample_rows (ABI 19) = rows between consecutive samples in qkv / o / dO: Rp on the padded layout, or R itself - the
 *   region rows of the batch packed back to back ([B * R, ld], what the Linear layers around the core read and write, no pad
 *   rows anywhere): only the per-(sample, head) statistics and the two maps keep the padded pitch Rp.  Rows past R of a
 *   sample are never written; the K-strided products that contract over Rp rows read up to Rp - R rows past a sample's last
 *   one (the next sample's, or - last sample - slack the caller provides: finite values, they meet exact zeros of the maps).
 * scores (ABI 20, nullable on both sides): f32 [B * n_heads, Rp, Rp], the log2-domain scaled + biased score of every (query,
 *   key) pair as the forward's matrix cores produced it (rows >= R of a map are not written).  Given to the forward AND to
 *   the backward maps, the latter LOADS S instead of multiplying Q K^T again (one product instead of two, and the backward's
 *   probabilities are those of the forward to the last bit); NULL on either side: the backward recomputes the product. */
int gvd_flash_attn_train_fwd_f32(const float* qkv, int64_t ld, float* o, int64_t ldo, float* lse, float* scores_out, int B,
                                 int Rp, int R, int sample_rows, int n_heads, int head_pad, float scale,
                                 const float* key_bias, float p_drop, uint64_t seed, gvd_stream_t stream);
int gvd_enc_attn_bwd_maps(const float* qkv, int64_t ld, const float* dO, const float* O, int64_t ldo, const float* lse2,
                          const float* key_bias, const float* scores, float* delta, float* Pd, float* dS, int B, int Rp,
                          int R, int sample_rows, int n_heads, int head_pad, float scale, float p_drop, uint64_t seed,
                          gvd_stream_t stream);
/* Test aid: the keep mask of that dropout, u8 [n_maps, Rp, Rp] (1 = kept), map row = map * Rp + query. */
int gvd_enc_dropout_mask(uint8_t* out, int64_t n_maps, int Rp, float p_drop, uint64_t seed, gvd_stream_t stream);

/* Per proposal row (model.py:336-364): p = softmax over the n_cls similarity logits (all -1e8 when the row is
 * masked: row_mask[(row / mask_rows_per_batch) * mask_ld + row % mask_rows_per_batch] != 0), written to sim_out
 * [rows,n_cls] (optional; logits rows are logits_ld >= n_cls apart); out[row] = [layer_norm(g_pool row, G=2048) | layer_norm(loc row, n_loc) |
 * layer_norm(p, n_cls) | zeros] with out leading dimension out_ld >= G + n_loc + n_cls (the zero pad makes the row a
 * 16-byte-aligned, 32-multiple K operand for the pool_embed GEMM, model.py:384).  F.layer_norm semantics (biased
 * variance, eps inside the sqrt, no affine). */
int gvd_region_feature_rows(const float* g_pool, const float* loc, int n_loc, const float* sim_logits, int n_cls,
                            int64_t logits_ld, const uint8_t* row_mask, int64_t mask_rows_per_batch, int64_t mask_ld,
                            float* out, int64_t out_ld, float* sim_out, int64_t rows, const int* rows_dev, int G,
                            float ln_eps, gvd_stream_t stream);

/* Backward of gvd_region_feature_rows for the training path (autograd of model.py:336-364): from d_out [rows,d_out_ld]
 * (gradient of the concatenated row; d_out_ld a multiple of 4, 16-byte aligned) and the optional direct gradient d_sim
 * [rows,n_cls] of the class distribution (the region-classification loss reads it), with sim = the distribution the
 * forward wrote: d_gpool [rows,G], d_loc [rows,n_loc], d_logits [rows,d_logits_ld] (columns >= n_cls zeroed; masked rows
 * zero: their logits were the constant -1e8).  Layer-norm statistics are recomputed from the inputs. */
int gvd_region_feature_rows_bwd(const float* g_pool, const float* loc, int n_loc, const float* sim, int n_cls,
                                const uint8_t* row_mask, int64_t mask_rows_per_batch, int64_t mask_ld,
                                const float* d_out, int64_t d_out_ld, const float* d_sim, float* d_gpool, float* d_loc,
                                float* d_logits, int64_t d_logits_ld, int64_t rows, int G, float ln_eps,
                                gvd_stream_t stream);

/* Fused multi-head self-attention of the obj_interact encoder (transformer.py:90-123), flash-style in fp32 on the matrix
 * cores (no [B,R,R] score maps in HBM), over PADDED heads (the inference path of the encoder): head h of q, k, v occupies
 * columns [h*head_pad, (h+1)*head_pad) of rows with stride ld, real columns first, pad columns exactly zero (the fused
 * QKV projection against row-permuted, zero-padded weights writes them that way), so every head is 16-byte aligned.
 * o: [B,R,ldo] in the same padded layout (pad columns come out zero).  scores = scale * q.k (scale = 1/sqrt(d_model),
 * transformer.py:92,104).  head_pad must be 176; ld, ldo multiples of 4; all pointers 16-byte aligned. */
int gvd_flash_attn_padded_f32(const float* q, const float* k, const float* v, int64_t ld, float* o, int64_t ldo, int B,
                              int R, int n_heads, int head_pad, float scale, const int* row_off,
                              const float* last_key_log2_weight, void* workspace, gvd_stream_t stream);
/* workspace: ragged form only (NULL otherwise) - gvd_flash_attn_workspace_bytes(B, R) bytes, 4-byte aligned: the tile map
 * the launch builds on the device from row_off, so that the grid (sized for R rows per sample) keeps its live workgroups
 * in front and walks them in (sample, head, query tile) order. */
size_t gvd_flash_attn_workspace_bytes(int B, int R);
/* Ragged form (row_off != NULL, the compacted preamble below): sample b owns rows row_off[b] .. row_off[b+1]-1 (<= R of
 * them) of q/k/v/o, and its LAST row stands for n identical rows: as a key its score gets + last_key_log2_weight[b]
 * (= log2 n in the kernel's log2 domain; -inf = no such rows, the key is ignored).  Both arrays live on the device. */

/* ---------------------------------------------------------------------------------------------
 * Masked-proposal compaction of the per-segment preamble (csrc/compact.hip has the derivation): masked proposals are
 * zeroed by the loader (dataloader_anet.py:343-344), so all masked rows of a segment are one row as far as the
 * preamble is concerned; it runs on [valid rows | one representative] per segment and is expanded at the end.
 * ------------------------------------------------------------------------------------------- */

/* mask: u8 [B, >= R] (ld_mask between segments; pass pnt_mask + 1 to skip the legacy pad column), != 0 = masked.
 * Outputs (device): off i32 [B+1] first compact row of each segment, off[B] = total compact rows; nvalid i32 [B];
 * src_row i32 [B*(R+1)] dense row (b*R + r) of each compact row (representative = the segment's first masked row);
 * cidx i32 [B*R] compact row of each dense row (masked rows -> the representative); rep_w f32 [B] = log2(#masked)
 * or -inf; cmask u8 [B*(R+1)] = 1 for representative rows. */
int gvd_compact_index(const uint8_t* mask, int64_t ld_mask, int B, int R, int* off, int* nvalid, int* src_row, int* cidx,
                      float* rep_w, uint8_t* cmask, gvd_stream_t stream);
/* out[i, 0:D] = in[idx[i], 0:D] for i < n_rows (or < *n_rows_dev when given): compaction of inputs, expansion of outputs */
int gvd_gather_rows_f32(const float* in, int64_t in_ld, const int* idx, float* out, int64_t out_ld, int D, int64_t n_rows,
                        const int* n_rows_dev, gvd_stream_t stream);
/* Small fused kernels of the inference preamble (each replaces a chain of 4-10 ATen launches; they matter at batch_size = 4):
 *  gvd_fc_feature: out[b, 0:D] = layer_norm(mean_t segs[b,t,:]), out[b, D:D+S] = layer_norm(relu(w_seg num[b,3:7] + b_seg)),
 *    out[b, D+S:ldo] = 0 (the K pad of the fc_embed GEMM) - model.py:306-308; segs f32 [B,Ft,D], num i64 [B,7], w_seg [S,4].
 *  gvd_loc_features: out[i] = [x1,y1,x2,y2]/720, frame/n_frames, 0 pad to ldo columns, of proposal row src_row[i] (or i) for
 *    i < rows (or < *rows_dev) - model.py:357-360 on the compacted row set; ppls f32 [*,7].
 *  gvd_affine_relu_rows: x[r, c] = relu(x[r, c] * scale[c] + shift[c]) in place - BatchNorm1d(eval) + ReLU of the frame
 *    embeddings as a per-channel affine of the last axis (model.py:114,397).
 *  gvd_zero_rows_outside_window: x[b, t, :] = 0 for t outside [sample_idx[b,0], sample_idx[b,1]) - model.py:303-305,401. */
int gvd_fc_feature(const float* segs, const int64_t* num, const float* w_seg, const float* b_seg, float* out, int B, int Ft,
                   int D, int S, int ldo, float eps, gvd_stream_t stream);
int gvd_loc_features(const float* ppls, const int* src_row, const int* rows_dev, float* out, int64_t rows, int ldo,
                     float n_frames, gvd_stream_t stream);
int gvd_affine_relu_rows(float* x, const float* scale, const float* shift, int64_t rows, int D, gvd_stream_t stream);
int gvd_zero_rows_outside_window(float* x, const int64_t* sample_idx, int B, int Ft, int D, gvd_stream_t stream);
/* flag[0] |= 1 when some masked row of x [B*R, D] is not all-zero (the precondition of the compaction) */
int gvd_check_masked_rows_zero(const float* x, int D, const uint8_t* mask, int64_t ld_mask, int B, int R, int* flag,
                               gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Frame-wise context encoder: one bidirectional GRU layer as a persistent cooperative kernel
 * (nn.GRU(1024, 512, 2, bidirectional, batch_first), model.py:150-154,399; gate order r,z,n).
 * gi [B,T,2,3*Hh] = X [W_ih_fw ; W_ih_bw]^T + [b_ih_fw ; b_ih_bw] (one gvd_gemm_nt_f32 call, N = 6*Hh);
 * out [B,T,2*Hh] = [h_fw | h_bw] exactly like torch's batch_first bidirectional output.  Hh must be 512.
 * ------------------------------------------------------------------------------------------- */
/* Size (uint32 words) of one grid-barrier object used by the persistent kernels of this library. */
int gvd_grid_sync_words(void);

/* sync_ws: NULL -> grid-wide synchronisation by the HIP cooperative-groups library; otherwise a device buffer of
 * gvd_grid_sync_words()*ceil(B/256) uint32 ZEROED by the caller before every call -> hand-rolled two-level counter
 * barrier (no cache fences: the recurrent state is exchanged with agent-coherent accesses; bounded spin).  After
 * the call word [32] of slice i's object != 0 means slice i timed out (results invalid). */
int gvd_gru_bidir_layer(const float* gi, const float* w_hh_fw, const float* b_hh_fw, const float* w_hh_bw,
                        const float* b_hh_bw, float* out, int B, int T, int Hh, void* sync_ws,
                        gvd_stream_t stream);

/* `--t_attn_mode bilstm` (opts.py:60; model.py:145-149,399): one bidirectional LSTM layer of the frame-wise context encoder
 * nn.LSTM(1024, 512, 2, bidirectional, batch_first) as a persistent kernel (csrc/lstm_seq.hip; gate order i,f,g,o).
 * gi [B,T,2,4*Hh] = X [W_ih_fw ; W_ih_bw]^T + [b_ih_fw ; b_ih_bw] (one gvd_gemm_nt_f32 call, N = 8*Hh); out [B,T,2*Hh] =
 * [h_fw | h_bw]; c_state [B,2,Hh]: scratch for the running cell state (any content).  Training: gates_seq [B,T,2,4*Hh]
 * (post-activation i,f,g,o of every step) and c_seq [B,T,2*Hh] (cell state after every step, laid out like out) for the
 * BPTT through gvd_lstm_cell_bwd; NULL at inference.  sync_ws: as gvd_gru_bidir_layer (required; zeroed by the caller;
 * word [32] != 0 after the call = barrier timeout, results invalid).  Hh must be 512. */
int gvd_lstm_bidir_layer(const float* gi, const float* w_hh_fw, const float* b_hh_fw, const float* w_hh_bw,
                         const float* b_hh_bw, float* out, float* c_state, float* gates_seq, float* c_seq, int B, int T,
                         int Hh, void* sync_ws, gvd_stream_t stream);

/* One reverse step of a layer's BPTT, both directions at once (training backward of nn.GRU, model.py:150-154,399).
 * Direction 0 is at time t_fw (walking T-1..0), direction 1 at t_bw (walking 0..T-1).  gi / gh / d_gi / d_gh are
 * [B,T,2,3*Hh] (gh = h_{t-1} W_hh^T + b_hh for every step, formed by one GEMM per direction from the layer output),
 * dout / out [B,T,2*Hh], carry_mm / carry_z [2,B,Hh]: carry_mm = d_gh[previous step] W_hh (caller's GEMM), carry_z is
 * read and rewritten (dh z); first != 0 ignores both carries.  Gate order r,z,n. */
int gvd_gru_bwd_step(const float* dout, const float* gi, const float* gh, const float* out, const float* carry_mm,
                     float* carry_z, float* d_gi, float* d_gh, int B, int T, int Hh, int t_fw, int t_bw, int first,
                     gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the teacher-forced decoder loop (hand-scheduled BPTT; replaces autograd's per-op backward of
 * AttModel.py:33-53,71-108,138-160).  The dX / dW products are plain library GEMMs done by the caller.
 * ------------------------------------------------------------------------------------------- */

/* Pointwise backward of nn.LSTMCell: from dh, dc_next (nullable) and the saved post-activation gates
 * (i,f,g,o), c_prev, c_new -> d(pre-activation gates) [B,4H] and dc_prev [B,H].  dh2 (nullable): a second addend of the
 * hidden-state gradient (the recurrent contribution of step t+1), added to dh on load. */
int gvd_lstm_cell_bwd(const float* dh, int64_t lddh, const float* dh2, int64_t lddh2, const float* dc_next, int64_t lddc,
                      const float* gates, int64_t ldg, const float* c_prev, int64_t ldcp, const float* c_new, int64_t ldcn,
                      int B, int H, float* dgates, int64_t lddg, float* dc_prev, int64_t lddcp, gvd_stream_t stream);

/* One attention side, one step: streams feats/p_feats once.  alpha = softmax weights of the step [B,N]
 * (from the saved scores), ctx = the side's context [B,H], d_ctx / d_logits = incoming gradients.
 * Writes de_out[b,n] = dLoss/d e[n] and per-chunk partial sums dq_part/dw_part [B,NC,A], dab_part [B,NC]
 * with NC = gvd_attn_bwd_chunks(N, B) (deterministic: the caller sums over NC). */
int gvd_attn_bwd_chunks(int N, int B);
int gvd_attn_bwd_step(const gvd_attn_side* side, int B, int A, int H, const float* alpha, int64_t ld_alpha,
                      const float* ctx, int64_t ld_ctx, const float* d_ctx, int64_t ld_dctx,
                      const float* d_logits, int64_t ld_dlogits, float* de_out, int64_t ld_de, float* dq_part,
                      float* dw_part, float* dab_part, gvd_stream_t stream);

/* After the loop: d_p_feats[b,n,:] = sum_t de_all[t][b,n] * w * (1 - tanh^2(p_feats[b,n,:] + q_all[t][b,:])).
 * q_all + t*q_step_stride + b*ldq, de_all + t*de_step_stride + b*ld_de + n.  Lc <= 40.
 * score_mode (gvd_attn_side.score_mode): GVD_SCORE_MUL -> sum_t de w (1 - tanh^2(p q)) q; GVD_SCORE_DOT -> sum_t de q
 * (w NULL).  gvd_attn_bwd_step reads the mode from its side. */
int gvd_attn_bwd_pfeats(const float* p_feats, int B, int N, int A, const float* q_all, int64_t q_step_stride,
                        int64_t ldq, const float* de_all, int64_t de_step_stride, int64_t ld_de, const float* w,
                        int Lc, float* d_p_feats, int score_mode, gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Vocabulary head
 * ------------------------------------------------------------------------------------------- */

/* Greedy token rule on one step's logits [B,V] (model.py:587-608): log_softmax, top-2, take the runner-up
 * when the winner is UNK; writes token ids (int64) and their log-probs with the given strides (so they land
 * in seq[:,t] / seqLogprobs[:,t]) and the next input embedding xt_next[b,:] = relu(embed[it[b],:])
 * (model.py:79-82,605).  Ties resolve to the lowest index. */
int gvd_logsoftmax_top2_embed(const float* logits, int64_t ld_logits, int B, int V, int unk_idx,
                              int64_t* it_out, int64_t it_stride, float* lp_out, int64_t lp_stride,
                              const float* embed, int E, float* xt_next, int64_t ld_xt,
                              gvd_stream_t stream);

/* xt[b,:] = relu(embed[it[b],:])   (model.py:79-82,428,605) */
int gvd_embed_relu(const int64_t* it, int64_t it_stride, int B, const float* embed, int E,
                   float* xt, int64_t ld_xt, gvd_stream_t stream);

/* lse[row] = logsumexp(logits[row,:]);  optionally picked[row] = logits[row, target[row]] - lse[row]
 * (= log_softmax gathered at the target: utils.py:131-132) and top-K (values as log-probs, indices) per row
 * for beam search (CaptionModelBU.py:45,125).  target/picked/topk_* may be NULL. */
int gvd_logsoftmax_rows(const float* logits, int64_t ld_logits, int rows, int V, float* lse,
                        const int64_t* target, float* picked, int topk, float* topk_val, int64_t* topk_idx,
                        gvd_stream_t stream);

/* Beam-search bookkeeping of decode step t for all samples in one launch (CaptionModelBU.py:49-96,154-166; replaces the
 * reference's per-sample host-side sort): merges the K x K candidates (ys / ix = the sorted top-K log-probs / word ids of
 * every beam row, from gvd_logsoftmax_rows), stable-sorts them by descending summed log-prob, forks the histories
 * beam_seq / beam_lps / beam_att [L,B,K] from the chosen parents, records beam_att[t] = att2_ind[parent], tracks the best
 * finished beam (best_p [B] starts at -inf; best_seq / best_lps [B,L], best_vix [B]) and writes parent [B*K] (row of the
 * recurrent state each new beam continues from) and word [B*K] (its token).  sums [B,K] is updated in place (finished
 * beams: -1000).  Only beam 0 expands at t = 0.  K <= 8. */
typedef struct {
  const float* ys; const int64_t* ix;          /* [B,K,K] */
  float* sums; const int64_t* att2_ind;        /* [B,K] */
  int64_t* beam_seq; float* beam_lps; int64_t* beam_att;   /* [L,B,K] */
  float* best_p; int64_t* best_seq; float* best_lps; int64_t* best_vix;
  int64_t* parent; int64_t* word;              /* [B*K] */
  int B, K, L, t;
} gvd_beam_step_args;

int gvd_beam_step(const gvd_beam_step_args* args, gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole greedy decode: L x (embed, att-LSTM, 2 attentions, lang-LSTM, logit, token rule) on one stream,
 * no host round trip per token (AttModel._sample, model.py:580-624, sample_max=1, beam_size=1).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  /* per-segment features from the preamble (model.py:504-568) */
  const float* fc;      /* [B,H]   fc_embed output */
  const float* conv;    /* [B,Ft,H] */
  const float* p_conv;  /* [B,Ft,A] */
  const float* pool;    /* [B,R,H] */
  const float* p_pool;  /* [B,R,A] */
  const uint8_t* pnt_mask; /* [B,R+1] (column 0 is the legacy pad, main.py:227) */
  const int* pool_row_map; /* optional [B,R]: pool / p_pool are the COMPACTED flat arrays [rows,H] / [rows,A] and this maps
                              (b, r) to their row (gvd_attn_side.row_map); B > 4 only (the persistent decode-batch kernel
                              reads the dense layout); NULL = dense */
  /* parameters (state_dict tensors) */
  const float* embed;                                  /* embed.0.weight [V,E] */
  const float *att_w_ih, *att_w_hh, *att_b_ih, *att_b_hh;    /* core.att_lstm.*  [4H,E+H],[4H,H] */
  const float *lang_w_ih, *lang_w_hh, *lang_b_ih, *lang_b_hh; /* core.lang_lstm.* [4H,2H],[4H,H] */
  const float *att1_h2att_w, *att1_h2att_b, *att1_alpha_w, *att1_alpha_b; /* core.attention.*  */
  const float *att2_h2att_w, *att2_h2att_b, *att2_alpha_w, *att2_alpha_b; /* core.attention2.* */
  const float *logit_w, *logit_b;                       /* logit.* [V,H] */
  int B, Ft, R, H, A, E, V, L, unk_idx;
  int no_persistent;   /* non-zero: always the kernel-per-op loop, never the persistent decode-batch kernel (the caller's
                          retry after that kernel's grid barrier timed out, att_model.TopDownModel) */
  /* outputs */
  int64_t* seq;        /* [B,L] */
  float* seq_logprobs; /* [B,L] */
  float* att2_weights; /* [B,L,R] masked pre-softmax logits */
  void* workspace;     /* gvd_greedy_workspace_bytes(...) bytes */
  gvd_prof* prof;      /* optional: times the attention streaming kernel of every step (multi-kernel loop only) */
  int* status;         /* optional device int: set to 0, or to 1 when a grid barrier of the persistent decode-batch
                          kernel timed out (token ids are then all -1) */
  uint64_t* trace;     /* optional device buffer of 1 + 7*L words: 100 MHz wall-clock stamps of workgroup 0 at every
                          phase boundary of the persistent decode-batch kernel (profiling aid) */
  int att_input_mode;  /* opts.py:58 `--att_input_mode` (AttModel.py:140-151): what the language LSTM is fed -
                          GVD_ATT_INPUT_BOTH (0) att + att2; GVD_ATT_INPUT_FEATMAP (1) the frame-wise context alone (the
                          region attention still runs: its logits are the grounding output); GVD_ATT_INPUT_REGION (2) the
                          region context alone - conv / p_conv are not read (may be NULL, Ft is ignored).  Non-zero modes
                          always run the kernel-per-op loop */
  int region_attn_mode; /* opts.py:63 `--region_attn_mode`: GVD_SCORE_ADD ('mix', README) / GVD_SCORE_MUL ('mix_mul') /
                           GVD_SCORE_DOT ('dp': att2_alpha_w / att2_alpha_b are not read) - the score function of the region
                           attention (gvd_attn_side.score_mode).  Non-zero modes always run the kernel-per-op loop */
} gvd_greedy_args;
#define GVD_ATT_INPUT_BOTH 0
#define GVD_ATT_INPUT_FEATMAP 1
#define GVD_ATT_INPUT_REGION 2

size_t gvd_greedy_workspace_bytes(int B, int Ft, int R, int H, int A, int E, int V);
int gvd_greedy_decode(const gvd_greedy_args* args, gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Feature ingest (dataloader_anet.py:317-344: zero-padded proposal / feature rows, masked rows zeroed)
 * ------------------------------------------------------------------------------------------- */

/* In place: x[row,:] = 0 for every row whose mask byte is non-zero.  x f32 [rows, D] with rows = batch * rows_per_batch;
 * the mask byte of row (b, r) is mask[b*mask_ld + mask_off + r] (so the model's pnt_mask [B,R+1] with its legacy pad
 * column is passed with mask_ld = R+1, mask_off = 1).  Rows that are kept are not touched. */
int gvd_zero_masked_rows(float* x, int64_t rows, int D, const uint8_t* mask, int64_t rows_per_batch, int64_t mask_ld,
                         int64_t mask_off, gvd_stream_t stream);

/* Host helper of the ingest (no GPU work): pread `rows` back-to-back file rows of `row_bytes` bytes at `file_off` into
 * destination rows `dst_stride` bytes apart (pinned staging; a column block of wider rows when dst_stride > row_bytes:
 * the <vid>_resnet.npy / _bn.npy blocks of segs_feat, dataloader_anet.py:198-206).  Returns the bytes read (the caller
 * checks == rows * row_bytes) or -errno.  Called through ctypes it runs without the GIL. */
int64_t gvd_pread_rows(int fd, int64_t file_off, void* dst, int64_t rows, int64_t row_bytes, int64_t dst_stride);

/* One file of the feature layout in one native call: open `path` (.npy, float32, C order, last dimension D), parse the header,
 * read the first min(rows_in_file, max_rows) rows into destination rows `dst_stride` bytes apart.  Returns rows_in_file
 * (*rows_read = rows copied) or < 0: -errno, or -1000 - k for a malformed / unsupported header.  Replaces np.load of
 * dataloader_anet.py:189,198-199 for the pinned-staging pipeline (no Python object per file, GIL released). */
int64_t gvd_npy_read_rows_f32(const char* path, void* dst, int64_t max_rows, int64_t D, int64_t dst_stride,
                              int64_t* rows_read);

/* Every feature file of one batch in ONE call: job i reads paths[i] like gvd_npy_read_rows_f32 (rows_file[i] = its result,
 * rows_read[i] = rows copied) on the caller + n_threads - 1 PERSISTENT native threads (created at the first call, asleep on
 * a condition variable between batches; they inherit the first caller's CPU affinity).  Returns the number of failed jobs
 * (< 0: -EINVAL).  One GIL-free call per batch instead of three per segment from Python threads.
 * mode: GVD_READ_PREAD (0) = pread / preadv straight into the destination rows; GVD_READ_MAPPED (1) = map the file, copy
 * its rows in user space, unmap (a file shorter than its header promises is -1007 in both; a file truncated by another
 * process WHILE mapped raises SIGBUS - use 0 for files that are being rewritten).  job_ns (nullable): wall nanoseconds per
 * job (timeline diagnostics).  Replaces the DataLoader workers' np.load calls (dataloader_anet.py:189,198-199). */
#define GVD_READ_PREAD 0
#define GVD_READ_MAPPED 1
int gvd_npy_read_batch_f32(const char* const* paths, void* const* dsts, const int64_t* max_rows, const int64_t* D,
                           const int64_t* dst_stride, int n, int n_threads, int mode, int64_t* rows_read, int64_t* rows_file,
                           int64_t* job_ns);

/* ---------------------------------------------------------------------------------------------
 * Training targets and losses
 * ------------------------------------------------------------------------------------------- */

/* IoU with the '+1' pixel convention, frame/proposal masking and the zero-area rules
 * (utils.py:293-297 -> bbox_transform.py:224-269; mask = frm_mask | pnt_mask[:,1:], model.py:317-318).
 * ppls [B,R,ppl_ld>=5], gt [B,K,gt_ld>=5], frm_mask u8 [B,R,K], pnt_mask u8 [B,R+1] -> overlaps f32 [B,R,K].
 * Also sim_target i64 [B,K,R] = (IoU>0.5) * cls (utils.py:299-305) when non-NULL. */
int gvd_iou_targets(const float* ppls, int ppl_ld, const float* gt, int gt_ld, const uint8_t* frm_mask,
                    const uint8_t* pnt_mask, int B, int R, int K, float* overlaps, int64_t* sim_target,
                    gvd_stream_t stream);

/* Per-step region labels and frame masks for all Lc steps at once (utils.py:307-328; model.py:431-440):
 * roi_labels f32 [B,Lc,R] = max_k(IoU * (mask_boxes[b,0,k,t+1]==0)) > 0.5
 * frm_masks  u8  [B,Lc,R+1] = [0 | no selected box on this proposal's frame] | pnt_mask */
int gvd_step_targets(const float* overlaps, const uint8_t* mask_boxes, const uint8_t* frm_mask,
                     const uint8_t* pnt_mask, int B, int R, int K, int Lp1, int Lc, float* roi_labels,
                     uint8_t* frm_masks, gvd_stream_t stream);

/* sum and count of -log_softmax(x[row,:])[n] over entries with label[row,n] != 0 (utils.py:139,142):
 * acc[0] = sum, acc[1] = count; acc must hold 2 + 2*rows floats (per-row partials follow the two totals and are added
 * in a fixed order: no atomics, bit-reproducible run to run); row_lse (optional) keeps each row's logsumexp for the
 * backward. */
int gvd_masked_lsm_loss(const float* x, int64_t ldx, const float* label, int64_t ld_label, int rows, int N,
                        float* acc, float* row_lse, gvd_stream_t stream);

/* Region-classification loss (model.py:345-350, T2): acc[0] = sum over (b,k,r) with sim_target > 0 of
 * -max(log sim_mat[b, sim_target[b,k,r], r], -100), acc[1] = their count; sim_mat f32 [B,D1,R] (class softmax) addressed
 * through its three element strides (the training path keeps it class-last in memory),
 * sim_target i64 [B,K,R]; acc holds 2 + 2*ceil(B*K*R/256) floats (ordered partials, no atomics). */
int gvd_cls_loss(const float* sim_mat, int64_t stride_b, int64_t stride_cls, int64_t stride_r, const int64_t* sim_target,
                 int B, int D1, int R, int K, float* acc, gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused elementwise passes of the training step (csrc/train_fused.hip) and the optimiser (csrc/optim.hip)
 * ------------------------------------------------------------------------------------------- */

/* y[i] = x[i] * keep[i] / (1 - p_drop) over a contiguous tensor of n floats (n % 4 == 0, 16-byte aligned; y may alias
 * x): F.dropout in training mode (model.py:312,363,384,393-395; AttModel.py:161) with the Philox mask described above. */
int gvd_dropout_rows(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, gvd_stream_t stream);

/* Backward of y = dropout(relu(z)) (z = a Linear's output, e.g. model.py:312 `ctx2pool_grd`) from y alone - y > 0 iff
 * z > 0 and the element was kept: dz[m,n] = y[m,n] > 0 ? dy[m,n] / (1 - p_drop) : 0, and partials
 * [gvd_relu_dropout_bwd_parts(M), N] = per-workgroup column sums of dz (the bias gradient: the caller adds them over the
 * first axis, in order).  dy, y, dz: contiguous [M, N], N % 4 == 0.  p_drop = 0: plain ReLU backward + bias gradient. */
int gvd_relu_dropout_bwd_parts(int64_t M);
int gvd_relu_dropout_bwd_colsum(const float* dy, const float* y, float* dz, float* partials, int64_t M, int N,
                                float p_drop, gvd_stream_t stream);

/* out[b, 0:A] = sum_c a[b,c,:], out[b, A:2A] = sum_c r[b,c,:]  (a [B,nca,A], r [B,ncr,A] contiguous; out row stride
 * ldo): the per-chunk query-gradient partials of gvd_attn_bwd_step's two sides of one BPTT step in one launch. */
int gvd_sum_chunks_pair(const float* a, int nca, const float* r, int ncr, int B, int A, float* out, int64_t ldo,
                        gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Streaming products between the few rows of a segment (caption words / decoder steps, M <= 32) and its [R, N] region
 * tensor (csrc/stream_mm.hip): each reads or writes the [B, R, N] tensor exactly once - HBM streams with an MFMA tail.
 * ------------------------------------------------------------------------------------------- */

/* `AttModel._grounder`, dot-product branch (model.py:262-278) as the training / grounding drivers call it
 * (model.py:469-480): out[b,m,r] = xt[b,m,:] . feats[b,r,:] + mbias[b,m] + rowbias[b,m,r]; out[b,m,r] = GVD_MIN_VALUE where
 * mask[b*mask_batch_stride + m*ld_mask + r] != 0 (ld_mask = 0: one mask row per sample).  feats [B,R,K] (ldf, K % 32 == 0),
 * xt [B,M,K] (M <= 32), out [B,M,R].  mbias / rowbias / mask may be NULL. */
int gvd_grounder_fwd_f32(const float* feats, int64_t ldf, int64_t f_batch_stride, const float* xt, int64_t ldxt,
                         int64_t xt_batch_stride, const float* mbias, int64_t mbias_batch_stride, const float* rowbias,
                         int64_t rowbias_ld, int64_t rowbias_batch_stride, const uint8_t* mask, int64_t ld_mask,
                         int64_t mask_batch_stride, float* out, int64_t ldo, int64_t out_batch_stride, int B, int M, int R,
                         int K, gvd_stream_t stream);

/* out[b,m,:] = sum_r S[b,m,r] F[b,r,:]  (S entries under `mask` count as 0): the gradient of the grounder's output w.r.t.
 * the word embeddings xt (autograd of model.py:262-265; the masked_fill blocks the gradient).  S [B,M,R] (lds, M <= 32),
 * F [B,R,N] (ldf, N % 128 == 0), out [B,M,N].  S_t (optional): the same matrix transposed, [B,R,32] contiguous with the
 * mask already applied and the columns m >= M zero (gvd_masked_copy_rowsum writes it) - the kernel then reads S through
 * it (coalesced) and ignores S / mask. */
int gvd_rows_contract_f32(const float* S, int64_t lds, int64_t s_batch_stride, const uint8_t* mask, int64_t ld_mask,
                          int64_t mask_batch_stride, const float* S_t, const float* F, int64_t ldf, int64_t f_batch_stride, float* out,
                          int64_t ldo, int64_t out_batch_stride, int B, int M, int R, int N, gvd_stream_t stream);

/* out[b,r,:] = sum_m S[b,m,r] X[b,m,:]  (S entries under `mask` count as 0): the gradient of the grounder's output w.r.t.
 * the region features (autograd of model.py:262-265), and d pool / d conv = alpha^T d_ctx of the two attention contexts
 * over all decoder steps (autograd of AttModel.py:50,96).  S [B,M,R] (M <= 32), X [B,M,N] (N % 128 == 0), out [B,R,N]. */
int gvd_rank_update_f32(const float* S, int64_t lds, int64_t s_batch_stride, const uint8_t* mask, int64_t ld_mask,
                        int64_t mask_batch_stride, const float* X, int64_t ldx, int64_t x_batch_stride, float* out,
                        int64_t ldo, int64_t out_batch_stride, int B, int M, int R, int N, gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Small-M dX products of the token loop's BPTT (csrc/gemm_dxs.hip): out_g[M, ncols_g] = A_g[M, Kred_g] . W_g[Kred_g, ncols_g]
 * (+ addend_g) for up to GVD_DX_MAX_GROUPS products in ONE launch, W consumed in place with the contraction index as its
 * slow axis - the gradients w.r.t. the inputs of nn.LSTMCell / nn.Linear (autograd of AttModel.py:139,160,39,77):
 * dgates . weight_ih, dgates . weight_hh, dq . h2att.weight.  Kred % 128 == 0, ncols % 128 == 0, 16-byte aligned rows.
 * workspace: gvd_gemm_dx_small_workspace_bytes(M, total columns) bytes, ZERO before the first launch that uses it (every
 * launch leaves its tile counters zero again); one workspace per stream.
 * ------------------------------------------------------------------------------------------- */
#define GVD_DX_MAX_GROUPS 4
typedef struct {
  const float* A; int64_t lda;          /* [M, Kred] */
  const float* W; int64_t ldw;          /* [Kred, ncols] (a column block of a wider matrix: ldw >= ncols) */
  int Kred; int ncols;
  float* out; int64_t ldo;              /* [M, ncols] */
  const float* addend; int64_t ld_add;  /* optional [M, ncols] added to the product (may alias nothing written here) */
} gvd_dx_group;
size_t gvd_gemm_dx_small_workspace_bytes(int M, int total_cols);
int gvd_gemm_dx_small_f32(const gvd_dx_group* groups, int ngroups, int M, void* workspace, size_t workspace_bytes,
                          gvd_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Row / column kernels of the training step (csrc/train_rows.hip)
 * ------------------------------------------------------------------------------------------- */

/* out[r, :] = softmax(x[r, :]) over N columns: alpha = softmax(e) of the saved attention scores for the BPTT
 * (autograd of AttModel.py:46,92). */
int gvd_softmax_rows(const float* x, int64_t ldx, int rows, int N, float* out, int64_t ldo, gvd_stream_t stream);

/* Backward of gvd_masked_lsm_loss (utils.py:139,142): g[r,n] = (exp(x[r,n] - row_lse[r]) * count_r - [label[r,n] != 0]) *
 * (*dloss / total), count_r / total read from the forward's `acc` (acc[3 + 2 r], acc[1]). */
int gvd_masked_lsm_bwd(const float* x, int64_t ldx, const float* label, int64_t ld_label, int rows, int N, const float* acc,
                       const float* row_lse, const float* dloss, float* g, int64_t ldg, gvd_stream_t stream);

/* Backward of log_softmax(logits)[target] per row (utils.py:131-132; forward: gvd_logsoftmax_rows with `target`):
 * g[r,v] = dpicked[r] * ([v == target[r]] - exp(logits[r,v] - lse[r])). */
int gvd_nll_gather_bwd(const float* logits, int64_t ldx, int rows, int V, const int64_t* target, const float* lse,
                       const float* dpicked, float* g, int64_t ldg, gvd_stream_t stream);

/* y[b,m,r] = mask ? 0 : x[b,m,r] (the masked_fill of model.py:274-278 applied to the grounder's output gradient) and,
 * when rowsum != NULL, rowsum[b,m] = sum_r y[b,m,r] (the gradient of the per-word class bias).  x [B,M,R] (ldx, batch stride),
 * mask u8 (ld_mask = 0: one row per sample) or NULL, y [B,M,R] contiguous.  y_t (optional, M <= 32): the transposed copy
 * [B,R,32] (y_t[b,r,m] = y[b,m,r], columns m >= M zero) gvd_rows_contract_f32 reads. */
int gvd_masked_copy_rowsum(const float* x, int64_t ldx, int64_t x_batch_stride, const uint8_t* mask, int64_t ld_mask,
                           int64_t mask_batch_stride, int B, int M, int R, float* y, float* rowsum, float* y_t,
                           gvd_stream_t stream);

/* nn.BatchNorm1d(C) + ReLU in TRAIN mode (model.py:114,397 `att_embed_aux`) over x [rows, C] (rows = B * Ft; C % 4 == 0):
 * batch mean / biased variance per column by two ordered passes, y = relu((x - mean) * invstd * weight + bias); the running
 * statistics (nullable) are updated as F.batch_norm does: r = (1 - momentum) r + momentum * (mean | UNBIASED variance).
 * stat [4 C] receives mean | invstd | scale | shift (kept for the backward); parts: gvd_bn_parts(rows) * 2 * C floats of
 * scratch.  Backward: dz = dy * [y > 0]; dx = scale * (dz - mean_r(dz) - xhat * mean_r(dz * xhat)); sums [2 C] receives
 * sum_r dz (= d bias) | sum_r dz * xhat (= d weight). */
int gvd_bn_parts(int64_t rows);
int gvd_bn_train_fwd(const float* x, int64_t rows, int C, const float* weight, const float* bias, float eps, float momentum,
                     float* running_mean, float* running_var, float* stat, float* parts, float* y, gvd_stream_t stream);
int gvd_bn_train_bwd(const float* x, const float* y, const float* dy, const float* stat, int64_t rows, int C, float* parts,
                     float* sums, float* dx, gvd_stream_t stream);

/* Up to GVD_OPT_MAX_TENSORS parameter tensors of one optimiser launch, passed by value.  Tensor t owns workgroups
 * chunk0[t] .. chunk0[t+1]-1 (chunk0[0] = 0; ceil(n[t] / gvd_opt_chunk()) each).  vec_ok[t]: all of the tensor's
 * pointers are 16-byte aligned (16-byte accesses; scalar otherwise).  p / m / v may be NULL for gvd_sumsq_partials. */
#define GVD_OPT_MAX_TENSORS 32
#define GVD_OPT_CHUNK 16384
typedef struct {
  float* p[GVD_OPT_MAX_TENSORS]; const float* g[GVD_OPT_MAX_TENSORS];
  float* m[GVD_OPT_MAX_TENSORS]; float* v[GVD_OPT_MAX_TENSORS];     /* Adam's exp_avg / exp_avg_sq */
  int64_t n[GVD_OPT_MAX_TENSORS];
  int chunk0[GVD_OPT_MAX_TENSORS + 1];
  float lr[GVD_OPT_MAX_TENSORS];
  float bc1[GVD_OPT_MAX_TENSORS];            /* 1 - beta1^step */
  float bc2_sqrt[GVD_OPT_MAX_TENSORS];       /* sqrt(1 - beta2^step) */
  uint8_t vec_ok[GVD_OPT_MAX_TENSORS];
  int count;
  int part0;                                 /* gvd_sumsq_partials: index of this launch's first partial */
} gvd_opt_group;

int gvd_opt_chunk(void);

/* main.py:265 `clip_grad_norm_(model.parameters(), opt.grad_clip)` without touching the gradients:
 * gvd_sumsq_partials: partials[part0 + w] = sum of squares of workgroup w's chunk of the group's gradients;
 * gvd_clip_coef (after all groups): out[0] = sqrt(sum of the n partials, added in a fixed order in fp64) = the total
 * L2 norm, out[1] = min(1, max_norm / (out[0] + 1e-6)) - the factor clip_grad_norm_ would scale every gradient by. */
int gvd_sumsq_partials(const gvd_opt_group* g, float* partials, gvd_stream_t stream);
int gvd_clip_coef(const float* partials, int n, float max_norm, float* out, gvd_stream_t stream);

/* main.py:266 `optimizer.step()` for torch.optim.Adam (L2 weight decay, no amsgrad) over the group, reading the clip
 * factor from clip[1] (clip = gvd_clip_coef's out, or NULL for no clipping):
 *   g' = clip * g (+ weight_decay * p);  m = beta1 m + (1 - beta1) g';  v = beta2 v + (1 - beta2) g'^2;
 *   p -= (lr / bc1) * m / (sqrt(v) / bc2_sqrt + eps).
 * skip (nullable): n_skip (<= 64) device int32 words read by the kernel; if any is non-zero the launch changes NOTHING.
 * The caller can so enqueue the step before it has read the flags that decide whether the step is valid (kernel-status
 * words of the persistent kernels, the data-parallel reducer's MAX-reduced status word) instead of draining the queue
 * for that read first. */
int gvd_adam_step(const gvd_opt_group* g, const float* clip, const int* skip, int n_skip, float beta1, float beta2,
                  float eps, float weight_decay, gvd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GVD_HIP_H */
