// Vocabulary head kernels: log-softmax statistics per row, greedy token rule (top-2 with UNK suppression),
// word-embedding gather + ReLU, and top-K for beam search.
// Reference: model.py:464,587-608,615 (log_softmax + topk(2) + UNK rule + embed), utils.py:131-132
// (gather of the target log-prob), CaptionModelBU.py:45,125 (sorted log-probs for beam expansion).
// One 256-thread workgroup per row; rows are V floats (V ~ 5k): latency-bound, launched once per step.
#include "gvd_common.h"
#include "top2.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

__device__ __forceinline__ float block_max(float v, float* s_red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

__global__ __launch_bounds__(256) void top2_embed_kernel(const float* __restrict__ logits, int64_t ld, int V,
                                                         int unk, int64_t* it_out, int64_t it_stride,
                                                         float* lp_out, int64_t lp_stride,
                                                         const float* __restrict__ embed, int E, float* xt,
                                                         int64_t ld_xt) {
  __shared__ float s_red[4];
  __shared__ Top2 s_top[4];
  __shared__ int s_it;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (int64_t)b * ld;
  Top2 t = {-INFINITY, 0x7fffffff, -INFINITY, 0x7fffffff};
  for (int i = tid; i < V; i += 256) top2_insert(t, x[i], i);
  t = top2_wave(t);
  if ((tid & 63) == 0) s_top[tid >> 6] = t;
  __syncthreads();
  Top2 g = top2_merge(top2_merge(s_top[0], s_top[1]), top2_merge(s_top[2], s_top[3]));
  const float mx = g.v1;
  float se = 0.f;
  for (int i = tid; i < V; i += 256) se += expf(x[i] - mx);
  se = block_sum(se, s_red);
  if (tid == 0) {
    const float lse = logf(se);
    bool keep;
    const int it = top2_token(g, unk, V, &keep);
    const float lp = keep ? (g.v1 - mx) - lse : (g.v2 - mx) - lse;
    it_out[(int64_t)b * it_stride] = it;
    lp_out[(int64_t)b * lp_stride] = lp;
    s_it = it;
  }
  __syncthreads();
  if (xt) {
    const float* e = embed + (int64_t)s_it * E;
    for (int i = tid; i < E; i += 256) xt[(int64_t)b * ld_xt + i] = fmaxf(e[i], 0.f);
  }
}

// Same contract as top2_embed_kernel for V <= 8 * 1024: 16 waves per row, the row is read ONCE (8 independent
// loads per lane, all in flight together) and stays in registers for the exp pass.  The per-step token rule is a
// pure latency chain (logits -> max -> sum -> id -> embedding row), so the work is spread as wide as it goes.
constexpr int T2W_THREADS = 1024;
constexpr int T2W_SLOTS = 8;

__global__ __launch_bounds__(T2W_THREADS) void top2_embed_wide_kernel(const float* __restrict__ logits, int64_t ld,
                                                                      int V, int unk, int64_t* it_out,
                                                                      int64_t it_stride, float* lp_out,
                                                                      int64_t lp_stride,
                                                                      const float* __restrict__ embed, int E,
                                                                      float* xt, int64_t ld_xt) {
  __shared__ float s_red[16];
  __shared__ Top2 s_top[16];
  __shared__ int s_it;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const float* x = logits + (int64_t)b * ld;
  float v[T2W_SLOTS];
#pragma unroll
  for (int u = 0; u < T2W_SLOTS; ++u) {
    const int i = tid + u * T2W_THREADS;
    v[u] = i < V ? x[i] : -INFINITY;
  }
  Top2 t = {-INFINITY, 0x7fffffff, -INFINITY, 0x7fffffff};
#pragma unroll
  for (int u = 0; u < T2W_SLOTS; ++u) {
    const int i = tid + u * T2W_THREADS;
    if (i < V) top2_insert(t, v[u], i);
  }
  t = top2_wave(t);
  if ((tid & 63) == 0) s_top[wave] = t;
  __syncthreads();
  Top2 g = s_top[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) g = top2_merge(g, s_top[w]);
  const float mx = g.v1;
  float se = 0.f;
#pragma unroll
  for (int u = 0; u < T2W_SLOTS; ++u) {
    const int i = tid + u * T2W_THREADS;
    if (i < V) se += expf(v[u] - mx);
  }
  se = wave_sum(se);
  if ((tid & 63) == 0) s_red[wave] = se;
  bool keep;
  const int it = top2_token(g, unk, V, &keep);
  // the embedding row does not depend on the sum: gather it while the reduction finishes
  if (xt) {
    const float* e = embed + (int64_t)it * E;
    for (int i = tid; i < E; i += T2W_THREADS) xt[(int64_t)b * ld_xt + i] = fmaxf(e[i], 0.f);
  }
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += s_red[w];
    const float lse = logf(tot);
    it_out[(int64_t)b * it_stride] = it;
    lp_out[(int64_t)b * lp_stride] = keep ? (g.v1 - mx) - lse : (g.v2 - mx) - lse;
  }
}

__global__ __launch_bounds__(256) void embed_relu_kernel(const int64_t* it, int64_t it_stride,
                                                         const float* __restrict__ embed, int E, float* xt,
                                                         int64_t ld_xt) {
  const int b = blockIdx.x;
  const float* e = embed + it[(int64_t)b * it_stride] * E;
  for (int i = threadIdx.x; i < E; i += 256) xt[(int64_t)b * ld_xt + i] = fmaxf(e[i], 0.f);
}

constexpr int MAX_TOPK = 16;

__global__ __launch_bounds__(256) void lsm_rows_kernel(const float* __restrict__ logits, int64_t ld, int V,
                                                       float* lse_out, const int64_t* target, float* picked,
                                                       int topk, float* topk_val, int64_t* topk_idx) {
  __shared__ float s_red[4];
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* x = logits + (int64_t)row * ld;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 256) mx = fmaxf(mx, x[i]);
  mx = block_max(mx, s_red);
  float se = 0.f;
  for (int i = tid; i < V; i += 256) se += expf(x[i] - mx);
  se = block_sum(se, s_red);
  const float lse = logf(se);
  if (tid == 0) {
    if (lse_out) lse_out[row] = mx + lse;
    if (picked) picked[row] = (x[target[row]] - mx) - lse;
  }
  // top-K by K rounds of block arg-max over values not yet taken (K <= 16, V ~ 5k: cheap)
  float last_v = INFINITY; int last_i = -1;
  for (int k = 0; k < topk; ++k) {
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
      const float v = x[i];
      // candidates strictly after (last_v, last_i) in the (value desc, index asc) order
      const bool after = (v < last_v) || (v == last_v && i > last_i);
      if (after && better(v, i, bv, bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, GVD_WAVE); const int oi = __shfl_xor(bi, off, GVD_WAVE);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { s_v[tid >> 6] = bv; s_i[tid >> 6] = bi; }
    __syncthreads();
    bv = s_v[0]; bi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) if (better(s_v[w], s_i[w], bv, bi)) { bv = s_v[w]; bi = s_i[w]; }
    if (tid == 0) {
      topk_val[(int64_t)row * topk + k] = (bv - mx) - lse;
      topk_idx[(int64_t)row * topk + k] = bi;
    }
    last_v = bv; last_i = bi;
  }
}

}  // namespace

extern "C" int gvd_logsoftmax_top2_embed(const float* logits, int64_t ld_logits, int B, int V, int unk_idx,
                                         int64_t* it_out, int64_t it_stride, float* lp_out, int64_t lp_stride,
                                         const float* embed, int E, float* xt_next, int64_t ld_xt,
                                         gvd_stream_t stream) {
  if (!logits || !it_out || !lp_out || B <= 0 || V < 2 || (xt_next && !embed)) return GVD_EINVAL;
  if (V <= T2W_SLOTS * T2W_THREADS && B <= 64)   // decode batches: one wide workgroup per row (latency chain)
    hipLaunchKernelGGL(top2_embed_wide_kernel, dim3((unsigned)B), dim3(T2W_THREADS), 0, gvd_s(stream), logits,
                       ld_logits, V, unk_idx, it_out, it_stride, lp_out, lp_stride, embed, E, xt_next, ld_xt);
  else
    hipLaunchKernelGGL(top2_embed_kernel, dim3((unsigned)B), dim3(256), 0, gvd_s(stream), logits, ld_logits, V,
                       unk_idx, it_out, it_stride, lp_out, lp_stride, embed, E, xt_next, ld_xt);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_embed_relu(const int64_t* it, int64_t it_stride, int B, const float* embed, int E, float* xt,
                              int64_t ld_xt, gvd_stream_t stream) {
  if (!it || !embed || !xt || B <= 0 || E <= 0) return GVD_EINVAL;
  hipLaunchKernelGGL(embed_relu_kernel, dim3((unsigned)B), dim3(256), 0, gvd_s(stream), it, it_stride, embed, E,
                     xt, ld_xt);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_logsoftmax_rows(const float* logits, int64_t ld_logits, int rows, int V, float* lse,
                                   const int64_t* target, float* picked, int topk, float* topk_val,
                                   int64_t* topk_idx, gvd_stream_t stream) {
  if (!logits || rows <= 0 || V <= 0 || topk < 0 || topk > MAX_TOPK || topk > V) return GVD_EINVAL;
  if (picked && !target) return GVD_EINVAL;
  if (topk > 0 && (!topk_val || !topk_idx)) return GVD_EINVAL;
  hipLaunchKernelGGL(lsm_rows_kernel, dim3((unsigned)rows), dim3(256), 0, gvd_s(stream), logits, ld_logits, V, lse,
                     target, picked, topk, topk_val, topk_idx);
  GVD_CHECK_LAUNCH();
  return 0;
}
