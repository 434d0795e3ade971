// Masked-proposal compaction for the per-segment preamble (inference).
//
// The loader zeroes every masked proposal (score <= prop_thresh, or padding): its fc6 features and its box are 0
// (dataloader_anet.py:343-344), and its class-similarity row is the all-masked constant (model.py:336-340).  So ALL masked
// proposals of a segment enter the preamble with the SAME row, and every per-row stage (fc7, class logits, layer
// norms, pool_embed, the encoder's projections / feed-forward / layer norms, ctx2pool) maps equal rows to equal rows
// - bit for bit, each output row depends on its own input row only.  The one place rows meet is the encoder's
// self-attention (transformer.py:90-123, no padding mask): there the n_m masked rows are n_m identical keys, i.e. ONE
// key whose softmax weight is multiplied by n_m (its score gets + log2 n_m in the log2 domain), and n_m identical
// queries with one common answer.  The preamble therefore runs on the COMPACT row set of each segment
//     [ its n_v valid proposals in order | one representative masked row ]
// (typically 80 % of the rows: 20 % fewer GEMM flops, ~36 % fewer attention flops) and the dense [B,R,.] tensors the
// token loop streams are restored at the end by a row gather.  Results equal the dense path up to the fp32 summation
// order inside the attention softmax.  Precondition (the loader contract above): masked rows of ppls_feat / ppls are
// zero; gvd_check_masked_rows_zero raises a device flag otherwise.
//
//   gvd_compact_index        per-segment offsets / counts, dense->compact and compact->dense row maps, key weight
//   gvd_gather_rows_f32      out[i,:] = in[idx[i],:] (compaction of the inputs, expansion of the outputs)
//   gvd_check_masked_rows_zero
#include "gvd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void compact_index_kernel(const uint8_t* __restrict__ mask, int64_t ld_mask, int R,
                                                            int* __restrict__ off, int* __restrict__ nvalid,
                                                            int* __restrict__ src_row, int* __restrict__ cidx,
                                                            float* __restrict__ rep_w, uint8_t* __restrict__ cmask,
                                                            int B) {
  __shared__ int s_cnt[4];
  __shared__ int s_base;
  __shared__ int s_first_masked;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // rows of the segments before this one: sum over b' < b of (valid rows + 1 representative)
  int before = 0;
  for (int64_t i = tid; i < (int64_t)b * R; i += 256) {
    const int bb = (int)(i / R), r = (int)(i % R);
    before += mask[bb * ld_mask + r] == 0;
  }
  before = (int)wave_sum((float)before);      // counts < 2^24: exact in fp32
  if (lane == 0) s_cnt[wave] = before;
  if (tid == 0) s_first_masked = R;
  __syncthreads();
  const int base = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3] + b;
  __syncthreads();
  const uint8_t* m = mask + (int64_t)b * ld_mask;
  int run = 0;                                 // valid rows of this segment seen so far
  for (int r0 = 0; r0 < R; r0 += 256) {
    const int r = r0 + tid;
    const bool valid = r < R && m[r] == 0;
    const unsigned long long bal = __ballot(valid);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += s_cnt[w];
    const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (r < R) {
      if (valid) {
        const int c = base + run + wbase + in_wave;
        cidx[(int64_t)b * R + r] = c;
        src_row[c] = b * R + r;
        cmask[c] = 0;
      } else {
        atomicMin(&s_first_masked, r);
      }
    }
    run += tot;
    __syncthreads();
  }
  const int rep = base + run;                   // the representative masked row
  const int nm = R - run;
  for (int r = tid; r < R; r += 256)
    if (m[r] != 0) cidx[(int64_t)b * R + r] = rep;
  if (tid == 0) {
    off[b] = base;
    if (b == B - 1) off[B] = rep + 1;
    nvalid[b] = run;
    src_row[rep] = b * R + (nm > 0 ? s_first_masked : 0);
    cmask[rep] = 1;
    rep_w[b] = nm > 0 ? log2f((float)nm) : -INFINITY;
  }
}

// one wave per output row
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in, int64_t in_ld,
                                                          const int* __restrict__ idx, float* __restrict__ out,
                                                          int64_t out_ld, int D, int64_t n_rows,
                                                          const int* __restrict__ n_dev) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n = n_dev ? (int64_t)*n_dev : n_rows;
  if (row >= n) return;
  const float* src = in + (int64_t)idx[row] * in_ld;
  float* dst = out + row * out_ld;
  if (((D | in_ld | out_ld) & 3) == 0) {
    for (int c = 4 * lane; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c);
  } else {
    for (int c = lane; c < D; c += 64) dst[c] = src[c];
  }
}

// flag[0] |= 1 when a masked row of x [rows_total, D] holds a non-zero
__global__ __launch_bounds__(256) void check_masked_zero_kernel(const float* __restrict__ x, int D,
                                                                const uint8_t* __restrict__ mask, int64_t ld_mask, int R,
                                                                int64_t rows, int* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (mask[(row / R) * ld_mask + (row % R)] == 0) return;
  const float* p = x + row * D;
  bool bad = false;
  for (int c = lane; c < D; c += 64) bad |= p[c] != 0.f;
  if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

}  // namespace

extern "C" int gvd_compact_index(const uint8_t* mask, int64_t ld_mask, int B, int R, int* off, int* nvalid, int* src_row,
                                 int* cidx, float* rep_w, uint8_t* cmask, gvd_stream_t stream) {
  if (!mask || !off || !nvalid || !src_row || !cidx || !rep_w || !cmask || B <= 0 || R <= 0 ||
      (int64_t)B * (R + 1) >= (1 << 24))
    return GVD_EINVAL;
  hipLaunchKernelGGL(compact_index_kernel, dim3((unsigned)B), dim3(256), 0, gvd_s(stream), mask, ld_mask, R, off, nvalid,
                     src_row, cidx, rep_w, cmask, B);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_gather_rows_f32(const float* in, int64_t in_ld, const int* idx, float* out, int64_t out_ld, int D,
                                   int64_t n_rows, const int* n_rows_dev, gvd_stream_t stream) {
  if (!in || !idx || !out || D <= 0 || n_rows <= 0) return GVD_EINVAL;
  if (((D | in_ld | out_ld) & 3) == 0 && (!gvd_aligned16(in) || !gvd_aligned16(out))) return GVD_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, gvd_s(stream), in, in_ld, idx,
                     out, out_ld, D, n_rows, n_rows_dev);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_check_masked_rows_zero(const float* x, int D, const uint8_t* mask, int64_t ld_mask, int B, int R,
                                          int* flag, gvd_stream_t stream) {
  if (!x || !mask || !flag || D <= 0 || B <= 0 || R <= 0) return GVD_EINVAL;
  const int64_t rows = (int64_t)B * R;
  hipLaunchKernelGGL(check_masked_zero_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, gvd_s(stream), x, D, mask,
                     ld_mask, R, rows, flag);
  GVD_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Small fused kernels of the INFERENCE preamble.  At batch_size = 4 (BASELINE configs[1]) a 'sample' call is 3.2 ms,
// of which ~0.25 ms were ~50 ATen launches of 4-6 us each (profiles/r03/b4_dispatch_trace_q.txt): the fc feature
// (mean, tiny Linear, two layer norms, concat, K pad), the location features (gather, three scalings, concat, K pad),
// BatchNorm(eval) + ReLU of the frame embeddings and the sampling-window mask (arange, two compares, and, not, fill).
// ---------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// model.py:306-308: out[b] = [ layer_norm(mean_t segs[b,t,:]) | layer_norm(relu(W num[b,3:7] + bias)) | 0 pad ]
constexpr int FC_MAXC = 16;     // columns per thread: D <= 4096
__global__ __launch_bounds__(256) void fc_feature_kernel(const float* __restrict__ segs, const int64_t* __restrict__ num,
                                                         const float* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ out, int Ft, int D, int S, int ldo,
                                                         float eps) {
  __shared__ float s_red[4];
  __shared__ float s_seg[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* sb = segs + (int64_t)b * Ft * D;
  float v[FC_MAXC];
  float sum = 0.f;
  // mean over the Ft frames, per column in frame order (the same sequential sums as before: same bits) - but with FOUR
  // frames x all of the thread's columns (<= 64 loads) in flight before the first add: at the reference-default 480 frames x
  // 3072 columns the loop with one dependent 4-byte load per add ran at 0.7 TB/s (2.1 ms per B = 256 call, 1.8 % of that step)
#pragma unroll
  for (int i = 0; i < FC_MAXC; ++i) v[i] = 0.f;
  int t = 0;
  for (; t + 4 <= Ft; t += 4) {
    float x[FC_MAXC][4];
#pragma unroll
    for (int i = 0; i < FC_MAXC; ++i) {
      const int c = tid + 256 * i;
#pragma unroll
      for (int u = 0; u < 4; ++u) x[i][u] = c < D ? __builtin_nontemporal_load(sb + (int64_t)(t + u) * D + c) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < FC_MAXC; ++i)
#pragma unroll
      for (int u = 0; u < 4; ++u) v[i] += x[i][u];
  }
  for (; t < Ft; ++t) {
#pragma unroll
    for (int i = 0; i < FC_MAXC; ++i) {
      const int c = tid + 256 * i;
      if (c < D) v[i] += sb[(int64_t)t * D + c];
    }
  }
#pragma unroll
  for (int i = 0; i < FC_MAXC; ++i) {
    const int c = tid + 256 * i;
    v[i] = c < D ? v[i] / (float)Ft : 0.f;
    sum += v[i];
  }
  const float mean = block_sum_256(sum, s_red) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < FC_MAXC; ++i) {
    const int c = tid + 256 * i;
    if (c < D) { const float d = v[i] - mean; q = fmaf(d, d, q); }
  }
  const float inv = rsqrtf(block_sum_256(q, s_red) / (float)D + eps);
  float* ob = out + (int64_t)b * ldo;
#pragma unroll
  for (int i = 0; i < FC_MAXC; ++i) {
    const int c = tid + 256 * i;
    if (c < D) ob[c] = (v[i] - mean) * inv;
  }
  // segment-position embedding: S <= 64 outputs from 4 inputs
  float e = 0.f;
  if (tid < S) {
    const int64_t* nb = num + (int64_t)b * 7 + 3;
    e = bias[tid];
#pragma unroll
    for (int k = 0; k < 4; ++k) e = fmaf(W[tid * 4 + k], (float)nb[k], e);
    e = fmaxf(e, 0.f);
    s_seg[tid] = e;
  }
  __syncthreads();
  if (tid < 64) {
    float x = tid < S ? s_seg[tid] : 0.f;
    const float m2 = wave_sum(x) / (float)S;
    const float d = tid < S ? x - m2 : 0.f;
    const float iv = rsqrtf(wave_sum(d * d) / (float)S + eps);
    if (tid < S) ob[D + tid] = d * iv;
  }
  for (int c = D + S + tid; c < ldo; c += 256) ob[c] = 0.f;
}

// model.py:357-360 on the compacted row set: out[i] = [x1, y1, x2, y2] / 720, frame / T, zero pad to ldo columns
__global__ __launch_bounds__(256) void loc_features_kernel(const float* __restrict__ ppls, const int* __restrict__ src_row,
                                                           const int* __restrict__ m_dev, float* __restrict__ out,
                                                           int64_t rows, int ldo, float T) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t live = m_dev ? min((int64_t)*m_dev, rows) : rows;
  if (i >= live) return;
  const float* p = ppls + (int64_t)(src_row ? src_row[i] : i) * 7;
  float* o = out + i * ldo;
  // (IEEE divisions through double: hipcc's fp32 `/` was measured 1 ulp off the correctly rounded quotient ATen produces)
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (float)((double)p[k] / 720.0);
  o[4] = (float)((double)(p[4] * 1.0f) / (double)T);
  for (int k = 5; k < ldo; ++k) o[k] = 0.f;
}

// BatchNorm1d(eval) + ReLU over the LAST axis: x = relu(x * scale[c] + shift[c]) in place
__global__ __launch_bounds__(256) void affine_relu_kernel(float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int64_t n4, int D4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)(i % D4) * 4;
  gvd_f32x4 v = *reinterpret_cast<gvd_f32x4*>(x + i * 4);
  const gvd_f32x4 s = *reinterpret_cast<const gvd_f32x4*>(scale + c), t = *reinterpret_cast<const gvd_f32x4*>(shift + c);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k] * s[k] + t[k], 0.f);
  *reinterpret_cast<gvd_f32x4*>(x + i * 4) = v;
}

// model.py:303-305,401: rows t outside the sampling window [s, e) of their segment are cleared (one wave per row)
__global__ __launch_bounds__(256) void zero_outside_window_kernel(float* __restrict__ x, const int64_t* __restrict__ sidx,
                                                                  int64_t rows, int Ft, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t b = row / Ft, t = row % Ft;
  if (t >= sidx[2 * b] && t < sidx[2 * b + 1]) return;
  float* xr = x + row * D;
  const gvd_f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (int i = lane; i < D / 4; i += 64) reinterpret_cast<gvd_f32x4*>(xr)[i] = z;
}

}  // namespace

extern "C" int gvd_fc_feature(const float* segs, const int64_t* num, const float* w_seg, const float* b_seg, float* out, int B,
                              int Ft, int D, int S, int ldo, float eps, gvd_stream_t stream) {
  if (!segs || !num || !w_seg || !b_seg || !out || B <= 0 || Ft <= 0 || D <= 0 || D > 256 * FC_MAXC || S <= 0 || S > 64 ||
      ldo < D + S)
    return GVD_EINVAL;
  hipLaunchKernelGGL(fc_feature_kernel, dim3((unsigned)B), dim3(256), 0, gvd_s(stream), segs, num, w_seg, b_seg, out, Ft, D, S,
                     ldo, eps);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_loc_features(const float* ppls, const int* src_row, const int* rows_dev, float* out, int64_t rows, int ldo,
                                float n_frames, gvd_stream_t stream) {
  if (!ppls || !out || rows <= 0 || ldo < 5 || n_frames <= 0.f) return GVD_EINVAL;
  hipLaunchKernelGGL(loc_features_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, gvd_s(stream), ppls, src_row,
                     rows_dev, out, rows, ldo, n_frames);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_affine_relu_rows(float* x, const float* scale, const float* shift, int64_t rows, int D,
                                    gvd_stream_t stream) {
  if (!x || !scale || !shift || rows <= 0 || D <= 0 || (D & 3) || !gvd_aligned16(x) || !gvd_aligned16(scale) ||
      !gvd_aligned16(shift))
    return GVD_EINVAL;
  const int64_t n4 = rows * (D / 4);
  hipLaunchKernelGGL(affine_relu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, gvd_s(stream), x, scale, shift, n4,
                     D / 4);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_zero_rows_outside_window(float* x, const int64_t* sample_idx, int B, int Ft, int D, gvd_stream_t stream) {
  if (!x || !sample_idx || B <= 0 || Ft <= 0 || D <= 0 || (D & 3) || !gvd_aligned16(x)) return GVD_EINVAL;
  const int64_t rows = (int64_t)B * Ft;
  hipLaunchKernelGGL(zero_outside_window_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, gvd_s(stream), x, sample_idx,
                     rows, Ft, D);
  GVD_CHECK_LAUNCH();
  return 0;
}
