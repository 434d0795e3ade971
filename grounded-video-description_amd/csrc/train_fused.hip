// Fused elementwise kernels of the TRAINING step (the dropout / ReLU-mask / bias-gradient / small-sum passes that ran
// as ATen elementwise kernels through round 3: ~8 % of a batch_size = 64 step's GPU time in ~300 launches).
//
//  gvd_dropout_rows             y = x * keep / (1 - p) on a contiguous tensor (in place or out of place), keep from
//                               Philox4x32-10 (philox.h: counter = flat index / 4, key = the site's seed).  Used right after
//                               a Linear + ReLU product (model.py:312 `ctx2pool_grd`, 384 `pool_embed`, 363 `loc_fc`, 393-395
//                               `att_embed`), whose backward needs NO mask: y > 0  <=>  pre-activation > 0 and kept.
//  gvd_relu_dropout_bwd_colsum  backward of such a site in ONE pass: dz = y > 0 ? dy / (1 - p) : 0 (the gradient w.r.t.
//                               the Linear's pre-activation output) AND the per-workgroup column sums of dz (the bias
//                               gradient, added over workgroups in order by the caller) - instead of masked_scale +
//                               threshold_backward + sum(0): 2 reads + 1 write of the tensor instead of 5 + 2.
//  gvd_sum_chunks_pair          out[b, 0:A] = sum_c a[b, c, :], out[b, A:2A] = sum_c r[b, c, :]: the two per-chunk query
//                               gradients of a BPTT step (gvd_attn_bwd_step, temporal + region side) summed in one launch
//                               straight into the step's [B, 2A] slot.
#include "gvd_common.h"
#include "philox.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4,
                                                           uint32_t thresh, float keep_scale, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
  const GvdU4 u = gvd_philox4x32_10((uint64_t)i, seed);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = u.v[k] >= thresh ? v[k] * keep_scale : 0.f;
  *reinterpret_cast<f32x4*>(y + 4 * i) = o;
}

constexpr int CS_ROWS = 64;     // rows per workgroup of the column-sum pass

__global__ __launch_bounds__(256) void relu_dropout_bwd_colsum_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                      float* __restrict__ dz, float* __restrict__ part,
                                                                      int64_t M, int N, float keep_scale) {
  const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS;
  const int nr = (int)((M - r0) < CS_ROWS ? (M - r0) : CS_ROWS);
  for (int c = 4 * threadIdx.x; c < N; c += 1024) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* dp = dy + r0 * N + c;
    const float* yp = y + r0 * N + c;
    float* zp = dz + r0 * N + c;
    int r = 0;
    for (; r + 4 <= nr; r += 4) {          // four rows in flight per thread
      f32x4 g[4], v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dp + (int64_t)(r + u) * N));
        v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(yp + (int64_t)(r + u) * N));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[u][k] > 0.f ? g[u][k] * keep_scale : 0.f;
        acc += o;
        *reinterpret_cast<f32x4*>(zp + (int64_t)(r + u) * N) = o;
      }
    }
    for (; r < nr; ++r) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(dp + (int64_t)r * N);
      const f32x4 v = *reinterpret_cast<const f32x4*>(yp + (int64_t)r * N);
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? g[k] * keep_scale : 0.f;
      acc += o;
      *reinterpret_cast<f32x4*>(zp + (int64_t)r * N) = o;
    }
    *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.x * N + c) = acc;
  }
}

__global__ __launch_bounds__(256) void sum_chunks_pair_kernel(const float* __restrict__ a, int nca, const float* __restrict__ r,
                                                              int ncr, int A, float* __restrict__ out, int64_t ldo) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;        // column of [0, 2A)
  if (j >= 2 * A) return;
  const bool second = j >= A;
  const float* src = second ? r + (int64_t)b * ncr * A + (j - A) : a + (int64_t)b * nca * A + j;
  const int nc = second ? ncr : nca;
  float s = 0.f;
  for (int c = 0; c < nc; ++c) s += src[(int64_t)c * A];
  out[(int64_t)b * ldo + j] = s;
}

}  // namespace

extern "C" int gvd_dropout_rows(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, gvd_stream_t stream) {
  if (!x || !y || n <= 0 || (n % 4) != 0 || !(p_drop >= 0.f) || !(p_drop < 1.f) || !gvd_aligned16(x) || !gvd_aligned16(y))
    return GVD_EINVAL;
  const int64_t n4 = n / 4;
  if ((n4 + 255) / 256 > 0x7fffffffLL) return GVD_EINVAL;
  hipLaunchKernelGGL(dropout_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, gvd_s(stream), x, y, n4,
                     gvd_drop_thresh(p_drop), 1.0f / (1.0f - p_drop), seed);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_relu_dropout_bwd_parts(int64_t M) { return (int)((M + CS_ROWS - 1) / CS_ROWS); }

extern "C" int gvd_relu_dropout_bwd_colsum(const float* dy, const float* y, float* dz, float* partials, int64_t M, int N,
                                           float p_drop, gvd_stream_t stream) {
  if (!dy || !y || !dz || !partials || M <= 0 || N <= 0 || (N % 4) != 0 || !(p_drop >= 0.f) || !(p_drop < 1.f) ||
      !gvd_aligned16(dy) || !gvd_aligned16(y) || !gvd_aligned16(dz) || !gvd_aligned16(partials))
    return GVD_EINVAL;
  const int64_t nwg = (M + CS_ROWS - 1) / CS_ROWS;
  if (nwg > 0x7fffffffLL) return GVD_EINVAL;
  hipLaunchKernelGGL(relu_dropout_bwd_colsum_kernel, dim3((unsigned)nwg), dim3(256), 0, gvd_s(stream), dy, y, dz, partials,
                     M, N, 1.0f / (1.0f - p_drop));
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_sum_chunks_pair(const float* a, int nca, const float* r, int ncr, int B, int A, float* out, int64_t ldo,
                                   gvd_stream_t stream) {
  if (!a || !r || !out || nca <= 0 || ncr <= 0 || B <= 0 || A <= 0 || ldo < 2 * (int64_t)A) return GVD_EINVAL;
  hipLaunchKernelGGL(sum_chunks_pair_kernel, dim3((unsigned)((2 * A + 255) / 256), (unsigned)B), dim3(256), 0, gvd_s(stream),
                     a, nca, r, ncr, A, out, ldo);
  GVD_CHECK_LAUNCH();
  return 0;
}
