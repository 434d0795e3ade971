// Row / column kernels of the training step that used to be chains of ATen elementwise ops:
//
//   gvd_softmax_rows          softmax over the last axis of the saved attention scores (the BPTT's alpha = softmax(e):
//                             autograd of AttModel.py:46,92)
//   gvd_masked_lsm_bwd        backward of -mean(log_softmax(x)[label != 0]) (utils.py:139,142): one pass over [rows, N]
//   gvd_nll_gather_bwd        backward of log_softmax(logits)[target] per row (utils.py:131-132)
//   gvd_bn_train_fwd / _bwd   nn.BatchNorm1d(1024) + ReLU of the frame embeddings in TRAIN mode (model.py:114,397:
//                             `att_embed_aux`), on the [B Ft, C] layout the frame projections write - batch statistics by
//                             two ordered column passes (mean, then centred second moment), running statistics updated
//                             with momentum and the unbiased variance exactly like torch.nn.functional.batch_norm; no
//                             [B, C, Ft] permute copies around a library kernel.
#include "gvd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float block_max(float v, float* s_red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// one WAVE per row (4 rows per workgroup), the row held in registers (N <= 64 * SM_PER): one read, one write, shuffle-only
// reductions - these rows are short (R = 1000 regions / Ft frames) and the launch is latency-, not bandwidth-bound
template <int SM_PER>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int64_t ldx, int rows, int N,
                                                           float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (int64_t)row * ldx;
  float* o = out + (int64_t)row * ldo;
  if (N <= 64 * SM_PER) {
    float v[SM_PER];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SM_PER; ++k) {
      const int i = lane + 64 * k;
      v[k] = i < N ? xr[i] : -INFINITY;
      mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < SM_PER; ++k) {
      v[k] = __expf(v[k] - mx);            // exp(-inf) = 0 on the pad lanes
      se += v[k];
    }
    se = wave_sum(se);
    const float inv = 1.0f / se;
#pragma unroll
    for (int k = 0; k < SM_PER; ++k) {
      const int i = lane + 64 * k;
      if (i < N) o[i] = v[k] * inv;
    }
    return;
  }
  float mx = -INFINITY;
  for (int i = lane; i < N; i += 64) mx = fmaxf(mx, xr[i]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int i = lane; i < N; i += 64) se += expf(xr[i] - mx);
  se = wave_sum(se);
  const float inv = 1.0f / se;
  for (int i = lane; i < N; i += 64) o[i] = expf(xr[i] - mx) * inv;
}

// g[row, n] = (exp(x - lse_row) * cnt_row - [label != 0]) * (dloss / total);  cnt_row / total = the forward's accumulators
__global__ __launch_bounds__(256) void masked_lsm_bwd_kernel(const float* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ label, int64_t ldl, int N,
                                                             const float* __restrict__ acc, const float* __restrict__ row_lse,
                                                             const float* __restrict__ dloss, float* __restrict__ g, int64_t ldg) {
  const int row = blockIdx.x;
  const float* xr = x + (int64_t)row * ldx;
  const float* lr = label + (int64_t)row * ldl;
  float* gr = g + (int64_t)row * ldg;
  const float scale = *dloss / acc[1];
  const float cnt = acc[3 + 2 * row], lse = row_lse[row];
  for (int i = threadIdx.x; i < N; i += 256)
    gr[i] = (expf(xr[i] - lse) * cnt - (lr[i] != 0.f ? 1.f : 0.f)) * scale;
}

// g[row, v] = dpicked[row] * ([v == target[row]] - exp(logits[row, v] - lse[row]))
__global__ __launch_bounds__(256) void nll_gather_bwd_kernel(const float* __restrict__ logits, int64_t ldx, int V,
                                                             const int64_t* __restrict__ target, const float* __restrict__ lse,
                                                             const float* __restrict__ dpicked, float* __restrict__ g, int64_t ldg) {
  const int row = blockIdx.x;
  const float* xr = logits + (int64_t)row * ldx;
  float* gr = g + (int64_t)row * ldg;
  const float d = dpicked[row], l = lse[row];
  const int tg = (int)target[row];
  for (int i = threadIdx.x; i < V; i += 256) gr[i] = d * ((i == tg ? 1.f : 0.f) - expf(xr[i] - l));
}

// y[b,m,:] = x[b,m,:] with the entries under mask zeroed; rowsum[b,m] = sum of y[b,m,:] (one workgroup per row)
__global__ __launch_bounds__(256) void masked_copy_rowsum_kernel(const float* __restrict__ x, int64_t ldx, int64_t xbs,
                                                                 const uint8_t* __restrict__ mask, int64_t ldmk, int64_t mbs,
                                                                 int M, int R, float* __restrict__ y, float* __restrict__ rowsum,
                                                                 float* __restrict__ yt) {
  __shared__ float s_red[4];
  const int b = blockIdx.x / M, m = blockIdx.x % M;
  const float* xr = x + (int64_t)b * xbs + (int64_t)m * ldx;
  const uint8_t* mr = mask ? mask + (int64_t)b * mbs + (int64_t)m * ldmk : nullptr;
  float* yr = y + (int64_t)blockIdx.x * R;
  float s = 0.f;
  for (int i = threadIdx.x; i < R; i += 256) {
    const float v = (mr && mr[i]) ? 0.f : xr[i];
    yr[i] = v;
    s += v;
    if (yt) {
      float* t = yt + ((int64_t)b * R + i) * 32;
      t[m] = v;
      if (m == 0)                                   // the row-0 workgroup also clears the pad columns m >= M
        for (int k = M; k < 32; ++k) t[k] = 0.f;
    }
  }
  if (rowsum) {
    s = block_sum(s, s_red);
    if (threadIdx.x == 0) rowsum[blockIdx.x] = s;
  }
}

// ---- BatchNorm1d (train mode) over x [rows, C]: column statistics.  BN_RB rows per workgroup and column group of 256.
constexpr int BN_RB = 64;

// MODE 0: sum x;  MODE 1: sum (x - mean)^2;  MODE 2: (sum dz, sum dz * xhat) with dz = dy * [y > 0], xhat = (x - mean) invstd
template <int MODE>
__global__ __launch_bounds__(256) void bn_colsum_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ dy, const float* __restrict__ stat,
                                                        int64_t rows, int C, float* __restrict__ parts) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= C) return;
  const int64_t r0 = (int64_t)blockIdx.y * BN_RB, r1 = min(rows, r0 + BN_RB);
  float s0 = 0.f, s1 = 0.f;
  const float mean = MODE >= 1 ? stat[col] : 0.f;
  const float invstd = MODE == 2 ? stat[C + col] : 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    const float v = x[r * C + col];
    if (MODE == 0) s0 += v;
    if (MODE == 1) { const float d = v - mean; s0 = fmaf(d, d, s0); }
    if (MODE == 2) {
      const float dz = y[r * C + col] > 0.f ? dy[r * C + col] : 0.f;
      s0 += dz;
      s1 = fmaf(dz, (v - mean) * invstd, s1);
    }
  }
  parts[(int64_t)blockIdx.y * (MODE == 2 ? 2 : 1) * C + col] = s0;
  if (MODE == 2) parts[((int64_t)blockIdx.y * 2 + 1) * C + col] = s1;
}

// phase 0: stat[0:C] = mean.  phase 1: stat[C:2C] = invstd, stat[2C:3C] = scale, stat[3C:4C] = shift; running statistics.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ parts, int nparts, int64_t rows, int C,
                                                          int phase, const float* __restrict__ weight,
                                                          const float* __restrict__ bias, float eps, float momentum,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ stat) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= C) return;
  float s = 0.f;
  for (int k = 0; k < nparts; ++k) s += parts[(int64_t)k * C + col];          // ordered: bit-reproducible
  if (phase == 0) {
    stat[col] = s / (float)rows;
    return;
  }
  const float mean = stat[col];
  const float var = s / (float)rows;                                           // biased: what normalises (F.batch_norm)
  const float invstd = 1.0f / sqrtf(var + eps);
  const float sc = weight[col] * invstd;
  stat[C + col] = invstd;
  stat[2 * C + col] = sc;
  stat[3 * C + col] = bias[col] - mean * sc;
  if (running_mean) running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mean;
  if (running_var) {
    const float unbiased = rows > 1 ? s / (float)(rows - 1) : var;
    running_var[col] = (1.f - momentum) * running_var[col] + momentum * unbiased;
  }
}

// y = relu(x * scale + shift)
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                            int64_t n4, int C, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int col = (int)((i * 4) % C);
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  const f32x4 sc = *reinterpret_cast<const f32x4*>(stat + 2 * C + col);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(stat + 3 * C + col);
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaf(v[k], sc[k], sh[k]), 0.f);
  reinterpret_cast<f32x4*>(y)[i] = o;
}

// dx = (gamma invstd / n) (n dz - sum dz - xhat sum(dz xhat)); dgamma = sum(dz xhat); dbeta = sum dz  (sums[0:C], sums[C:2C])
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dy, const float* __restrict__ stat,
                                                           const float* __restrict__ sums, int64_t n4, int64_t rows, int C,
                                                           float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int col = (int)((i * 4) % C);
  const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
  const f32x4 yv = reinterpret_cast<const f32x4*>(y)[i];
  const f32x4 dv = reinterpret_cast<const f32x4*>(dy)[i];
  const f32x4 mean = *reinterpret_cast<const f32x4*>(stat + col);
  const f32x4 invstd = *reinterpret_cast<const f32x4*>(stat + C + col);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(stat + 2 * C + col);
  const f32x4 sdz = *reinterpret_cast<const f32x4*>(sums + col);
  const f32x4 sdzx = *reinterpret_cast<const f32x4*>(sums + C + col);
  const float inv_n = 1.0f / (float)rows;
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dz = yv[k] > 0.f ? dv[k] : 0.f;
    const float xh = (xv[k] - mean[k]) * invstd[k];
    o[k] = sc[k] * (dz - inv_n * (sdz[k] + xh * sdzx[k]));
  }
  reinterpret_cast<f32x4*>(dx)[i] = o;
}

// sums[j * C + col] = sum over k of parts[(k * nj + j) * C + col], in order
__global__ __launch_bounds__(256) void bn_sum_parts_kernel(const float* __restrict__ parts, int nparts, int nj, int C,
                                                           float* __restrict__ sums) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= C) return;
  for (int j = 0; j < nj; ++j) {
    float s = 0.f;
    for (int k = 0; k < nparts; ++k) s += parts[((int64_t)k * nj + j) * C + col];
    sums[(int64_t)j * C + col] = s;
  }
}

}  // namespace

extern "C" int gvd_softmax_rows(const float* x, int64_t ldx, int rows, int N, float* out, int64_t ldo, gvd_stream_t stream) {
  if (!x || !out || rows <= 0 || N <= 0) return GVD_EINVAL;
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = gvd_s(stream);
  if (N <= 256) hipLaunchKernelGGL(softmax_rows_kernel<4>, grid, dim3(256), 0, st, x, ldx, rows, N, out, ldo);
  else if (N <= 512) hipLaunchKernelGGL(softmax_rows_kernel<8>, grid, dim3(256), 0, st, x, ldx, rows, N, out, ldo);
  else if (N <= 1024) hipLaunchKernelGGL(softmax_rows_kernel<16>, grid, dim3(256), 0, st, x, ldx, rows, N, out, ldo);
  else hipLaunchKernelGGL(softmax_rows_kernel<32>, grid, dim3(256), 0, st, x, ldx, rows, N, out, ldo);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_masked_lsm_bwd(const float* x, int64_t ldx, const float* label, int64_t ld_label, int rows, int N,
                                  const float* acc, const float* row_lse, const float* dloss, float* g, int64_t ldg,
                                  gvd_stream_t stream) {
  if (!x || !label || !acc || !row_lse || !dloss || !g || rows <= 0 || N <= 0) return GVD_EINVAL;
  hipLaunchKernelGGL(masked_lsm_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, gvd_s(stream), x, ldx, label, ld_label, N,
                     acc, row_lse, dloss, g, ldg);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_nll_gather_bwd(const float* logits, int64_t ldx, int rows, int V, const int64_t* target, const float* lse,
                                  const float* dpicked, float* g, int64_t ldg, gvd_stream_t stream) {
  if (!logits || !target || !lse || !dpicked || !g || rows <= 0 || V <= 0) return GVD_EINVAL;
  hipLaunchKernelGGL(nll_gather_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, gvd_s(stream), logits, ldx, V, target, lse,
                     dpicked, g, ldg);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_masked_copy_rowsum(const float* x, int64_t ldx, int64_t x_batch_stride, const uint8_t* mask, int64_t ld_mask,
                                      int64_t mask_batch_stride, int B, int M, int R, float* y, float* rowsum, float* y_t,
                                      gvd_stream_t stream) {
  if (!x || !y || B <= 0 || M <= 0 || R <= 0 || (y_t && M > 32)) return GVD_EINVAL;
  hipLaunchKernelGGL(masked_copy_rowsum_kernel, dim3((unsigned)(B * M)), dim3(256), 0, gvd_s(stream), x, ldx, x_batch_stride,
                     mask, ld_mask, mask_batch_stride, M, R, y, rowsum, y_t);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_bn_parts(int64_t rows) { return (int)((rows + BN_RB - 1) / BN_RB); }

extern "C" int gvd_bn_train_fwd(const float* x, int64_t rows, int C, const float* weight, const float* bias, float eps,
                                float momentum, float* running_mean, float* running_var, float* stat, float* parts, float* y,
                                gvd_stream_t stream) {
  if (!x || !weight || !bias || !stat || !parts || !y || rows <= 0 || C <= 0 || (C % 4) || !gvd_aligned16(x) ||
      !gvd_aligned16(y) || !gvd_aligned16(stat))
    return GVD_EINVAL;
  hipStream_t st = gvd_s(stream);
  const int np = gvd_bn_parts(rows);
  const dim3 gcs((unsigned)((C + 255) / 256), (unsigned)np), gc((unsigned)((C + 255) / 256));
  hipLaunchKernelGGL(bn_colsum_kernel<0>, gcs, dim3(256), 0, st, x, nullptr, nullptr, nullptr, rows, C, parts);
  hipLaunchKernelGGL(bn_finalize_kernel, gc, dim3(256), 0, st, parts, np, rows, C, 0, weight, bias, eps, momentum,
                     (float*)nullptr, (float*)nullptr, stat);
  hipLaunchKernelGGL(bn_colsum_kernel<1>, gcs, dim3(256), 0, st, x, nullptr, nullptr, stat, rows, C, parts);
  hipLaunchKernelGGL(bn_finalize_kernel, gc, dim3(256), 0, st, parts, np, rows, C, 1, weight, bias, eps, momentum,
                     running_mean, running_var, stat);
  const int64_t n4 = rows * C / 4;
  hipLaunchKernelGGL(bn_apply_relu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, stat, n4, C, y);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_bn_train_bwd(const float* x, const float* y, const float* dy, const float* stat, int64_t rows, int C,
                                float* parts, float* sums, float* dx, gvd_stream_t stream) {
  if (!x || !y || !dy || !stat || !parts || !sums || !dx || rows <= 0 || C <= 0 || (C % 4) || !gvd_aligned16(x) ||
      !gvd_aligned16(y) || !gvd_aligned16(dy) || !gvd_aligned16(dx) || !gvd_aligned16(stat) || !gvd_aligned16(sums))
    return GVD_EINVAL;
  hipStream_t st = gvd_s(stream);
  const int np = gvd_bn_parts(rows);
  const dim3 gcs((unsigned)((C + 255) / 256), (unsigned)np), gc((unsigned)((C + 255) / 256));
  hipLaunchKernelGGL(bn_colsum_kernel<2>, gcs, dim3(256), 0, st, x, y, dy, stat, rows, C, parts);
  hipLaunchKernelGGL(bn_sum_parts_kernel, gc, dim3(256), 0, st, parts, np, 2, C, sums);
  const int64_t n4 = rows * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, y, dy, stat, sums, n4,
                     rows, C, dx);
  GVD_CHECK_LAUNCH();
  return 0;
}
