// Training-path row kernels of the encoder's self-attention core (transformer.py:90-117: scores / sqrt(d_model),
// softmax over keys, dropout(0.2) on the attention weights) for the MATERIALISED score maps the backward needs.
//
// The six per-head products of one layer (S = Q K^T, O = P V forward; dP = dO V^T, dV = P^T dO, dQ = dS K, dK = dS^T Q
// backward) run on the pipelined fp32-MFMA GEMM (gemm_pipe.hip) over zero-padded operands: heads padded to 192 columns,
// the region axis padded to Rp = a multiple of 32 with ZERO pad rows / columns in every map, so that the K = Rp
// contractions need no tail handling.  Between the products sit these two HBM-bound row kernels (one wave per row of a
// [B, heads, Rp, Rp] map):
//   forward:  y = softmax(scale * S[:, :R]) in place (pad rows / columns written as 0), Pd = y * keep / (1 - p)
//             (Philox4x32-10 counter RNG keyed by a host-drawn seed: one 128-bit block per 4 consecutive columns);
//   backward: dS = scale * y * (dY - sum_j dY_j y_j) with dY = dPd * keep / (1 - p), in place over dPd; the keep mask is
//             read back as Pd != 0 (where y underflowed to 0 the gradient is 0 either way), so no mask tensor exists.
// The reference (ATen) makes 4 elementwise passes forward and 6 backward over the same maps.
#include "gvd_common.h"
#include "philox.h"

#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

using U4 = GvdU4;
__device__ __forceinline__ U4 philox4x32_10(uint64_t ctr, uint64_t seed) { return gvd_philox4x32_10(ctr, seed); }

constexpr int MAXNV = 8;      // Rp <= 2048

// key_bias (nullable; the compacted training layout, train_compact.py): [n_maps / maps_per_sample, Rp] added to the SCALED
// scores of a key for every query of the sample's maps - log n on a key that stands for n identical keys, -inf on a key
// that does not exist, 0 elsewhere.  Applied as s + bias / scale before the kernel's own exp(scale (s - max)), so that a
// zero bias leaves the row bit-identical to the unbiased kernel.
template <int NV>
__global__ __launch_bounds__(256) void enc_softmax_dropout_fwd_kernel(float* __restrict__ S, float* __restrict__ Pd,
                                                                      int64_t nrows, int Rp, int R, float scale_log2e,
                                                                      uint32_t thresh, float keep_scale, uint64_t seed,
                                                                      const float* __restrict__ key_bias,
                                                                      int maps_per_sample, float inv_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const float* kb = key_bias ? key_bias + (row / Rp / maps_per_sample) * Rp : nullptr;
  float* sr = S + row * Rp;
  float* pr = Pd ? Pd + row * Rp : nullptr;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if ((int)(row % Rp) >= R) {                      // pad row of the map: all zero (it is a K index of dV / dK)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 256 + 4 * lane;
      if (c < Rp) {
        *reinterpret_cast<f32x4*>(sr + c) = z;
        if (pr) *reinterpret_cast<f32x4*>(pr + c) = z;
      }
    }
    return;
  }
  f32x4 v[NV];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + 4 * lane;
    v[i] = c < Rp ? *reinterpret_cast<const f32x4*>(sr + c) : z;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c + k >= R) v[i][k] = -INFINITY;
      else if (kb) {
        const float b = kb[c + k];
        if (b != 0.f) v[i][k] = b == -INFINITY ? -INFINITY : fmaf(b, inv_scale, v[i][k]);
      }
      mx = fmaxf(mx, v[i][k]);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[i][k] = exp2f((v[i][k] - mx) * scale_log2e);       // exp(scale * (s - max)); -inf -> 0 on the pad columns
      sum += v[i][k];
    }
  const float inv = 1.0f / wave_sum(sum);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + 4 * lane;
    if (c >= Rp) continue;
    f32x4 y = {v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv};
    *reinterpret_cast<f32x4*>(sr + c) = y;
    if (pr) {
      const U4 u = philox4x32_10((uint64_t)row * (uint64_t)(Rp / 4) + (uint64_t)(c >> 2), seed);
      f32x4 d;
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] = u.v[k] >= thresh ? y[k] * keep_scale : 0.f;
      *reinterpret_cast<f32x4*>(pr + c) = d;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void enc_softmax_dropout_bwd_kernel(float* __restrict__ dP, const float* __restrict__ Pd,
                                                                      const float* __restrict__ Y, int64_t nrows, int Rp,
                                                                      int R, float scale, float keep_scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  float* dr = dP + row * Rp;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  if ((int)(row % Rp) >= R) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = i * 256 + 4 * lane;
      if (c < Rp) *reinterpret_cast<f32x4*>(dr + c) = z;
    }
    return;
  }
  const float* yr = Y + row * Rp;
  const float* pr = Pd ? Pd + row * Rp : nullptr;
  f32x4 g[NV], y[NV];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + 4 * lane;
    g[i] = z; y[i] = z;
    if (c >= Rp) continue;
    g[i] = *reinterpret_cast<const f32x4*>(dr + c);
    y[i] = *reinterpret_cast<const f32x4*>(yr + c);           // pad columns hold y = 0
    if (pr) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(pr + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) g[i][k] = p[k] != 0.f ? g[i][k] * keep_scale : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (c + k >= R) g[i][k] = 0.f;                          // the GEMM wrote only the first R columns
      dot = fmaf(g[i][k], y[i][k], dot);
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = i * 256 + 4 * lane;
    if (c >= Rp) continue;
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = scale * y[i][k] * (g[i][k] - dot);
    *reinterpret_cast<f32x4*>(dr + c) = o;
  }
}

}  // namespace

#define GVD_NV_SWITCH(NVV, CALL)                        \
  switch (NVV) {                                        \
    case 1: { constexpr int NV_ = 1; CALL; break; }     \
    case 2: { constexpr int NV_ = 2; CALL; break; }     \
    case 3: { constexpr int NV_ = 3; CALL; break; }     \
    case 4: { constexpr int NV_ = 4; CALL; break; }     \
    case 5: { constexpr int NV_ = 5; CALL; break; }     \
    case 6: { constexpr int NV_ = 6; CALL; break; }     \
    case 7: { constexpr int NV_ = 7; CALL; break; }     \
    case 8: { constexpr int NV_ = 8; CALL; break; }     \
    default: return GVD_EINVAL;                         \
  }

extern "C" int gvd_enc_softmax_dropout_fwd(float* S, float* Pd, int64_t n_maps, int Rp, int R, float scale, float p_drop,
                                           uint64_t seed, const float* key_bias, int maps_per_sample, gvd_stream_t stream) {
  if (key_bias && (maps_per_sample <= 0 || (n_maps % maps_per_sample) != 0 || !(scale > 0.f))) return GVD_EINVAL;
  if (!S || n_maps <= 0 || Rp <= 0 || (Rp % 32) != 0 || Rp > 256 * MAXNV || R <= 0 || R > Rp || !(p_drop >= 0.f) ||
      !(p_drop < 1.f) || !gvd_aligned16(S) || (Pd && !gvd_aligned16(Pd)) || (p_drop > 0.f && !Pd))
    return GVD_EINVAL;
  const int64_t nrows = n_maps * Rp;
  const int nv = (Rp + 255) / 256;
  const uint32_t thresh = (uint32_t)fmin(4294967295.0, (double)p_drop * 4294967296.0);
  const float keep_scale = 1.0f / (1.0f - p_drop);
  float* pd = p_drop > 0.f ? Pd : nullptr;
  const dim3 grid((unsigned)((nrows + 3) / 4));
  GVD_NV_SWITCH(nv, hipLaunchKernelGGL(enc_softmax_dropout_fwd_kernel<NV_>, grid, dim3(256), 0, gvd_s(stream), S, pd, nrows,
                                       Rp, R, scale * 1.44269504088896340736f, thresh, keep_scale, seed, key_bias,
                                       maps_per_sample, key_bias ? 1.0f / scale : 0.f));
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_enc_softmax_dropout_bwd(float* dP, const float* Pd, const float* Y, int64_t n_maps, int Rp, int R,
                                           float scale, float p_drop, gvd_stream_t stream) {
  if (!dP || !Y || n_maps <= 0 || Rp <= 0 || (Rp % 32) != 0 || Rp > 256 * MAXNV || R <= 0 || R > Rp || !(p_drop >= 0.f) ||
      !(p_drop < 1.f) || !gvd_aligned16(dP) || !gvd_aligned16(Y) || (Pd && !gvd_aligned16(Pd)) || (p_drop > 0.f && !Pd))
    return GVD_EINVAL;
  const int64_t nrows = n_maps * Rp;
  const int nv = (Rp + 255) / 256;
  const float keep_scale = 1.0f / (1.0f - p_drop);
  const float* pd = p_drop > 0.f ? Pd : nullptr;
  const dim3 grid((unsigned)((nrows + 3) / 4));
  GVD_NV_SWITCH(nv, hipLaunchKernelGGL(enc_softmax_dropout_bwd_kernel<NV_>, grid, dim3(256), 0, gvd_s(stream), dP, pd, Y,
                                       nrows, Rp, R, scale, keep_scale));
  GVD_CHECK_LAUNCH();
  return 0;
}
