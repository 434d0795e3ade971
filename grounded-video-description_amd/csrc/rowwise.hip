// Fused row kernels of the per-segment preamble (inference path), one 64-lane wave per row, no LDS:
//
//  gvd_add_layernorm_unbiased   out = gamma * (s - mean(s)) / (std_unbiased(s) + eps) + beta,  s = x + y
//      = ResidualBlock + the encoder's custom LayerNorm (transformer.py:66-88: unbiased std, eps added to std)
//      which the reference runs as ~8 separate [B,R,1024] elementwise/reduce ATen ops, 4x per forward.
//  gvd_region_feature_rows      per proposal row (model.py:336-364):
//      p = softmax_classes(sim_logits_row  (filled with -1e8 when the proposal is masked))   -> sim_out row
//      out = [ layer_norm(g_pool_row, 2048) | layer_norm(loc_row, 300) | layer_norm(p, D1) ]      -> [2781]
//      i.e. `_grounder` mask + F.softmax(dim=1) + permute + 3x F.layer_norm + torch.cat in ONE pass
//      (reads 11 KB, writes 13 KB per row instead of ~8 passes over [B,R,2781]-sized tensors).
// HBM-bound streaming kernels; values stay in registers between the statistics and the normalisation pass.
#include "gvd_common.h"
#include "philox.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// DROP (training): s = x + y * keep / (1 - p), the branch dropout of the ResidualBlock (transformer.py:84-87) applied while
// y streams through - keep from Philox (philox.h: counter = flat index / 4 of the contiguous [rows, D] tensor, the same
// mask gvd_dropout_rows draws for the same seed); the backward kernel regenerates it.
template <int D, bool DROP>   // D = 64 * 4 * NV
__global__ __launch_bounds__(256) void add_ln_unbiased_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ out,
                                                              int64_t rows, const int* __restrict__ rows_dev, float eps,
                                                              uint32_t thresh, float keep_scale, uint64_t seed) {
  constexpr int NV = D / 256;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (rows_dev ? (int64_t)*rows_dev : rows)) return;
  const float* xr = x + row * D;
  const float* yr = y ? y + row * D : nullptr;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + i * 256 + 4 * lane));   // read-once streams
    if (yr) {
      f32x4 w = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(yr + i * 256 + 4 * lane));
      if constexpr (DROP) {
        const GvdU4 u = gvd_philox4x32_10((uint64_t)row * (uint64_t)(D / 4) + (uint64_t)(i * 64 + lane), seed);
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = u.v[k] >= thresh ? w[k] * keep_scale : 0.f;
      }
      v[i][0] += w[0]; v[i][1] += w[1]; v[i][2] += w[2]; v[i][3] += w[3];
    }
    s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q = fmaf(d, d, q); }
  const float stdv = sqrtf(wave_sum(q) / (D - 1));
  const float inv = 1.0f / (stdv + eps);
  float* o = out + row * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + i * 256 + 4 * lane);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + i * 256 + 4 * lane);
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = g[k] * (v[i][k] - mean) * inv + b[k];
    __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(o + i * 256 + 4 * lane));
  }
}

// Backward of out = gamma * (s - mean) / (std_unbiased + eps) + beta, s = x + y (the encoder's ResidualBlock LayerNorm,
// transformer.py:66-88): ds (= dx = dy_branch) per row, and per-workgroup partial sums of dgamma / dbeta (a workgroup
// walks ROWS_PER_WG rows with one wave per row; lanes keep the partials of their 16 columns in registers; the host adds
// the [nwg, D] partials in order).  Statistics are recomputed from s: nothing but x, y is kept from the forward.
// DROP: s is recomputed as x + y * keep / (1 - p) with the forward's mask, and the branch gradient dyb = ds * keep / (1 - p)
// is written next to ds (without dropout both addends share ds).
constexpr int LNB_ROWS = 64;
template <int D, bool DROP>
__global__ __launch_bounds__(256) void add_ln_unbiased_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                  const float* __restrict__ dout,
                                                                  const float* __restrict__ gamma, float* __restrict__ ds,
                                                                  float* __restrict__ dyb, float* __restrict__ part,
                                                                  int64_t rows, float eps, uint32_t thresh, float keep_scale,
                                                                  uint64_t seed) {
  constexpr int NV = D / 256;
  __shared__ float s_acc[4][2][D / 4 + 4];      // staged one quarter of the columns at a time (see below)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 g[NV], dg[NV], db[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    g[i] = *reinterpret_cast<const f32x4*>(gamma + i * 256 + 4 * lane);
    dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int64_t r0 = (int64_t)blockIdx.x * LNB_ROWS;
  for (int rr = wave; rr < LNB_ROWS; rr += 4) {
    const int64_t row = r0 + rr;
    if (row >= rows) break;
    f32x4 v[NV], d[NV];
    f32x4 km[DROP ? NV : 1];       // keep / (1 - p) per element (DROP only)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = *reinterpret_cast<const f32x4*>(x + row * D + i * 256 + 4 * lane);
      if constexpr (DROP) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(y + row * D + i * 256 + 4 * lane);
        const GvdU4 u = gvd_philox4x32_10((uint64_t)row * (uint64_t)(D / 4) + (uint64_t)(i * 64 + lane), seed);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          km[i][k] = u.v[k] >= thresh ? keep_scale : 0.f;
          v[i][k] = fmaf(w[k], km[i][k], v[i][k]);
        }
      } else if (y) v[i] += *reinterpret_cast<const f32x4*>(y + row * D + i * 256 + 4 * lane);
      d[i] = *reinterpret_cast<const f32x4*>(dout + row * D + i * 256 + 4 * lane);
      sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(sum) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[i][k] -= mean; q = fmaf(v[i][k], v[i][k], q); }
    const float stdv = sqrtf(wave_sum(q) / (D - 1));
    const float inv = 1.0f / (stdv + eps);
    // dxhat_j = gamma_j dout_j inv;  dstd = -sum_j gamma_j dout_j xhat_j inv^2
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gd = g[i][k] * d[i][k];
        t = fmaf(gd, v[i][k], t);
        dg[i][k] = fmaf(d[i][k], v[i][k] * inv, dg[i][k]);
        db[i][k] += d[i][k];
      }
    const float dstd = -wave_sum(t) * inv * inv;
    const float c = stdv > 0.f ? dstd / ((D - 1) * stdv) : 0.f;
    // grad_j = gamma_j dout_j inv + c xhat_j; ds_j = grad_j - mean(grad)   (sum_j xhat_j = 0)
    float gs = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[i][k] = fmaf(g[i][k] * d[i][k], inv, c * v[i][k]);
        gs += v[i][k];
      }
    const float gm = wave_sum(gs) / D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 o = {v[i][0] - gm, v[i][1] - gm, v[i][2] - gm, v[i][3] - gm};
      *reinterpret_cast<f32x4*>(ds + row * D + i * 256 + 4 * lane) = o;
      if constexpr (DROP) {
        const f32x4 ob = {o[0] * km[i][0], o[1] * km[i][1], o[2] * km[i][2], o[3] * km[i][3]};
        *reinterpret_cast<f32x4*>(dyb + row * D + i * 256 + 4 * lane) = ob;
      }
    }
  }
  // reduce the 4 waves' partials through LDS, one 256-column slab (index i) at a time
  float* pg = part + (int64_t)blockIdx.x * 2 * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(&s_acc[wave][0][4 * lane]) = dg[i];
    *reinterpret_cast<f32x4*>(&s_acc[wave][1][4 * lane]) = db[i];
    __syncthreads();
    if (wave == 0) {
      f32x4 a = *reinterpret_cast<f32x4*>(&s_acc[0][0][4 * lane]);
      f32x4 b = *reinterpret_cast<f32x4*>(&s_acc[0][1][4 * lane]);
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        a += *reinterpret_cast<f32x4*>(&s_acc[w][0][4 * lane]);
        b += *reinterpret_cast<f32x4*>(&s_acc[w][1][4 * lane]);
      }
      *reinterpret_cast<f32x4*>(pg + i * 256 + 4 * lane) = a;
      *reinterpret_cast<f32x4*>(pg + D + i * 256 + 4 * lane) = b;
    }
  }
}

constexpr int RF_G = 2048;   // fc7 feature size
constexpr int RF_MAXC = 8;   // per-lane slots for the loc (<= 512) and class (<= 512) segments

__global__ __launch_bounds__(256) void region_feature_rows_kernel(const float* __restrict__ g_pool,
                                                                  const float* __restrict__ loc, int n_loc,
                                                                  const float* __restrict__ logits, int n_cls,
                                                                  int64_t logits_ld,
                                                                  const uint8_t* __restrict__ row_mask,
                                                                  int64_t mask_row_div, int64_t mask_ld,
                                                                  float* __restrict__ out, int64_t out_ld,
                                                                  float* __restrict__ sim_out, int64_t rows,
                                                                  const int* __restrict__ rows_dev, float ln_eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (rows_dev ? (int64_t)*rows_dev : rows)) return;
  float* o = out + row * out_ld;
  // pad columns (out_ld > 2048 + n_loc + n_cls: the K-padded operand of the pool_embed GEMM) are zero
  for (int c = RF_G + n_loc + n_cls + lane; c < out_ld; c += 64) o[c] = 0.f;

  // ---- segment 1: F.layer_norm over the 2048 fc7 features (biased variance, eps inside the sqrt, no affine)
  {
    const float* gr = g_pool + row * RF_G;
    f32x4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(gr + i * 256 + 4 * lane));
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) / RF_G;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = v[i][k] - mean; q = fmaf(d, d, q); }
    const float rstd = rsqrtf(wave_sum(q) / RF_G + ln_eps);
    if ((out_ld & 3) == 0) {               // padded rows are 16-byte aligned: vector stores
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = (v[i][k] - mean) * rstd;
        *reinterpret_cast<f32x4*>(o + i * 256 + 4 * lane) = r;
      }
    } else {
      // rows of 2781 floats are not 16-byte aligned -> scalar stores, lane-contiguous (coalesced)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) o[i * 256 + 4 * lane + k] = (v[i][k] - mean) * rstd;
    }
  }
  // ---- segment 2: layer_norm over the location embedding (n_loc = 300)
  {
    const float* lr = loc + row * n_loc;
    float v[RF_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int c = i * 64 + lane;
      v[i] = c < n_loc ? lr[c] : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) / n_loc;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < n_loc) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    }
    const float rstd = rsqrtf(wave_sum(q) / n_loc + ln_eps);
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < n_loc) o[RF_G + c] = (v[i] - mean) * rstd;
    }
  }
  // ---- segment 3: class softmax of the (masked) similarity logits, then layer_norm of the distribution
  {
    const float* sr = logits + row * logits_ld;
    const bool masked = row_mask && row_mask[(row / mask_row_div) * mask_ld + (row % mask_row_div)];
    float v[RF_MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int c = i * 64 + lane;
      v[i] = c < n_cls ? (masked ? GVD_MIN_VALUE : sr[c]) : -INFINITY;
      mx = fmaxf(mx, v[i]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int c = i * 64 + lane;
      v[i] = c < n_cls ? expf(v[i] - mx) : 0.f;
      se += v[i];
    }
    const float inv = 1.0f / wave_sum(se);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) { v[i] *= inv; s += v[i]; }
    if (sim_out) {
#pragma unroll
      for (int i = 0; i < RF_MAXC; ++i) {
        const int c = i * 64 + lane;
        if (c < n_cls) sim_out[row * n_cls + c] = v[i];
      }
    }
    const float mean = wave_sum(s) / n_cls;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < n_cls) { const float d = v[i] - mean; q = fmaf(d, d, q); }
    }
    const float rstd = rsqrtf(wave_sum(q) / n_cls + ln_eps);
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int c = i * 64 + lane;
      if (c < n_cls) o[RF_G + n_loc + c] = (v[i] - mean) * rstd;
    }
  }
}

// Backward of region_feature_rows_kernel (training): per proposal row, from d_out [2048 | n_loc | n_cls | pads] and the
// optional direct gradient of the class distribution (the region-classification loss reads sim_out):
//   F.layer_norm without affine:  dx = rstd (dy - mean(dy) - y mean(dy y)),  y = (x - mean) rstd  (statistics recomputed)
//   class softmax:                dl = p (dp - sum p dp),  dp = LN-backward + d_sim;  masked rows had constant logits: dl = 0
// Reads g_pool / loc / p / d_out once, writes d_gpool / d_loc / d_logits once (one wave per row, values in registers).
__global__ __launch_bounds__(256) void region_feature_rows_bwd_kernel(
    const float* __restrict__ g_pool, const float* __restrict__ loc, int n_loc, const float* __restrict__ sim, int n_cls,
    const uint8_t* __restrict__ row_mask, int64_t mask_row_div, int64_t mask_ld, const float* __restrict__ dout,
    int64_t dout_ld, const float* __restrict__ dsim, float* __restrict__ d_gpool, float* __restrict__ d_loc,
    float* __restrict__ d_logits, int64_t dl_ld, int64_t rows, float ln_eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* dor = dout + row * dout_ld;
  {
    const float* gr = g_pool + row * RF_G;
    f32x4 v[8], d[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = *reinterpret_cast<const f32x4*>(gr + i * 256 + 4 * lane);
      d[i] = *reinterpret_cast<const f32x4*>(dor + i * 256 + 4 * lane);
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    const float mean = wave_sum(s) / RF_G;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[i][k] -= mean; q = fmaf(v[i][k], v[i][k], q); }
    const float rstd = rsqrtf(wave_sum(q) / RF_G + ln_eps);
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[i][k] *= rstd; a += d[i][k]; c = fmaf(d[i][k], v[i][k], c); }
    const float m1 = wave_sum(a) / RF_G, m2 = wave_sum(c) / RF_G;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = rstd * (d[i][k] - m1 - v[i][k] * m2);
      *reinterpret_cast<f32x4*>(d_gpool + row * RF_G + i * 256 + 4 * lane) = o;
    }
  }
  {
    const float* lr = loc + row * n_loc;
    float v[RF_MAXC], d[RF_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int cc = i * 64 + lane;
      v[i] = cc < n_loc ? lr[cc] : 0.f;
      d[i] = cc < n_loc ? dor[RF_G + cc] : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) / n_loc;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int cc = i * 64 + lane;
      v[i] = cc < n_loc ? v[i] - mean : 0.f;
      q = fmaf(v[i], v[i], q);
    }
    const float rstd = rsqrtf(wave_sum(q) / n_loc + ln_eps);
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) { v[i] *= rstd; a += d[i]; c = fmaf(d[i], v[i], c); }
    const float m1 = wave_sum(a) / n_loc, m2 = wave_sum(c) / n_loc;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int cc = i * 64 + lane;
      if (cc < n_loc) d_loc[row * n_loc + cc] = rstd * (d[i] - m1 - v[i] * m2);
    }
  }
  {
    const bool masked = row_mask && row_mask[(row / mask_row_div) * mask_ld + (row % mask_row_div)];
    float* dl = d_logits + row * dl_ld;
    for (int cc = n_cls + lane; cc < dl_ld; cc += 64) dl[cc] = 0.f;       // pad classes of the K-padded operand
    if (masked) {
#pragma unroll
      for (int i = 0; i < RF_MAXC; ++i) {
        const int cc = i * 64 + lane;
        if (cc < n_cls) dl[cc] = 0.f;
      }
      return;
    }
    float pv[RF_MAXC], v[RF_MAXC], d[RF_MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int cc = i * 64 + lane;
      pv[i] = cc < n_cls ? sim[row * n_cls + cc] : 0.f;
      d[i] = cc < n_cls ? dor[RF_G + n_loc + cc] : 0.f;
      s += pv[i];
    }
    const float mean = wave_sum(s) / n_cls;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int cc = i * 64 + lane;
      v[i] = cc < n_cls ? pv[i] - mean : 0.f;
      q = fmaf(v[i], v[i], q);
    }
    const float rstd = rsqrtf(wave_sum(q) / n_cls + ln_eps);
    float a = 0.f, c = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) { v[i] *= rstd; a += d[i]; c = fmaf(d[i], v[i], c); }
    const float m1 = wave_sum(a) / n_cls, m2 = wave_sum(c) / n_cls;
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int cc = i * 64 + lane;
      float dp = cc < n_cls ? rstd * (d[i] - m1 - v[i] * m2) : 0.f;
      if (dsim && cc < n_cls) dp += dsim[row * n_cls + cc];
      d[i] = dp;
      dot = fmaf(pv[i], dp, dot);
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int i = 0; i < RF_MAXC; ++i) {
      const int cc = i * 64 + lane;
      if (cc < n_cls) dl[cc] = pv[i] * (d[i] - dot);
    }
  }
}

}  // namespace

extern "C" int gvd_add_layernorm_unbiased(const float* x, const float* y, const float* gamma, const float* beta,
                                          float* out, int64_t rows, const int* rows_dev, int D, float eps,
                                          gvd_stream_t stream) {
  if (!x || !gamma || !beta || !out || rows <= 0 || D != 1024) return GVD_EINVAL;
  if (!gvd_aligned16(x) || (y && !gvd_aligned16(y)) || !gvd_aligned16(out) || !gvd_aligned16(gamma) || !gvd_aligned16(beta))
    return GVD_EINVAL;
  hipLaunchKernelGGL((add_ln_unbiased_kernel<1024, false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, gvd_s(stream), x,
                     y, gamma, beta, out, rows, rows_dev, eps, 0u, 1.0f, (uint64_t)0);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_add_layernorm_unbiased_drop(const float* x, const float* y, const float* gamma, const float* beta,
                                               float* out, int64_t rows, int D, float eps, float p_drop, uint64_t seed,
                                               gvd_stream_t stream) {
  if (!x || !y || !gamma || !beta || !out || rows <= 0 || D != 1024 || !(p_drop > 0.f) || !(p_drop < 1.f)) return GVD_EINVAL;
  if (!gvd_aligned16(x) || !gvd_aligned16(y) || !gvd_aligned16(out) || !gvd_aligned16(gamma) || !gvd_aligned16(beta))
    return GVD_EINVAL;
  hipLaunchKernelGGL((add_ln_unbiased_kernel<1024, true>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, gvd_s(stream), x,
                     y, gamma, beta, out, rows, (const int*)nullptr, eps, gvd_drop_thresh(p_drop), 1.0f / (1.0f - p_drop),
                     seed);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_add_layernorm_unbiased_bwd_parts(int64_t rows) { return (int)((rows + LNB_ROWS - 1) / LNB_ROWS); }

extern "C" int gvd_add_layernorm_unbiased_bwd(const float* x, const float* y, const float* dout, const float* gamma,
                                              float* ds, float* partials, int64_t rows, int D, float eps,
                                              gvd_stream_t stream) {
  if (!x || !dout || !gamma || !ds || !partials || rows <= 0 || D != 1024) return GVD_EINVAL;
  if (!gvd_aligned16(x) || (y && !gvd_aligned16(y)) || !gvd_aligned16(dout) || !gvd_aligned16(ds) || !gvd_aligned16(gamma) ||
      !gvd_aligned16(partials))
    return GVD_EINVAL;
  hipLaunchKernelGGL((add_ln_unbiased_bwd_kernel<1024, false>), dim3((unsigned)((rows + LNB_ROWS - 1) / LNB_ROWS)), dim3(256),
                     0, gvd_s(stream), x, y, dout, gamma, ds, (float*)nullptr, partials, rows, eps, 0u, 1.0f, (uint64_t)0);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_add_layernorm_unbiased_drop_bwd(const float* x, const float* y, const float* dout, const float* gamma,
                                                   float* ds, float* dy, float* partials, int64_t rows, int D, float eps,
                                                   float p_drop, uint64_t seed, gvd_stream_t stream) {
  if (!x || !y || !dout || !gamma || !ds || !dy || !partials || rows <= 0 || D != 1024 || !(p_drop > 0.f) || !(p_drop < 1.f))
    return GVD_EINVAL;
  if (!gvd_aligned16(x) || !gvd_aligned16(y) || !gvd_aligned16(dout) || !gvd_aligned16(ds) || !gvd_aligned16(dy) ||
      !gvd_aligned16(gamma) || !gvd_aligned16(partials))
    return GVD_EINVAL;
  hipLaunchKernelGGL((add_ln_unbiased_bwd_kernel<1024, true>), dim3((unsigned)((rows + LNB_ROWS - 1) / LNB_ROWS)), dim3(256),
                     0, gvd_s(stream), x, y, dout, gamma, ds, dy, partials, rows, eps, gvd_drop_thresh(p_drop),
                     1.0f / (1.0f - p_drop), seed);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_region_feature_rows(const float* g_pool, const float* loc, int n_loc, const float* sim_logits,
                                       int n_cls, int64_t logits_ld, const uint8_t* row_mask, int64_t mask_rows_per_batch,
                                       int64_t mask_ld, float* out, int64_t out_ld, float* sim_out, int64_t rows,
                                       const int* rows_dev, int G, float ln_eps, gvd_stream_t stream) {
  if (!g_pool || !loc || !sim_logits || !out || rows <= 0 || G != RF_G || n_loc <= 0 || n_loc > 64 * RF_MAXC ||
      n_cls <= 0 || n_cls > 64 * RF_MAXC || logits_ld < n_cls || !gvd_aligned16(g_pool) || out_ld < G + n_loc + n_cls ||
      ((out_ld & 3) == 0 && !gvd_aligned16(out)))
    return GVD_EINVAL;
  if (row_mask && mask_rows_per_batch <= 0) return GVD_EINVAL;
  hipLaunchKernelGGL(region_feature_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, gvd_s(stream), g_pool,
                     loc, n_loc, sim_logits, n_cls, logits_ld, row_mask, row_mask ? mask_rows_per_batch : 1, mask_ld, out,
                     out_ld, sim_out, rows, rows_dev, ln_eps);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_region_feature_rows_bwd(const float* g_pool, const float* loc, int n_loc, const float* sim, int n_cls,
                                           const uint8_t* row_mask, int64_t mask_rows_per_batch, int64_t mask_ld,
                                           const float* d_out, int64_t d_out_ld, const float* d_sim, float* d_gpool,
                                           float* d_loc, float* d_logits, int64_t d_logits_ld, int64_t rows, int G,
                                           float ln_eps, gvd_stream_t stream) {
  if (!g_pool || !loc || !sim || !d_out || !d_gpool || !d_loc || !d_logits || rows <= 0 || G != RF_G || n_loc <= 0 ||
      n_loc > 64 * RF_MAXC || n_cls <= 0 || n_cls > 64 * RF_MAXC || d_out_ld < G + n_loc + n_cls || (d_out_ld & 3) ||
      d_logits_ld < n_cls || !gvd_aligned16(g_pool) || !gvd_aligned16(d_out) || !gvd_aligned16(d_gpool))
    return GVD_EINVAL;
  if (row_mask && mask_rows_per_batch <= 0) return GVD_EINVAL;
  hipLaunchKernelGGL(region_feature_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, gvd_s(stream), g_pool,
                     loc, n_loc, sim, n_cls, row_mask, row_mask ? mask_rows_per_batch : 1, mask_ld, d_out, d_out_ld, d_sim,
                     d_gpool, d_loc, d_logits, d_logits_ld, rows, ln_eps);
  GVD_CHECK_LAUNCH();
  return 0;
}
