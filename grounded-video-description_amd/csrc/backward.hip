// Backward kernels of the teacher-forced decoder loop (hand-scheduled BPTT, decoder_bwd.py).
//
//  gvd_lstm_cell_bwd    pointwise part of nn.LSTMCell's backward (the dX / dW GEMMs are plain library GEMMs)
//  gvd_attn_bwd_step    one pass over feats / p_feats per (step, side): softmax + tanh backward, producing
//                       de[n] (kept for the final pass) and per-chunk partials of d_q, d_w, d_alpha_bias
//  gvd_attn_bwd_pfeats  after the loop: d_p_feats[b,n,:] = sum_t de_t[n] * w * (1 - tanh^2(p_feats[n]+q_t))
//                       — reads p_feats once and writes d_p_feats once for ALL steps (tanh recomputed, never
//                       stored: SURVEY.md §7 "BPTT memory")
//
// d_feats[b] = alpha[b]^T d_ctx[b] over all steps is a plain batched GEMM done by the host side.
#include "gvd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ATT_A = 512;
constexpr int ATT_H = 1024;

__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* __restrict__ dh, int64_t lddh,
                                                       const float* __restrict__ dh2, int64_t lddh2,
                                                       const float* __restrict__ dc_next, int64_t lddc,
                                                       const float* __restrict__ gates, int64_t ldg,
                                                       const float* __restrict__ c_prev, int64_t ldcp,
                                                       const float* __restrict__ c_new, int64_t ldcn, int B, int H,
                                                       float* __restrict__ dgates, int64_t lddg,
                                                       float* __restrict__ dc_prev, int64_t lddcp) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * H) return;
  const int b = idx / H, j = idx % H;
  const float* g = gates + (int64_t)b * ldg;
  const float gi = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
  const float tc = tanhf(c_new[(int64_t)b * ldcn + j]);
  float dhv = dh[(int64_t)b * lddh + j];
  if (dh2) dhv += dh2[(int64_t)b * lddh2 + j];
  float dc = dhv * go * (1.f - tc * tc);
  if (dc_next) dc += dc_next[(int64_t)b * lddc + j];
  float* dg = dgates + (int64_t)b * lddg;
  dg[j] = dc * gg * gi * (1.f - gi);
  dg[H + j] = dc * c_prev[(int64_t)b * ldcp + j] * gf * (1.f - gf);
  dg[2 * H + j] = dc * gi * (1.f - gg * gg);
  dg[3 * H + j] = dhv * tc * go * (1.f - go);
  dc_prev[(int64_t)b * lddcp + j] = dc * gf;
}

struct BwdStepParams {
  const float* feats; const float* p_feats; const float* q; int64_t ldq;
  const float* w;
  const float* alpha; int64_t ld_alpha;        // softmax weights of this step [B,N]
  const float* ctx; int64_t ld_ctx;            // this side's context [B,H]
  const float* d_ctx; int64_t ld_dctx;         // [B,H]
  const float* d_logits; int64_t ld_dlogits;   // [B,N] or NULL
  const uint8_t* att_mask; int64_t ld_att_mask;
  const uint8_t* pnt_mask; int64_t ld_pnt_mask;
  float* de_out; int64_t ld_de;                // [B,N]
  float* dq_part; float* dw_part; float* dab_part;   // [B,NC,A], [B,NC,A], [B,NC]
  int N, chunk, nchunks;
};

// MODE = the side's score function (gvd_attn_side.score_mode): with u = x + q (ADD) or u = x q (MUL), t = tanh(u):
//   ADD  d q += de w (1 - t^2)          d w += de t     d alpha_bias += de
//   MUL  d q += de w (1 - t^2) x        d w += de t     d alpha_bias += de
//   DOT  d q += de x                    (no alpha_net: the d w / d alpha_bias partials are written as zeros)
template <int MODE>
__global__ __launch_bounds__(256) void attn_bwd_step_kernel(const BwdStepParams p) {
  __shared__ float s_red[8];
  __shared__ float s_acc[4][2 * ATT_A];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y, c = blockIdx.x;
  const int n0 = c * p.chunk;
  const int rows = min(p.chunk, p.N - n0);

  // d_ctx (16 values per lane: columns 4*lane + 256*j) and ctx . d_ctx
  const float* dcb = p.d_ctx + (int64_t)b * p.ld_dctx;
  const float* cb = p.ctx + (int64_t)b * p.ld_ctx;
  f32x4 dcv[4];
  float dot = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dcv[j] = *reinterpret_cast<const f32x4*>(dcb + 256 * j + 4 * lane);
    const f32x4 cv = *reinterpret_cast<const f32x4*>(cb + 256 * j + 4 * lane);
    dot += cv[0] * dcv[j][0] + cv[1] * dcv[j][1] + cv[2] * dcv[j][2] + cv[3] * dcv[j][3];
  }
  dot = wave_sum(dot);   // every wave holds the full 1024-term dot product

  const float* qb = p.q + (int64_t)b * p.ldq;
  const f32x4 q0 = *reinterpret_cast<const f32x4*>(qb + 4 * lane);
  const f32x4 q1 = *reinterpret_cast<const f32x4*>(qb + 256 + 4 * lane);
  f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
  if constexpr (MODE != GVD_SCORE_DOT) {
    w0 = *reinterpret_cast<const f32x4*>(p.w + 4 * lane);
    w1 = *reinterpret_cast<const f32x4*>(p.w + 256 + 4 * lane);
  }
  const float* fb = p.feats + ((int64_t)b * p.N + n0) * ATT_H;
  const float* pf = p.p_feats + ((int64_t)b * p.N + n0) * ATT_A;
  const float* al = p.alpha + (int64_t)b * p.ld_alpha + n0;
  const float* dl = p.d_logits ? p.d_logits + (int64_t)b * p.ld_dlogits + n0 : nullptr;
  const uint8_t* am = p.att_mask ? p.att_mask + (int64_t)b * p.ld_att_mask + n0 : nullptr;
  const uint8_t* pm = p.pnt_mask ? p.pnt_mask + (int64_t)b * p.ld_pnt_mask + n0 : nullptr;
  float* deo = p.de_out + (int64_t)b * p.ld_de + n0;

  f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = dq0, dw0 = dq0, dw1 = dq0;
  float dab = 0.f;
  for (int r = wave; r < rows; r += 4) {
    // A row the attention mask removed has de = 0 exactly (masked_fill blocks both gradient paths): it adds 0 to every partial
    // sum below - its 6 KB of features are not fetched (wave-uniform; a fifth of the region rows at the loader's masking rate,
    // like the forward kernel's row skip)
    if (am && am[r]) {
      if (lane == 0) deo[r] = 0.f;
      continue;
    }
    // d alpha[r] = feats[r,:] . d_ctx
    float da = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 f = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(fb + (int64_t)r * ATT_H + 256 * j + 4 * lane));   // read-once stream
      da += f[0] * dcv[j][0] + f[1] * dcv[j][1] + f[2] * dcv[j][2] + f[3] * dcv[j][3];
    }
    const f32x4 x0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pf + (int64_t)r * ATT_A + 4 * lane));
    const f32x4 x1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pf + (int64_t)r * ATT_A + 256 + 4 * lane));
    da = wave_sum(da);
    float de = al[r] * (da - dot);                           // softmax backward
    if (dl && !(pm && pm[r])) de += dl[r];                    // gradient through `att2_weight` (pre-softmax logits)
    if (lane == 0) deo[r] = de;
    dab += de;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (MODE == GVD_SCORE_ADD) {
        const float t0 = tanh_fast(x0[k] + q0[k]), t1 = tanh_fast(x1[k] + q1[k]);
        dq0[k] = fmaf(de * w0[k], 1.f - t0 * t0, dq0[k]);
        dq1[k] = fmaf(de * w1[k], 1.f - t1 * t1, dq1[k]);
        dw0[k] = fmaf(de, t0, dw0[k]);
        dw1[k] = fmaf(de, t1, dw1[k]);
      } else if constexpr (MODE == GVD_SCORE_MUL) {
        const float t0 = tanh_fast(x0[k] * q0[k]), t1 = tanh_fast(x1[k] * q1[k]);
        dq0[k] = fmaf(de * w0[k] * x0[k], 1.f - t0 * t0, dq0[k]);
        dq1[k] = fmaf(de * w1[k] * x1[k], 1.f - t1 * t1, dq1[k]);
        dw0[k] = fmaf(de, t0, dw0[k]);
        dw1[k] = fmaf(de, t1, dw1[k]);
      } else {
        dq0[k] = fmaf(de, x0[k], dq0[k]);
        dq1[k] = fmaf(de, x1[k], dq1[k]);
      }
    }
  }
  if constexpr (MODE == GVD_SCORE_DOT) dab = 0.f;
  // cross-wave reduction of the per-lane partials through LDS, then one deterministic partial per chunk
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s_acc[wave][4 * lane + k] = dq0[k];
    s_acc[wave][256 + 4 * lane + k] = dq1[k];
    s_acc[wave][ATT_A + 4 * lane + k] = dw0[k];
    s_acc[wave][ATT_A + 256 + 4 * lane + k] = dw1[k];
  }
  if (lane == 0) s_red[wave] = dab;
  __syncthreads();
  const int64_t pc = (int64_t)b * p.nchunks + c;
  for (int i = tid; i < 2 * ATT_A; i += 256) {
    const float v = s_acc[0][i] + s_acc[1][i] + s_acc[2][i] + s_acc[3][i];
    if (i < ATT_A) p.dq_part[pc * ATT_A + i] = v;
    else p.dw_part[pc * ATT_A + (i - ATT_A)] = v;
  }
  if (tid == 0) p.dab_part[pc] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

struct BwdPfParams {
  const float* p_feats;        // [B,N,A]
  const float* q_all; int64_t q_step_stride; int64_t ldq;   // q of step t, sample b: q_all + t*step + b*ldq
  const float* de_all; int64_t de_step_stride; int64_t ld_de;
  const float* w;
  float* d_p_feats;            // [B,N,A]
  int N, Lc, chunk;
};
// MODE as in attn_bwd_step_kernel: d x = sum_t de w (1 - t^2) [ADD], ... (1 - t^2) q [MUL], sum_t de q [DOT]

constexpr int PF_MAX_L = 40;   // seq_length 20 (40 for GT-sentence grounding, README.md:115)

template <int MODE>
__global__ __launch_bounds__(256) void attn_bwd_pfeats_kernel(const BwdPfParams p) {
  extern __shared__ __attribute__((aligned(16))) float s_q[];   // [Lc][A]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y, c = blockIdx.x;
  const int n0 = c * p.chunk;
  const int rows = min(p.chunk, p.N - n0);
  for (int i = tid; i < p.Lc * (ATT_A / 4); i += 256) {
    const int t = i / (ATT_A / 4), a4 = i % (ATT_A / 4);
    *reinterpret_cast<f32x4*>(s_q + t * ATT_A + 4 * a4) =
        *reinterpret_cast<const f32x4*>(p.q_all + (int64_t)t * p.q_step_stride + (int64_t)b * p.ldq + 4 * a4);
  }
  __syncthreads();
  f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = w0;
  if constexpr (MODE != GVD_SCORE_DOT) {
    w0 = *reinterpret_cast<const f32x4*>(p.w + 4 * lane);
    w1 = *reinterpret_cast<const f32x4*>(p.w + 256 + 4 * lane);
  }
  for (int r = wave; r < rows; r += 4) {
    const int n = n0 + r;
    const float* px = p.p_feats + ((int64_t)b * p.N + n) * ATT_A;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(px + 4 * lane);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(px + 256 + 4 * lane);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int t = 0; t < p.Lc; ++t) {
      const float de = p.de_all[(int64_t)t * p.de_step_stride + (int64_t)b * p.ld_de + n];
      if (de == 0.f) continue;     // wave-uniform: masked rows / steps without gradient skip the tanh work
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(s_q + t * ATT_A + 4 * lane);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(s_q + t * ATT_A + 256 + 4 * lane);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if constexpr (MODE == GVD_SCORE_ADD) {
          const float t0 = tanh_fast(x0[k] + q0[k]), t1 = tanh_fast(x1[k] + q1[k]);
          a0[k] = fmaf(de * w0[k], 1.f - t0 * t0, a0[k]);
          a1[k] = fmaf(de * w1[k], 1.f - t1 * t1, a1[k]);
        } else if constexpr (MODE == GVD_SCORE_MUL) {
          const float t0 = tanh_fast(x0[k] * q0[k]), t1 = tanh_fast(x1[k] * q1[k]);
          a0[k] = fmaf(de * w0[k] * q0[k], 1.f - t0 * t0, a0[k]);
          a1[k] = fmaf(de * w1[k] * q1[k], 1.f - t1 * t1, a1[k]);
        } else {
          a0[k] = fmaf(de, q0[k], a0[k]);
          a1[k] = fmaf(de, q1[k], a1[k]);
        }
      }
    }
    float* o = p.d_p_feats + ((int64_t)b * p.N + n) * ATT_A;
    *reinterpret_cast<f32x4*>(o + 4 * lane) = a0;
    *reinterpret_cast<f32x4*>(o + 256 + 4 * lane) = a1;
  }
}

}  // namespace

extern "C" int gvd_lstm_cell_bwd(const float* dh, int64_t lddh, const float* dh2, int64_t lddh2, const float* dc_next,
                                 int64_t lddc, const float* gates, int64_t ldg, const float* c_prev, int64_t ldcp,
                                 const float* c_new, int64_t ldcn, int B, int H, float* dgates, int64_t lddg,
                                 float* dc_prev, int64_t lddcp, gvd_stream_t stream) {
  if (!dh || !gates || !c_prev || !c_new || !dgates || !dc_prev || B <= 0 || H <= 0) return GVD_EINVAL;
  const int64_t n = (int64_t)B * H;
  hipLaunchKernelGGL(lstm_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, gvd_s(stream), dh, lddh,
                     dh2, lddh2, dc_next, lddc, gates, ldg, c_prev, ldcp, c_new, ldcn, B, H, dgates, lddg, dc_prev, lddcp);
  GVD_CHECK_LAUNCH();
  return 0;
}

static int bwd_chunk(int N, int B) {
  int chunk = 48;
  while (chunk > 12 && (long)B * ((N + chunk - 1) / chunk) < 1024) chunk /= 2;
  if (chunk > N) chunk = N;
  return chunk;
}

extern "C" int gvd_attn_bwd_chunks(int N, int B) {
  if (N <= 0 || B <= 0) return 0;
  const int c = bwd_chunk(N, B);
  return (N + c - 1) / c;
}

extern "C" int gvd_attn_bwd_step(const gvd_attn_side* side, int B, int A, int H, const float* alpha,
                                 int64_t ld_alpha, const float* ctx, int64_t ld_ctx, const float* d_ctx,
                                 int64_t ld_dctx, const float* d_logits, int64_t ld_dlogits, float* de_out,
                                 int64_t ld_de, float* dq_part, float* dw_part, float* dab_part,
                                 gvd_stream_t stream) {
  if (!side || A != ATT_A || H != ATT_H || B <= 0 || !alpha || !ctx || !d_ctx || !de_out || !dq_part || !dw_part ||
      !dab_part || side->N <= 0 || side->score_mode < GVD_SCORE_ADD || side->score_mode > GVD_SCORE_DOT ||
      (side->score_mode != GVD_SCORE_DOT && !side->w))
    return GVD_EINVAL;
  if (!gvd_aligned16(side->feats) || !gvd_aligned16(side->p_feats) || !gvd_aligned16(side->q) || !gvd_aligned16(ctx) ||
      !gvd_aligned16(d_ctx) || (ld_ctx % 4) || (ld_dctx % 4) || (side->ldq % 4))
    return GVD_EINVAL;
  BwdStepParams p = {};
  p.feats = side->feats; p.p_feats = side->p_feats; p.q = side->q; p.ldq = side->ldq; p.w = side->w;
  p.alpha = alpha; p.ld_alpha = ld_alpha; p.ctx = ctx; p.ld_ctx = ld_ctx; p.d_ctx = d_ctx; p.ld_dctx = ld_dctx;
  p.d_logits = d_logits; p.ld_dlogits = ld_dlogits;
  p.att_mask = side->att_mask; p.ld_att_mask = side->ld_att_mask;
  p.pnt_mask = side->pnt_mask; p.ld_pnt_mask = side->ld_pnt_mask;
  p.de_out = de_out; p.ld_de = ld_de; p.dq_part = dq_part; p.dw_part = dw_part; p.dab_part = dab_part;
  p.N = side->N;
  p.chunk = bwd_chunk(side->N, B);
  p.nchunks = (side->N + p.chunk - 1) / p.chunk;
  const dim3 grid((unsigned)p.nchunks, (unsigned)B);
  if (side->score_mode == GVD_SCORE_ADD) hipLaunchKernelGGL(attn_bwd_step_kernel<GVD_SCORE_ADD>, grid, dim3(256), 0, gvd_s(stream), p);
  else if (side->score_mode == GVD_SCORE_MUL) hipLaunchKernelGGL(attn_bwd_step_kernel<GVD_SCORE_MUL>, grid, dim3(256), 0, gvd_s(stream), p);
  else hipLaunchKernelGGL(attn_bwd_step_kernel<GVD_SCORE_DOT>, grid, dim3(256), 0, gvd_s(stream), p);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_attn_bwd_pfeats(const float* p_feats, int B, int N, int A, const float* q_all,
                                   int64_t q_step_stride, int64_t ldq, const float* de_all, int64_t de_step_stride,
                                   int64_t ld_de, const float* w, int Lc, float* d_p_feats, int score_mode,
                                   gvd_stream_t stream) {
  if (!p_feats || !q_all || !de_all || !d_p_feats || A != ATT_A || B <= 0 || N <= 0 || Lc <= 0 || Lc > PF_MAX_L ||
      score_mode < GVD_SCORE_ADD || score_mode > GVD_SCORE_DOT || (score_mode != GVD_SCORE_DOT && !w))
    return GVD_EINVAL;
  if (!gvd_aligned16(p_feats) || !gvd_aligned16(q_all) || !gvd_aligned16(d_p_feats) || (ldq % 4) || (q_step_stride % 4))
    return GVD_EINVAL;
  BwdPfParams p = {};
  p.p_feats = p_feats; p.q_all = q_all; p.q_step_stride = q_step_stride; p.ldq = ldq;
  p.de_all = de_all; p.de_step_stride = de_step_stride; p.ld_de = ld_de; p.w = w; p.d_p_feats = d_p_feats;
  p.N = N; p.Lc = Lc;
  int chunk = 32;
  while (chunk > 8 && (long)B * ((N + chunk - 1) / chunk) < 1024) chunk /= 2;
  p.chunk = chunk;
  const size_t lds = (size_t)Lc * ATT_A * sizeof(float);
  const void* fn = score_mode == GVD_SCORE_ADD ? reinterpret_cast<const void*>(attn_bwd_pfeats_kernel<GVD_SCORE_ADD>)
                   : score_mode == GVD_SCORE_MUL ? reinterpret_cast<const void*>(attn_bwd_pfeats_kernel<GVD_SCORE_MUL>)
                                                 : reinterpret_cast<const void*>(attn_bwd_pfeats_kernel<GVD_SCORE_DOT>);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid((unsigned)((N + chunk - 1) / chunk), (unsigned)B);
  if (score_mode == GVD_SCORE_ADD) hipLaunchKernelGGL(attn_bwd_pfeats_kernel<GVD_SCORE_ADD>, grid, dim3(256), lds, gvd_s(stream), p);
  else if (score_mode == GVD_SCORE_MUL) hipLaunchKernelGGL(attn_bwd_pfeats_kernel<GVD_SCORE_MUL>, grid, dim3(256), lds, gvd_s(stream), p);
  else hipLaunchKernelGGL(attn_bwd_pfeats_kernel<GVD_SCORE_DOT>, grid, dim3(256), lds, gvd_s(stream), p);
  GVD_CHECK_LAUNCH();
  return 0;
}
