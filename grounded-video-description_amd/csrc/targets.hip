// Training targets and loss reductions of the GVD train step.
//   gvd_iou_targets   utils.py:293-305 -> bbox_transform.py:224-269 (IoU '+1' convention, masks, zero-area
//                     rules) and sim_mat_target
//   gvd_step_targets  utils.py:307-328 (bbox_target) + model.py:431-440 (per-step frame mask), all Lc
//                     steps in one launch instead of once per token
//   gvd_masked_lsm_loss  utils.py:139,142: -mean(log_softmax(x)[label]) as (sum, count) accumulators
// All tiny / HBM-light: one thread per output element or one workgroup per row, fp32 arithmetic in the
// same operation order as the reference so the integer/boolean outputs are bit-exact.
#include "gvd_common.h"

namespace {

__global__ void iou_kernel(const float* __restrict__ ppls, int ppl_ld, const float* __restrict__ gt, int gt_ld,
                           const uint8_t* __restrict__ frm_mask, const uint8_t* __restrict__ pnt_mask, int B, int R,
                           int K, float* __restrict__ overlaps, int64_t* __restrict__ sim_target) {
#pragma clang fp contract(off)   // the reference rounds every product/sum separately: no fma fusion here
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * R * K) return;
  const int k = idx % K;
  const int r = (idx / K) % R;
  const int b = idx / ((int64_t)K * R);
  const float* a = ppls + ((int64_t)b * R + r) * ppl_ld;
  const float* g = gt + ((int64_t)b * K + k) * gt_ld;
  const float gx = g[2] - g[0] + 1.f, gy = g[3] - g[1] + 1.f;
  const float ax = a[2] - a[0] + 1.f, ay = a[3] - a[1] + 1.f;
  // opaque barriers: each product is rounded on its own like the reference's tensor ops (no fma fusion
  // with the following add/sub, whatever -ffp-contract says)
  float g_area = gx * gy, a_area = ax * ay;
  asm volatile("" : "+v"(g_area));
  asm volatile("" : "+v"(a_area));
  float iw = fminf(a[2], g[2]) - fmaxf(a[0], g[0]) + 1.f;
  if (iw < 0.f) iw = 0.f;
  float ih = fminf(a[3], g[3]) - fmaxf(a[1], g[1]) + 1.f;
  if (ih < 0.f) ih = 0.f;
  // no fp contraction: the reference evaluates (a_area + g_area) - iw*ih and (iw*ih)/ua with separate roundings
  float inter = iw * ih;
  asm volatile("" : "+v"(inter));
  float ua = a_area + g_area;
  asm volatile("" : "+v"(ua));
  ua = ua - inter;
  // The CPU reference divides with IEEE-correct rounding; hipcc's fp32 '/' expansion was measured 1 ulp off.
  // Divide in fp64 (correctly rounded, and 53 >= 2*24+2 bits makes the double->float rounding exact); the
  // empty asm keeps LLVM from folding fptrunc(fdiv(fpext, fpext)) back into an fp32 divide.
  double qd = (double)inter / (double)ua;
  asm volatile("" : "+v"(qd));
  float ov = (float)qd;
  const bool masked = frm_mask[idx] | pnt_mask[(int64_t)b * (R + 1) + 1 + r];
  ov = ov * (masked ? 0.f : 1.f);
  if (gx == 1.f && gy == 1.f) ov = 0.f;
  if (ax == 1.f && ay == 1.f) ov = -1.f;
  overlaps[idx] = ov;
  if (sim_target) sim_target[((int64_t)b * K + k) * R + r] = (ov > 0.5f) ? (int64_t)g[5] : 0;
}

__global__ void step_targets_kernel(const float* __restrict__ overlaps, const uint8_t* __restrict__ mask_boxes,
                                    const uint8_t* __restrict__ frm_mask, const uint8_t* __restrict__ pnt_mask,
                                    int B, int R, int K, int Lp1, int Lc, float* __restrict__ roi_labels,
                                    uint8_t* __restrict__ frm_masks) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * Lc * (R+1)
  if (idx >= (int64_t)B * Lc * (R + 1)) return;
  const int c = idx % (R + 1);
  const int t = (idx / (R + 1)) % Lc;
  const int b = idx / ((int64_t)(R + 1) * Lc);
  if (c == 0) { frm_masks[idx] = pnt_mask[(int64_t)b * (R + 1)]; return; }   // [0 | ...] | pnt_mask col 0
  const int r = c - 1;
  const uint8_t* mb = mask_boxes + (int64_t)b * K * Lp1 + (t + 1);   // [B,1,K,L+1]: element (k, t+1)
  float best = -INFINITY;
  int on = 0;
  for (int k = 0; k < K; ++k) {
    const uint8_t m = mb[(int64_t)k * Lp1];
    const float ov = m ? 0.f : overlaps[((int64_t)b * R + r) * K + k];
    best = fmaxf(best, ov);
    on += (m | frm_mask[((int64_t)b * R + r) * K + k]) ? 0 : 1;
  }
  roi_labels[((int64_t)b * Lc + t) * R + r] = (best > 0.5f) ? 1.f : 0.f;
  frm_masks[idx] = ((on <= 0) ? 1 : 0) | pnt_mask[(int64_t)b * (R + 1) + c];
}

__global__ __launch_bounds__(256) void masked_lsm_loss_kernel(const float* __restrict__ x, int64_t ldx,
                                                              const float* __restrict__ label, int64_t ldl, int N,
                                                              float* acc, float* row_lse) {
  __shared__ float s_red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (int64_t)row * ldx;
  const float* lr = label + (int64_t)row * ldl;
  float mx = -INFINITY;
  for (int i = tid; i < N; i += 256) mx = fmaxf(mx, xr[i]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float se = 0.f;
  for (int i = tid; i < N; i += 256) se += expf(xr[i] - mx);
  se = wave_sum(se);
  if ((tid & 63) == 0) s_red[tid >> 6] = se;
  __syncthreads();
  se = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  __syncthreads();
  const float lse = logf(se);
  float s = 0.f, cnt = 0.f;
  for (int i = tid; i < N; i += 256)
    if (lr[i] != 0.f) { s -= (xr[i] - mx) - lse; cnt += 1.f; }
  s = wave_sum(s); cnt = wave_sum(cnt);
  if ((tid & 63) == 0) { s_red[tid >> 6] = s; }
  __syncthreads();
  const float st = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  __syncthreads();
  if ((tid & 63) == 0) { s_red[tid >> 6] = cnt; }
  __syncthreads();
  if (tid == 0) {
    const float ct = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    // per-row partials; masked_lsm_reduce_kernel adds them in row order (no atomics: run-to-run bit-reproducible)
    acc[2 + 2 * row] = ct > 0.f ? st : 0.f;
    acc[3 + 2 * row] = ct > 0.f ? ct : 0.f;
    if (row_lse) row_lse[row] = mx + lse;
  }
}

// acc[0] = sum of the row sums, acc[1] = sum of the row counts, in a fixed order: lane-strided partial sums of one wave,
// then the xor-shuffle tree
__global__ __launch_bounds__(64) void masked_lsm_reduce_kernel(float* acc, int rows) {
  float s = 0.f, c = 0.f;
  for (int r = threadIdx.x; r < rows; r += 64) { s += acc[2 + 2 * r]; c += acc[3 + 2 * r]; }
  s = wave_sum(s); c = wave_sum(c);
  if (threadIdx.x == 0) { acc[0] = s; acc[1] = c; }
}

// Region-classification loss (model.py:345-350): over the (box k, proposal r) pairs with sim_target > 0,
// -mean( clamp(log sim_mat[b, sim_target[b,k,r], r], -100) ): gather + log + masked mean in one pass, per-block partials
// (sum, count) reduced in a fixed order by masked_lsm_reduce_kernel.
__global__ __launch_bounds__(256) void cls_loss_kernel(const float* __restrict__ sim, int64_t sb, int64_t sc, int64_t sr,
                                                       const int64_t* __restrict__ tgt, int R, int K, int64_t n,
                                                       float* acc) {
  __shared__ float s_red[8];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float s = 0.f, c = 0.f;
  if (i < n) {
    const int64_t t = tgt[i];
    if (t > 0) {
      const int r = (int)(i % R);
      const int64_t b = i / ((int64_t)R * K);
      s = -fmaxf(logf(sim[b * sb + t * sc + r * sr]), -100.f);
      c = 1.f;
    }
  }
  s = wave_sum(s); c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = s; s_red[4 + (threadIdx.x >> 6)] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    acc[2 + 2 * blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    acc[3 + 2 * blockIdx.x] = s_red[4] + s_red[5] + s_red[6] + s_red[7];
  }
}

}  // namespace

extern "C" int gvd_cls_loss(const float* sim_mat, int64_t stride_b, int64_t stride_cls, int64_t stride_r,
                            const int64_t* sim_target, int B, int D1, int R, int K, float* acc, gvd_stream_t stream) {
  if (!sim_mat || !sim_target || !acc || B <= 0 || D1 <= 0 || R <= 0 || K <= 0) return GVD_EINVAL;
  const int64_t n = (int64_t)B * K * R;
  const int nblk = (int)((n + 255) / 256);
  hipLaunchKernelGGL(cls_loss_kernel, dim3((unsigned)nblk), dim3(256), 0, gvd_s(stream), sim_mat, stride_b, stride_cls,
                     stride_r, sim_target, R, K, n, acc);
  hipLaunchKernelGGL(masked_lsm_reduce_kernel, dim3(1), dim3(64), 0, gvd_s(stream), acc, nblk);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_iou_targets(const float* ppls, int ppl_ld, const float* gt, int gt_ld, const uint8_t* frm_mask,
                               const uint8_t* pnt_mask, int B, int R, int K, float* overlaps, int64_t* sim_target,
                               gvd_stream_t stream) {
  if (!ppls || !gt || !frm_mask || !pnt_mask || !overlaps || B <= 0 || R <= 0 || K <= 0 || ppl_ld < 5 || gt_ld < 6)
    return GVD_EINVAL;
  const int64_t n = (int64_t)B * R * K;
  hipLaunchKernelGGL(iou_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, gvd_s(stream), ppls, ppl_ld, gt,
                     gt_ld, frm_mask, pnt_mask, B, R, K, overlaps, sim_target);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_step_targets(const float* overlaps, const uint8_t* mask_boxes, const uint8_t* frm_mask,
                                const uint8_t* pnt_mask, int B, int R, int K, int Lp1, int Lc, float* roi_labels,
                                uint8_t* frm_masks, gvd_stream_t stream) {
  if (!overlaps || !mask_boxes || !frm_mask || !pnt_mask || !roi_labels || !frm_masks || B <= 0 || R <= 0 ||
      K <= 0 || Lc <= 0 || Lc + 1 > Lp1)
    return GVD_EINVAL;
  const int64_t n = (int64_t)B * Lc * (R + 1);
  hipLaunchKernelGGL(step_targets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, gvd_s(stream), overlaps,
                     mask_boxes, frm_mask, pnt_mask, B, R, K, Lp1, Lc, roi_labels, frm_masks);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_masked_lsm_loss(const float* x, int64_t ldx, const float* label, int64_t ld_label, int rows,
                                   int N, float* acc, float* row_lse, gvd_stream_t stream) {
  if (!x || !label || !acc || rows <= 0 || N <= 0) return GVD_EINVAL;
  hipLaunchKernelGGL(masked_lsm_loss_kernel, dim3((unsigned)rows), dim3(256), 0, gvd_s(stream), x, ldx, label,
                     ld_label, N, acc, row_lse);
  hipLaunchKernelGGL(masked_lsm_reduce_kernel, dim3(1), dim3(64), 0, gvd_s(stream), acc, rows);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" const char* gvd_version(void) { return "gvd_hip 0.1 (gfx950, fp32 MFMA)"; }
extern "C" int gvd_abi_version(void) { return GVD_ABI_VERSION; }
