// Optimiser step of the training recipe (main.py:263-266: clip_grad_norm_(parameters, 0.1) then Adam.step()) as own
// multi-tensor kernels: the total gradient norm from ordered per-chunk partial sums (no float atomics: reproducible run
// to run), the clip coefficient on the device, and ONE pass per parameter that scales the gradient, updates both moments
// and the parameter - the clipped gradient is never written back (clip_grad_norm_ + the fused Adam read the gradients
// three times and write them once; here they are read twice).
//
// A launch takes up to GVD_OPT_MAX_TENSORS tensors by value (pointers + sizes in the kernel arguments, no device-side
// table to keep in sync with autograd's freshly allocated .grad tensors); workgroup <-> tensor through the prefix sums of
// the tensors' chunk counts.
#include "gvd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int CHUNK = GVD_OPT_CHUNK;       // elements per workgroup

__device__ __forceinline__ int find_tensor(const gvd_opt_group& g, int chunk) {
  int t = 0;
  while (t + 1 < g.count && g.chunk0[t + 1] <= chunk) ++t;     // <= 32 uniform scalar compares
  return t;
}

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const gvd_opt_group g, float* __restrict__ partials) {
  __shared__ float s_red[4];
  const int t = find_tensor(g, (int)blockIdx.x);
  const int64_t n = g.n[t];
  const int64_t e0 = (int64_t)((int)blockIdx.x - g.chunk0[t]) * CHUNK;
  const int64_t e1 = e0 + CHUNK < n ? e0 + CHUNK : n;
  const float* gr = g.g[t];
  float s = 0.f;
  if (g.vec_ok[t]) {
    const int64_t v1 = e0 + ((e1 - e0) & ~(int64_t)3);
    for (int64_t i = e0 + 4 * threadIdx.x; i < v1; i += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(gr + i);
      s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
    }
    for (int64_t i = v1 + threadIdx.x; i < e1; i += 256) s = fmaf(gr[i], gr[i], s);
  } else {
    for (int64_t i = e0 + threadIdx.x; i < e1; i += 256) s = fmaf(gr[i], gr[i], s);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[g.part0 + blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// one workgroup: ordered sum of the partials -> out[0] = total L2 norm, out[1] = min(1, max_norm / (norm + 1e-6))
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partials, int n, float max_norm,
                                                        float* __restrict__ out) {
  __shared__ double s_red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  s_red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) s_red[threadIdx.x] += s_red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(s_red[0]);
    const float coef = max_norm / (norm + 1e-6f);
    out[0] = norm;
    out[1] = coef < 1.0f ? coef : 1.0f;
  }
}

struct AdamHyper { float beta1, beta2, eps, weight_decay; };

__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float gr, float coef, const AdamHyper& h, float step,
                                          float inv_bc2s) {
  gr *= coef;
  if (h.weight_decay != 0.f) gr = fmaf(h.weight_decay, p, gr);
  m = fmaf(h.beta1, m, (1.f - h.beta1) * gr);
  v = fmaf(h.beta2, v, (1.f - h.beta2) * gr * gr);
  const float denom = sqrtf(v) * inv_bc2s + h.eps;
  p -= step * (m / denom);
}

// skip (nullable): n_skip device words; any non-zero word -> the launch leaves every tensor untouched.  Lets the host
// enqueue the optimiser BEFORE it reads the step's kernel-status / collective flags (train.Trainer): a step whose flags turn
// out raised has not moved the parameters, and the host-side read no longer drains the queue ahead of the optimiser.
__global__ __launch_bounds__(256) void adam_step_kernel(const gvd_opt_group g, const float* __restrict__ clip,
                                                        const int* __restrict__ skip, int n_skip, AdamHyper h) {
  if (skip) {
    int any = 0;
    for (int i = 0; i < n_skip; ++i) any |= skip[i];
    if (any) return;
  }
  const int t = find_tensor(g, (int)blockIdx.x);
  const int64_t n = g.n[t];
  const int64_t e0 = (int64_t)((int)blockIdx.x - g.chunk0[t]) * CHUNK;
  const int64_t e1 = e0 + CHUNK < n ? e0 + CHUNK : n;
  const float coef = clip ? clip[1] : 1.0f;
  const float step = g.lr[t] / g.bc1[t];              // lr / (1 - beta1^t)
  const float inv_bc2s = 1.0f / g.bc2_sqrt[t];        // 1 / sqrt(1 - beta2^t)
  float* P = g.p[t]; const float* G = g.g[t]; float* M = g.m[t]; float* V = g.v[t];
  int64_t s0 = e0;
  if (g.vec_ok[t]) {
    const int64_t v1 = e0 + ((e1 - e0) & ~(int64_t)3);
    for (int64_t i = e0 + 4 * threadIdx.x; i < v1; i += 1024) {
      f32x4 p = *reinterpret_cast<const f32x4*>(P + i), m = *reinterpret_cast<const f32x4*>(M + i),
            v = *reinterpret_cast<const f32x4*>(V + i);
      const f32x4 gr = *reinterpret_cast<const f32x4*>(G + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        adam_elem(pk, mk, vk, gr[k], coef, h, step, inv_bc2s);
        p[k] = pk; m[k] = mk; v[k] = vk;
      }
      *reinterpret_cast<f32x4*>(P + i) = p;
      *reinterpret_cast<f32x4*>(M + i) = m;
      *reinterpret_cast<f32x4*>(V + i) = v;
    }
    s0 = v1;
  }
  for (int64_t i = s0 + threadIdx.x; i < e1; i += 256) {
    float p = P[i], m = M[i], v = V[i];
    adam_elem(p, m, v, G[i], coef, h, step, inv_bc2s);
    P[i] = p; M[i] = m; V[i] = v;
  }
}

int check_group(const gvd_opt_group* g, bool need_state) {
  if (!g || g->count <= 0 || g->count > GVD_OPT_MAX_TENSORS || g->chunk0[0] != 0) return GVD_EINVAL;
  for (int t = 0; t < g->count; ++t) {
    if (!g->g[t] || g->n[t] <= 0) return GVD_EINVAL;
    if (need_state && (!g->p[t] || !g->m[t] || !g->v[t])) return GVD_EINVAL;
    const int64_t nchunks = (g->n[t] + CHUNK - 1) / CHUNK;
    if ((int64_t)g->chunk0[t + 1] - g->chunk0[t] != nchunks) return GVD_EINVAL;
    if (g->vec_ok[t] && (!gvd_aligned16(g->g[t]) || (need_state && (!gvd_aligned16(g->p[t]) || !gvd_aligned16(g->m[t]) ||
                                                                    !gvd_aligned16(g->v[t])))))
      return GVD_EINVAL;
  }
  return 0;
}

}  // namespace

extern "C" int gvd_opt_chunk(void) { return CHUNK; }

extern "C" int gvd_sumsq_partials(const gvd_opt_group* g, float* partials, gvd_stream_t stream) {
  if (check_group(g, false) || !partials || g->part0 < 0) return GVD_EINVAL;
  hipLaunchKernelGGL(sumsq_partials_kernel, dim3((unsigned)g->chunk0[g->count]), dim3(256), 0, gvd_s(stream), *g, partials);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_clip_coef(const float* partials, int n, float max_norm, float* out, gvd_stream_t stream) {
  if (!partials || n <= 0 || !out || !(max_norm > 0.f)) return GVD_EINVAL;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, gvd_s(stream), partials, n, max_norm, out);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_adam_step(const gvd_opt_group* g, const float* clip, const int* skip, int n_skip, float beta1,
                             float beta2, float eps, float weight_decay, gvd_stream_t stream) {
  if (check_group(g, true) || (skip && n_skip <= 0) || n_skip > 64) return GVD_EINVAL;
  for (int t = 0; t < g->count; ++t)
    if (!(g->bc1[t] > 0.f) || !(g->bc2_sqrt[t] > 0.f)) return GVD_EINVAL;
  const AdamHyper h = {beta1, beta2, eps, weight_decay};
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)g->chunk0[g->count]), dim3(256), 0, gvd_s(stream), *g, clip, skip,
                     skip ? n_skip : 0, h);
  GVD_CHECK_LAUNCH();
  return 0;
}
