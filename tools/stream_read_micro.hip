// Micro-benchmark: read-only HBM streaming rate on this GPU (the ceiling the attention kernel's roofline fraction
// should be read against).  Each workgroup sums a contiguous slab with 16-byte loads, UNROLL loads in flight per lane.
// hipcc --offload-arch=gfx950 -O3 stream_read_micro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ x, size_t n_per_wg, float* out) {
  const f32x4* p = x + (size_t)blockIdx.x * n_per_wg + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = 0; i < n_per_wg; i += 256 * UNROLL) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + 256 * u) : p[i + 256 * u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
  }
  const float s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s == 12345.678f) out[blockIdx.x] = s;      // keep the loads alive
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int UNROLL, bool NT>
int run(const f32x4* x, size_t bytes, int nwg, float* out) {
  const size_t n_per_wg = bytes / 16 / nwg / (256 * UNROLL) * (256 * UNROLL);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((read_kernel<UNROLL, NT>), dim3(nwg), dim3(256), 0, 0, x, n_per_wg, out);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("read %.2f GB  %5d workgroups  %d x16B in flight/lane %s: %.3f ms  %.0f GB/s\n", n_per_wg * nwg * 16 / 1e9, nwg,
         UNROLL, NT ? "nontemporal" : "           ", best, n_per_wg * (double)nwg * 16 / best / 1e6);
  return 0;
}

int main() {
  const size_t bytes = (size_t)1600 << 20;       // 1.6 GB: the size of one attention launch at B = 256
  f32x4* x; float* out;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(x, 0, bytes));
  for (int nwg : {2048, 8192, 32768}) {
    if (run<4, false>(x, bytes, nwg, out)) return 1;
    if (run<8, false>(x, bytes, nwg, out)) return 1;
    if (run<8, true>(x, bytes, nwg, out)) return 1;
  }
  return 0;
}
