// Probe of the gfx950 lane-swap instructions (semantics used by flash_attn_pad.hip: rows_max / rows_sum).
//   hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  u2 s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap r[0]", "permlane16_swap r[1]", "permlane32_swap r[0]", "permlane32_swap r[1]"};
  for (int j = 0; j < 4; ++j) {
    printf("%s:", names[j]);
    for (int i = 0; i < 64; i += 4) printf(" %u", h[64 * j + i]);
    printf("\n");
  }
  return 0;
}
