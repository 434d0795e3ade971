// Micro-benchmark: cost of one grid-wide barrier (256 workgroups, cooperative launch) on gfx950, two forms:
//   mode 0  agent-scope release fence (L2 write-back) + counter + acquire fence (L1/L2 invalidate)  [csrc/gru.hip]
//   mode 1  no fences: exchanged data is written/read with agent-scope relaxed atomics (sc1 accesses that bypass the
//           non-coherent cache levels), the barrier only drains the stores and bumps/polls the counter
// Every round each workgroup publishes one value per thread-quad and, after the barrier, checks the value of another
// workgroup (so a broken barrier shows up as errors).   hipcc --offload-arch=gfx950 -O3 barrier_micro.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr unsigned SPIN_LIMIT = 4000000u;

template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* sync, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) { __hip_atomic_store(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// mode 2: one arrival counter, the LAST arriver stores a release word that the others poll (arrivals and polls
//         do not fight over one address)
// mode 3: two levels, 16 groups of 16 workgroups (group = id & 15): group counter -> top counter -> per-group
//         release words written by the last arriver (16 pollers per word)
// mode 4: same with 8 groups of 32 (group = id & 7 = the XCD the workgroup runs on)
// layout of `sync` (unsigned, 32 words = 128 B apart): [0] top counter, [32] error flag, [64 + 32 g] group counter,
// [64 + 32*16 + 32 g] group release word
template <int MODE>
__device__ __forceinline__ void grid_barrier_h(unsigned* sync, unsigned round, unsigned nwg) {
  constexpr unsigned NG = MODE == 4 ? 8u : 16u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = MODE == 2 ? 0u : (blockIdx.x & (NG - 1));
    unsigned* rel = sync + 64 + 32 * 16 + 32 * g;
    bool last;
    if (MODE == 2) {
      last = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (round + 1) * nwg - 1;
      if (last) __hip_atomic_store(rel, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned per = nwg / NG;
      last = __hip_atomic_fetch_add(sync + 64 + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
             (round + 1) * per - 1;
      if (last) {
        last = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (round + 1) * NG - 1;
        if (last)
          for (unsigned k = 0; k < NG; ++k)
            __hip_atomic_store(sync + 64 + 32 * 16 + 32 * k, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    unsigned spins = 0;
    while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) { __hip_atomic_store(sync + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}

// mode 5: 16 x 16 tree arrivals, every workgroup polls the TOP counter itself (no release store hop)
__device__ __forceinline__ void grid_barrier_t5(unsigned* sync, unsigned round, unsigned nwg) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 15u, per = nwg / 16u;
    if (__hip_atomic_fetch_add(sync + 64 + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (round + 1) * per - 1)
      __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (round + 1) * 16u) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SPIN_LIMIT) { __hip_atomic_store(sync + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}

template <int MODE>
__global__ void bench_kernel(unsigned* sync, float* buf, int rounds, int per_wg, unsigned* errors, float* dirty,
                             int dirty_per_thread) {
  const unsigned nwg = gridDim.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    float* cur = buf + (size_t)(r & 1) * nwg * per_wg;
    if ((int)threadIdx.x < per_wg) {
      const float v = (float)(r * 1000 + (int)blockIdx.x);
      if (MODE == 0) cur[blockIdx.x * per_wg + threadIdx.x] = v;
      else __hip_atomic_store(cur + blockIdx.x * per_wg + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // optional plain (cached) stores that make this XCD's L2 dirty: what a release fence has to write back
    for (int i = 0; i < dirty_per_thread; ++i)
      dirty[((size_t)blockIdx.x * blockDim.x + threadIdx.x) * dirty_per_thread + i] = (float)r;
    if (MODE <= 1) grid_barrier<MODE>(sync, (unsigned)(r + 1) * nwg);
    else if (MODE == 5) grid_barrier_t5(sync, (unsigned)r, nwg);
    else grid_barrier_h<MODE>(sync, (unsigned)r, nwg);
    const unsigned other = (blockIdx.x * 37u + 11u + (unsigned)r) % nwg;
    if ((int)threadIdx.x < per_wg) {
      float got;
      if (MODE == 0) got = cur[other * per_wg + threadIdx.x];
      else got = __hip_atomic_load(cur + other * per_wg + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (got != (float)(r * 1000 + (int)other)) ++bad;
    }
  }
  if (bad) atomicAdd(errors, bad);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
int run(int nwg, int threads, int rounds, int per_wg, int dirty_per_thread) {
  unsigned *sync, *errors; float *buf, *dirty;
  CK(hipMalloc(&sync, 4096 * 2)); CK(hipMalloc(&errors, 4));
  CK(hipMalloc(&buf, sizeof(float) * 2 * nwg * per_wg));
  CK(hipMalloc(&dirty, sizeof(float) * (size_t)nwg * threads * (dirty_per_thread > 0 ? dirty_per_thread : 1)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(sync, 0, 4096 * 2)); CK(hipMemset(errors, 0, 4));
    void* args[] = {&sync, &buf, &rounds, &per_wg, &errors, &dirty, &dirty_per_thread};
    CK(hipEventRecord(e0, 0));
    CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(bench_kernel<MODE>), dim3(nwg), dim3(threads), args, 0, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned h_err = 0, h_sync[64];
  CK(hipMemcpy(&h_err, errors, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h_sync, sync, 256, hipMemcpyDeviceToHost));
  printf("mode %d  wg %d x %d thr  publish %d floats/wg  dirty %d floats/thr : %.3f us / round   (errors %u, timeout flag %u)\n",
         MODE, nwg, threads, per_wg, dirty_per_thread, 1e3f * best / rounds, h_err, MODE <= 1 ? h_sync[1] : h_sync[32]);
  hipFree(sync); hipFree(errors); hipFree(buf); hipFree(dirty);
  return 0;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  for (int threads : {256, 512}) {
    for (int per_wg : {16, 256}) {
      if (run<1>(256, threads, rounds, per_wg, 0)) return 1;
      if (run<3>(256, threads, rounds, per_wg, 0)) return 1;
      if (run<5>(256, threads, rounds, per_wg, 0)) return 1;
    }
  }
  if (run<3>(128, 256, rounds, 16, 0)) return 1;
  if (run<5>(128, 256, rounds, 16, 0)) return 1;
  return 0;
}
