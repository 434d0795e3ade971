// Shared device/host helpers for libgvd_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gvd_hip.h"

#define GVD_WAVE 64

#define GVD_CHECK_LAUNCH()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

static inline hipStream_t gvd_s(gvd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool gvd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, GVD_WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, GVD_WAVE));
  return v;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// XCD-aware bijective remap of a linear workgroup id (cdna_hip_programming.md T1): hardware round-robins
// consecutive ids over the 8 XCDs; after the remap each XCD walks a contiguous chunk of logical ids so
// neighbouring tiles (which share an operand panel) hit the same private L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  unsigned q = nwg / nx, r = nwg % nx;
  unsigned xcd = bid % nx, idx = bid / nx;
  unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// event-pair recorder (prof.hip); no-ops when p == nullptr
void gvd_prof_begin(gvd_prof* p, hipStream_t st);
void gvd_prof_end(gvd_prof* p, hipStream_t st);
