// Philox4x32-10 counter RNG of the training path's dropout sites (enc_attn_train.hip, train_fused.hip, rowwise.hip):
// one 128-bit block per 4 consecutive elements of a contiguous tensor, counter = flat element index / 4, key = a
// host-drawn 64-bit seed per site and step.  A site's forward and backward kernels regenerate the same keep mask from
// (seed, counter): no mask tensor is written or read.
#pragma once
#include <stdint.h>

struct GvdU4 { uint32_t v[4]; };

__device__ __forceinline__ GvdU4 gvd_philox4x32_10(uint64_t ctr, uint64_t seed) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
    const uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return GvdU4{{c0, c1, c2, c3}};
}

// keep threshold of a drop probability p: element kept iff its 32-bit draw >= thresh
static inline uint32_t gvd_drop_thresh(float p_drop) {
  const double t = (double)p_drop * 4294967296.0;
  return (uint32_t)(t > 4294967295.0 ? 4294967295.0 : t);
}
