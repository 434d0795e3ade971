// Keep mask of the dropout on the encoder's attention weights (transformer.py:95,104: nn.Dropout(0.2) on softmax(QK^T / sqrt d))
// for the TRAINING attention core (flash_attn_pad.hip forward, enc_attn_bwd.hip backward).
//
// The forward is a flash-style kernel (no [B, heads, R, R] map exists), the backward recomputes the probabilities tile by
// tile in another register layout - so the mask must be a pure function of (seed, map row, key) that either kernel can
// evaluate for whatever elements its lanes hold: a counter-based hash, one 32-bit draw per element.
//     add, flip = two hashes of row_id               row_id = (sample * heads + head) * Rp + query   (once per query)
//     draw      = mix32(add + key) ^ flip            keep iff draw >= thresh = p * 2^32
// mix32 is the two-multiply xorshift finaliser ("lowbias32", bias 0.17 bits over all 2^32 inputs): 9 VALU instructions per
// element, two of them quarter-rate - against 12.5 per element for Philox4x32-10, which also hands out its draws in blocks
// of four CONSECUTIVE elements (a layout only one of the two kernels could use without wasting three of four draws).
// Statistics (keep rate, row / column independence) are checked on the device by tests/test_gpu_kernels.py.
#pragma once
#include <stdint.h>

__host__ __device__ __forceinline__ uint32_t gvd_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// Row key = 64 bits: `add` enters the element hash by addition, `flip` is XORed onto the draw.  With the additive word alone
// two map rows whose keys differ by d < Rp carry the SAME mask shifted by d keys (at 393 k rows per layer tens of thousands of
// such pairs exist: a birthday bound on 32 bits); the second word, an independent hash of the row, turns the shared draw m into
// m ^ flip_A and m ^ flip_B, which are independent uniform values over the population of row pairs.  One full-rate v_xor per
// element on top of the 9 instructions of the hash.
struct gvd_encdrop_key { uint32_t add, flip; };

// per query row, once
__host__ __device__ __forceinline__ gvd_encdrop_key gvd_encdrop_row(uint32_t row_id, uint32_t seed_lo, uint32_t seed_hi) {
  gvd_encdrop_key k;
  k.add = gvd_mix32(row_id ^ seed_hi) ^ seed_lo;
  k.flip = gvd_mix32((row_id + 0x9e3779b9u) ^ seed_lo) ^ seed_hi;
  return k;
}
// per element
__host__ __device__ __forceinline__ bool gvd_encdrop_keep(gvd_encdrop_key rowkey, uint32_t key, uint32_t thresh) {
  return (gvd_mix32(rowkey.add + key) ^ rowkey.flip) >= thresh;
}
