// Running top-2 of (value, index) pairs under the strict order "larger value first, ties -> lower index" (torch CPU
// max/topk behaviour on distinct data); shared by vocab.hip and decode_persistent.hip.
#pragma once
#include "gvd_common.h"

struct Top2 { float v1; int i1; float v2; int i2; };

__device__ __forceinline__ bool better(float va, int ia, float vb, int ib) {
  return (va > vb) || (va == vb && ia < ib);
}

__device__ __forceinline__ void top2_insert(Top2& t, float v, int i) {
  if (better(v, i, t.v1, t.i1)) { t.v2 = t.v1; t.i2 = t.i1; t.v1 = v; t.i1 = i; }
  else if (better(v, i, t.v2, t.i2)) { t.v2 = v; t.i2 = i; }
}

__device__ __forceinline__ Top2 top2_merge(Top2 a, const Top2& b) {
  top2_insert(a, b.v1, b.i1);
  top2_insert(a, b.v2, b.i2);
  return a;
}

__device__ __forceinline__ Top2 top2_wave(Top2 t) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Top2 o;
    o.v1 = __shfl_xor(t.v1, off, GVD_WAVE); o.i1 = __shfl_xor(t.i1, off, GVD_WAVE);
    o.v2 = __shfl_xor(t.v2, off, GVD_WAVE); o.i2 = __shfl_xor(t.i2, off, GVD_WAVE);
    t = top2_merge(t, o);
  }
  return t;
}

// Token rule of model.py:590-594 on a merged top-2: the best word unless it is UNK, then the runner-up.  A row of
// all-NaN logits (diverged weights) leaves the sentinel index 0x7fffffff in place; clamp it to END (0) so the
// next-token embedding gather can never leave the table.
__device__ __forceinline__ int top2_token(const Top2& g, int unk, int V, bool* keep_out) {
  const bool keep = g.i1 != unk;
  *keep_out = keep;
  const int it = keep ? g.i1 : g.i2;
  return (unsigned)it < (unsigned)V ? it : 0;
}
