// Masked-proposal compaction for the per-segment preamble (inference).
//
// The loader zeroes every masked proposal (score <= prop_thresh, or padding): its fc6 features and its box are 0
// (dataloader_anet.py:343-344), and its class-similarity row is the all-masked constant (model.py:336-340).  So ALL masked
// proposals of a segment enter the preamble with the SAME row, and every per-row stage (fc7, class logits, layer
// norms, pool_embed, the encoder's projections / feed-forward / layer norms, ctx2pool) maps equal rows to equal rows
// - bit for bit, each output row depends on its own input row only.  The one place rows meet is the encoder's
// self-attention (transformer.py:90-123, no padding mask): there the n_m masked rows are n_m identical keys, i.e. ONE
// key whose softmax weight is multiplied by n_m (its score gets + log2 n_m in the log2 domain), and n_m identical
// queries with one common answer.  The preamble therefore runs on the COMPACT row set of each segment
//     [ its n_v valid proposals in order | one representative masked row ]
// (typically 80 % of the rows: 20 % fewer GEMM flops, ~36 % fewer attention flops) and the dense [B,R,.] tensors the
// token loop streams are restored at the end by a row gather.  Results equal the dense path up to the fp32 summation
// order inside the attention softmax.  Precondition (the loader contract above): masked rows of ppls_feat / ppls are
// zero; gvd_check_masked_rows_zero raises a device flag otherwise.
//
//   gvd_compact_index        per-segment offsets / counts, dense->compact and compact->dense row maps, key weight
//   gvd_gather_rows_f32      out[i,:] = in[idx[i],:] (compaction of the inputs, expansion of the outputs)
//   gvd_check_masked_rows_zero
#include "gvd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void compact_index_kernel(const uint8_t* __restrict__ mask, int64_t ld_mask, int R,
                                                            int* __restrict__ off, int* __restrict__ nvalid,
                                                            int* __restrict__ src_row, int* __restrict__ cidx,
                                                            float* __restrict__ rep_w, uint8_t* __restrict__ cmask,
                                                            int B) {
  __shared__ int s_cnt[4];
  __shared__ int s_base;
  __shared__ int s_first_masked;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // rows of the segments before this one: sum over b' < b of (valid rows + 1 representative)
  int before = 0;
  for (int64_t i = tid; i < (int64_t)b * R; i += 256) {
    const int bb = (int)(i / R), r = (int)(i % R);
    before += mask[bb * ld_mask + r] == 0;
  }
  before = (int)wave_sum((float)before);      // counts < 2^24: exact in fp32
  if (lane == 0) s_cnt[wave] = before;
  if (tid == 0) s_first_masked = R;
  __syncthreads();
  const int base = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3] + b;
  __syncthreads();
  const uint8_t* m = mask + (int64_t)b * ld_mask;
  int run = 0;                                 // valid rows of this segment seen so far
  for (int r0 = 0; r0 < R; r0 += 256) {
    const int r = r0 + tid;
    const bool valid = r < R && m[r] == 0;
    const unsigned long long bal = __ballot(valid);
    const int in_wave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += s_cnt[w];
    const int tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (r < R) {
      if (valid) {
        const int c = base + run + wbase + in_wave;
        cidx[(int64_t)b * R + r] = c;
        src_row[c] = b * R + r;
        cmask[c] = 0;
      } else {
        atomicMin(&s_first_masked, r);
      }
    }
    run += tot;
    __syncthreads();
  }
  const int rep = base + run;                   // the representative masked row
  const int nm = R - run;
  for (int r = tid; r < R; r += 256)
    if (m[r] != 0) cidx[(int64_t)b * R + r] = rep;
  if (tid == 0) {
    off[b] = base;
    if (b == B - 1) off[B] = rep + 1;
    nvalid[b] = run;
    src_row[rep] = b * R + (nm > 0 ? s_first_masked : 0);
    cmask[rep] = 1;
    rep_w[b] = nm > 0 ? log2f((float)nm) : -INFINITY;
  }
}

// one wave per output row
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in, int64_t in_ld,
                                                          const int* __restrict__ idx, float* __restrict__ out,
                                                          int64_t out_ld, int D, int64_t n_rows,
                                                          const int* __restrict__ n_dev) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n = n_dev ? (int64_t)*n_dev : n_rows;
  if (row >= n) return;
  const float* src = in + (int64_t)idx[row] * in_ld;
  float* dst = out + row * out_ld;
  if (((D | in_ld | out_ld) & 3) == 0) {
    for (int c = 4 * lane; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c);
  } else {
    for (int c = lane; c < D; c += 64) dst[c] = src[c];
  }
}

// flag[0] |= 1 when a masked row of x [rows_total, D] holds a non-zero
__global__ __launch_bounds__(256) void check_masked_zero_kernel(const float* __restrict__ x, int D,
                                                                const uint8_t* __restrict__ mask, int64_t ld_mask, int R,
                                                                int64_t rows, int* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (mask[(row / R) * ld_mask + (row % R)] == 0) return;
  const float* p = x + row * D;
  bool bad = false;
  for (int c = lane; c < D; c += 64) bad |= p[c] != 0.f;
  if (__any(bad) && lane == 0) atomicOr(flag, 1);
}

}  // namespace

extern "C" int gvd_compact_index(const uint8_t* mask, int64_t ld_mask, int B, int R, int* off, int* nvalid, int* src_row,
                                 int* cidx, float* rep_w, uint8_t* cmask, gvd_stream_t stream) {
  if (!mask || !off || !nvalid || !src_row || !cidx || !rep_w || !cmask || B <= 0 || R <= 0 ||
      (int64_t)B * (R + 1) >= (1 << 24))
    return GVD_EINVAL;
  hipLaunchKernelGGL(compact_index_kernel, dim3((unsigned)B), dim3(256), 0, gvd_s(stream), mask, ld_mask, R, off, nvalid,
                     src_row, cidx, rep_w, cmask, B);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_gather_rows_f32(const float* in, int64_t in_ld, const int* idx, float* out, int64_t out_ld, int D,
                                   int64_t n_rows, const int* n_rows_dev, gvd_stream_t stream) {
  if (!in || !idx || !out || D <= 0 || n_rows <= 0) return GVD_EINVAL;
  if (((D | in_ld | out_ld) & 3) == 0 && (!gvd_aligned16(in) || !gvd_aligned16(out))) return GVD_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, gvd_s(stream), in, in_ld, idx,
                     out, out_ld, D, n_rows, n_rows_dev);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_check_masked_rows_zero(const float* x, int D, const uint8_t* mask, int64_t ld_mask, int B, int R,
                                          int* flag, gvd_stream_t stream) {
  if (!x || !mask || !flag || D <= 0 || B <= 0 || R <= 0) return GVD_EINVAL;
  const int64_t rows = (int64_t)B * R;
  hipLaunchKernelGGL(check_masked_zero_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, gvd_s(stream), x, D, mask,
                     ld_mask, R, rows, flag);
  GVD_CHECK_LAUNCH();
  return 0;
}
