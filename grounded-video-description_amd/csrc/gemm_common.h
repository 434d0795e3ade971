// Shared pieces of the fp32-MFMA GEMM kernels (gemm_f32.hip: general tiles + LSTM epilogue; gemm_pipe.hip: the
// software-pipelined 128x128 kernel for the large per-segment projections).
#pragma once
#include "gvd_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GVD_GEMM_BK_MIN = 32;   // K granularity every segment must be a multiple of

struct KParams {
  const float* A[3]; int64_t lda[3]; int64_t abs_[3];
  const float* W[3]; int64_t ldw[3]; int64_t wbs[3];
  int K[3]; int nseg;
  const float* nbias; const float* nbias2;
  const float* mbias; int64_t mbias_bs;
  const float* rowbias; int64_t rowbias_ld; int64_t rowbias_bs;
  const uint8_t* mask; int64_t mask_ldm; int64_t mask_bs;
  float* C; int64_t ldc; int64_t cbs;
  int M, N, act;
  const int* m_dev;      // when set: the row count is read on the device (<= M; tiles past it exit)
  const int* a_rmap;     // when set (pipelined kernel, one segment): output row m reads row a_rmap[m] of A
  int a_t, w_t;          // operand is K-strided: A given as [K, M] (lda >= M), W as [K, N] (ldw >= N)
  int ktail;             // pipelined kernel, direct-to-LDS plain products, one segment: K % 32 == 16 (see gemm_pipe.hip)
  int binner;            // two-level batch: b = outer * binner + inner (0 / 1 = flat)
  int64_t abs2, wbs2, cbs2;   // inner strides of A / W (every segment) / C
  // LSTM epilogue
  const float* c_prev; int64_t ldcp;
  float* h_out; int64_t ldh;
  float* c_out; int64_t ldco;
  float* gates_out; int64_t ldg;
  int H;
  int ntn, ntm;
};


// Base offset of batch entry bz: flat (bz * s1) or two-level (outer * s1 + inner * s2).
__device__ __forceinline__ int64_t gvd_boff(const KParams& p, int bz, int64_t s1, int64_t s2) {
  if (p.binner > 1) {
    const int bo = bz / p.binner;
    return (int64_t)bo * s1 + (int64_t)(bz - bo * p.binner) * s2;
  }
  return (int64_t)bz * s1;
}

// Plain epilogue of a wave's TM x TN grid of 32x32 accumulator tiles (bias / per-row bias / 2-D bias / ReLU / masked
// fill), straight from the MFMA layout: lane (r = l&31, half = l>>5) holds column r, rows (e&3) + 8(e>>2) + 4 half.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_plain(const KParams& p, int M, const f32x16 (&acc)[TM][TN], int bz, int mw0,
                                                    int nw0, int r, int half) {
  float* Cb = p.C + gvd_boff(p, bz, p.cbs, p.cbs2);
  const float* rb = p.rowbias ? p.rowbias + (int64_t)bz * p.rowbias_bs : nullptr;
  const float* mb = p.mbias ? p.mbias + (int64_t)bz * p.mbias_bs : nullptr;
  const uint8_t* mk = p.mask ? p.mask + (int64_t)bz * p.mask_bs : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int gn = nw0 + j * 32 + r;
      if (gn >= p.N) continue;
      float nb = 0.f;
      if (p.nbias) nb += p.nbias[gn];
      if (p.nbias2) nb += p.nbias2[gn];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
        const int gm = mw0 + i * 32 + row;
        if (gm < M) {
          float v = acc[i][j][e] + nb;
          if (mb) v += mb[gm];
          if (rb) v += rb[(int64_t)gm * p.rowbias_ld + gn];
          if (p.act == 1) v = fmaxf(v, 0.f);
          if (mk && mk[(int64_t)gm * p.mask_ldm + gn]) v = GVD_MIN_VALUE;
          Cb[(int64_t)gm * p.ldc + gn] = v;
        }
      }
    }
}

// gemm_pipe.hip
int gvd_gemm_pipe_launch(KParams& p, int batch, hipStream_t st);
bool gvd_gemm_pipe_takes_ktail();      // built with the direct-to-LDS plain path (the only one with the K tail of 16)
// gemm_small.hip: pipelined 64 x 64 kernel for the token-loop products (plain / LSTM-cell epilogue)
bool gvd_gemm_small_ok(const KParams& p, int batch);
int gvd_gemm_small_launch(KParams& p, bool lstm, hipStream_t st);
// gemm_ks.hip: K-split 32 x 32 tiles for the LSTM cells of batches of 17 .. 64 rows
bool gvd_gemm_ks_ok(const KParams& p, int batch);
int gvd_gemm_ks_lstm_launch(KParams& p, hipStream_t st);
// gemm_n192.hip: K-strided-W products with 129..192 output columns (backward of the training attention core)
bool gvd_gemm_n192_ok(const KParams& p);
int gvd_gemm_n192_launch(KParams& p, int batch, hipStream_t st);
