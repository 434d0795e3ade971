// Internal interface between gemm_f32.hip (C-ABI entry points + dispatch) and gemv_f32.hip (M <= 16 path).
#pragma once
#include "gvd_common.h"

struct GemvParams {
  const float* A[3]; int64_t lda[3];
  const float* W[3]; int64_t ldw[3];
  int K[3]; int nseg;
  const float* nbias; const float* nbias2;
  const float* rowbias; int64_t rowbias_ld;
  float* C; int64_t ldc;
  int M, N, act;
  // LSTM epilogue
  const float* c_prev; int64_t ldcp;
  float* h_out; int64_t ldh;
  float* c_out; int64_t ldco;
  float* gates_out; int64_t ldg;
  int H;
};

int gvd_gemv_plain(const GemvParams& p, hipStream_t st);
int gvd_gemv_lstm(const GemvParams& p, hipStream_t st);
