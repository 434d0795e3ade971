// Small-batch (M <= 16) path of the dense projections: weight-streaming "skinny" GEMM.
//
// At decode batch sizes the LSTM / query / logit products are weight-bandwidth bound (92 MB of LSTM
// weights per token at any B; L3-resident across steps): MFMA tiles would waste 7/8 of their rows AND leave
// only N/128 workgroups to pull the weights.  Here every wave owns RPW weight rows (for the LSTM: the four
// gate rows i,f,g,o of ONE hidden unit, so the cell epilogue needs no cross-wave exchange), its 64 lanes
// split K in 16-byte slices (fully coalesced 1 KiB row segments, several rows in flight), the few
// activation rows are re-read through L1, and the per-(row, m) partial sums are reduced with xor-shuffles.
// Grid = N / (4 waves * RPW) workgroups (256 for the LSTM) so every CU streams weights.
// Same C-ABI semantics as gemm_nt_kernel (csrc/gemm_f32.hip); called from gvd_gemm_nt_f32 / gvd_lstm_cell_fwd.
#include <stdlib.h>
#include "gvd_common.h"
#include "gemv_f32.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int MB, int RPW, int KS, bool LSTM>
__global__ __launch_bounds__(256 * KS) void gemv_nt_kernel(const GemvParams p) {
  // 4 column groups x KS K-slices of waves: the K-slices put KS x more weight bytes in flight per CU (the
  // products are latency-bound otherwise: a wave only has RPW x 1 KiB outstanding per iteration) and are
  // summed through LDS by the slice-0 waves, which also run the epilogue.
  __shared__ float s_part[(KS > 1 ? KS - 1 : 1) * 4 * RPW * MB];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int cg = wave & 3, kq = wave >> 2;
  // weight rows of this wave
  int wrow[RPW];
  bool rok[RPW];
  int j = 0;
  if (LSTM) {
    j = blockIdx.x * 4 + cg;                   // hidden unit; RPW == 4 gate rows
#pragma unroll
    for (int r = 0; r < RPW; ++r) { rok[r] = j < p.H; wrow[r] = rok[r] ? r * p.H + j : 0; }
  } else {
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int n = (blockIdx.x * 4 + cg) * RPW + r;
      rok[r] = n < p.N;
      wrow[r] = rok[r] ? n : 0;
    }
  }
  float acc[RPW][MB];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[r][m] = 0.f;

  for (int s = 0; s < p.nseg; ++s) {
    const float* A = p.A[s];
    const float* W = p.W[s];
    const int64_t lda = p.lda[s], ldw = p.ldw[s];
    const int K = p.K[s];
    for (int k = (kq * 64 + lane) * 4; k < K; k += 256 * KS) {
      f32x4 w[RPW];
#pragma unroll
      for (int r = 0; r < RPW; ++r) w[r] = *reinterpret_cast<const f32x4*>(W + (int64_t)wrow[r] * ldw + k);
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        if (m < p.M) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(A + (int64_t)m * lda + k);
#pragma unroll
          for (int r = 0; r < RPW; ++r)
            acc[r][m] = fmaf(w[r][3], a[3], fmaf(w[r][2], a[2], fmaf(w[r][1], a[1], fmaf(w[r][0], a[0], acc[r][m]))));
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[r][m] = wave_sum(acc[r][m]);

  if (KS > 1) {
    if (kq > 0 && lane == 0) {
      float* dst = s_part + ((kq - 1) * 4 + cg) * RPW * MB;
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) dst[r * MB + m] = acc[r][m];
    }
    __syncthreads();
    if (kq > 0) return;
#pragma unroll
    for (int q = 0; q < KS - 1; ++q) {
      const float* src = s_part + (q * 4 + cg) * RPW * MB;
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[r][m] += src[r * MB + m];
    }
  }

  if (!LSTM) {
    // lane m writes row m of the RPW columns
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      if (!rok[r]) continue;
      const int n = wrow[r];
      float nb = 0.f;
      if (p.nbias) nb += p.nbias[n];
      if (p.nbias2) nb += p.nbias2[n];
      float v = 0.f;
#pragma unroll
      for (int m = 0; m < MB; ++m) if (lane == m) v = acc[r][m];
      if (lane < p.M) {
        v += nb;
        if (p.rowbias) v += p.rowbias[(int64_t)lane * p.rowbias_ld + n];
        if (p.act == 1) v = fmaxf(v, 0.f);
        p.C[(int64_t)lane * p.ldc + n] = v;
      }
    }
  } else {
    if (j >= p.H) return;
    float g[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = 0.f;
#pragma unroll
      for (int m = 0; m < MB; ++m) if (lane == m) v = acc[r][m];
      const int row = r * p.H + j;
      if (p.nbias) v += p.nbias[row];
      if (p.nbias2) v += p.nbias2[row];
      if (p.rowbias && lane < p.M) v += p.rowbias[(int64_t)lane * p.rowbias_ld + row];
      g[r] = v;
    }
    if (lane < p.M) {
      const int m = lane;
      const float gi = sigmoid_f(g[0]), gf = sigmoid_f(g[1]), gg = tanhf(g[2]), go = sigmoid_f(g[3]);
      const float c = gf * p.c_prev[(int64_t)m * p.ldcp + j] + gi * gg;
      p.c_out[(int64_t)m * p.ldco + j] = c;
      p.h_out[(int64_t)m * p.ldh + j] = go * tanhf(c);
      if (p.gates_out) {
        float* go_ = p.gates_out + (int64_t)m * p.ldg;
        go_[j] = gi; go_[p.H + j] = gf; go_[2 * p.H + j] = gg; go_[3 * p.H + j] = go;
      }
    }
  }
}

// K-slices per workgroup: 4 where the register budget allows it (1 and 2 were measured slower: 19.2 -> 16.3 us per LSTM launch)
constexpr int gemv_ks() { return 4; }

template <int MB, int RPW, int KS, bool LSTM>
int launch_ks(const GemvParams& p, hipStream_t st) {
  const int cols_per_wg = 4 * (LSTM ? 1 : RPW);
  const int total = LSTM ? p.H : p.N;
  dim3 grid((unsigned)((total + cols_per_wg - 1) / cols_per_wg));
  hipLaunchKernelGGL((gemv_nt_kernel<MB, RPW, KS, LSTM>), grid, dim3(256 * KS), 0, st, p);
  GVD_CHECK_LAUNCH();
  return 0;
}

template <int MB, int RPW, bool LSTM>
int launch(const GemvParams& p, hipStream_t st) {
  // 4 slices = 1024 threads = a 128-VGPR budget: only the variants with <= 16 accumulators fit without spilling
  constexpr int KS_MAX = (RPW * MB <= 16) ? 4 : 2;
  const int ks = gemv_ks() < KS_MAX ? gemv_ks() : KS_MAX;
  switch (ks) {
    case 1: return launch_ks<MB, RPW, 1, LSTM>(p, st);
    case 2: return launch_ks<MB, RPW, 2, LSTM>(p, st);
    default: return launch_ks<MB, RPW, KS_MAX, LSTM>(p, st);
  }
}

template <int RPW>
int dispatch_plain(const GemvParams& p, hipStream_t st) {
  if (p.M <= 4) return launch<4, RPW, false>(p, st);
  if (p.M <= 8) return launch<8, RPW, false>(p, st);
  return launch<16, RPW, false>(p, st);
}

template <bool LSTM>
int dispatch(const GemvParams& p, hipStream_t st) {
  if (LSTM) {
    if (p.M <= 4) return launch<4, 4, true>(p, st);
    if (p.M <= 8) return launch<8, 4, true>(p, st);
    return launch<16, 4, true>(p, st);
  }
  // rows per wave: keep >= 256 workgroups (one per CU) whenever N allows it
  if (p.N >= 4096) return dispatch_plain<4>(p, st);
  if (p.N >= 2048) return dispatch_plain<2>(p, st);
  return dispatch_plain<1>(p, st);
}

}  // namespace

int gvd_gemv_plain(const GemvParams& p, hipStream_t st) { return dispatch<false>(p, st); }
int gvd_gemv_lstm(const GemvParams& p, hipStream_t st) { return dispatch<true>(p, st); }
