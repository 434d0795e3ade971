// fp32 "NT" GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fma chains, 157 TF peak)
// with fused epilogues:  plain (bias / per-row bias / 2-D bias / ReLU / masked fill)  and  LSTM cell.
//
//   C[b][M,N] = epi( sum_s A_s[b][M,K_s] * W_s[b][N,K_s]^T )
//
// Replaces the implicit cuBLAS GEMM + elementwise ATen chains of the reference hot path
// (SURVEY.md §2.3 P2,P4,P8,S1,S4,S6,T4; reference call sites cited in include/gvd_hip.h).
//
// Design (MI355X): 256 threads = 4 waves (one per SIMD) per workgroup; each wave owns a grid of 32x32
// MFMA tiles.  Operand tiles [rows][32 k] are staged global -> registers -> LDS (float4, 128-B row
// segments: fully coalesced), double-buffered so one barrier per k-tile suffices; LDS rows are padded to
// 36 floats, which makes the per-lane ds_read_b128 fragment reads bank-conflict free.  Each lane reads 4
// consecutive k of its row once (16 B) and feeds them to 4 successive MFMAs: lane-half h supplies
// k = 8*kb + 4*h + t at MFMA step t, so the two k-slots of a 32x32x2 MFMA carry k and k+4 — a
// permutation of the summation order only.  The f32 MFMA issues every 64 cycles per SIMD, so LDS and
// global traffic hide completely behind the matrix pipe.  Workgroup ids are remapped XCD-aware so the
// tiles that share an A panel land on one XCD's L2.
#include "gemm_common.h"
#include "gemv_f32.h"
#include <stdlib.h>

namespace {

constexpr int BK_MIN = GVD_GEMM_BK_MIN;

template <int BM, int BN, int WGM, int WGN, bool LSTM, int NBUF = 2, int BK = 32>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const KParams p) {
  constexpr int LDK = BK + 4;                     // padded LDS row (floats): conflict-free ds_read_b128
  constexpr int F4R = BK / 4;                     // float4 pieces per tile row
  constexpr int WTM = BM / WGM, WTN = BN / WGN;   // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;     // MFMA tiles per wave
  constexpr int NA = BM * F4R / 256, NW = BN * F4R / 256;   // float4 loads per thread per k-tile
  constexpr int HU = BN / 4;                      // LSTM: hidden units per tile
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(TM >= 1 && TN >= 1, "tile");
  // NBUF = 2: double-buffered operand tiles (one barrier per k-tile); NBUF = 1: single buffer + register prefetch
  // (two barriers per k-tile, half the LDS -> one more workgroup per CU)
  __shared__ __attribute__((aligned(16))) float smem[NBUF * (BM + BN) * LDK];
  float* As = smem;
  float* Ws = smem + NBUF * BM * LDK;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, half = lane >> 5;
  const int wm = wave / WGN, wn = wave % WGN;
  const unsigned nwg = gridDim.x;
  const unsigned lid = xcd_remap(blockIdx.x, nwg);
  const int tn_ = lid % p.ntn, tm_ = lid / p.ntn;
  const int bz = blockIdx.y;
  const int m0 = tm_ * BM;
  const int n0 = LSTM ? 0 : tn_ * BN;
  const int M = p.m_dev ? min(*p.m_dev, p.M) : p.M;          // device-side row count (compacted preamble)
  if (m0 >= M) return;

  // per-thread global row pointers are recomputed per segment; row validity is segment independent
  int a_row[NA], w_row[NW];
  bool a_ok[NA], w_ok[NW];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    int row = (tid + i * 256) / F4R;
    int gm = m0 + row;
    a_ok[i] = gm < M;
    a_row[i] = a_ok[i] ? gm : 0;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    int nl = (tid + i * 256) / F4R;
    if (LSTM) {
      w_row[i] = (nl / HU) * p.H + tn_ * HU + (nl % HU);
      w_ok[i] = true;
    } else {
      int gn = n0 + nl;
      w_ok[i] = gn < p.N;
      w_row[i] = w_ok[i] ? gn : 0;
    }
  }
  const int kq4 = (tid % F4R) * 4;

  int nkt = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (s < p.nseg) nkt += p.K[s] / BK;

  f32x4 ra[NA], rw[NW];
  auto load_tile = [&](int kt) {
    int s = 0, k0 = kt * BK;
    while (s + 1 < p.nseg && k0 >= p.K[s]) { k0 -= p.K[s]; ++s; }
    const float* Ab = p.A[s] + gvd_boff(p, bz, p.abs_[s], p.abs2) + k0 + kq4;
    const float* Wb = p.W[s] + gvd_boff(p, bz, p.wbs[s], p.wbs2) + k0 + kq4;
    const int64_t lda = p.lda[s], ldw = p.ldw[s];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a_ok[i]) v = *reinterpret_cast<const f32x4*>(Ab + (int64_t)a_row[i] * lda);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (w_ok[i]) v = *reinterpret_cast<const f32x4*>(Wb + (int64_t)w_row[i] * ldw);
      rw[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int row = (tid + i * 256) / F4R;
      *reinterpret_cast<f32x4*>(&As[(buf * BM + row) * LDK + kq4]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      int row = (tid + i * 256) / F4R;
      *reinterpret_cast<f32x4*>(&Ws[(buf * BN + row) * LDK + kq4]) = rw[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  load_tile(0);
  store_tile(0);
  __syncthreads();
  int buf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_tile(kt + 1);
    const float* Ab = &As[(buf * BM + wm * WTM + r) * LDK + half * 4];
    const float* Wb = &Ws[(buf * BN + wn * WTN + r) * LDK + half * 4];
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + kb * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Wb + j * 32 * LDK + kb * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    if (NBUF == 2) {
      if (kt + 1 < nkt) store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    } else {
      __syncthreads();                 // every wave finished reading the tile
      if (kt + 1 < nkt) store_tile(0);
      __syncthreads();
    }
  }

  if (!LSTM) {
    gemm_epilogue_plain<TM, TN>(p, M, acc, bz, m0 + wm * WTM, n0 + wn * WTN, r, half);
  } else {
    // gates -> LDS tile G[BM][BN+1] (columns grouped i|f|g|o, HU units each), then the pointwise cell.
    constexpr int LDG = BN + 1;
    static_assert(!LSTM || BM * LDG <= NBUF * (BM + BN) * LDK, "G tile fits");
    float* G = smem;   // all waves passed the loop's final barrier: operand tiles are dead
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int nl = wn * WTN + j * 32 + r;
        const int wrow = (nl / HU) * p.H + tn_ * HU + (nl % HU);
        float nb = 0.f;
        if (p.nbias) nb += p.nbias[wrow];
        if (p.nbias2) nb += p.nbias2[wrow];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
          const int ml = wm * WTM + i * 32 + row;
          const int gm = m0 + ml;
          float v = acc[i][j][e] + nb;
          if (p.rowbias && gm < M) v += p.rowbias[(int64_t)gm * p.rowbias_ld + wrow];
          G[ml * LDG + nl] = v;
        }
      }
    __syncthreads();
    for (int idx = tid; idx < BM * HU; idx += 256) {
      const int ml = idx / HU, jl = idx % HU;
      const int gm = m0 + ml;
      if (gm >= M) continue;
      const int j = tn_ * HU + jl;
      const float gi = sigmoid_f(G[ml * LDG + jl]);
      const float gf = sigmoid_f(G[ml * LDG + HU + jl]);
      const float gg = tanhf(G[ml * LDG + 2 * HU + jl]);
      const float go = sigmoid_f(G[ml * LDG + 3 * HU + jl]);
      const float c = gf * p.c_prev[(int64_t)gm * p.ldcp + j] + gi * gg;
      p.c_out[(int64_t)gm * p.ldco + j] = c;
      p.h_out[(int64_t)gm * p.ldh + j] = go * tanhf(c);
      if (p.gates_out) {
        float* g = p.gates_out + (int64_t)gm * p.ldg;
        g[j] = gi; g[p.H + j] = gf; g[2 * p.H + j] = gg; g[3 * p.H + j] = go;
      }
    }
  }
}

bool seg_ok(const gvd_gemm_seg& s) {
  return s.A && s.W && s.K > 0 && (s.K % BK_MIN) == 0 && gvd_aligned16(s.A) && gvd_aligned16(s.W) &&
         (s.lda % 4) == 0 && (s.ldw % 4) == 0 && (s.a_batch_stride % 4) == 0 && (s.w_batch_stride % 4) == 0;
}

template <int BM, int BN, int WGM, int WGN, bool LSTM, int NBUF = 2, int BK = 32>
int launch(KParams& p, int batch, hipStream_t st) {
  p.ntm = (p.M + BM - 1) / BM;
  p.ntn = LSTM ? p.H / (BN / 4) : (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(p.ntm * p.ntn), (unsigned)batch);
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WGM, WGN, LSTM, NBUF, BK>), grid, dim3(256), 0, st, p);
  GVD_CHECK_LAUNCH();
  return 0;
}

}  // namespace

extern "C" int gvd_gemm_nt_f32(const gvd_gemm_args* a, gvd_stream_t stream) {
  if (!a || a->nseg < 1 || a->nseg > 3 || !a->C || a->M <= 0 || a->N <= 0 || a->batch <= 0) return GVD_EINVAL;
  KParams p = {};
  p.nseg = a->nseg;
  // one plain segment with K % 32 == 16 (K >= 48) is taken by the pipelined kernel's shifted tail tile, and only by it
  const long big_tiles = (long)((a->M + 127) / 128) * ((a->N + 127) / 128) * a->batch;
  const bool ktail = a->nseg == 1 && (a->seg[0].K % BK_MIN) == 16 && a->seg[0].K >= 48 && !a->a_kstrided && !a->w_kstrided &&
                     a->M > 32 && big_tiles >= 256 && gvd_gemm_pipe_takes_ktail();
  for (int s = 0; s < a->nseg; ++s) {
    gvd_gemm_seg sg = a->seg[s];
    if (ktail) sg.K += 16;                      // (passes the 32-multiple check below; the kernel gets the real K)
    if (!seg_ok(sg)) return GVD_EINVAL;
    p.A[s] = a->seg[s].A; p.lda[s] = a->seg[s].lda; p.abs_[s] = a->seg[s].a_batch_stride;
    p.W[s] = a->seg[s].W; p.ldw[s] = a->seg[s].ldw; p.wbs[s] = a->seg[s].w_batch_stride;
    p.K[s] = a->seg[s].K;
  }
  p.nbias = a->nbias; p.nbias2 = a->nbias2;
  p.mbias = a->mbias; p.mbias_bs = a->mbias_batch_stride;
  p.rowbias = a->rowbias; p.rowbias_ld = a->rowbias_ld; p.rowbias_bs = a->rowbias_batch_stride;
  p.mask = a->mask; p.mask_ldm = a->mask_ldm; p.mask_bs = a->mask_batch_stride;
  p.C = a->C; p.ldc = a->ldc; p.cbs = a->c_batch_stride;
  p.M = a->M; p.N = a->N; p.act = a->act; p.m_dev = a->m_dev;
  p.a_t = a->a_kstrided; p.w_t = a->w_kstrided;
  p.ktail = ktail ? 1 : 0;                      // (set before any route to the pipelined kernel, incl. the row-gather one)
  if (a->batch_inner > 1) {
    if (a->batch % a->batch_inner || a->mbias || a->rowbias || a->mask || a->a_row_map || (a->a_inner_stride % 4) ||
        (a->w_inner_stride % 4) || (a->c_inner_stride % 4))
      return GVD_EINVAL;
    p.binner = a->batch_inner; p.abs2 = a->a_inner_stride; p.wbs2 = a->w_inner_stride; p.cbs2 = a->c_inner_stride;
  }
  hipStream_t st = gvd_s(stream);
  if (a->a_row_map) {
    // fused row gather: pipelined kernel only, one plain segment, row offsets within the 32-bit buffer offset
    if (a->nseg != 1 || a->batch != 1 || p.a_t || p.w_t || a->a_src_rows <= 0 ||
        (double)a->a_src_rows * (double)a->seg[0].lda * 4.0 >= 4294967296.0)
      return GVD_EINVAL;
    p.a_rmap = a->a_row_map;
    return gvd_gemm_pipe_launch(p, a->batch, st);
  }
  // K-strided W with one head slot of output columns (the backward products of the training attention core): own kernel
  if (gvd_gemm_n192_ok(p) && !p.mbias && !p.rowbias && !p.mask && !p.nbias && !p.nbias2 && !p.act)
    return gvd_gemm_n192_launch(p, a->batch, st);
  if (p.a_t || p.w_t) return gvd_gemm_pipe_launch(p, a->batch, st);      // backward products: pipelined kernel only
  if (ktail) return gvd_gemm_pipe_launch(p, a->batch, st);
  if (a->M <= 16 && a->batch == 1 && !a->mbias && !a->mask && !a->m_dev) {   // decode batch: weight-streaming skinny kernel
    GemvParams v = {};
    v.nseg = a->nseg;
    for (int s = 0; s < a->nseg; ++s) {
      v.A[s] = a->seg[s].A; v.lda[s] = a->seg[s].lda; v.W[s] = a->seg[s].W; v.ldw[s] = a->seg[s].ldw; v.K[s] = a->seg[s].K;
    }
    v.nbias = a->nbias; v.nbias2 = a->nbias2; v.rowbias = a->rowbias; v.rowbias_ld = a->rowbias_ld;
    v.C = a->C; v.ldc = a->ldc; v.M = a->M; v.N = a->N; v.act = a->act;
    return gvd_gemv_plain(v, st);
  }
  // (the K-split kernel of the LSTM cells, gemm_ks.hip, is NOT used for plain products: every kernel on this path adds the k
  // terms of an output in the same ascending order, so a row's result does not depend on the batch - and hence the kernel - it
  // travels in: tests/test_gpu_kernels.py::test_lstm_persistent_kernel compares a 5-row head against the full batch bit for bit)
  // 17..32 rows: the pipelined 64 x 64 kernel where it is eligible (half of its MFMA rows idle, but the K loop is software
  // pipelined: LSTM cell 99 -> ~45 us, queries / logits 40 -> ~25 us at B = 32, profiles/r03/b32_w_kernel_stats.md), the
  // general 32 x 128 kernel otherwise
  if (a->M <= 32) {
    if (a->M > 16 && gvd_gemm_small_ok(p, a->batch)) return gvd_gemm_small_launch(p, false, st);
    return launch<32, 128, 1, 4, false>(p, a->batch, st);
  }
  // >= 256 tiles of 128 x 128: the software-pipelined kernel with direct global->LDS operand loads (gemm_pipe.hip).  Measured
  // and removed: a double-buffered register-staged 128 x 128 form, 64-deep K tiles, a 256 x 128 tile.
  const long big = (long)((a->M + 127) / 128) * ((a->N + 127) / 128) * a->batch;
  if (big >= 256) return gvd_gemm_pipe_launch(p, a->batch, st);
  if (gvd_gemm_small_ok(p, a->batch)) return gvd_gemm_small_launch(p, false, st);
  return launch<64, 64, 2, 2, false>(p, a->batch, st);
}

extern "C" int gvd_lstm_cell_fwd(const gvd_lstm_args* a, gvd_stream_t stream) {
  if (!a || a->nseg < 1 || a->nseg > 3 || a->B <= 0 || a->H <= 0 || (a->H % 32) != 0 || !a->c_prev ||
      !a->h_out || !a->c_out)
    return GVD_EINVAL;
  KParams p = {};
  p.nseg = a->nseg;
  for (int s = 0; s < a->nseg; ++s) {
    if (!seg_ok(a->seg[s])) return GVD_EINVAL;
    p.A[s] = a->seg[s].A; p.lda[s] = a->seg[s].lda; p.abs_[s] = 0;
    p.W[s] = a->seg[s].W; p.ldw[s] = a->seg[s].ldw; p.wbs[s] = 0;
    p.K[s] = a->seg[s].K;
  }
  p.nbias = a->b_ih; p.nbias2 = a->b_hh;
  p.rowbias = a->rowbias; p.rowbias_ld = a->rowbias_ld;
  p.M = a->B; p.N = 4 * a->H; p.H = a->H;
  p.c_prev = a->c_prev; p.ldcp = a->ldc_prev;
  p.h_out = a->h_out; p.ldh = a->ldh;
  p.c_out = a->c_out; p.ldco = a->ldc_out;
  p.gates_out = a->gates_out; p.ldg = a->ldg;
  hipStream_t st = gvd_s(stream);
  if (a->B <= 16) {
    GemvParams v = {};
    v.nseg = a->nseg;
    for (int s = 0; s < a->nseg; ++s) {
      v.A[s] = a->seg[s].A; v.lda[s] = a->seg[s].lda; v.W[s] = a->seg[s].W; v.ldw[s] = a->seg[s].ldw; v.K[s] = a->seg[s].K;
    }
    v.nbias = a->b_ih; v.nbias2 = a->b_hh; v.rowbias = a->rowbias; v.rowbias_ld = a->rowbias_ld;
    v.M = a->B; v.N = 4 * a->H; v.H = a->H;
    v.c_prev = a->c_prev; v.ldcp = a->ldc_prev; v.h_out = a->h_out; v.ldh = a->ldh;
    v.c_out = a->c_out; v.ldco = a->ldc_out; v.gates_out = a->gates_out; v.ldg = a->ldg;
    return gvd_gemv_lstm(v, st);
  }
  if (gvd_gemm_ks_ok(p, 1) && (a->H % 8) == 0) return gvd_gemm_ks_lstm_launch(p, st);           // 17 .. 128 rows
  if (gvd_gemm_small_ok(p, 1) && (a->H % 16) == 0) return gvd_gemm_small_launch(p, true, st);   // (B = 17..32 too)
  if (a->B <= 32) return launch<32, 128, 1, 4, true>(p, 1, st);
  return launch<64, 64, 2, 2, true>(p, 1, st);
}
