// fp32-MFMA GEMM for the backward products of the training attention core (ops._EncAttnCoreFn.backward):
//     dV_h = Pd_h^T dO_h,   dQ_h = dS_h K_h,   dK_h = dS_h^T Q_h          (per sample and head, transformer.py:90-117)
// i.e. C[M, N] = A W with N = one 176-column head slot, a long contraction (K = Rp = 1024 queries / keys), W K-STRIDED
// ([K, N] row-major: K / Q / dO rows are the contraction index) and A either plain ([M, K]: dS rows) or K-strided ([K, M]:
// Pd / dS read "transposed").  On the 128 x 128 tiles of gemm_pipe.hip these ran at 0.55 of the fp32 MFMA peak (1.55 ms per
// product at batch_size = 64): N = 176 is one full column tile + one "narrow" tile with 48 live columns that re-loads the
// whole A tile for half the MFMAs per k tile - and A IS the [Rp, Rp] map, 1.6 GB per product.
//
// Here ONE workgroup owns all N <= 192 columns of its 128 rows: the map is read once, every k tile feeds 6 (not 4 or 2)
// 32 x 32 accumulator blocks per wave (waves 2 x 2, 64 rows x 96 columns each; the 16 columns past 176 are the only waste),
// k tiles are 16 deep so that two LDS stages are 46 KB and two workgroups share a CU.  Pipeline per k tile (register
// staged, one barrier): global loads of tile t+2 | MFMAs of the first 8 k | LDS writes of tile t+1 (fetched one tile
// earlier) spread between the MFMAs of the second 8 k | barrier | first fragments of tile t+1.  K-strided tiles sit in LDS as [k][row] (ds_read_b32
// fragments, rows padded by 4 floats: the two halves of a wave read k rows 4 apart = 16 banks apart), a plain A tile as
// [row][16 k + 4] (ds_read_b128).  Same v_mfma_f32_32x32x2_f32 chain in the same ascending k order per output as the other
// GEMM kernels of the library: bitwise interchangeable with them.
#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 192, BK = 16;
constexpr int LDA_T = BM + 4;            // K-strided A tile [16 k][132]
constexpr int LDA_P = BK + 4;            // plain A tile [128 rows][20]
constexpr int LDW = BN + 4;              // W tile [16 k][196]
constexpr int A_FLOATS = (BK * LDA_T > BM * LDA_P) ? BK * LDA_T : BM * LDA_P;      // 2560
constexpr int W_FLOATS = BK * LDW;                                                 // 3136
constexpr int STAGE = A_FLOATS + W_FLOATS;                                         // 5696 floats = 22,784 B

template <bool AT>
__global__ __launch_bounds__(256, 2) void gemm_n192_kernel(const KParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];                   // 45,568 B
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int rb = wm * 64, cb = wn * 96;
  // XCD-aware: the row tiles of one batch entry (sample, head) run on ONE XCD, so its W operand (K / Q / dO rows of the head:
  // 0.7 MB) is fetched into that L2 once instead of once per XCD
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int bz = lid / p.ntm;
  const int m0 = (lid - bz * p.ntm) * BM;
  const int M = p.M, N = p.N, K = p.K[0];

  // ---- staging roles
  // W (and a K-strided A): thread covers k row tk + 4 i (W: i < 4 over 48 chunks; A^T: i < 2 over 32 chunks... see below)
  // W tile: 16 k rows x 48 16-byte chunks = 768 pieces = 3 per thread: piece = tid + 256 i -> (k row = piece / 48, chunk = piece % 48)
  // A^T tile: 16 x 32 = 512 pieces = 2 per thread: (k row = piece / 32, chunk = piece % 32)
  // plain A tile: 128 rows x 4 chunks = 512 pieces = 2 per thread: (row = piece / 4, chunk = piece % 4)
  const float* Wb = p.W[0] + gvd_boff(p, bz, p.wbs[0], p.wbs2);
  const float* Ab = p.A[0] + gvd_boff(p, bz, p.abs_[0], p.abs2);
  const int64_t lda = p.lda[0], ldw = p.ldw[0];
  int wk[3], wc[3], wl[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int piece = tid + 256 * i;
    wk[i] = piece / 48;
    const int ch = piece % 48;
    wc[i] = min(4 * ch, N - 4);                          // chunks past the last column re-read the last whole chunk
    wl[i] = wk[i] * LDW + 4 * ch;
  }
  int ak[2], ac[2], al[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int piece = tid + 256 * i;
    if (AT) {
      ak[i] = piece / 32;
      const int ch = piece % 32;
      ac[i] = min(m0 + 4 * ch, M - 4);                   // (M % 4 == 0: whole chunks; rows past M are clamped, never stored)
      al[i] = ak[i] * LDA_T + 4 * ch;
    } else {
      const int row = piece / 4, ch = piece % 4;
      ak[i] = min(m0 + row, M - 1);                      // (row index; `ac` = the k chunk)
      ac[i] = 4 * ch;
      al[i] = row * LDA_P + 4 * ch;
    }
  }
  // two register sets: tile kt+1 waits in one (staged into LDS during tile kt's second half) while tile kt+2 is in flight
  // into the other - one tile of lookahead left the LDS write pass waiting on HBM (the A operand is a streamed 1.6 GB map)
  struct GSet { f32x4 a[2], w[3]; };
  GSet s0, s1;
  auto fetch = [&](GSet& g, int k0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) g.w[i] = *reinterpret_cast<const f32x4*>(Wb + (int64_t)(k0 + wk[i]) * ldw + wc[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      g.a[i] = AT ? *reinterpret_cast<const f32x4*>(Ab + (int64_t)(k0 + ak[i]) * lda + ac[i])
                  : *reinterpret_cast<const f32x4*>(Ab + (int64_t)ak[i] * lda + k0 + ac[i]);
  };
  auto stage_a = [&](const GSet& g, int buf, int i) { *reinterpret_cast<f32x4*>(smem + buf * STAGE + al[i]) = g.a[i]; };
  auto stage_w = [&](const GSet& g, int buf, int i) { *reinterpret_cast<f32x4*>(smem + buf * STAGE + A_FLOATS + wl[i]) = g.w[i]; };

  // ---- fragments of quarter q (8 k values) of the tile in `buf`: lane (r, half) takes k = 8 q + 4 half + t for MFMA step t
  auto frags = [&](f32x4 (&a)[2], f32x4 (&b)[3], int buf, int q) {
    const float* As = smem + buf * STAGE;
    const float* Ws = As + A_FLOATS;
    if (AT) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[0][t] = As[(8 * q + 4 * half + t) * LDA_T + rb + r];
        a[1][t] = As[(8 * q + 4 * half + t) * LDA_T + rb + 32 + r];
      }
    } else {
      a[0] = *reinterpret_cast<const f32x4*>(As + (rb + r) * LDA_P + 8 * q + 4 * half);
      a[1] = *reinterpret_cast<const f32x4*>(As + (rb + 32 + r) * LDA_P + 8 * q + 4 * half);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 3; ++j) b[j][t] = Ws[(8 * q + 4 * half + t) * LDW + cb + 32 * j + r];
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // every 32 x 32 block is multiplied, also one that lies wholly past M or N: its operands are clamped re-reads of live
  // rows / columns and the epilogue stores nothing of it (a per-block "live" flag is VGPR-derived, so guarding made EVERY
  // MFMA a saveexec + branch + branch-back in the ISA)
  auto mfma_t = [&](const f32x4 (&a)[2], const f32x4 (&b)[3], int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
  };
  const int nkt = K / BK;
  f32x4 a0[2], b0[3], a1[2], b1[3];
  fetch(s0, 0);
  fetch(s1, BK);                               // (K >= 2 BK: gvd_gemm_n192_ok)
#pragma unroll
  for (int i = 0; i < 2; ++i) stage_a(s0, 0, i);
#pragma unroll
  for (int i = 0; i < 3; ++i) stage_w(s0, 0, i);
  __syncthreads();
  frags(a0, b0, 0, 0);
  int buf = 0;
  // one k tile: multiplies tile kt (in LDS buffer `buf`), stages tile kt+1 (waiting in `cur`) and fetches tile kt+2 into `nxt`
  auto tile = [&](int kt, const GSet& cur, GSet& nxt) {
    if (kt + 2 < nkt) fetch(nxt, (kt + 2) * BK);
    frags(a1, b1, buf, 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) mfma_t(a0, b0, t);
    // second quarter, with the LDS write pass of tile kt+1 spread between its MFMA groups (nobody reads buf^1: its last
    // reads preceded the previous barrier)
    mfma_t(a1, b1, 0);
    stage_a(cur, buf ^ 1, 0); stage_w(cur, buf ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_t(a1, b1, 1);
    stage_a(cur, buf ^ 1, 1); stage_w(cur, buf ^ 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_t(a1, b1, 2);
    stage_w(cur, buf ^ 1, 2);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    frags(a0, b0, buf ^ 1, 0);
    mfma_t(a1, b1, 3);
    buf ^= 1;
  };
  int kt = 0;
#pragma unroll 1
  for (; kt + 2 < nkt; kt += 2) {              // (two tiles per trip: the register sets swap roles without copies)
    tile(kt, s1, s0);
    tile(kt + 1, s0, s1);
  }
  if (kt + 1 < nkt) tile(kt, s1, s0);
  frags(a1, b1, buf, 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) mfma_t(a0, b0, t);
#pragma unroll
  for (int t = 0; t < 4; ++t) mfma_t(a1, b1, t);

  gemm_epilogue_plain<2, 3>(p, M, acc, bz, m0 + rb, cb, r, half);
}

}  // namespace

// One K-strided-W product with at most 192 output columns, a 16-multiple contraction and no epilogue extras beyond what
// gemm_epilogue_plain applies.
bool gvd_gemm_n192_ok(const KParams& p) {
  return p.w_t && p.nseg == 1 && p.N > 128 && p.N <= BN && (p.N % 4) == 0 && (p.K[0] % BK) == 0 && p.K[0] >= 2 * BK &&
         !p.m_dev && !p.a_rmap && p.M >= 4 && (!p.a_t || (p.M % 4) == 0) && (p.abs_[0] % 4) == 0 && (p.wbs[0] % 4) == 0 &&
         (p.lda[0] % 4) == 0 && (p.ldw[0] % 4) == 0;
}

int gvd_gemm_n192_launch(KParams& p, int batch, hipStream_t st) {
  p.ntm = (p.M + BM - 1) / BM;
  p.ntn = 1;
  dim3 grid((unsigned)(p.ntm * batch));
  if (p.a_t) hipLaunchKernelGGL(gemm_n192_kernel<true>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(gemm_n192_kernel<false>, grid, dim3(256), 0, st, p);
  GVD_CHECK_LAUNCH();
  return 0;
}
