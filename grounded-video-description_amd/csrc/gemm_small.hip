// Software-pipelined fp32-MFMA GEMM for the token-loop products (M = decode batch of 17..512 rows: the two LSTM cells with
// their fused cell epilogue, both attention queries, the vocabulary logits).
//
// These launches have one 64 x 64 tile per CU (M = 256: 4 x 64 tiles) and a long K loop (1536 .. 3072) over weights that
// stream from L2 / Infinity Cache: with one wave per SIMD nothing hides a stall, and the general kernel (gemm_f32.hip)
// ran them at 37 % of the MFMA rate (LSTM 91 us at B = 256) - every 32-deep k tile paid its global-load latency, a
// register->LDS pass and a barrier in the open.  Same structure as gemm_pipe.hip, re-dimensioned for the small tile:
//   * 64-deep k tiles (2048 MFMA cycles per barrier), operands through buffer loads with precomputed offsets;
//   * register prefetch TWO tiles ahead (a tile has ~1.75 iterations = ~3500 cycles to arrive from L2 / MALL);
//   * the LDS write pass of the next tile interleaved with the MFMAs of the 7th of 8 k-quarters, the tile's only barrier
//     and the first fragment reads of the next tile under the 8th;
//   * fragment registers double-buffered and pinned ahead of the MFMAs.
// Numerics: the k order per output element is the one of gemm_f32.hip / gemm_pipe.hip (k = 8q + 4 half + t per MFMA
// step, ascending tiles) - bitwise the same results.
#include "gemm_common.h"

namespace {

constexpr int SB = 64, SK = 64, SLD = SK + 4;     // tile, k depth, padded LDS row (floats)
constexpr int SNLD = 4;                           // 16-byte loads per thread per operand per k tile (64 rows x 16 / 256)
constexpr int SHU = SB / 4;                       // LSTM: hidden units per tile (4 gates x 16)

template <bool LSTM>
__global__ __launch_bounds__(256, 1) void gemm_small_kernel(const KParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * SB * SLD];      // [buf][A|W][64][68] = 69,632 B
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // consecutive ids walk the M tiles of one weight panel: the (few) row tiles sharing a W panel sit on one XCD's L2
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm_ = lid % p.ntm, tn_ = lid / p.ntm;
  const int m0 = tm_ * SB, n0 = LSTM ? 0 : tn_ * SB;
  // device-side row count (compacted preamble of a small batch: the few-tile products of its 4000 region rows): the grid is
  // sized for the worst case, row tiles past the live count leave at once
  const int M = p.m_dev ? min(*p.m_dev, p.M) : p.M;
  if (m0 >= M) return;

  const int srow = tid >> 4, kq = tid & 15;       // thread covers rows srow + 16 i, 16-byte chunk kq of a 256-byte k slice
  int arow[SNLD], wrow[SNLD];
#pragma unroll
  for (int i = 0; i < SNLD; ++i) {
    arow[i] = min(m0 + srow + 16 * i, M - 1);
    const int nl = srow + 16 * i;
    wrow[i] = LSTM ? (nl / SHU) * p.H + tn_ * SHU + (nl % SHU) : min(n0 + nl, p.N - 1);
  }
  __amdgpu_buffer_rsrc_t ra, rw;
  unsigned voa[SNLD], vow[SNLD];
  int seg = 0, kpos = 0;
  const int kseg1 = p.K[1], kseg2 = p.K[2], nseg = p.nseg;
  int kend = p.K[0];
  auto seg_setup = [&](int s) {
    ra = gvd_rsrc(p.A[s]);
    rw = gvd_rsrc(p.W[s]);
    const unsigned lda4 = (unsigned)p.lda[s] * 4u, ldw4 = (unsigned)p.ldw[s] * 4u;
#pragma unroll
    for (int i = 0; i < SNLD; ++i) {
      voa[i] = (unsigned)arow[i] * lda4 + 16u * kq;
      vow[i] = (unsigned)wrow[i] * ldw4 + 16u * kq;
    }
  };
  seg_setup(0);
  int nkt = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (s < p.nseg) nkt += p.K[s] / SK;

  f32x4 ga[2][SNLD], gw[2][SNLD];
  auto fetch = [&](int set) {
    const unsigned so = 4u * (unsigned)kpos;
#pragma unroll
    for (int i = 0; i < SNLD; ++i) ga[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa[i], so, 0));
#pragma unroll
    for (int i = 0; i < SNLD; ++i) gw[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, vow[i], so, 0));
    kpos += SK;
    if (kpos == kend && seg + 1 < nseg) {
      ++seg;
      kpos = 0;
      kend = seg == 1 ? kseg1 : kseg2;
      seg_setup(seg);
    }
  };
  float* Ast = smem + srow * SLD + 4 * kq;
  float* Wst = smem + SB * SLD + srow * SLD + 4 * kq;
  auto stage_part = [&](int set, int buf, int i) {
    *reinterpret_cast<f32x4*>(Ast + buf * (2 * SB * SLD) + 16 * i * SLD) = ga[set][i];
    *reinterpret_cast<f32x4*>(Wst + buf * (2 * SB * SLD) + 16 * i * SLD) = gw[set][i];
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* Afr = smem + (wm * 32 + r) * SLD + half * 4;
  const float* Wfr = smem + SB * SLD + (wn * 32 + r) * SLD + half * 4;
  auto frags = [&](f32x4& a, f32x4& b, int buf, int q) {
    a = *reinterpret_cast<const f32x4*>(Afr + buf * (2 * SB * SLD) + q * 8);
    b = *reinterpret_cast<const f32x4*>(Wfr + buf * (2 * SB * SLD) + q * 8);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma4 = [&](const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
  };

  // prologue: tile 0 -> LDS, tile 1 -> registers
  fetch(0);
#pragma unroll
  for (int i = 0; i < SNLD; ++i) stage_part(0, 0, i);
  if (nkt > 1) fetch(1);
  __syncthreads();
  f32x4 a0, b0, a1, b1;
  frags(a0, b0, 0, 0);
  // two tiles per trip so that the register set / LDS buffer indices are compile-time constants
  // cur = kt & 1: LDS buffer of tile kt; registers hold tile kt+1 in set cur^1.  do_fetch: tile kt+2 exists (its set was
  // staged into LDS one tile ago); more: tile kt+1 exists.  Both are literals at every call site (straight-line bodies).
  auto tile = [&](const int cur, const bool do_fetch, const bool more) {
    if (do_fetch) fetch(cur);
    // quarters 0..5: (a0,b0) holds the even, (a1,b1) the odd quarter; q+1 is read while q multiplies
#pragma unroll
    for (int q = 0; q < 6; q += 2) {
      frags(a1, b1, cur, q + 1);
      mfma4(a0, b0);
      frags(a0, b0, cur, q + 2);
      mfma4(a1, b1);
    }
    frags(a1, b1, cur, 7);
    if (more) {
      // quarter 6 with the LDS write pass of tile kt+1 spread between its MFMAs, then the tile's only barrier, then the
      // first fragments of tile kt+1 - all under the MFMAs of quarters 6 and 7
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
        stage_part(cur ^ 1, cur ^ 1, t);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      frags(a0, b0, cur ^ 1, 0);
    } else {
      mfma4(a0, b0);
    }
    mfma4(a1, b1);
  };
  int kt = 0;
#pragma unroll 1
  for (; kt + 3 < nkt; kt += 2) {                 // steady state: no conditionals inside
    tile(0, true, true);
    tile(1, true, true);
  }
  // tail: 1..3 tiles left, tile kt sits in LDS buffer 0
  const int left = nkt - kt;
  if (left == 3) { tile(0, true, true); tile(1, false, true); tile(0, false, false); }
  else if (left == 2) { tile(0, false, true); tile(1, false, false); }
  else { tile(0, false, false); }

  if (!LSTM) {
    gemm_epilogue_plain<1, 1>(p, M, reinterpret_cast<const f32x16(&)[1][1]>(acc), 0, m0 + wm * 32, n0 + wn * 32, r, half);
    return;
  }
  // ---- LSTM cell epilogue (nn.LSTMCell, AttModel.py:139,160): gates -> LDS tile G[64][65] (columns grouped i|f|g|o,
  // 16 units each), then the pointwise cell
  constexpr int LDG = SB + 1;
  __syncthreads();                                // operand tiles are dead
  float* G = smem;
  {
    const int nl = wn * 32 + r;
    const int wr = (nl / SHU) * p.H + tn_ * SHU + (nl % SHU);
    float nb = 0.f;
    if (p.nbias) nb += p.nbias[wr];
    if (p.nbias2) nb += p.nbias2[wr];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ml = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      const int gm = m0 + ml;
      float v = acc[e] + nb;
      if (p.rowbias && gm < M) v += p.rowbias[(int64_t)gm * p.rowbias_ld + wr];
      G[ml * LDG + nl] = v;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < SB * SHU; idx += 256) {
    const int ml = idx / SHU, jl = idx % SHU;
    const int gm = m0 + ml;
    if (gm >= M) continue;
    const int j = tn_ * SHU + jl;
    const float gi = sigmoid_f(G[ml * LDG + jl]);
    const float gf = sigmoid_f(G[ml * LDG + SHU + jl]);
    const float gg = tanhf(G[ml * LDG + 2 * SHU + jl]);
    const float go = sigmoid_f(G[ml * LDG + 3 * SHU + jl]);
    const float c = gf * p.c_prev[(int64_t)gm * p.ldcp + j] + gi * gg;
    p.c_out[(int64_t)gm * p.ldco + j] = c;
    p.h_out[(int64_t)gm * p.ldh + j] = go * tanhf(c);
    if (p.gates_out) {
      float* g = p.gates_out + (int64_t)gm * p.ldg;
      g[j] = gi; g[p.H + j] = gf; g[2 * p.H + j] = gg; g[3 * p.H + j] = go;
    }
  }
}

}  // namespace

// eligibility: every K segment a multiple of 64; plain: single batch
bool gvd_gemm_small_ok(const KParams& p, int batch) {
  if (batch != 1 || p.a_t || p.w_t) return false;
  for (int s = 0; s < p.nseg; ++s)
    if (p.K[s] % SK) return false;
  return true;
}

int gvd_gemm_small_launch(KParams& p, bool lstm, hipStream_t st) {
  p.ntm = (p.M + SB - 1) / SB;
  p.ntn = lstm ? p.H / SHU : (p.N + SB - 1) / SB;
  dim3 grid((unsigned)(p.ntm * p.ntn));
  if (lstm) hipLaunchKernelGGL(gemm_small_kernel<true>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(gemm_small_kernel<false>, grid, dim3(256), 0, st, p);
  GVD_CHECK_LAUNCH();
  return 0;
}
