// Software-pipelined fp32-MFMA "NT" GEMM for the large per-segment projections of the hot path (fc7, class logits,
// pool_embed, the obj_interact projections / feed-forward, ctx2pool: M = B*R rows, hundreds of tiles per CU).
//
//   C[b][M,N] = epi( sum_s A_s[b][M,K_s] * W_s[b][N,K_s]^T ),  128 x 128 output tile per workgroup, 32-deep k tiles
//
// Why a second kernel (gemm_f32.hip keeps the small/odd shapes and the LSTM epilogue): PMC on the general kernel showed
// 81 % MFMA-busy against 95 % for the library kernel on the same shape.  Its k-loop spends the gap between two k tiles
// in scalar segment look-ups, 64-bit address arithmetic, eight exec-masked loads, a register->LDS pass and two
// barriers, all with the matrix pipe of that wave idle.  Here the loop carries nothing but loads, LDS traffic and MFMAs:
//   * operands come through buffer loads: one SGPR descriptor per operand and segment, one precomputed 32-bit offset
//     per (thread, row) and the k position as the scalar offset - no per-tile address VALU, no branches.  Rows past
//     M / N are CLAMPED to the last valid row instead of zero-filled: their accumulators are never stored;
//   * one barrier per k tile, placed BEFORE the tile's last quarter: order per tile is
//         global loads (tile t+1) | MFMA q0 | MFMA q1 | MFMA q2 + LDS writes (tile t+1) | barrier |
//         fragment reads (tile t+1, q0) | MFMA q3
//     so the barrier wait, the LDS write pass and the first fragment reads of the next tile all sit under 16 MFMAs
//     (1024 cycles) of the current one; fragment registers are double-buffered (q+1 is read while q multiplies);
//   * epilogue through LDS: the wave's 64 x 64 block is transposed in its own LDS slice and leaves as 16-byte stores
//     (4 rows x 256 B per instruction instead of 2 rows x 128 B of dword stores) when the output allows it.
// Numerics are identical to gemm_f32.hip: the same v_mfma_f32_32x32x2_f32 chain in the same k order per output.
#include "gemm_common.h"

// DIRECT-TO-LDS OPERANDS: operand tiles of the plain (not K-strided) products go global -> LDS by direct loads
// (buffer_load_dwordx4 ... lds, 16 bytes per lane on gfx950), without a register round trip and its ds_write_b128 pass (6 of
// the register-staged kernel's 10 idle points in an ablation: 141.1 -> 149.5 TF/s without that pass).  A
// direct load writes lane l's 16 bytes to LDS at base + 16 l, so the tile is stored UNPADDED (32 floats per row, 8 rows per
// wave instruction) with an XOR swizzle instead of 36-float rows: 16-byte slot s of row m holds the k-chunk s ^ (m & 7); the
// loading lane fetches the k-chunk that belongs at its slot, the fragment reads of 8 consecutive rows hit 8 different slots.
// Bitwise equal to the register-staged form it replaced (same k order).  K-strided operands (the backward products) keep the
// register-staged path below: their tiles are [32 k][128] with ds_read_b32 fragments, and direct loads measured +1 % there.

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDK = BK + 4;
constexpr int NLD = 4;                 // 16-byte loads per thread per operand per k tile (128 rows x 8 / 256)
constexpr int EPI_LD = 68;             // padded row of the epilogue transpose slice (floats)
constexpr int LDT = BM + 4;            // LDS row (floats) of a K-STRIDED operand tile [32 k][128 rows]
// K-strided ("transposed") operands, for the backward products of nn.Linear: dX = dY W (W: [N(contraction), K]) and
// dW = dY^T X (both operands [M(contraction), .]).  Such an operand's tile is 32 memory rows (k) of 128 contiguous
// floats: loaded with the same 16-byte buffer loads, stored row-major [k][row] in LDS, and a lane takes its MFMA values
// with four ds_read_b32 (LDS[8q + 4 half + t][row]) instead of one ds_read_b128 - same k order, same accumulators.

struct Seg {
  __amdgpu_buffer_rsrc_t ra, rw;
  unsigned voa[NLD], vow[NLD];
};

// EDGE: launches whose last M or N tile is at most half full (e.g. the N = 192 head blocks of the training attention
// core) skip the MFMAs of 32 x 32 sub-tiles that lie wholly outside the output - a wave with nothing to multiply leaves its
// SIMD's matrix pipe to the other resident workgroup, so a half-empty tile costs about half a tile.
template <bool EPI_LDS, bool AT = false, bool BT = false, bool EDGE = false>
__global__ __launch_bounds__(256, 2) void gemm_pipe_kernel(const KParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];     // 73,728 B -> two workgroups per CU
  constexpr bool DMA = !AT && !BT;               // plain products: direct-to-LDS operand loads
  constexpr int DLD = BK;                        // DMA layout of a plain operand tile: unpadded rows of 32 floats
  constexpr int A_STRIDE = DMA ? BM * DLD : BM * LDK;      // floats per LDS buffer of the A / W tile
  constexpr int W_STRIDE = DMA ? BN * DLD : BN * LDK;
  float* As = smem;
  float* Ws = smem + 2 * A_STRIDE;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // device-side row count (compacted preamble): only the first ceil(M/128) * ntn workgroups work, and THEY are remapped
  // XCD-aware among themselves (hardware deals consecutive ids round-robin to the XCDs, so the live ones stay balanced)
  const int M = p.m_dev ? min(*p.m_dev, p.M) : p.M;
  const unsigned nlive = p.m_dev ? (unsigned)((M + BM - 1) / BM) * p.ntn : gridDim.x;
  if (blockIdx.x >= nlive) return;
  const unsigned lid = xcd_remap(blockIdx.x, nlive);
  const int tn_ = lid % p.ntn, tm_ = lid / p.ntn;
  const int bz = blockIdx.y;
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  // EDGE: an N tile with at most 64 live columns is split 4 x 1 over the waves (32 rows x 64 columns each, all four SIMDs
  // at half the MFMA count) instead of 2 x 2 with two idle waves; `rb` / `cb` = the wave's block origin inside the tile
  const bool narrow = EDGE && n0 + BN / 2 >= p.N;
  const int rb = narrow ? wave * 32 : wm * 64, cb = narrow ? 0 : wn * 64;

  // staging role: thread covers rows srow + 32 i (i < 4), 16-byte chunk kq of the 128-byte k slice of a row
  const int srow = tid >> 3, kq = tid & 7;
  // DMA layout: this lane's 16 bytes land in slot kq of tile row srow + 32 i, which holds the k-chunk kq ^ (row & 7)
  const int kq_sw = kq ^ (srow & 7);
  int arow[NLD], wrow[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    arow[i] = min(m0 + srow + 32 * i, M - 1) - m0;       // clamp: rows past the edge re-read the last valid row
    wrow[i] = min(n0 + srow + 32 * i, p.N - 1) - n0;
  }
  // fused row gather (fc7 over the compacted proposal set): row m of the product reads row a_rmap[m] of A; the operand
  // descriptor then starts at A itself (the host checked that every row offset fits the 32-bit buffer offset)
  const bool gathered = !AT && p.a_rmap != nullptr;
  if (gathered) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) arow[i] = p.a_rmap[m0 + arow[i]];
  }
  // K-strided operands: thread covers memory rows (k) tk + 8 i, 16-byte column chunk tc (columns = output rows / cols;
  // chunks past the edge are clamped to the last whole chunk - M, N are multiples of 4 there)
  const int tk = tid >> 5, tc = tid & 31;
  const int acol = min(m0 + 4 * tc, M - 4) - m0, wcol = min(n0 + 4 * tc, p.N - 4) - n0;

  Seg sg;
  int seg = 0, kpos = 0;                                    // position of the NEXT tile to fetch
  const int kseg1 = p.K[1], kseg2 = p.K[2], nseg = p.nseg;  // (kept in SGPRs: no kernarg reload inside the k loop)
  int kend = p.K[0];
  const float* pa_t = nullptr;                              // K-strided operands: running base of the current k tile
  const float* pw_t = nullptr;
  auto seg_setup = [&](int s) {
    const unsigned lda4 = (unsigned)p.lda[s] * 4u, ldw4 = (unsigned)p.ldw[s] * 4u;
    if (AT) {
      pa_t = p.A[s] + gvd_boff(p, bz, p.abs_[s], p.abs2) + m0;
#pragma unroll
      for (int i = 0; i < NLD; ++i) sg.voa[i] = (unsigned)(tk + 8 * i) * lda4 + 4u * (unsigned)acol;
    } else {
      sg.ra = gvd_rsrc(p.A[s] + gvd_boff(p, bz, p.abs_[s], p.abs2) + (gathered ? 0 : (int64_t)m0 * p.lda[s]));
#pragma unroll
      for (int i = 0; i < NLD; ++i) sg.voa[i] = (unsigned)arow[i] * lda4 + 16u * (DMA ? kq_sw : kq);
    }
    if (BT) {
      pw_t = p.W[s] + gvd_boff(p, bz, p.wbs[s], p.wbs2) + n0;
#pragma unroll
      for (int i = 0; i < NLD; ++i) sg.vow[i] = (unsigned)(tk + 8 * i) * ldw4 + 4u * (unsigned)wcol;
    } else {
      sg.rw = gvd_rsrc(p.W[s] + gvd_boff(p, bz, p.wbs[s], p.wbs2) + (int64_t)n0 * p.ldw[s]);
#pragma unroll
      for (int i = 0; i < NLD; ++i) sg.vow[i] = (unsigned)wrow[i] * ldw4 + 16u * (DMA ? kq_sw : kq);
    }
  };
  seg_setup(0);
  const int64_t lda_t = p.lda[0], ldw_t = p.ldw[0];         // (K-strided operands use a single segment)

  int nkt = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (s < p.nseg) nkt += p.K[s] / BK;
  // K TAIL OF 16 (host sets p.ktail only for one-segment plain products on the direct-to-LDS path; used by the training attention
  // core with its 176-column head slots - verified on the device against the 192-slot form and by the reference gradient goldens,
  // tests/test_gpu_kernels.py::test_enc_attn_core_176_column_head_slots): the last tile is fetched SHIFTED BACK by 16
  // columns (k = K-32 .. K-1: always inside the operand rows, never past them) and only its last two quarters are
  // multiplied - the first two repeat columns the previous tile already covered.  Same ascending k order per output.
  const bool ktail = DMA && p.ktail != 0;
  if (ktail) ++nkt;

  f32x4 ga[NLD], gw[NLD];
  auto fetch = [&]() {
    const unsigned so = 4u * (unsigned)kpos;
    if (AT) {                                   // the descriptor follows the k tile (row offsets can exceed 32 bits)
      const __amdgpu_buffer_rsrc_t ra = gvd_rsrc(pa_t + (int64_t)kpos * lda_t);
#pragma unroll
      for (int i = 0; i < NLD; ++i) ga[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, sg.voa[i], 0, 0));
    } else {
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        ga[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sg.ra, sg.voa[i], so, 0));
    }
    if (BT) {
      const __amdgpu_buffer_rsrc_t rw = gvd_rsrc(pw_t + (int64_t)kpos * ldw_t);
#pragma unroll
      for (int i = 0; i < NLD; ++i) gw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, sg.vow[i], 0, 0));
    } else {
#pragma unroll
      for (int i = 0; i < NLD; ++i)
        gw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sg.rw, sg.vow[i], so, 0));
    }
    kpos += BK;
    if (kpos == kend && seg + 1 < nseg) {                  // wave-uniform, taken nseg-1 times per workgroup
      ++seg;
      kpos = 0;
      kend = seg == 1 ? kseg1 : kseg2;
      seg_setup(seg);
    }
  };
  // direct global -> LDS loads of the tile at kpos into buffer `buf` (DMA layout): wave w's instruction i covers tile rows
  // 8 w + 32 i .. + 7 (one KiB of LDS, lane l at + 16 l)
  auto dma = [&](int buf) {
    const unsigned so = 4u * (unsigned)((ktail && kpos + BK > kend) ? kend - BK : kpos);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sg.ra, (__attribute__((address_space(3))) void*)&As[buf * A_STRIDE + (8 * wv + 32 * i) * DLD],
                                               16, sg.voa[i], so, 0, 0);
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(sg.rw, (__attribute__((address_space(3))) void*)&Ws[buf * W_STRIDE + (8 * wv + 32 * i) * DLD],
                                               16, sg.vow[i], so, 0, 0);
    kpos += BK;
    if (kpos == kend && seg + 1 < nseg) {
      ++seg;
      kpos = 0;
      kend = seg == 1 ? kseg1 : kseg2;
      seg_setup(seg);
    }
  };
  float* Ast = AT ? &As[tk * LDT + 4 * tc] : &As[srow * LDK + 4 * kq];
  float* Wst = BT ? &Ws[tk * LDT + 4 * tc] : &Ws[srow * LDK + 4 * kq];
  auto stage_part = [&](int buf, int i) {      // one A row and one W row of this thread's share of the tile
    *reinterpret_cast<f32x4*>(Ast + (AT ? buf * BM * LDK + 8 * i * LDT : (buf * BM + 32 * i) * LDK)) = ga[i];
    *reinterpret_cast<f32x4*>(Wst + (BT ? buf * BN * LDK + 8 * i * LDT : (buf * BN + 32 * i) * LDK)) = gw[i];
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) stage_part(buf, i);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment q (8 k values) of the tile in `buf`: lane (r, half) takes k = 8q + 4 half + t for MFMA step t
  static_assert(BK * LDT <= BM * LDK, "a K-strided tile fits the operand buffer");
  const float* Afr = AT ? &As[half * 4 * LDT + rb + r] : &As[(rb + r) * LDK + half * 4];
  const float* Wfr = BT ? &Ws[half * 4 * LDT + cb + r] : &Ws[(cb + r) * LDK + half * 4];
  // DMA layout of a plain tile: k-chunk 2 q + half of row m sits in slot (2 q + half) ^ (m & 7); rb / cb are multiples of
  // 32, so m & 7 = r & 7.
  const float* Adm = &As[(rb + r) * DLD];
  const float* Wdm = &Ws[(cb + r) * DLD];
  const int rsw = r & 7;
  auto frags = [&](f32x4 (&a)[2], f32x4 (&b)[2], int buf, int q) {
    if (DMA) {
      const int so4 = ((2 * q + half) ^ rsw) * 4;
      a[0] = *reinterpret_cast<const f32x4*>(Adm + buf * A_STRIDE + so4);
      a[1] = *reinterpret_cast<const f32x4*>(Adm + buf * A_STRIDE + 32 * DLD + so4);
      b[0] = *reinterpret_cast<const f32x4*>(Wdm + buf * W_STRIDE + so4);
      b[1] = *reinterpret_cast<const f32x4*>(Wdm + buf * W_STRIDE + 32 * DLD + so4);
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
    if (AT) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[0][t] = Afr[buf * BM * LDK + (q * 8 + t) * LDT];
        a[1][t] = Afr[buf * BM * LDK + (q * 8 + t) * LDT + 32];
      }
    } else {
      a[0] = *reinterpret_cast<const f32x4*>(Afr + buf * BM * LDK + q * 8);
      a[1] = *reinterpret_cast<const f32x4*>(Afr + buf * BM * LDK + 32 * LDK + q * 8);
    }
    if (BT) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        b[0][t] = Wfr[buf * BN * LDK + (q * 8 + t) * LDT];
        b[1][t] = Wfr[buf * BN * LDK + (q * 8 + t) * LDT + 32];
      }
    } else {
      b[0] = *reinterpret_cast<const f32x4*>(Wfr + buf * BN * LDK + q * 8);
      b[1] = *reinterpret_cast<const f32x4*>(Wfr + buf * BN * LDK + 32 * LDK + q * 8);
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the reads AHEAD of the MFMAs that follow in program order
  };
  bool lv[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      lv[i][j] = !EDGE || (!(narrow && i == 1) && m0 + __builtin_amdgcn_readfirstlane(rb) + i * 32 < M &&
                           n0 + __builtin_amdgcn_readfirstlane(cb) + j * 32 < p.N);
  auto mfma4 = [&](const f32x4 (&a)[2], const f32x4 (&b)[2], int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (!EDGE || lv[i][j]) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
  };
  auto mfma16 = [&](const f32x4 (&a)[2], const f32x4 (&b)[2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) mfma4(a, b, t);
  };

  f32x4 a0[2], b0[2], a1[2], b1[2];
  int buf = 0;
  if (DMA) {
    dma(0);
    __syncthreads();                             // (s_waitcnt vmcnt(0) + barrier: every wave's direct loads have landed)
    frags(a0, b0, 0, 0);
#pragma unroll 1
    for (int kt = 0; kt + 1 < nkt; ++kt) {
      dma(buf ^ 1);                              // tile kt+1 straight into the other buffer: its last reads preceded the
      frags(a1, b1, buf, 1);                     // previous barrier; three quarters (~3000 cycles) to land
      mfma16(a0, b0);
      frags(a0, b0, buf, 2);
      mfma16(a1, b1);
      frags(a1, b1, buf, 3);
      mfma16(a0, b0);
      __syncthreads();
      if (ktail && kt + 2 == nkt) frags(a0, b0, buf ^ 1, 2);     // the shifted tail tile starts at its third quarter
      else frags(a0, b0, buf ^ 1, 0);
      mfma16(a1, b1);
      buf ^= 1;
    }
  } else {
  fetch();
  stage(0);
  __syncthreads();
  frags(a0, b0, 0, 0);
#pragma unroll 1
  for (int kt = 0; kt + 1 < nkt; ++kt) {
    fetch();                                   // tile kt+1: in flight under the first three quarters
    frags(a1, b1, buf, 1);
    mfma16(a0, b0);
    frags(a0, b0, buf, 2);
    mfma16(a1, b1);
    frags(a1, b1, buf, 3);
    // third quarter, with the LDS write pass of tile kt+1 spread between its MFMAs (nobody reads buf^1: its last reads
    // preceded the previous barrier; the global loads were issued three quarters = ~3000 cycles ago)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      mfma4(a0, b0, t);
      stage_part(buf ^ 1, t);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    frags(a0, b0, buf ^ 1, 0);
    mfma16(a1, b1);
    buf ^= 1;
  }
  }
  if (ktail) {                      // last tile = columns K-32 .. K-1: quarters 2 and 3 are the 16 new ones
    frags(a1, b1, buf, 3);
    mfma16(a0, b0);
    mfma16(a1, b1);
  } else {
    frags(a1, b1, buf, 1);
    mfma16(a0, b0);
    frags(a0, b0, buf, 2);
    mfma16(a1, b1);
    frags(a1, b1, buf, 3);
    mfma16(a0, b0);
    mfma16(a1, b1);
  }

  if (!EPI_LDS) {
    // (narrow tiles: the wave's second row block is not part of the tile - push it past M so nothing is stored)
    if (narrow) gemm_epilogue_plain<1, 2>(p, M, reinterpret_cast<const f32x16(&)[1][2]>(acc[0]), bz, m0 + rb, n0 + cb, r, half);
    else gemm_epilogue_plain<2, 2>(p, M, acc, bz, m0 + rb, n0 + cb, r, half);
    return;
  }
  // ---- epilogue through LDS (bias / bias2 / ReLU only; N % 4 == 0, ldc % 4 == 0, C 16-byte aligned)
  __syncthreads();                                             // every wave finished reading the operand tiles
  float* T = smem + wave * 64 * EPI_LD;                        // this wave's private 64 x 64 slice
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        T[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * EPI_LD + j * 32 + r] = acc[i][j][e];
  const int c4 = (lane & 15) * 4, rsub = lane >> 4;
  const int gn = n0 + cb + c4;
  f32x4 nb = {0.f, 0.f, 0.f, 0.f};
  if (gn < p.N) {
    if (p.nbias) nb = *reinterpret_cast<const f32x4*>(p.nbias + gn);
    if (p.nbias2) nb += *reinterpret_cast<const f32x4*>(p.nbias2 + gn);
  }
  float* Cb = p.C + gvd_boff(p, bz, p.cbs, p.cbs2);
  // [M, N] addend (16-byte aligned rows; the launcher checked): dX = dY W + the gradient the same tensor receives through
  // its other consumer (residual branches of the encoder, decoder_bwd) - one 16-byte read per store instead of a separate
  // elementwise pass over both
  const float* Rb = p.rowbias ? p.rowbias + (int64_t)bz * p.rowbias_bs : nullptr;
  const bool relu = p.act == 1;
  // (DS operations of one wave execute in order: its reads below see its own writes above)
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int row = it * 4 + rsub;
    const int gm = (narrow && row >= 32) ? M : m0 + rb + row;        // narrow tiles: 32 rows per wave
    f32x4 v = *reinterpret_cast<const f32x4*>(&T[row * EPI_LD + c4]) + nb;
    const bool live = gm < M && gn < p.N;
    if (Rb && live) v += *reinterpret_cast<const f32x4*>(Rb + (int64_t)gm * p.rowbias_ld + gn);
    if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    if (live) *reinterpret_cast<f32x4*>(Cb + (int64_t)gm * p.ldc + gn) = v;
  }
}

}  // namespace

namespace {
template <bool EDGE>
int pipe_launch_t(const KParams& p, dim3 grid, bool lds_epi, hipStream_t st) {
  if (p.a_t && p.w_t) {
    if (lds_epi) hipLaunchKernelGGL((gemm_pipe_kernel<true, true, true, EDGE>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_pipe_kernel<false, true, true, EDGE>), grid, dim3(256), 0, st, p);
  } else if (p.w_t) {
    if (lds_epi) hipLaunchKernelGGL((gemm_pipe_kernel<true, false, true, EDGE>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_pipe_kernel<false, false, true, EDGE>), grid, dim3(256), 0, st, p);
  } else {
    if (lds_epi) hipLaunchKernelGGL((gemm_pipe_kernel<true, false, false, EDGE>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_pipe_kernel<false, false, false, EDGE>), grid, dim3(256), 0, st, p);
  }
  GVD_CHECK_LAUNCH();
  return 0;
}
}  // namespace

bool gvd_gemm_pipe_takes_ktail() { return true; }

// ---- measurement hook (bench.py `roofline_mfma`): while armed, every pipelined-GEMM launch is bracketed by an event pair on
// its stream and a one-thread kernel adds the launch's flops - 2 x rows x N x K x batch with the DEVICE-side row count where
// the launch has one - to a device counter.  Process-global and not thread-safe: armed by bench.py for a dedicated
// measurement pass (outside its timed region), never by the product path.
namespace {
gvd_prof* g_gemm_prof = nullptr;
double* g_gemm_flops = nullptr;
int* g_gemm_rows = nullptr;            // optional [n_rows]: live row count of launch i (the device-side count where it has one)
int g_gemm_nrows = 0;

__global__ void gemm_flops_kernel(double* acc, const int* m_dev, int M, double per_row, int* rows_out) {
  const int m = m_dev ? min(*m_dev, M) : M;
  *acc += per_row * (double)m;
  if (rows_out) *rows_out = m;
}
}  // namespace

extern "C" int gvd_gemm_prof_set(gvd_prof* prof, double* dev_flops, int* dev_rows, int n_rows) {
  g_gemm_prof = prof;
  g_gemm_flops = prof ? dev_flops : nullptr;
  g_gemm_rows = (prof && dev_rows && n_rows > 0) ? dev_rows : nullptr;
  g_gemm_nrows = g_gemm_rows ? n_rows : 0;
  return 0;
}

int gvd_gemm_pipe_launch(KParams& p, int batch, hipStream_t st) {
  p.ntm = (p.M + BM - 1) / BM;
  p.ntn = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(p.ntm * p.ntn), (unsigned)batch);
  const bool rb_vec = !p.rowbias || (gvd_aligned16(p.rowbias) && (p.rowbias_ld % 4) == 0 && (p.rowbias_bs % 4) == 0);
  const bool lds_epi = !p.mbias && rb_vec && !p.mask && (p.N % 4) == 0 && (p.ldc % 4) == 0 && (p.cbs % 4) == 0 &&
                       gvd_aligned16(p.C) && (!p.nbias || gvd_aligned16(p.nbias)) && (!p.nbias2 || gvd_aligned16(p.nbias2));
  if (p.a_t || p.w_t) {
    // backward products: one segment, 16-byte aligned K-strided operands, whole 4-column chunks, no device row count
    const bool ok = p.nseg == 1 && !p.m_dev && (!p.a_t || ((p.M % 4) == 0 && p.M >= 4)) &&
                    (!p.w_t || ((p.N % 4) == 0 && p.N >= 4)) && (p.abs_[0] % 4) == 0 && (p.wbs[0] % 4) == 0;
    if (!ok || (p.a_t && !p.w_t)) return GVD_EINVAL;   // (A K-strided with W K-major is not a product the path needs)
  }
  // a last tile at most half full in N (few N tiles) or in M (few M tiles) is worth the sub-tile skip
  const int remn = p.N - (p.ntn - 1) * BN, remm = p.M - (p.ntm - 1) * BM;
  // (round 6, profiles/r06/gemm_edge_b.txt: the skip at ANY tile count - N = 1056, 9 column tiles, the last 32 wide - changes
  // nothing for dX (1237 -> 1229 us) and costs dW 10 %: the partial tile is not what holds those shapes at 0.70)
  const bool edge = (remn <= BN / 2 && p.ntn <= 4) || (!p.m_dev && remm <= BM / 2 && p.ntm <= 4);
  gvd_prof* prof = g_gemm_prof;
  if (prof) {
    double ktot = 0.0;
    for (int s = 0; s < p.nseg; ++s) ktot += (double)p.K[s];
    const int pair = gvd_prof_next(prof);
    if (g_gemm_flops)
      hipLaunchKernelGGL(gemm_flops_kernel, dim3(1), dim3(1), 0, st, g_gemm_flops, p.m_dev, p.M,
                         2.0 * (double)p.N * ktot * (double)batch,
                         (g_gemm_rows && pair >= 0 && pair < g_gemm_nrows) ? g_gemm_rows + pair : (int*)nullptr);
    // what this pair times: rows (upper bound; the live count is in dev_rows), columns, contraction, batch, operand forms
    const int64_t words[GVD_PROF_TAG_WORDS] = {p.M, p.N, (int64_t)ktot, batch, p.a_t, p.w_t, p.m_dev ? 1 : 0, p.rowbias ? 1 : 0};
    gvd_prof_tag(prof, words);
    gvd_prof_begin(prof, st);
  }
  const int rc = edge ? pipe_launch_t<true>(p, grid, lds_epi, st) : pipe_launch_t<false>(p, grid, lds_epi, st);
  if (prof) gvd_prof_end(prof, st);
  return rc;
}
