// K-split fp32-MFMA GEMM for the two LSTM cells of the token loop (gate product + fused cell epilogue) at SMALL batches: 17 ..
// 128 rows - a training step at batch_size = 64, a decode batch of 32 or 96.
//
// gemm_small.hip gives such a launch one 64 x 64 tile per workgroup: at 64 rows x 4096 gate columns that is 64 workgroups on
// 256 CUs, one wave per SIMD on a quarter of the chip, and every wave issues its 1792 MFMAs back to back - the launch is
// MFMA-ISSUE-bound per wave (56 k tiles x 2048 cycles = 47.8 us, measured 47.8) while three quarters of the matrix pipes idle.
// Here a workgroup owns a 32 x 32 tile and its four waves split the CONTRACTION: wave w multiplies the w-th quarter of K for
// the whole tile (448 MFMAs at K = 3584: 12 us), the four partial tiles are summed through LDS in a fixed order (w = 0, 1, 2,
// 3: deterministic), then the LSTM-cell epilogue runs on the summed tile.  4096 gate columns x 64 rows = 256 workgroups: every
// CU works.  Each wave runs its OWN pipeline - no workgroup barrier in the K loop: groups of 32 k (one 128-byte line per operand
// row) go global -> LDS by direct 16-byte loads into a wave-private ring of four 8 KB slots (eight lanes fetch one whole line;
// unpadded XOR-swizzled rows as in gemm_pipe.hip), completion is tracked with explicit s_waitcnt vmcnt counts (loads return in
// order: a group is ready when at most the loads of the younger groups are outstanding), fragments are read one group ahead
// under the 16 MFMAs of the current one.  A first form loaded the fragments straight into registers in the MFMA layout
// (lane = its own row: 64 cache-line requests per load instruction): 28 us per cell at 64 rows; this form 21 us (launch to
// launch, tools/gemm_ks_micro.py), gemm_small.hip 48 - 54; at 65 .. 128 rows 37 - 39 against 54 (two workgroups per CU in turn).
//
// Numerics: the k order per output is NOT the ascending order of the other GEMM kernels (four quarter sums added in order);
// the result of a row does not depend on the batch it travels in (tile-local arithmetic), but a batch of <= 128 rows and one
// of > 128 rows differ in the last bits like any two fp32 summation orders do (the skinny kernel of <= 16 rows has its own
// order too).  That is why only the cells use this kernel: the plain products of gvd_gemm_nt_f32 all add an output's k terms in
// the same ascending order, so that the preamble's results do not depend on the batch size (see gemm_f32.hip).
#include "gemm_common.h"

namespace {

constexpr int KT = 32;                 // tile rows = tile columns
constexpr int KNS = 4;                 // ring slots per wave (groups of 32 k in flight)
constexpr int KLDP = KT + 1;           // padded LDS row of a partial tile
constexpr int KHU = KT / 4;            // LSTM: hidden units per tile (4 gates x 8)
constexpr int KSLOT = 2 * KT * 32;     // floats of one ring slot: [A | W][32 rows][32 k], unpadded XOR-swizzled 128-byte rows

#define GVD_KS_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ __launch_bounds__(256, 1) void gemm_ks_lstm_kernel(const KParams p) {
  __shared__ __attribute__((aligned(16))) float ring[4 * KNS * KSLOT];      // 131,072 B: wave-private rings, then the partial tiles
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 31, half = lane >> 5;
  // consecutive ids walk the (<= 4) row tiles of one weight panel: they share it in one XCD's L2
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm_ = lid % p.ntm, tn_ = lid / p.ntm;
  const int m0 = tm_ * KT;
  const int M = p.M;

  // ---- this wave's quarter of the concatenated K axis in groups of 32 k; (seg, kpos) = position of the next group to FETCH
  int ktot = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s)
    if (s < p.nseg) ktot += p.K[s];
  const int ngr = ktot / 128;                          // groups per wave (every segment is a multiple of 128: see _ok)
  int seg = 0, kpos = wave * ngr * 32;
  while (seg + 1 < p.nseg && kpos >= p.K[seg]) { kpos -= p.K[seg]; ++seg; }
  // Staging role (direct global -> LDS loads, 16 bytes per lane): instruction i of a group covers tile rows 8 i .. 8 i + 7, lane
  // l = row 8 i + (l >> 3), LDS slot l & 7 of that row, which holds the 16-byte k chunk (l & 7) ^ (row & 7) - eight lanes fetch
  // ONE whole 128-byte line (four tag lookups per instruction instead of the 64 of a load in the MFMA layout, where every lane
  // touches its own row: measured 28 us per cell at 64 rows against 8.5 us of MFMAs - the vector cache's request rate).
  const int srow = lane >> 3, chunk = (lane & 7) ^ srow;
  __amdgpu_buffer_rsrc_t ra, rw;
  unsigned voa[4], vow[4];
  int kend;
  auto seg_setup = [&](int s) {
    ra = gvd_rsrc(p.A[s]);
    rw = gvd_rsrc(p.W[s]);
    const unsigned lda4 = (unsigned)p.lda[s] * 4u, ldw4 = (unsigned)p.ldw[s] * 4u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      voa[i] = (unsigned)min(m0 + 8 * i + srow, M - 1) * lda4 + 16u * (unsigned)chunk;
      // tile column 8 i + srow = gate i of hidden unit 8 tn + srow
      vow[i] = (unsigned)(i * p.H + tn_ * KHU + srow) * ldw4 + 16u * (unsigned)chunk;
    }
    kend = p.K[s];
  };
  seg_setup(seg);
  float* mine = ring + wave * (KNS * KSLOT);
  auto dma = [&](int slot) {
    const unsigned so = 4u * (unsigned)kpos;
    float* dst = mine + slot * KSLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(dst + i * 256), 16, voa[i], so, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(dst + KT * 32 + i * 256), 16, vow[i], so, 0, 0);
    kpos += 32;
    if (kpos == kend && seg + 1 < p.nseg) {            // wave-uniform
      ++seg;
      kpos = 0;
      seg_setup(seg);
    }
  };
  // fragments of a slot: lane (r, half) takes, for MFMA steps 4 j .. 4 j + 3, the chunk 2 j + half of row r = slot
  // (2 j + half) ^ (r & 7) of that row
  const float* frA = mine + r * 32;
  const int rsw = r & 7;
  auto frags = [&](f32x4 (&a)[4], f32x4 (&b)[4], int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = ((2 * j + half) ^ rsw) * 4;
      a[j] = *reinterpret_cast<const f32x4*>(frA + slot * KSLOT + o);
      b[j] = *reinterpret_cast<const f32x4*>(frA + slot * KSLOT + KT * 32 + o);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  auto mfma16 = [&](const f32x4 (&a)[4], const f32x4 (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][t], b[j][t], acc, 0, 0, 0);
  };
#pragma unroll
  for (int g = 0; g < KNS; ++g) dma(g);                // (ngr >= KNS: _ok)
  // Group q sits in slot q % KNS.  Before its fragments are read, its 8 loads must have landed: loads complete in order, so it
  // is enough that at most the loads of the YOUNGER groups in flight (8 each, at most KNS - 1 groups) are outstanding.
  auto wait_for = [&](int q) {
    const int younger = min(KNS - 1, ngr - 1 - q);     // wave-uniform
    if (younger >= 3) GVD_KS_WAIT_VM(24);
    else if (younger == 2) GVD_KS_WAIT_VM(16);
    else if (younger == 1) GVD_KS_WAIT_VM(8);
    else GVD_KS_WAIT_VM(0);
  };
  // one group: (cur) holds the fragments of group q.  Refill q's slot with group q + KNS (its fragments are in registers: the
  // previous step ended with lgkmcnt(0)), read the fragments of group q + 1 into (nxt) under the 16 MFMAs of group q.
  auto step = [&](int q, const f32x4 (&ca)[4], const f32x4 (&cb)[4], f32x4 (&na)[4], f32x4 (&nb)[4]) {
    if (q + KNS < ngr) dma(q & (KNS - 1));
    if (q + 1 < ngr) {
      wait_for(q + 1);
      frags(na, nb, (q + 1) & (KNS - 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma16(ca, cb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x4 a0[4], b0[4], a1[4], b1[4];
  wait_for(0);
  frags(a0, b0, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  int q = 0;
#pragma unroll 1
  for (; q + 1 < ngr; q += 2) {
    step(q, a0, b0, a1, b1);
    step(q + 1, a1, b1, a0, b0);
  }
  if (q < ngr) step(q, a0, b0, a1, b1);

  // ---- the four quarter sums -> LDS (each wave into its OWN ring region: no other wave reads or writes it), added in the order
  // w = 0, 1, 2, 3
  float* pw = mine;
#pragma unroll
  for (int e = 0; e < 16; ++e) pw[((e & 3) + 8 * (e >> 2) + 4 * half) * KLDP + r] = acc[e];
  __syncthreads();
  // ---- LSTM cell epilogue (nn.LSTMCell, AttModel.py:139,160): thread = (row, hidden unit) of the 32 x 8 cell tile; the tile's
  // columns are grouped i | f | g | o, 8 units each
  const int ml = tid >> 3, jl = tid & 7;
  const int gm = m0 + ml;
  if (gm >= M) return;
  const int j = tn_ * KHU + jl;
  float g4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int nl = q * KHU + jl;
    const int wr = q * p.H + j;
    float nb = 0.f;
    if (p.nbias) nb += p.nbias[wr];
    if (p.nbias2) nb += p.nbias2[wr];
    const float* pp = ring + ml * KLDP + nl;
    float v = ((pp[0] + pp[KNS * KSLOT]) + pp[2 * KNS * KSLOT]) + pp[3 * KNS * KSLOT];
    v += nb;
    if (p.rowbias) v += p.rowbias[(int64_t)gm * p.rowbias_ld + wr];
    g4[q] = v;
  }
  const float gi = sigmoid_f(g4[0]);
  const float gf = sigmoid_f(g4[1]);
  const float gg = tanhf(g4[2]);
  const float go = sigmoid_f(g4[3]);
  const float c = gf * p.c_prev[(int64_t)gm * p.ldcp + j] + gi * gg;
  p.c_out[(int64_t)gm * p.ldco + j] = c;
  p.h_out[(int64_t)gm * p.ldh + j] = go * tanhf(c);
  if (p.gates_out) {
    float* g = p.gates_out + (int64_t)gm * p.ldg;
    g[j] = gi; g[p.H + j] = gf; g[2 * p.H + j] = gg; g[3 * p.H + j] = go;
  }
}

}  // namespace

// eligibility: 17 .. 128 rows, plain operands, one batch entry, every K segment a multiple of 128 and every operand row
// 16-byte aligned (the 16-byte loads start at multiples of 4 floats)
bool gvd_gemm_ks_ok(const KParams& p, int batch) {
  if (batch != 1 || p.a_t || p.w_t || p.m_dev || p.a_rmap || p.M < 17 || p.M > 128) return false;
  int ktot = 0;
  for (int s = 0; s < p.nseg; ++s) {
    ktot += p.K[s];
    if ((p.K[s] % 128) || (p.lda[s] % 4) || (p.ldw[s] % 4) || !gvd_aligned16(p.A[s]) || !gvd_aligned16(p.W[s])) return false;
  }
  return ktot >= 4 * KNS * 32;                         // every wave's quarter fills its ring
}

int gvd_gemm_ks_lstm_launch(KParams& p, hipStream_t st) {
  p.ntm = (p.M + KT - 1) / KT;
  p.ntn = p.H / KHU;
  hipLaunchKernelGGL(gemm_ks_lstm_kernel, dim3((unsigned)(p.ntm * p.ntn)), dim3(256), 0, st, p);
  GVD_CHECK_LAUNCH();
  return 0;
}
