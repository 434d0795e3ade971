// Fused multi-head self-attention of the `obj_interact` encoder (transformer.py:90-123 as configured at
// model.py:126-135) over PADDED heads: the fused QKV projection (one GEMM against the row-permuted weights
// [wq | wk | wv], att_model.py) writes every head of q, k and v into its own 176-column slot (171 / 169 real columns,
// the rest exactly zero because the matching weight rows are zero), so every head starts on a 16-byte boundary.
// What the layout buys (the first kernels, on the uneven heads in place, sat at 62 % MFMA-busy):
//   * K/V tiles move in 16-byte pieces - 6 loads per lane and key tile instead of 48 + 48 dword accesses with per-element
//     column predicates (pads need no masking: they are zeros in memory; rows past R read as zero through the buffer
//     bound and are masked to -inf);
//   * eight waves (128 queries) share one K/V tile: half the staging work and half the K/V L2 traffic per query;
//   * LDS holds THREE tile buffers with ONE barrier per key tile, placed before the second half of the PV product, so
//     the barrier wait sits under 44 MFMAs (three buffers because that second half still reads the current tile after the
//     barrier: the buffer overwritten in iteration j held tile j-2, whose last reads every wave finished before it
//     arrived at barrier j-1); two waves per SIMD cover the softmax VALU work.
// Arithmetic: 16x16x4 fp32 MFMAs on swapped products (S^T = K Q^T, O^T = V^T P^T), so that a lane holds 8 keys of ONE
// query: lane-local online softmax in the log2 domain, exact skip of the identity rescale, the exp'd score register IS the
// B operand of the PV step.
//
// TRAINING forward (template TRAIN; ops._EncAttnCoreFn): the same kernel over the Rp-padded training layout with
//   * a per-key additive bias (the compacted training layout's key weights, train_compact.py), staged once into LDS,
//   * the dropout of transformer.py:95,104 applied to the probabilities in registers (enc_dropout.h: a counter-based hash
//     of (seed, map row, key) the backward re-evaluates) - the row sum keeps the UNdropped probabilities, as
//     dropout(softmax(s)) @ V requires,
//   * the log2-domain logsumexp of every query written out: the backward (enc_attn_bwd.hip) recomputes
//     P = exp2(s - lse) tile by tile instead of reading a [B, heads, R, R] map the forward would have to write.
#include "gvd_common.h"
#include "enc_dropout.h"
#include "philox.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// K / V tile staging: direct global -> LDS loads (buffer_load_dwordx4 ... lds).  A direct load writes lane l's 16 bytes at
// wave base + 16 l, i.e. LDS is filled in lane order and cannot be padded by the hardware - the 180-float rows (176 + one
// 16-byte pad chunk, which is what keeps the fragment reads conflict-free) are kept anyway by treating the tile as a linear
// array of 16-byte chunks, 45 per row: the lane whose chunk is a row's pad slot (or lies past the 32nd row) issues an
// out-of-range address (reads as zero, no memory traffic).  1536 chunks per operand tile = 8 waves x 3 instructions: the
// instruction count of the register-staged form's loads, without its LDS write pass and its 24 staging registers.  Measured
// against that form (round 4, profiles/r04/flash_glds_ab_a.log: bitwise equal outputs): dense B = 256 9.03 -> 8.71 ms per
// layer, the ragged compacted-preamble shape 7.27 -> 7.02 ms, the training forward 2.37 -> 2.29 ms.

namespace {

constexpr int DP = 176;             // padded head width: 11 MFMA k-blocks / output row tiles of 16
constexpr int LD = 180;             // LDS row stride (floats): 16-byte multiple, conflict-free fragment reads
constexpr int NSB = DP / 16;        // 11
constexpr int NLD = 3;              // direct-load instructions per wave and operand tile (both workgroup shapes)
constexpr int MAXNU = 2;            // 16-key sub-tiles per tile: 1 or 2
constexpr int CPR = LD / 4;         // 45 16-byte chunks per LDS row (the last one is the pad)
// Two shapes of the workgroup (template parameters NW = waves of 16 queries, TK = keys per tile):
//   8 waves x 32-key tiles: 128 queries share a K/V tile, three 49 KB buffers = 147 KB -> ONE workgroup per CU whose two waves
//     per SIMD are phase-skewed around the tile's barrier (the dense-batch shape of rounds 2-3; TRAIN keeps it: its 8 KB key
//     bias would push two of the small workgroups past the 160 KB of a CU);
//   4 waves x 16-key tiles: 64 queries per workgroup, three 24.6 KB buffers = 74 KB -> TWO workgroups per CU.  The two waves of
//     a SIMD now belong to different workgroups: their barriers are independent (one workgroup's barrier wait falls into the
//     other's MFMA stretch by itself), and the tiling wastes less of a ragged sample - 801 rows are 12.5 row tiles of 64 and
//     50.1 key tiles of 16 (13 x 51: 5 % waste) instead of 6.3 of 128 and 25.03 of 32 (7 x 26: 14 %); price: each K/V tile is
//     staged by twice as many workgroups (L2 -> LDS traffic x 2, far below the L2 rate).  Inference uses this shape (round 4,
//     profiles/r04/flash_shape_ab_d.log: dense B = 256 8.63 -> 8.50 ms per layer, ragged 6.95 -> 6.83 ms).
template <int NW, int TK> struct Shape {
  static constexpr int NT = NW * 64;
  static constexpr int NU = TK / 16;                                  // 16-key sub-tiles per tile
  static constexpr int CHUNKS = TK * CPR;                             // 16-byte chunks of one operand tile incl. pads
  static constexpr int NLD = (CHUNKS + NT - 1) / NT;                  // direct-load instructions per wave and operand: 3
  static constexpr int OPSZ = NLD * NT * 4;                           // floats of one operand tile's LDS region
  static constexpr int BUFSZ = 2 * OPSZ;
};

// Reduction over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with the gfx950 lane-swap instructions:
// v_permlane16_swap a, b exchanges the odd rows of a with the even rows of b -> [a0 b0 a2 b2] / [a1 b1 a3 b3]; with
// a = b = x one max (add) of the two registers reduces row pairs, v_permlane32_swap does the same for the 32-lane halves.
// No LDS traffic, no dependent LDS round trips.  Written as inline assembly: through the clang 22 builtin
// (__builtin_amdgcn_permlane16_swap) the second result was dropped when both operands held the same value (the ISA
// showed `v_mov b, a'` right after the swap); the `s_nop 1` is the wait state the hazard recogniser puts between a VALU
// write and these instructions (tools/permlane_probe.hip checks the semantics on the device).
__device__ __forceinline__ void row_swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void row_swap32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float rows_max(float v) {
  float a = v, b = v;
  row_swap16(a, b);
  float m = fmaxf(a, b), n = m;
  row_swap32(m, n);
  return fmaxf(m, n);
}
__device__ __forceinline__ float rows_sum(float v) {
  float a = v, b = v;
  row_swap16(a, b);
  float m = a + b, n = m;
  row_swap32(m, n);
  return m + n;
}

struct PParams {
  const float* q; const float* k; const float* v; float* o;
  int64_t ld, ldo;       // row strides (floats) of q/k/v and of o
  int B, R, n_heads;
  int rstride;           // rows between consecutive samples in q / k / v / o (R at inference; TRAIN: `sample_rows`)
  int mstride;           // TRAIN: entries per (sample, head) of lse / per sample of kbias, and the row pitch of the dropout
                         // hash = the padded query count Rp of the backward maps (csrc/enc_attn_bwd.hip)
  float qscale;          // 1/sqrt(d_model) (a power of two in the reference configuration) times log2(e)
  // ragged (compacted) batches: sample b owns rows off[b] .. off[b+1]-1 of q/k/v/o (at most R of them); its LAST row
  // stands for n identical rows: as a key its score gets + key_w[b] = log2(n) (-inf: no such rows, key ignored)
  const int* off; const float* key_w;
  // ragged batches: tile map built on the device by flash_tile_map_kernel - tmap[0 .. B] = exclusive prefix of the samples'
  // query-tile counts, tmap[B + 1 + s] = the sample of tile slot s.  The LIVE workgroups (tmap[B] * n_heads of them) are
  // the first ones of the grid, in (sample, head, query tile) order; the grid is sized for the longest possible sample.
  const int* tmap;
  // TRAIN only
  const float* kbias;    // nullable [B, mstride]: added to the SCALED scores of a key (natural-log units; -inf = no such key)
  float* lse;            // [B * n_heads, mstride]: log2-domain logsumexp of every query's (scaled, biased) scores
  float* s_out;          // nullable [B * n_heads, mstride, mstride]: the scaled + biased log2-domain scores themselves (rows < R)
  uint32_t thresh, seed_lo, seed_hi;   // dropout: element dropped iff its draw < thresh (0 = no dropout)
  float keep_scale;      // 1 / (1 - p)
};

constexpr int MAX_TRAIN_KEYS = 4096;      // LDS slice of the staged key bias (TRAIN): 3 x 48 KB tile buffers + 16 KB = the CU's 160 KB

template <bool TRAIN, int NW, int TK>
__global__ __launch_bounds__(512, 2) void flash_attn_pad_kernel(const PParams p) {
  using SH = Shape<NW, TK>;
  constexpr int NT = SH::NT, NU = SH::NU, OPSZ = SH::OPSZ, BUFSZ = SH::BUFSZ;
  static_assert(SH::NLD == NLD && NU <= MAXNU, "both shapes stage a tile with 3 direct loads per wave and operand");
  // (Register arrays are declared with the NON-dependent bounds NLD / MAXNU and walked up to NU: an array whose size depends
  // on a template parameter makes every AMDGPU builtin that takes one of its elements a type-dependent expression, and
  // the host-side instantiation of the kernel template - which exists only to emit the launch stub - then fails without a
  // diagnostic and leaves the stub undefined.)
  // [buf][K|V][TK rows of 180 floats, as a linear chunk array] (+ the key bias of the sample, TRAIN: 8 KB)
  __shared__ __attribute__((aligned(16))) float smem[3 * BUFSZ + (TRAIN ? MAX_TRAIN_KEYS : 0)];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c16 = lane & 15, g = lane >> 4;
  // linear workgroup id, XCD-aware: the query tiles of one (sample, head) run on ONE XCD so its K/V slices are fetched
  // into that L2 once
  int qt, h, b;
  if (p.tmap) {
    // Ragged batch: walk the device-built tile map, so that every workgroup past the live count sits at the END of the grid.
    // With the plain (sample, head, tile) decomposition of a grid sized for the longest possible sample, the 2-3 empty
    // workgroups of every (sample, head) are interleaved with the live ones in dispatch order - measured: 6.91 instead of
    // 5.67 ms per layer at 801 rows per sample (profiles/r04/attn_micro_e.log; the dispatcher is in-order and every empty
    // workgroup waits for a 74 KB LDS slot before it can exit).
    const unsigned nlive = (unsigned)p.tmap[p.B] * (unsigned)p.n_heads;
    if (blockIdx.x >= nlive) return;
    const unsigned lid = xcd_remap(blockIdx.x, nlive);
    b = p.tmap[p.B + 1 + lid / p.n_heads];
    const int first = p.tmap[b], nt_b = p.tmap[b + 1] - first;
    const int idx = (int)lid - first * p.n_heads;
    h = idx / nt_b;
    qt = idx - h * nt_b;
  } else {
    const unsigned nqt = (unsigned)((p.R + 16 * NW - 1) / (16 * NW));
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    qt = lid % nqt;
    h = (lid / nqt) % p.n_heads;
    b = lid / (nqt * p.n_heads);
  }
  const int64_t ld = p.ld;
  const int64_t row0 = p.off ? (int64_t)p.off[b] : (int64_t)b * p.rstride;  // first row of this sample
  const int R = p.off ? p.off[b + 1] - p.off[b] : p.R;                       // its row count
  if (qt * (16 * NW) >= R) return;                                           // (ragged: grid sized for the longest sample)
  const int wkey = p.off ? R - 1 : -1;                                       // the weighted key, if any
  const float wval = p.off ? p.key_w[b] : 0.f;
  const unsigned span = (unsigned)((int64_t)R * ld * 4);                     // bytes of one sample's rows: loads past
  __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(            // row R-1 return zeros
      const_cast<float*>(p.q + row0 * ld + h * DP), 0, span - 4u * h * DP, 0x00020000);
  __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.k + row0 * ld + h * DP), 0, span - 4u * h * DP, 0x00020000);
  __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.v + row0 * ld + h * DP), 0, span - 4u * h * DP, 0x00020000);
  const int qrow = qt * (16 * NW) + wave * 16 + c16;
  const unsigned ld4 = (unsigned)ld * 4u;

  // Q as the MFMA B operand: qreg[sb][t] = Q[qrow][16 sb + 4 g + t], pre-multiplied into the log2 domain
  f32x4 qreg[NSB];
#pragma unroll
  for (int sb = 0; sb < NSB; ++sb) {
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, (unsigned)qrow * ld4 + 64u * sb + 16u * g, 0, 0));
    qreg[sb] = v * p.qscale;
  }

  // staging role: instruction i of wave w fills chunks (3 w + i) 64 + lane of the tile's linear chunk array
  unsigned doff[NLD];
  const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int ch = (wv * NLD + i) * 64 + lane;
    const int row = ch / CPR, c4 = ch - row * CPR;
    doff[i] = (row < TK && c4 < DP / 4) ? (unsigned)row * ld4 + 16u * c4 : 0x80000000u;   // pad chunk / past the tile: out of range
  }
  // fetch(key0, buf): tile key0 straight into LDS buffer `buf` (the tile's barrier waits for the loads: vmcnt(0))
  auto fetch = [&](int key0, int buf) {
    const unsigned so = (unsigned)key0 * ld4;
    float* kd = smem + buf * BUFSZ;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (__attribute__((address_space(3))) void*)(kd + (wv * NLD + i) * 256), 16,
                                               doff[i], so, 0, 0);
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(kd + OPSZ + (wv * NLD + i) * 256), 16,
                                               doff[i], so, 0, 0);
  };

  f32x4 oacc[NSB];
#pragma unroll
  for (int dt = 0; dt < NSB; ++dt) oacc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (R + TK - 1) / TK;
  fetch(0, 0);
  float* kb_s = smem + 3 * BUFSZ;           // TRAIN: the sample's key bias in log2 units (0 without one)
  if (TRAIN) {
    const float* kb = p.kbias ? p.kbias + (int64_t)b * p.mstride : nullptr;
    for (int i = tid; i < ntiles * TK; i += NT) kb_s[i] = (kb && i < R) ? kb[i] * 1.4426950408889634f : 0.f;
  }
  __syncthreads();
  int buf = 0;
  // A wave whose 16 queries all lie past the sample's last row (the tail of the last query tile; ragged batches: on
  // average half of that tile) only helps staging the K / V tiles and keeps the barrier count: it issues no MFMA, so its
  // SIMD's matrix pipe goes to the other resident waves.
  if (qt * (16 * NW) + wave * 16 >= R) {
#pragma unroll 1
    for (int jt = 0; jt < ntiles; ++jt) {
      const int nxt = buf == 2 ? 0 : buf + 1;
      if (jt + 1 < ntiles) fetch((jt + 1) * TK, nxt);       // (tile jt-2's buffer: every wave left it before the previous barrier)
      __syncthreads();
      buf = nxt;
    }
    return;
  }
  const bool skew = NW == 8 && wave >= NW / 2;     // wave-uniform (4-wave shape: the SIMD partner is another workgroup)
  // TRAIN: this lane's query in the dropout hash (enc_dropout.h): row id = (sample * heads + head) * mstride + query
  const gvd_encdrop_key dkey = TRAIN ? gvd_encdrop_row((uint32_t)(b * p.n_heads + h) * (uint32_t)p.mstride + (uint32_t)qrow, p.seed_lo, p.seed_hi) : gvd_encdrop_key{0u, 0u};
  const bool drop = TRAIN && p.thresh != 0u;       // wave-uniform
  const bool biased = TRAIN && p.kbias != nullptr;
  // first K fragments of the next tile, read right after the barrier that publishes it (under the last 44 MFMAs)
  f32x4 kpre[MAXNU];
#pragma unroll
  for (int u = 0; u < NU; ++u) kpre[u] = *reinterpret_cast<const f32x4*>(smem + (c16 + 16 * u) * LD + 4 * g);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int jt = 0; jt < ntiles; ++jt) {
    const int key0 = jt * TK;
    const bool more = jt + 1 < ntiles;                                     // wave-uniform
    const int nxt = buf == 2 ? 0 : buf + 1;
    // tile jt+1 flies under this tile's MFMAs, straight into the buffer that held tile jt-2 (which every wave left before
    // the previous barrier)
    if (more) fetch(key0 + TK, nxt);
    const float* sk = smem + buf * BUFSZ;
    const float* sv = sk + OPSZ;
    // Publishing tile jt+1 = barrier_next: the tile's one barrier (the compiler puts the s_waitcnt vmcnt(0) for this wave's
    // direct loads in front of it) + the first K fragments of tile jt+1.
    // PHASE SKEW: waves 4..7 publish right after their softmax (before the PV product), waves 0..3 in the middle of the
    // PV product, so that per SIMD (waves w and w + 4) one wave's softmax falls into the other's MFMA stretch (8.89 vs
    // 9.05 ms).  The buffer protocol only depends on the barrier order: every wave passes exactly one barrier per tile,
    // after its last read of tile jt-1's buffer and before its first read of tile jt+1's.
    auto barrier_next = [&]() {
      __syncthreads();
      const float* nk = smem + nxt * BUFSZ + c16 * LD + 4 * g;               // (stale but harmless after the last tile)
#pragma unroll
      for (int u = 0; u < NU; ++u) kpre[u] = *reinterpret_cast<const f32x4*>(nk + 16 * u * LD);
      __builtin_amdgcn_sched_barrier(0);
    };

    // ---- S^T for the NU 16-key sub-tiles.  With two sub-tiles their accumulator chains alternate (16x16x4: 40-cycle
    // dependent latency against a 32-cycle issue interval; with one, the SIMD's other wave fills the gaps); the K fragments
    // of k-block sb+1 are read while block sb multiplies (the compiler on its own reads each fragment right before its
    // use and stalls on it: that, not the softmax, is what held the first kernels at 62 % matrix-pipe utilisation).
    f32x4 sacc[MAXNU];
#pragma unroll
    for (int u = 0; u < NU; ++u) sacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* kp0 = sk + c16 * LD + 4 * g;
    f32x4 ka[2][MAXNU];
#pragma unroll
    for (int u = 0; u < NU; ++u) ka[0][u] = kpre[u];
#pragma unroll
    for (int sb = 0; sb < NSB; ++sb) {
      if (sb + 1 < NSB) {
#pragma unroll
        for (int u = 0; u < NU; ++u) ka[(sb + 1) & 1][u] = *reinterpret_cast<const f32x4*>(kp0 + 16 * u * LD + 16 * (sb + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u)
          sacc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[sb & 1][u][t], qreg[sb][t], sacc[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // V fragments of PV step 0 (key 4 g of sub-tile 0) do not depend on the softmax: fetch them under it
    float vf[2][NSB];
    auto vload = [&](float (&dst)[NSB], int step) {          // step = 4 u + s4 -> key 16 u + 4 g + s4
      const float* vp = sv + (16 * (step >> 2) + 4 * g + (step & 3)) * LD + c16;
#pragma unroll
      for (int dt = 0; dt < NSB; ++dt) dst[dt] = vp[16 * dt];
    };
    vload(vf[0], 0);
    // ---- online softmax: this lane holds keys 16 u + 4 g + reg of its query
    float mt = -INFINITY;
    if (biased) {                                  // per-key bias (log2 units; -inf removes the key)
#pragma unroll
      for (int u = 0; u < NU; ++u) sacc[u] += *reinterpret_cast<const f32x4*>(kb_s + key0 + 16 * u + 4 * g);
    }
    if (TRAIN && p.s_out && qrow < R) {
      // the scores as the backward maps kernel will read them (csrc/enc_attn_bwd.hip loads them instead of multiplying Q K^T
      // again): 16 bytes per lane and sub-tile, 64 contiguous bytes per query row and instruction; read once, by a later launch
      float* sp = p.s_out + ((int64_t)(b * p.n_heads + h) * p.mstride + qrow) * p.mstride + key0 + 4 * g;
#pragma unroll
      for (int u = 0; u < NU; ++u) __builtin_nontemporal_store(sacc[u], reinterpret_cast<f32x4*>(sp + 16 * u));
    }
    // masks only where they can apply (wave-uniform): the sample's last key tile (keys past R, the weighted key)
    if (key0 + TK > R || (wkey >= key0 && wkey < key0 + TK)) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + 16 * u + 4 * g + r;
          if (key >= R) sacc[u][r] = -INFINITY;
          else if (key == wkey) sacc[u][r] += wval;        // n identical keys = one key with n times the weight
        }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) mt = fmaxf(mt, sacc[u][r]);
    // max over the four 16-lane rows (g): gfx950 lane-swap instructions instead of two dependent LDS-permute round trips
    mt = rows_max(mt);
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sacc[u][r] = __builtin_amdgcn_exp2f(sacc[u][r] - m_new);
        psum += sacc[u][r];
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < NSB; ++dt) oacc[dt] *= alpha;
    }
    if (drop) {                                    // dropout on the probabilities that enter the PV product only
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          sacc[u][r] = gvd_encdrop_keep(dkey, (uint32_t)(key0 + 16 * u + 4 * g + r), p.thresh) ? sacc[u][r] * p.keep_scale : 0.f;
    }
    // ---- O^T += V^T P^T: step = 4 u + s4 contracts key 16 u + 4 g + s4 = score register s4 of sub-tile u; the V
    // fragments of step+1 are read while step multiplies.  In the middle of the product: the tile's only barrier (the reads
    // of the next step are already in flight; the second half still reads `buf`).
    if (skew) barrier_next();
#pragma unroll
    for (int step = 0; step < 4 * NU; ++step) {
      if (step + 1 < 4 * NU) vload(vf[(step + 1) & 1], step + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int dt = 0; dt < NSB; ++dt)
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[step & 1][dt], sacc[step >> 2][step & 3], oacc[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (step == 2 * NU - 1 && !skew) barrier_next();
    }
    buf = nxt;
  }

  const float l_tot = rows_sum(l_run);
  const float inv = 1.0f / l_tot;
  if (TRAIN && qrow < R && g == 0)
    p.lse[(int64_t)(b * p.n_heads + h) * p.mstride + qrow] = m_run + __builtin_amdgcn_logf(l_tot);   // (v_log_f32 = log2)
  if (qrow < R) {
    float* orow = p.o + (row0 + qrow) * p.ldo + h * DP + 4 * g;
#pragma unroll
    for (int dt = 0; dt < NSB; ++dt) *reinterpret_cast<f32x4*>(orow + 16 * dt) = oacc[dt] * inv;
  }
}

// Tile map of a ragged batch (one workgroup): per sample its number of query tiles of `rows_per_tile` rows, the exclusive
// prefix over the samples, and the sample of every tile slot.  B <= a few thousand: a serial scan by one thread is ~2 us.
__global__ void flash_tile_map_kernel(const int* __restrict__ off, int B, int rows_per_tile, int* __restrict__ tmap) {
  __shared__ int s_total;
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int b = 0; b < B; ++b) {
      tmap[b] = acc;
      acc += (off[b + 1] - off[b] + rows_per_tile - 1) / rows_per_tile;
    }
    tmap[B] = acc;
    s_total = acc;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const int first = tmap[b], last = b + 1 < B ? tmap[b + 1] : s_total;
    for (int s = first; s < last; ++s) tmap[B + 1 + s] = b;
  }
}

}  // namespace

extern "C" size_t gvd_flash_attn_workspace_bytes(int B, int R) {
  // ragged batches: (B + 1) prefix entries + one sample id per tile slot of 64 rows
  return sizeof(int) * ((size_t)B + 1 + (size_t)B * (size_t)((R + 63) / 64));
}

extern "C" int gvd_flash_attn_padded_f32(const float* q, const float* k, const float* v, int64_t ld, float* o, int64_t ldo,
                                         int B, int R, int n_heads, int head_pad, float scale, const int* row_off,
                                         const float* last_key_log2_weight, void* workspace, gvd_stream_t stream) {
  if (!q || !k || !v || !o || B <= 0 || R <= 0 || n_heads <= 0 || head_pad != DP || (ld % 4) != 0 || (ldo % 4) != 0 ||
      !gvd_aligned16(q) || !gvd_aligned16(k) || !gvd_aligned16(v) || !gvd_aligned16(o) || ld < (int64_t)n_heads * DP ||
      ldo < (int64_t)n_heads * DP || (int64_t)R * ld * 4 >= (int64_t)1 << 31 || (row_off && !last_key_log2_weight) ||
      (row_off && (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 3u))))
    return GVD_EINVAL;
  PParams p = {};
  p.q = q; p.k = k; p.v = v; p.o = o; p.ld = ld; p.ldo = ldo; p.B = B; p.R = R; p.n_heads = n_heads; p.rstride = R; p.mstride = R;
  p.qscale = 1.4426950408889634f * scale;
  p.off = row_off; p.key_w = last_key_log2_weight;
  if (row_off) {
    p.tmap = reinterpret_cast<int*>(workspace);
    hipLaunchKernelGGL(flash_tile_map_kernel, dim3(1), dim3(256), 0, gvd_s(stream), row_off, B, 64, reinterpret_cast<int*>(workspace));
    GVD_CHECK_LAUNCH();
  }
  // inference: the 4-wave / 16-key shape (measured against 8 x 32, round 4 session D: dense B = 256 8.63 -> 8.50 ms per layer,
  // the ragged compacted shape 6.95 -> 6.83 ms, the batch_size = 4 call unchanged)
  const unsigned nwg = (unsigned)((R + 63) / 64) * n_heads * B;
  hipLaunchKernelGGL((flash_attn_pad_kernel<false, 4, 16>), dim3(nwg), dim3(256), 0, gvd_s(stream), p);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_flash_attn_train_fwd_f32(const float* qkv, int64_t ld, float* o, int64_t ldo, float* lse, float* scores_out,
                                            int B, int Rp, int R, int sample_rows, int n_heads, int head_pad, float scale,
                                            const float* key_bias, float p_drop, uint64_t seed, gvd_stream_t stream) {
  if (!qkv || !o || !lse || B <= 0 || R <= 0 || Rp < R || sample_rows < R || Rp > MAX_TRAIN_KEYS || (Rp % 32) != 0 || n_heads <= 0 ||
      head_pad != DP || (ld % 4) != 0 || (ldo % 4) != 0 || !gvd_aligned16(qkv) || !gvd_aligned16(o) ||
      ld < (int64_t)3 * n_heads * DP || ldo < (int64_t)n_heads * DP || (int64_t)R * ld * 4 >= (int64_t)1 << 31 ||
      !(p_drop >= 0.f) || !(p_drop < 1.f) || (key_bias && !gvd_aligned16(key_bias)) ||
      (scores_out && !gvd_aligned16(scores_out)) || (int64_t)B * n_heads * Rp >= (int64_t)1 << 32)
    return GVD_EINVAL;
  PParams p = {};
  p.q = qkv; p.k = qkv + (int64_t)n_heads * DP; p.v = qkv + (int64_t)2 * n_heads * DP; p.o = o;
  p.ld = ld; p.ldo = ldo; p.B = B; p.R = R; p.n_heads = n_heads; p.rstride = sample_rows; p.mstride = Rp;
  p.qscale = 1.4426950408889634f * scale;
  p.kbias = key_bias; p.lse = lse; p.s_out = scores_out;
  p.thresh = p_drop > 0.f ? gvd_drop_thresh(p_drop) : 0u;
  p.keep_scale = 1.0f / (1.0f - p_drop);
  p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
  const unsigned nwg = (unsigned)((R + 127) / 128) * n_heads * B;
  hipLaunchKernelGGL((flash_attn_pad_kernel<true, 8, 32>), dim3(nwg), dim3(512), 0, gvd_s(stream), p);
  GVD_CHECK_LAUNCH();
  return 0;
}
