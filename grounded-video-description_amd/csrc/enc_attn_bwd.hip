// Backward of the encoder's self-attention core on the TRAINING path (transformer.py:90-117: softmax(Q K^T / sqrt d) ->
// dropout(0.2) -> @ V), after a flash-style forward (flash_attn_pad.hip, TRAIN) that kept only the log2-domain
// logsumexp of every query.
//
//   delta[q] = sum_d dO[q,d] O[q,d]                       (= sum_k dP[q,k] P[q,k]: the softmax backward's row term)
//   per (sample, head), per 128 x 128 tile of (queries, keys) - ONE kernel, two fp32-MFMA products into two accumulator
//   sets over the same LDS pipeline:
//       S  = Q K^T          P  = exp2(c S + bias2[k] - lse2[q])          (the forward's probabilities, recomputed)
//       dY = dO V^T         keep from the dropout hash of (seed, map row, key)            (enc_dropout.h)
//       Pd = keep ? P / (1-p) : 0                      -> map 1: the K-strided A operand of  dV = Pd^T dO
//       dS = scale P ((keep ? dY / (1-p) : 0) - delta[q])  -> map 2: operand of  dQ = dS K  and  dK = dS^T Q
//   (the three N = 176 products run on the one-head-slot GEMM, gemm_n192.hip).
// Rows / columns >= R of both maps are written as zeros: the consumers contract over the whole padded Rp.
// Round 6: when the forward handed over its scores (`scores`: the log2-domain c S + bias as its matrix cores produced it, 1.6
// GB per layer at batch_size = 64), the kernel LOADS them into the first accumulator set instead of multiplying Q K^T again:
// one product per tile, 2.65 -> 1.72 ms per launch (the kernel is MFMA-bound, HBM has the slack), and the backward's P is
// the forward's to the last bit.
//
// What this replaces (round 3): S and dY as two GEMM launches writing [B, heads, Rp, Rp] maps, a softmax + dropout row
// kernel forward (read S, write Y and Pd) and one backward (read dY, Pd, Y, write dS) - six trips of a 1.6 GB map through
// HBM per layer and direction at batch_size = 64; now Pd and dS are written once and read once / twice, and nothing is
// kept from the forward but [B, heads, Rp] statistics.
//
// Pipeline = gemm_pipe.hip's direct-to-LDS form (buffer_load ... lds into unpadded, XOR-swizzled 32-float rows, one barrier
// per k tile placed before the tile's last quarter, double-buffered fragments), specialised: one segment, K = head_pad =
// 176 = 5 k tiles + the shifted 16-column tail tile, the second product's first tile is fetched under the first product's
// tail.  Same ascending k order per output as every other GEMM kernel of the library.
#include "gemm_common.h"
#include "enc_dropout.h"
#include "philox.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int NLD = 4;
constexpr int TILE = BM * BK;              // floats of one operand tile (unpadded)

struct MapParams {
  const float* qkv; int64_t ld;            // packed q | k | v, [B, Rp, 3 * nh * HP]
  const float* dO; int64_t ldo;            // [B, Rp, nh * HP]
  const float* lse2;                       // [B * nh, Rp]
  const float* delta;                      // [B * nh, Rp]
  const float* kbias;                      // nullable [B, Rp] (natural-log units)
  const float* scores;                     // HAVE_S: [B * nh, Rp, Rp] log2-domain scaled + biased scores of the forward (rows < R written)
  float* Pd; float* dS;                    // [B * nh, Rp, Rp]
  int B, Rp, R, Rs, nh, HP;                // Rs: rows between consecutive samples in qkv / dO (>= R; Rp on the padded layout)
  float scale, c2, keep_scale;
  uint32_t thresh, seed_lo, seed_hi;
  int ntk;
};

// Round 6 (profiles/r06/maps_forms_e.txt, maps_ablate_f.txt): the epilogue is branch-free, reads lse / delta / the dropout row
// key (hashed once per query row by one thread) as one 16-byte LDS read per row, and stores both maps row by row straight from
// the accumulator layout - 2.93 -> 2.67 ms per launch at batch_size = 64, bit for bit the maps of the LDS-transposed form of
// rounds 4-5; the products alone take 2.35 ms.  A workgroup that walks the key tiles of its query tile (next tile's operands
// in flight during the epilogue) measured 2.80 ms and was dropped.
// HAVE_S (round 6): the forward handed over its scores (flash_attn_pad.hip, TRAIN, `s_out`): they are LOADED into the first
// accumulator set - straight in the accumulator layout, whole cache lines per instruction, in flight under the one remaining
// product dY = dO V^T - instead of multiplying Q K^T again.
template <bool HAVE_S>
__global__ __launch_bounds__(256, 2) void enc_attn_bwd_maps_kernel(const MapParams p) {
  __shared__ __attribute__((aligned(16))) float smem[4 * TILE + 640];      // operand buffers, per-row / per-key vectors
  float* As = smem;                       // [2][128][32]
  float* Ws = smem + 2 * TILE;            // [2][128][32]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tk_ = lid % p.ntk, tq_ = lid / p.ntk;
  const int bh = blockIdx.y, b = bh / p.nh, h = bh - b * p.nh;
  const int m0 = tq_ * BM, n0 = tk_ * BN;
  const int rb = wm * 64, cb = wn * 64;
  const int R = p.R, HP = p.HP;

  const int srow = tid >> 3, kq = tid & 7;
  const int kq_sw = kq ^ (srow & 7);
  const float* base = p.qkv + (int64_t)b * p.Rs * p.ld + (int64_t)h * HP;
  const float* dob = p.dO + (int64_t)b * p.Rs * p.ldo + (int64_t)h * HP;
  const unsigned ld4 = (unsigned)p.ld * 4u, ldo4 = (unsigned)p.ldo * 4u;
  unsigned vq[NLD], vk[NLD], vd[NLD];      // row offsets: Q / dO rows (queries), K / V rows (keys); edge rows clamped
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int qr = min(m0 + srow + 32 * i, R - 1), kr = min(n0 + srow + 32 * i, R - 1);
    vq[i] = (unsigned)qr * ld4 + 16u * kq_sw;
    vd[i] = (unsigned)qr * ldo4 + 16u * kq_sw;
    vk[i] = (unsigned)kr * ld4 + 16u * kq_sw;
  }
  const __amdgpu_buffer_rsrc_t rQ = gvd_rsrc(base), rK = gvd_rsrc(base + (int64_t)p.nh * HP),
                               rV = gvd_rsrc(base + (int64_t)2 * p.nh * HP), rD = gvd_rsrc(dob);
  const int nfull = HP / BK;               // 5
  const bool tail = (HP % BK) != 0;        // 16 more columns: the shifted tail tile
  const int wv = __builtin_amdgcn_readfirstlane(wave);

  // direct global -> LDS loads of one k tile (columns kcol .. kcol + 31) of an (A, W) operand pair into buffer `buf`
  auto dma = [&](const __amdgpu_buffer_rsrc_t& ra, const unsigned (&va)[NLD], const __amdgpu_buffer_rsrc_t& rw, int kcol, int buf) {
    const unsigned so = 4u * (unsigned)kcol;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)&As[buf * TILE + (8 * wv + 32 * i) * BK],
                                               16, va[i], so, 0, 0);
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)&Ws[buf * TILE + (8 * wv + 32 * i) * BK],
                                               16, vk[i], so, 0, 0);
  };
  const float* Adm = &As[(rb + r) * BK];
  const float* Wdm = &Ws[(cb + r) * BK];
  const int rsw = r & 7;
  auto frags = [&](f32x4 (&a)[2], f32x4 (&w)[2], int buf, int q) {
    const int so4 = ((2 * q + half) ^ rsw) * 4;
    a[0] = *reinterpret_cast<const f32x4*>(Adm + buf * TILE + so4);
    a[1] = *reinterpret_cast<const f32x4*>(Adm + buf * TILE + 32 * BK + so4);
    w[0] = *reinterpret_cast<const f32x4*>(Wdm + buf * TILE + so4);
    w[1] = *reinterpret_cast<const f32x4*>(Wdm + buf * TILE + 32 * BK + so4);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma16 = [&](f32x16 (&acc)[2][2], const f32x4 (&a)[2], const f32x4 (&w)[2]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], w[j][t], acc[i][j], 0, 0, 0);
  };

  f32x16 accS[2][2], accP[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { accS[i][j][e] = 0.f; accP[i][j][e] = 0.f; }

  f32x4 a0[2], w0[2], a1[2], w1[2];
  int buf = 0;
  // One product = nfull whole k tiles + (tail) the tile at columns HP-32 .. HP-1 of which only the last two quarters are
  // new.  On entry tile 0 of the product sits in `buf` and (a0, w0) hold its first quarter; on exit the same holds for the
  // NEXT product (whose first tile was fetched under this product's last one) when there is one.
  auto product = [&](f32x16 (&acc)[2][2], const __amdgpu_buffer_rsrc_t& ra, const unsigned (&va)[NLD],
                     const __amdgpu_buffer_rsrc_t& rw, bool more, const __amdgpu_buffer_rsrc_t& na,
                     const unsigned (&nva)[NLD], const __amdgpu_buffer_rsrc_t& nw) {
    const int nt = nfull + (tail ? 1 : 0);
#pragma unroll 1
    for (int kt = 0; kt < nt; ++kt) {
      const bool last = kt + 1 == nt;
      const bool is_tail = tail && last;
      if (!last) dma(ra, va, rw, (tail && kt + 2 == nt) ? HP - BK : (kt + 1) * BK, buf ^ 1);
      else if (more) dma(na, nva, nw, 0, buf ^ 1);
      if (!is_tail) {
        frags(a1, w1, buf, 1);
        mfma16(acc, a0, w0);
        frags(a0, w0, buf, 2);
        mfma16(acc, a1, w1);
      }
      frags(a1, w1, buf, 3);
      mfma16(acc, a0, w0);
      __syncthreads();                                   // (vmcnt(0) + barrier: the next tile has landed in buf ^ 1)
      if (!last) frags(a0, w0, buf ^ 1, (tail && kt + 2 == nt) ? 2 : 0);
      else if (more) frags(a0, w0, buf ^ 1, 0);
      mfma16(acc, a1, w1);
      buf ^= 1;
    }
  };
  if (HAVE_S) dma(rD, vd, rV, 0, 0);
  else dma(rQ, vq, rK, 0, 0);
  // The epilogue needs lse / delta of the tile's 128 query rows and the bias of its 128 key columns: one coalesced element
  // per thread, fetched under the first tile's load latency and parked in the 4 KB of LDS between the operand buffers and
  // the end of the epilogue slices (64 dependent scalar loads per thread in the epilogue cost 0.24 of the kernel's 3.4 ms:
  // ablation timings, profiles/r04/bwd_maps_ablate_b.log).
  float* rowv = smem + 4 * TILE;               // [128][lse, delta, dropout row key (2 words)]: one 16-byte read per row in the epilogue
  float* colv = rowv + 512;                    // [128] key bias (log2 units)
  const bool drop = p.thresh != 0u;
  if (tid < 128) {
    const int64_t mrow = (int64_t)bh * p.Rp;
    const int q = m0 + tid;
    f32x4 v = {1e30f, 0.f, 0.f, 0.f};          // (rows >= R: P = exp2(. - 1e30) = 0)
    if (q < R) {
      v[0] = p.lse2[mrow + q];
      v[1] = p.delta[mrow + q];
      if (drop) {
        const gvd_encdrop_key dk = gvd_encdrop_row((uint32_t)bh * (uint32_t)p.Rp + (uint32_t)q, p.seed_lo, p.seed_hi);
        v[2] = __uint_as_float(dk.add);
        v[3] = __uint_as_float(dk.flip);
      }
    }
    *reinterpret_cast<f32x4*>(&rowv[4 * tid]) = v;
  } else {
    const int k = n0 + tid - 128;
    // (keys >= R: P = 0; HAVE_S: the key bias is part of the loaded scores)
    colv[tid - 128] = k < R ? ((p.kbias && !HAVE_S) ? p.kbias[(int64_t)b * p.Rp + k] * 1.4426950408889634f : 0.f) : -1e30f;
  }
  if (HAVE_S) {
    // the descriptor ends after the map's R written rows: a row >= R (or a 32 x 32 block outside the map) reads as 0, whatever
    // the memory holds; the whole offset is in the lane register - that is what the range check looks at
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.scores) + (int64_t)bh * p.Rp * p.Rp, 0, R * p.Rp * 4, 0x00020000);
    const unsigned rowb_ = (unsigned)p.Rp * 4u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned base_ = (m0 + rb + i * 32 < p.Rp && n0 + cb + j * 32 < p.Rp)
                                   ? (unsigned)(m0 + rb + i * 32 + 4 * half) * rowb_ + (unsigned)(n0 + cb + j * 32 + r) * 4u : 0x80000000u;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          accS[i][j][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rT, base_ + (unsigned)((e & 3) + 8 * (e >> 2)) * rowb_, 0, 2));
      }
  }
  __syncthreads();
  frags(a0, w0, 0, 0);
  if (!HAVE_S) product(accS, rQ, vq, rK, true, rD, vd, rV);
  // the W rows of the second product are the V rows: same offsets as the K rows (vk), other descriptor
  product(accP, rD, vd, rV, false, rD, vd, rV);

  // ---- epilogue: P, keep mask, Pd and dS row by row, stored straight from the accumulator layout - a store instruction
  // writes 2 rows x 128 contiguous bytes (whole cache lines).  Both maps are read again only by later launches, 1.6 GB each
  // at batch_size = 64: nontemporal.
  float colb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) colb[j] = colv[cb + j * 32 + r];
  const unsigned rowb = (unsigned)p.Rp * 4u;
  // No guards: Rp is a multiple of 32, so a 32 x 32 accumulator block is inside the map or outside it as a whole; the lanes
  // of an outside block get an offset beyond the descriptor (which ends with this (sample, head)'s map) and the hardware
  // drops their stores.  (The range check sees the lane offset only - the uniform part stays inside a block's 32 rows.)
  unsigned voff[2][2];                                // lane part of the store offset, per block
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      voff[i][j] = (m0 + rb + i * 32 < p.Rp && n0 + cb + j * 32 < p.Rp)
                       ? (unsigned)(m0 + rb + i * 32 + 4 * half) * rowb + (unsigned)(n0 + cb + j * 32 + r) * 4u : 0x80000000u;
  const int64_t mapo = (int64_t)bh * p.Rp * p.Rp;
  const int mapb = p.Rp * p.Rp * 4;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(p.Pd + mapo, 0, mapb, 0x00020000),
                               rS = __builtin_amdgcn_make_buffer_rsrc(p.dS + mapo, 0, mapb, 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    unsigned so = 0u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ql = rb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      const f32x4 rv = *reinterpret_cast<const f32x4*>(&rowv[4 * ql]);
      const float lq = rv[0], dq = rv[1];
      const float ka = rv[2], kf = rv[3];      // (copies: a bit cast applied to a vector element expression reads element 0)
      const gvd_encdrop_key dkey = {__float_as_uint(ka), __float_as_uint(kf)};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = n0 + cb + j * 32 + r;
        // branch-free: rows / keys >= R carry lse = +1e30 / bias = -1e30, so their P is exactly 0 and both maps get zeros there;
        // thresh == 0 (no dropout) keeps every element.  Same roundings as the select form of rounds 4-5 (no contraction).
        const float pr = __builtin_amdgcn_exp2f(fmaf(accS[i][j][e], p.c2, colb[j]) - lq);
        const float ks = gvd_encdrop_keep(dkey, (uint32_t)k, p.thresh) ? p.keep_scale : 0.f;
        const float pd = pr * ks;
        const float ds = (p.scale * pr) * __fsub_rn(__fmul_rn(accP[i][j][e], ks), dq);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pd), rP, voff[i][j], so, 2);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ds), rS, voff[i][j], so, 2);
      }
      so += (e & 3) == 3 ? 5u * rowb : rowb;
      if (e & 1) __builtin_amdgcn_sched_barrier(0);      // (two rows in flight: the scheduler would hoist all 32 row reads + hashes)
    }
  }
}

// delta[map, q] = sum_d dO[b, q, h, d] * O[b, q, h, d]: one wave per (b, q) row, head after head (44 lanes x 16 bytes)
__global__ __launch_bounds__(256) void enc_attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O,
                                                             int64_t ld, float* __restrict__ delta, int B, int Rp, int R,
                                                             int Rs, int nh, int HP) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * Rp) return;
  const int b = (int)(row / Rp), q = (int)(row - (int64_t)b * Rp);
  const int64_t drow = (int64_t)b * Rs + min(q, R - 1);          // (rows q >= R are not read: their delta is 0)
  const float* d = dO + drow * ld;
  const float* o = O + drow * ld;
  for (int h = 0; h < nh; ++h) {
    float acc = 0.f;
    if (q < R)
      for (int c = 4 * lane; c < HP; c += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(d + h * HP + c), y = *reinterpret_cast<const f32x4*>(o + h * HP + c);
        acc = fmaf(x[0], y[0], fmaf(x[1], y[1], fmaf(x[2], y[2], fmaf(x[3], y[3], acc))));
      }
    acc = wave_sum(acc);
    if (lane == 0) delta[((int64_t)b * nh + h) * Rp + q] = acc;
  }
}

// test aid: the keep mask itself, u8 [n_maps, Rp, Rp] (1 = kept)
__global__ void enc_dropout_mask_kernel(uint8_t* out, int64_t n, int Rp, uint32_t thresh, uint32_t seed_lo, uint32_t seed_hi) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t row = (uint32_t)(i / Rp), key = (uint32_t)(i - (int64_t)row * Rp);
  out[i] = gvd_encdrop_keep(gvd_encdrop_row(row, seed_lo, seed_hi), key, thresh) ? 1 : 0;
}

}  // namespace

extern "C" int gvd_enc_attn_bwd_maps(const float* qkv, int64_t ld, const float* dO, const float* O, int64_t ldo,
                                     const float* lse2, const float* key_bias, const float* scores, float* delta, float* Pd,
                                     float* dS, int B, int Rp, int R, int sample_rows, int n_heads, int head_pad, float scale,
                                     float p_drop, uint64_t seed, gvd_stream_t stream) {
  if (!qkv || !dO || !O || !lse2 || !delta || !Pd || !dS || B <= 0 || R <= 0 || Rp < R || sample_rows < R || (Rp % 32) != 0 || n_heads <= 0 ||
      head_pad < 32 || (head_pad % 16) != 0 || (ld % 4) != 0 || (ldo % 4) != 0 || ld < (int64_t)3 * n_heads * head_pad ||
      ldo < (int64_t)n_heads * head_pad || !gvd_aligned16(qkv) || !gvd_aligned16(dO) || !gvd_aligned16(O) ||
      !gvd_aligned16(Pd) || !gvd_aligned16(dS) || (scores && !gvd_aligned16(scores)) || !(p_drop >= 0.f) || !(p_drop < 1.f) ||
      (int64_t)Rp * ld * 4 >= (int64_t)1 << 31 || (int64_t)Rp * ldo * 4 >= (int64_t)1 << 31 ||
      (int64_t)B * n_heads * Rp >= (int64_t)1 << 32 || (int64_t)B * n_heads > 65535 || (int64_t)Rp * Rp * 4 >= (int64_t)1 << 31)
    return GVD_EINVAL;
  hipStream_t st = gvd_s(stream);
  hipLaunchKernelGGL(enc_attn_delta_kernel, dim3((unsigned)(((int64_t)B * Rp + 3) / 4)), dim3(256), 0, st, dO, O, ldo, delta,
                     B, Rp, R, sample_rows, n_heads, head_pad);
  GVD_CHECK_LAUNCH();
  MapParams p = {};
  p.qkv = qkv; p.ld = ld; p.dO = dO; p.ldo = ldo; p.lse2 = lse2; p.delta = delta; p.kbias = key_bias; p.scores = scores; p.Pd = Pd; p.dS = dS;
  p.B = B; p.Rp = Rp; p.R = R; p.Rs = sample_rows; p.nh = n_heads; p.HP = head_pad;
  p.scale = scale; p.c2 = scores ? 1.0f : scale * 1.4426950408889634f;      // (loaded scores are in the log2 domain already)
  p.keep_scale = 1.0f / (1.0f - p_drop);
  p.thresh = p_drop > 0.f ? gvd_drop_thresh(p_drop) : 0u;
  p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
  const int nt = (Rp + BM - 1) / BM;
  p.ntk = nt;
  if (scores) hipLaunchKernelGGL(enc_attn_bwd_maps_kernel<true>, dim3((unsigned)(nt * nt), (unsigned)(B * n_heads)), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(enc_attn_bwd_maps_kernel<false>, dim3((unsigned)(nt * nt), (unsigned)(B * n_heads)), dim3(256), 0, st, p);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_enc_dropout_mask(uint8_t* out, int64_t n_maps, int Rp, float p_drop, uint64_t seed, gvd_stream_t stream) {
  if (!out || n_maps <= 0 || Rp <= 0 || !(p_drop >= 0.f) || !(p_drop < 1.f) || n_maps * Rp >= (int64_t)1 << 32) return GVD_EINVAL;
  const int64_t n = n_maps * Rp * Rp;
  hipLaunchKernelGGL(enc_dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, gvd_s(stream), out, n, Rp,
                     p_drop > 0.f ? gvd_drop_thresh(p_drop) : 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
  GVD_CHECK_LAUNCH();
  return 0;
}
