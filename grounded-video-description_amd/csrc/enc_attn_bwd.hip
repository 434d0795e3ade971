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
//   (the three N = 176 products run on the pipelined GEMM, gemm_pipe.hip, as before).
// Rows / columns >= R of both maps are written as zeros: the consumers contract over the whole padded Rp.
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
constexpr int EPI_LD = 68;
constexpr int TILE = BM * BK;              // floats of one operand tile (unpadded)

struct MapParams {
  const float* qkv; int64_t ld;            // packed q | k | v, [B, Rp, 3 * nh * HP]
  const float* dO; int64_t ldo;            // [B, Rp, nh * HP]
  const float* lse2;                       // [B * nh, Rp]
  const float* delta;                      // [B * nh, Rp]
  const float* kbias;                      // nullable [B, Rp] (natural-log units)
  float* Pd; float* dS;                    // [B * nh, Rp, Rp]
  int B, Rp, R, Rs, nh, HP;                // Rs: rows between consecutive samples in qkv / dO (>= R; Rp on the padded layout)
  float scale, c2, keep_scale;
  uint32_t thresh, seed_lo, seed_hi;
  int ntk;
};

__global__ __launch_bounds__(256, 2) void enc_attn_bwd_maps_kernel(const MapParams p) {
  __shared__ __attribute__((aligned(16))) float smem[4 * 64 * EPI_LD];          // 69,632 B: operand buffers, then the epilogue slices
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
  dma(rQ, vq, rK, 0, 0);
  // The epilogue needs lse / delta of the tile's 128 query rows and the bias of its 128 key columns: one coalesced element
  // per thread, fetched under the first tile's load latency and parked in the 4 KB of LDS between the operand buffers and
  // the end of the epilogue slices (64 dependent scalar loads per thread in the epilogue cost 0.24 of the kernel's 3.4 ms:
  // ablation timings, profiles/r04/bwd_maps_ablate_b.log).
  float* rowv = smem + 4 * TILE;               // [128][lse, delta]
  float* colv = rowv + 256;                    // [128] key bias (log2 units)
  {
    const int64_t mrow = (int64_t)bh * p.Rp;
    const int q = m0 + (tid & 127);
    float v = 0.f;
    if (q < R) v = tid < 128 ? p.lse2[mrow + q] : p.delta[mrow + q];
    rowv[2 * (tid & 127) + (tid >> 7)] = v;
    if (tid < 128) {
      const int k = n0 + tid;
      colv[tid] = (p.kbias && k < R) ? p.kbias[(int64_t)b * p.Rp + k] * 1.4426950408889634f : 0.f;
    }
  }
  __syncthreads();
  frags(a0, w0, 0, 0);
  product(accS, rQ, vq, rK, true, rD, vd, rV);
  {
    // the W rows of the second product are the V rows: same offsets as the K rows (vk), other descriptor
    product(accP, rD, vd, rV, false, rD, vd, rV);
  }

  // ---- epilogue: P, keep mask, Pd and dS in registers (accS <- Pd, accP <- dS), then two transposed store passes
  const bool drop = p.thresh != 0u;
  float colb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) colb[j] = colv[cb + j * 32 + r];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int ql = rb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
      const int q = m0 + ql;
      const bool qok = q < R;
      const float lq = rowv[2 * ql], dq = rowv[2 * ql + 1];
      const gvd_encdrop_key dkey = drop ? gvd_encdrop_row((uint32_t)bh * (uint32_t)p.Rp + (uint32_t)q, p.seed_lo, p.seed_hi) : gvd_encdrop_key{0u, 0u};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = n0 + cb + j * 32 + r;
        float pd = 0.f, ds = 0.f;
        if (qok && k < R) {
          const float bias2 = colb[j];
          const float pr = __builtin_amdgcn_exp2f(fmaf(accS[i][j][e], p.c2, bias2) - lq);
          const bool keep = !drop || gvd_encdrop_keep(dkey, (uint32_t)k, p.thresh);
          pd = keep ? pr * p.keep_scale : 0.f;
          ds = p.scale * pr * ((keep ? accP[i][j][e] * p.keep_scale : 0.f) - dq);
        }
        accS[i][j][e] = pd;
        accP[i][j][e] = ds;
      }
    }
  }
  __syncthreads();           // every wave finished reading the operand tiles (and rowv / colv, which wave 3's slice covers)
  float* T = smem + wave * 64 * EPI_LD;                        // this wave's private 64 x 64 slice
  const int c4 = (lane & 15) * 4, rsub = lane >> 4;
  const int gn = n0 + cb + c4;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* Cb = (pass == 0 ? p.Pd : p.dS) + (int64_t)bh * p.Rp * p.Rp;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          T[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half) * EPI_LD + j * 32 + r] = pass == 0 ? accS[i][j][e] : accP[i][j][e];
    // (DS operations of one wave execute in order: its reads below see its own writes above, and the next pass's writes
    // come after this pass's reads)
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int row = it * 4 + rsub;
      const int gm = m0 + rb + row;
      const f32x4 v = *reinterpret_cast<const f32x4*>(&T[row * EPI_LD + c4]);
      if (gm < p.Rp && gn < p.Rp) *reinterpret_cast<f32x4*>(Cb + (int64_t)gm * p.Rp + gn) = v;
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
                                     const float* lse2, const float* key_bias, float* delta, float* Pd, float* dS, int B,
                                     int Rp, int R, int sample_rows, int n_heads, int head_pad, float scale, float p_drop,
                                     uint64_t seed, gvd_stream_t stream) {
  if (!qkv || !dO || !O || !lse2 || !delta || !Pd || !dS || B <= 0 || R <= 0 || Rp < R || sample_rows < R || (Rp % 32) != 0 || n_heads <= 0 ||
      head_pad < 32 || (head_pad % 16) != 0 || (ld % 4) != 0 || (ldo % 4) != 0 || ld < (int64_t)3 * n_heads * head_pad ||
      ldo < (int64_t)n_heads * head_pad || !gvd_aligned16(qkv) || !gvd_aligned16(dO) || !gvd_aligned16(O) ||
      !gvd_aligned16(Pd) || !gvd_aligned16(dS) || !(p_drop >= 0.f) || !(p_drop < 1.f) ||
      (int64_t)Rp * ld * 4 >= (int64_t)1 << 31 || (int64_t)Rp * ldo * 4 >= (int64_t)1 << 31 ||
      (int64_t)B * n_heads * Rp >= (int64_t)1 << 32 || (int64_t)B * n_heads > 65535)
    return GVD_EINVAL;
  hipStream_t st = gvd_s(stream);
  hipLaunchKernelGGL(enc_attn_delta_kernel, dim3((unsigned)(((int64_t)B * Rp + 3) / 4)), dim3(256), 0, st, dO, O, ldo, delta,
                     B, Rp, R, sample_rows, n_heads, head_pad);
  GVD_CHECK_LAUNCH();
  MapParams p = {};
  p.qkv = qkv; p.ld = ld; p.dO = dO; p.ldo = ldo; p.lse2 = lse2; p.delta = delta; p.kbias = key_bias; p.Pd = Pd; p.dS = dS;
  p.B = B; p.Rp = Rp; p.R = R; p.Rs = sample_rows; p.nh = n_heads; p.HP = head_pad;
  p.scale = scale; p.c2 = scale * 1.4426950408889634f; p.keep_scale = 1.0f / (1.0f - p_drop);
  p.thresh = p_drop > 0.f ? gvd_drop_thresh(p_drop) : 0u;
  p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
  const int nt = (Rp + BM - 1) / BM;
  p.ntk = nt;
  hipLaunchKernelGGL(enc_attn_bwd_maps_kernel, dim3((unsigned)(nt * nt), (unsigned)(B * n_heads)), dim3(256), 0, st, p);
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
