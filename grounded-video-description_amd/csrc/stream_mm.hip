// Streaming fp32-MFMA products between a FEW rows per segment (the caption's words / decoder steps: M <= 32) and the
// segment's big per-region tensor ([R, N]: fc7 region features g_pool [R,2048], region / frame features [R,1024]) - each
// one reads or writes the [B, R, N] tensor exactly once, so each is an HBM stream with a small MFMA tail (10 - 16 flop per
// streamed byte against a ridge of ~20), not a GEMM: the operand that streams goes global -> registers (or global -> LDS by
// direct loads) with 512-byte contiguous row segments per half wave, the few-row operand stays in registers for the whole
// launch.
//
//   gvd_grounder_fwd_f32     out[b,m,r] = xt[b,m,:] . feats[b,r,:] + mbias[b,m] + rowbias[b,m,r]; out[mask] = -1e8
//                            `AttModel._grounder`, dot-product branch (model.py:262-278) as used at model.py:469-480
//   gvd_rows_contract_f32    out[b,m,:] = sum_r S[b,m,r] F[b,r,:]          d xt of the grounder (autograd of model.py:262-265)
//   gvd_rank_update_f32      out[b,r,:] = sum_m S[b,m,r] X[b,m,:]          d feats of the grounder; d pool / d conv of the
//                            attention contexts over all decoder steps (autograd of AttModel.py:50,96: alpha^T d_ctx)
//
// MFMA = v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains).  Lane (c = l & 31, half = l >> 5) supplies A[row c][k = half] and
// B[k = half][col c]; accumulator register e holds row (e & 3) + 8 (e >> 2) + 4 half, column c.  Which REAL index a k slot
// stands for is free as long as both operands agree, and so is the real column behind "col c": a lane's 16-byte load
// [4 c, 4 c + 4) of a row feeds FOUR interleaved column tiles (tile j = columns 4 c + j), which makes every streamed load a
// 16-byte-per-lane, 512-byte-per-half-wave contiguous access with no LDS transpose.
#include "gvd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}

// ------------------------------------------------------------------------------------------------------------------
// rank update:  out[b, r, n] = sum_{m < M} S[b, m, r] X[b, m, n]      (S entries under `mask` count as 0)
// grid (r groups, ceil(N / 512), B); wave w of a workgroup owns columns [n0, n0 + 128), its X fragments (M x 128) stay in
// registers; the workgroup walks `tiles` 32-row tiles of R.  k slot (i, half, t) <-> m = 8 i + 4 half + t.
// ------------------------------------------------------------------------------------------------------------------
struct RankUpdParams {
  const float* S; int64_t lds_; int64_t sbs;
  const uint8_t* mask; int64_t ldmk; int64_t mbs;
  const float* X; int64_t ldx; int64_t xbs;
  float* out; int64_t ldo; int64_t obs;
  int M, R, N, tiles;
};

template <int KP>
__global__ __launch_bounds__(256, 2) void rank_update_kernel(const RankUpdParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  const int b = blockIdx.z;
  const int n0 = (blockIdx.y * 4 + wave) * 128;
  if (n0 >= p.N) return;
  const float* Xb = p.X + (int64_t)b * p.xbs + n0 + 4 * c;
  f32x4 xb[KP][4];
#pragma unroll
  for (int i = 0; i < KP; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = min(8 * i + 4 * half + t, p.M - 1);          // rows past M: clamped (their S value is forced to 0)
      xb[i][t] = *reinterpret_cast<const f32x4*>(Xb + (int64_t)m * p.ldx);
    }
  const float* Sb = p.S + (int64_t)b * p.sbs;
  const uint8_t* Mb = p.mask ? p.mask + (int64_t)b * p.mbs : nullptr;
  float* Ob = p.out + (int64_t)b * p.obs + n0 + 4 * c;
  const int t0 = blockIdx.x * p.tiles;
  const int ntile = (p.R + 31) / 32;
  const int t1 = min(t0 + p.tiles, ntile);

  auto load_s = [&](float (&s)[KP][4], int tile) {
    const int rr = min(tile * 32 + c, p.R - 1);
#pragma unroll
    for (int i = 0; i < KP; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int m = 8 * i + 4 * half + t;
        const int mm = min(m, p.M - 1);
        float v = Sb[(int64_t)mm * p.lds_ + rr];
        if (Mb && Mb[(int64_t)mm * p.ldmk + rr]) v = 0.f;
        s[i][t] = m < p.M ? v : 0.f;
      }
  };
  float s[KP][4], sn[KP][4];
  if (t0 < t1) load_s(s, t0);
#pragma unroll 1
  for (int tile = t0; tile < t1; ++tile) {
    if (tile + 1 < t1) load_s(sn, tile + 1);                     // next tile's few values under this tile's MFMAs
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = zero16();
#pragma unroll
    for (int i = 0; i < KP; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[i][t], xb[i][t][j], acc[j], 0, 0, 0);
    const int r0 = tile * 32;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * half;
      if (row < p.R) {
        const f32x4 v = {acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(Ob + (int64_t)row * p.ldo));     // written once, read much later
      }
    }
#pragma unroll
    for (int i = 0; i < KP; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) s[i][t] = sn[i][t];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// contraction over the streamed rows:  out[b, m, n] = sum_r S[b, m, r] F[b, r, n]      (m < M <= 32)
// grid (N / 128, B); the four waves of a workgroup take a quarter of R each (in-workgroup reduction through LDS at the end).
// Per 8-row block: one 16-byte gather per lane of S (row m = c, columns r .. r + 3 of its half), four 16-byte loads of F
// (rows r + 4 half + t: 512 contiguous bytes per half wave), 16 MFMAs.  F loads run `PD` blocks ahead in a register ring.
// ------------------------------------------------------------------------------------------------------------------
struct ContractParams {
  const float* S; int64_t lds_; int64_t sbs;
  const uint8_t* mask; int64_t ldmk; int64_t mbs;
  const float* F; int64_t ldf; int64_t fbs;
  float* out; int64_t ldo; int64_t obs;
  int M, R, N, s_vec;          // s_vec: S rows are 16-byte aligned (lds_ % 4 == 0, base aligned)
  const float* St; int64_t stbs;   // optional transposed copy of S: [B][R][32] (mask applied, rows m >= M zero)
};

constexpr int CT_RING = 4, CT_PD = 3;     // (ring 8 / 6 ahead: same time, 252 registers - profiles/r05/stream_wide_k.txt)

__global__ __launch_bounds__(256, 2) void rows_contract_kernel(const ContractParams p) {
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 64 * 4];     // 64 KiB: [wave][e][lane][4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  const int b = blockIdx.y, n0 = blockIdx.x * 128;
  const int rq = (((p.R + 3) / 4) + 7) / 8 * 8;                  // rows per wave, a multiple of the 8-row block
  const int rbeg = wave * rq, rend = min(p.R, rbeg + rq);
  const int nblk = rend > rbeg ? (rend - rbeg + 7) / 8 : 0;
  const int mm = min(c, p.M - 1);
  const float* Srow = p.S + (int64_t)b * p.sbs + (int64_t)mm * p.lds_;
  const uint8_t* Mrow = p.mask ? p.mask + (int64_t)b * p.mbs + (int64_t)mm * p.ldmk : nullptr;
  const float* Fb = p.F + (int64_t)b * p.fbs + n0 + 4 * c;
  const bool live_m = c < p.M;
  const float* Stb = p.St ? p.St + (int64_t)b * p.stbs : nullptr;

  f32x4 fr[CT_RING][4];
  f32x4 sr[CT_RING];
  auto issue = [&](const int slot, int blk) {
    const int r = rbeg + 8 * blk + 4 * half;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      fr[slot][t] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Fb + (int64_t)min(r + t, p.R - 1) * p.ldf));
    f32x4 s;
    if (Stb) {
      // transposed copy [R][32]: the 32 lanes of a half read one 128-byte row per k slot (a gather over the [M][R] layout
      // touches 32 cache lines per instruction and doubled the L1 <-> L2 traffic of the kernel)
#pragma unroll
      for (int t = 0; t < 4; ++t) s[t] = (r + t < rend) ? Stb[(int64_t)(r + t) * 32 + c] : 0.f;
      sr[slot] = s;
      return;
    }
    if (p.s_vec && r + 3 < rend) {
      s = *reinterpret_cast<const f32x4*>(Srow + r);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) s[t] = (r + t < rend) ? Srow[r + t] : 0.f;
    }
    if (Mrow) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (r + t < rend && Mrow[r + t]) s[t] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (!live_m || r + t >= rend) s[t] = 0.f;
    sr[slot] = s;
  };
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = zero16();
#pragma unroll
  for (int k = 0; k < CT_PD; ++k)
    if (k < nblk) issue(k, k);
#pragma unroll 1
  for (int base = 0; base < nblk; base += CT_RING) {
#pragma unroll
    for (int k = 0; k < CT_RING; ++k) {
      const int blk = base + k;
      if (blk < nblk) {
        if (blk + CT_PD < nblk) issue((k + CT_PD) % CT_RING, blk + CT_PD);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(sr[k][t], fr[k][t][j], acc[j], 0, 0, 0);
      }
    }
  }
  // in-workgroup reduction over the four row quarters: wave w adds up accumulator registers 4 w .. 4 w + 3
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const f32x4 v = {acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
    *reinterpret_cast<f32x4*>(&red[((wave * 16 + e) * 64 + lane) * 4]) = v;
  }
  __syncthreads();
  float* Ob = p.out + (int64_t)b * p.obs + n0 + 4 * c;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = 4 * wave + k;
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[((0 * 16 + e) * 64 + lane) * 4]);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(&red[((w * 16 + e) * 64 + lane) * 4]);
    // accumulator row = the A operand's row = m
    const int m = (e & 3) + 8 * (e >> 2) + 4 * half;
    if (m < p.M) *reinterpret_cast<f32x4*>(Ob + (int64_t)m * p.ldo) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// grounder forward:  out[b, m, r] = xt[b, m, :] . feats[b, r, :] + mbias[b, m] + rowbias[b, m, r] ;  out[mask] = -1e8
// grid (ceil(R / 128), B), 4 waves x (32 region rows x 32 words) per workgroup, K in 32-deep tiles.  Both operand tiles go
// global -> LDS by direct loads (16 B per lane, rows of 32 floats, 16-byte slot s of row m holds k-chunk s ^ (m & 7): the
// layout of gemm_pipe.hip) into a ring of GS_STAGES buffers with GS_STAGES - 1 tiles in flight per workgroup: two resident
// workgroups keep ~80 KB of the stream in flight per CU, what 8 TB/s x the HBM latency asks for.  ONE barrier per k tile
// (a bare s_barrier after an explicit s_waitcnt vmcnt: the tile about to be multiplied has landed, the buffer about to be
// refilled was read by everyone).
// ------------------------------------------------------------------------------------------------------------------
struct GroundFwdParams {
  const float* feats; int64_t ldf; int64_t fbs;        // [B][R][K]
  const float* xt; int64_t ldxt; int64_t xbs;          // [B][M][K]
  const float* mbias; int64_t mb_bs;                   // [B][M] or NULL
  const float* rowbias; int64_t rb_ld; int64_t rb_bs;  // [B][M][R] or NULL
  const uint8_t* mask; int64_t ldmk; int64_t mk_bs;    // u8, ldmk may be 0 (same mask row for every m), or NULL
  float* out; int64_t ldo; int64_t obs;                // [B][M][R]
  int M, R, K;
};

constexpr int GS_X = 32 * 32;                           // floats per X tile (32 words x 32 k)

// NW waves x 32 region rows per workgroup, STAGES ring slots (STAGES - 1 tiles in flight)
template <int NW, int STAGES>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void grounder_fwd_kernel(const GroundFwdParams p) {
  constexpr int ROWS = 32 * NW, GS_A = ROWS * 32, SLOT = GS_A + GS_X;
  __shared__ __attribute__((aligned(16))) float smem[STAGES * SLOT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, half = lane >> 5;
  const int b = blockIdx.y, r0 = blockIdx.x * ROWS;
  const int srow = tid >> 3, kq = tid & 7;                // srow in [0, 8 NW)
  const int kq_sw = kq ^ (srow & 7);
  const __amdgpu_buffer_rsrc_t ra = gvd_rsrc(p.feats + (int64_t)b * p.fbs + (int64_t)r0 * p.ldf);
  const __amdgpu_buffer_rsrc_t rx = gvd_rsrc(p.xt + (int64_t)b * p.xbs);
  unsigned voa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    voa[i] = (unsigned)(min(r0 + srow + 8 * NW * i, p.R - 1) - r0) * (unsigned)p.ldf * 4u + 16u * kq_sw;
  const unsigned vox = (unsigned)min(srow & 31, p.M - 1) * (unsigned)p.ldxt * 4u + 16u * kq_sw;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const bool xload = wv < 4;                              // the X tile (32 rows x 8 chunks) is loaded by the first 256 lanes
  auto issue = [&](int kt) {                              // tile kt -> ring slot kt % STAGES (4 or 5 direct loads per lane)
    float* As = smem + (kt % STAGES) * SLOT;
    float* Xs = As + GS_A;
    const unsigned so = 128u * (unsigned)kt;              // 32 floats per k tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)&As[(8 * wv + 8 * NW * i) * 32], 16,
                                               voa[i], so, 0, 0);
    if (xload)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void*)&Xs[(8 * wv) * 32], 16, vox, so, 0, 0);
  };
  const int nkt = p.K / 32;
  f32x16 acc = zero16();
  const int rsw = r & 7;
#pragma unroll
  for (int k = 0; k < STAGES - 1; ++k)
    if (k < nkt) issue(k);
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt) {
    // tile kt has landed when at most the loads of the min(STAGES - 2, nkt - 1 - kt) younger tiles are outstanding
    const int younger = min(STAGES - 2, nkt - 1 - kt);
    if (xload) {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + STAGES - 1 < nkt) issue(kt + STAGES - 1);    // into the slot tile kt - 1 used: everyone is past its reads
    const float* As = smem + (kt % STAGES) * SLOT + (32 * wave + r) * 32;
    const float* Xs = smem + (kt % STAGES) * SLOT + GS_A + r * 32;
    f32x4 a[4], x[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int so4 = ((2 * q + half) ^ rsw) * 4;
      a[q] = *reinterpret_cast<const f32x4*>(As + so4);
      x[q] = *reinterpret_cast<const f32x4*>(Xs + so4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][t], x[q][t], acc, 0, 0, 0);
  }
  // epilogue: accumulator row = region row, column = word m = r (lane); registers 4 g .. 4 g + 3 are 4 consecutive regions
  const int m = r;
  if (m >= p.M) return;
  const float mb = p.mbias ? p.mbias[(int64_t)b * p.mb_bs + m] : 0.f;
  const float* rb = p.rowbias ? p.rowbias + (int64_t)b * p.rb_bs + (int64_t)m * p.rb_ld : nullptr;
  const uint8_t* mk = p.mask ? p.mask + (int64_t)b * p.mk_bs + (int64_t)m * p.ldmk : nullptr;
  float* ob = p.out + (int64_t)b * p.obs + (int64_t)m * p.ldo;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int rr = r0 + 32 * wave + 8 * g + 4 * half;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int reg = rr + t;
      if (reg < p.R) {
        float v = acc[4 * g + t] + mb;
        if (rb) v += rb[reg];
        if (mk && mk[reg]) v = GVD_MIN_VALUE;
        ob[reg] = v;
      }
    }
  }
}

}  // namespace

extern "C" int gvd_rank_update_f32(const float* S, int64_t lds_, int64_t s_batch_stride, const uint8_t* mask, int64_t ld_mask,
                                   int64_t mask_batch_stride, const float* X, int64_t ldx, int64_t x_batch_stride, float* out,
                                   int64_t ldo, int64_t out_batch_stride, int B, int M, int R, int N, gvd_stream_t stream) {
  if (!S || !X || !out || B <= 0 || M <= 0 || M > 32 || R <= 0 || N <= 0 || (N % 128) != 0) return GVD_EINVAL;
  if (!gvd_aligned16(X) || !gvd_aligned16(out) || (ldx % 4) || (ldo % 4) || (x_batch_stride % 4) || (out_batch_stride % 4))
    return GVD_EINVAL;
  RankUpdParams p = {S, lds_, s_batch_stride, mask, ld_mask, mask_batch_stride, X, ldx, x_batch_stride,
                     out, ldo, out_batch_stride, M, R, N, 1};
  const int ntile = (R + 31) / 32, ncol = (N + 511) / 512;
  // enough workgroups to fill the chip a few times over, but no fewer than 4 row tiles each (the X fragments are loaded
  // once per workgroup)
  int groups = ntile;
  while (groups > 1 && (long)groups * ncol * B > 2048 && (ntile + groups / 2 - 1) / (groups / 2) <= 8) groups /= 2;
  p.tiles = (ntile + groups - 1) / groups;
  groups = (ntile + p.tiles - 1) / p.tiles;
  const dim3 grid((unsigned)groups, (unsigned)ncol, (unsigned)B);
  const int kp = (M + 7) / 8;
  hipStream_t st = gvd_s(stream);
  switch (kp) {
    case 1: hipLaunchKernelGGL(rank_update_kernel<1>, grid, dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL(rank_update_kernel<2>, grid, dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL(rank_update_kernel<3>, grid, dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(rank_update_kernel<4>, grid, dim3(256), 0, st, p); break;
  }
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_rows_contract_f32(const float* S, int64_t lds_, int64_t s_batch_stride, const uint8_t* mask,
                                     int64_t ld_mask, int64_t mask_batch_stride, const float* S_t, const float* F, int64_t ldf,
                                     int64_t f_batch_stride, float* out, int64_t ldo, int64_t out_batch_stride, int B, int M,
                                     int R, int N, gvd_stream_t stream) {
  if ((!S && !S_t) || !F || !out || B <= 0 || M <= 0 || M > 32 || R <= 0 || N <= 0 || (N % 128) != 0) return GVD_EINVAL;
  if (!gvd_aligned16(F) || !gvd_aligned16(out) || (ldf % 4) || (ldo % 4) || (f_batch_stride % 4) || (out_batch_stride % 4))
    return GVD_EINVAL;
  ContractParams p = {S, lds_, s_batch_stride, mask, ld_mask, mask_batch_stride, F, ldf, f_batch_stride,
                      out, ldo, out_batch_stride, M, R, N, 0, S_t, (int64_t)R * 32};
  p.s_vec = (gvd_aligned16(S) && (lds_ % 4) == 0 && (s_batch_stride % 4) == 0) ? 1 : 0;
  hipLaunchKernelGGL(rows_contract_kernel, dim3((unsigned)(N / 128), (unsigned)B), dim3(256), 0, gvd_s(stream), p);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_grounder_fwd_f32(const float* feats, int64_t ldf, int64_t f_batch_stride, const float* xt, int64_t ldxt,
                                    int64_t xt_batch_stride, const float* mbias, int64_t mbias_batch_stride,
                                    const float* rowbias, int64_t rowbias_ld, int64_t rowbias_batch_stride, const uint8_t* mask,
                                    int64_t ld_mask, int64_t mask_batch_stride, float* out, int64_t ldo, int64_t out_batch_stride,
                                    int B, int M, int R, int K, gvd_stream_t stream) {
  if (!feats || !xt || !out || B <= 0 || M <= 0 || M > 32 || R <= 0 || K <= 0 || (K % 32) != 0) return GVD_EINVAL;
  if (!gvd_aligned16(feats) || !gvd_aligned16(xt) || (ldf % 4) || (ldxt % 4) || (f_batch_stride % 4) || (xt_batch_stride % 4))
    return GVD_EINVAL;
  // 32-bit buffer offsets inside one (sample, 128-row tile) / one sample's words
  if ((int64_t)256 * ldf * 4 >= (1ll << 31) || (int64_t)32 * ldxt * 4 >= (1ll << 31)) return GVD_EINVAL;
  GroundFwdParams p = {feats, ldf, f_batch_stride, xt, ldxt, xt_batch_stride, mbias, mbias_batch_stride,
                       rowbias, rowbias_ld, rowbias_batch_stride, mask, ld_mask, mask_batch_stride,
                       out, ldo, out_batch_stride, M, R, K};
  // 8 waves x 32 region rows per workgroup, 4 ring slots (three 36 KB tiles in flight per CU): measured best of
  // {4 waves, 8 waves} x {3, 4} slots at the training shape (profiles/r05/stream_variants_b.jsonl).  Measured and NOT adopted
  // (profiles/r05/stream_nomath_j.txt, stream_wide_*): this kernel with its MFMAs removed runs in the same time - the 128-byte-
  // per-row-and-tile access pattern, not the matrix pipe, sets its pace (4.7 TB/s); two 128-float-tile forms (64 rows x 512 B per
  // ring slot, words' tile in LDS or in registers) reach 5.1 TB/s without MFMAs but lose it again with them (one workgroup per
  // CU, 16 MFMAs per wave between barriers): within the box-to-box noise of this one, so the simpler kernel stays.
  hipLaunchKernelGGL((grounder_fwd_kernel<8, 4>), dim3((unsigned)((R + 255) / 256), (unsigned)B), dim3(512), 0, gvd_s(stream), p);
  GVD_CHECK_LAUNCH();
  return 0;
}
