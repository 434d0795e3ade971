// Fused multi-head self-attention of the `obj_interact` region encoder (transformer.py:90-123 as configured at
// model.py:126-135: 6 uneven heads = Tensor.chunk(1024, 6) -> 171 x5 + 169 columns, softmax(q k^T / sqrt(d_model)) v,
// no mask) in fp32 on the matrix cores, flash-style: the [B,R,R] score maps the reference materialises three
// times per head (bmm out, softmax in/out, bmm in: ~48 GB of HBM traffic per forward at B=256) never leave the CU.
//
// Work decomposition (MI355X): workgroup = (128 query rows, head, sample); each of the 4 waves owns 32 query rows
// and walks the R keys in tiles of 32.  The products are formed in the SWAPPED orientation so that everything the
// online softmax needs is lane-local:
//   S^T[key][q] = sum_d K[key][d] Q[q][d]      MFMA A = K tile (LDS), B = Q (88 registers, loaded once)
//        -> lane (q = l&31, half) holds the scores of its query against 16 keys: running max / sum are per lane,
//           only one xor-32 shuffle per tile combines the two half-waves
//   O^T[d][q]  += sum_key V[key][d] P[q][key]   MFMA A = V tile (LDS), B = P = the exp'd score registers AS THEY ARE
//        -> the accumulator column of a lane is its own query: the exp(m_old - m_new) rescale is per lane too
// (v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[row=(e&3)+8(e>>2)+4(l>>5)][col=l&31]; the k
// slots are assigned so that step s of the O^T product consumes exactly score register s.)
// K/V tiles (32 keys x head) are staged global -> registers -> LDS with lane-contiguous dword loads (head column
// offsets are only 4-byte aligned), double-buffered: one barrier per key tile; the next tile's loads fly under the
// 184 MFMAs of the current one.  Head widths are zero-padded to 176 (K/Q) / 192 (V) in LDS/registers only.
#include "gvd_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int FA_DK = 176;          // padded head width for the q.k contraction (22 x 8)
constexpr int FA_DV = 192;          // padded head width of the output (6 MFMA row tiles of 32)
constexpr int FA_KLD = 180;         // LDS row strides (floats): conflict-free ds_read_b128 over 16 rows
constexpr int FA_VLD = 196;
constexpr int FA_MAXH = 8;

struct FaParams {
  const float* q; const float* k; const float* v; float* o;
  int64_t ld;            // row stride (floats) of q/k/v/o ([B,R,ld])
  int B, R, n_heads;
  int c0[FA_MAXH];       // first column of each head
  int dh[FA_MAXH];       // width of each head (<= 176)
};

template <bool GLDS>
__global__ __launch_bounds__(256, 1) void flash_attn_kernel(const FaParams p) {
  __shared__ __attribute__((aligned(16))) float s_k[2][32 * FA_KLD];
  __shared__ float s_v[2][32 * FA_VLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int c0 = p.c0[h], dh = p.dh[h];
  const int R = p.R;
  const int64_t ld = p.ld;
  const float* qb = p.q + (int64_t)b * R * ld + c0;
  const float* kb_ = p.k + (int64_t)b * R * ld + c0;
  const float* vb = p.v + (int64_t)b * R * ld + c0;
  float* ob = p.o + (int64_t)b * R * ld + c0;
  const int qrow = blockIdx.x * 128 + wave * 32 + col;

  // ---- this lane's query row, laid out for the MFMA B operand: qreg[kb][t] = Q[qrow][8*kb + 4*half + t]
  f32x4 qreg[FA_DK / 8];
#pragma unroll
  for (int kb = 0; kb < FA_DK / 8; ++kb) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (qrow < R) {
      const float* src = qb + (int64_t)qrow * ld + 8 * kb + 4 * half;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (8 * kb + 4 * half + t < dh) v[t] = src[t] * 1.4426950408889634f;   // scores in the log2 domain: exp -> exp2
    }
    qreg[kb] = v;
  }

  // ---- staging roles: wave w stages key rows 8w..8w+7 of a tile; lane covers columns lane, lane+64, lane+128
  //      (lane-contiguous dword loads: head column offsets are only 4-byte aligned)
  float rk[24], rv[24];
  auto load_tile = [&](int key0) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int key = key0 + wave * 8 + rr;
      const bool ok = key < R;
      const float* kr = kb_ + (int64_t)(ok ? key : 0) * ld + lane;
      const float* vr = vb + (int64_t)(ok ? key : 0) * ld + lane;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const bool in = ok && (lane + 64 * c) < dh;
        rk[rr * 3 + c] = in ? kr[64 * c] : 0.f;
        rv[rr * 3 + c] = in ? vr[64 * c] : 0.f;
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float* kd = &s_k[buf][(wave * 8 + rr) * FA_KLD + lane];
      float* vd = &s_v[buf][(wave * 8 + rr) * FA_VLD + lane];
      kd[0] = rk[rr * 3]; kd[64] = rk[rr * 3 + 1];
      if (lane + 128 < FA_DK) kd[128] = rk[rr * 3 + 2];
      vd[0] = rv[rr * 3]; vd[64] = rv[rr * 3 + 1]; vd[128] = rv[rr * 3 + 2];
    }
  };

  // GLDS variant: the tiles go global -> LDS by asynchronous DMA (global_load_lds_dword: LDS address = wave-uniform
  // base + 4*lane, exactly the lane-contiguous row pieces staged above) — no staging registers, no ds_write pass.
  // Masked lanes (columns >= head width, keys >= R) do not write: the pad columns are zeroed once below; rows of
  // invalid keys keep finite stale data whose scores are set to -inf / whose probabilities are exactly 0.
  auto stage_async = [&](int key0, int buf) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int key = key0 + wave * 8 + rr;
      if (key < R) {
        const float* kr = kb_ + (int64_t)key * ld + lane;
        const float* vr = vb + (int64_t)key * ld + lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (lane + 64 * c < dh) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kr + 64 * c),
                                             (__attribute__((address_space(3))) void*)&s_k[buf][(wave * 8 + rr) * FA_KLD + 64 * c],
                                             4, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vr + 64 * c),
                                             (__attribute__((address_space(3))) void*)&s_v[buf][(wave * 8 + rr) * FA_VLD + 64 * c],
                                             4, 0, 0);
          }
        }
      }
    }
  };

  f32x16 oacc[FA_DV / 32];
#pragma unroll
  for (int dt = 0; dt < FA_DV / 32; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (R + 31) / 32;
  if (GLDS) {
    for (int i = tid; i < 2 * 32 * FA_KLD; i += 256) (&s_k[0][0])[i] = 0.f;
    for (int i = tid; i < 2 * 32 * FA_VLD; i += 256) (&s_v[0][0])[i] = 0.f;
    __syncthreads();
    stage_async(0, 0);
  } else {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
#pragma unroll 1
  for (int jt = 0; jt < ntiles; ++jt) {
    const int buf = jt & 1;
    const int key0 = jt * 32;
    if (jt + 1 < ntiles) {                              // next tile's transfer flies under this tile's MFMAs
      if (GLDS) stage_async(key0 + 32, buf ^ 1); else load_tile(key0 + 32);
    }

    // S^T tile: scores of this lane's query against keys (e&3) + 8*(e>>2) + 4*half of the tile.
    // LDS fragment reads run one step ahead of the MFMAs; sched barriers keep the compiler from hoisting all of them.
    f32x16 sacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = 0.f;
    const float* kp = &s_k[buf][col * FA_KLD + 4 * half];
    f32x4 a_cur = *reinterpret_cast<const f32x4*>(kp);
#pragma unroll
    for (int kb = 0; kb < FA_DK / 8; ++kb) {
      f32x4 a_nxt = a_cur;
      if (kb + 1 < FA_DK / 8) a_nxt = *reinterpret_cast<const f32x4*>(kp + 8 * (kb + 1));
#pragma unroll
      for (int t = 0; t < 4; ++t) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[t], qreg[kb][t], sacc, 0, 0, 0);
      a_cur = a_nxt;
      // issue order inside the step: MFMA, then the next fragment's LDS read in its shadow, then the other MFMAs
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // online softmax (per lane = per query row)
    float mt = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * half;
      if (key >= R) sacc[e] = -INFINITY;                // padded keys of the last tile
      mt = fmaxf(mt, sacc[e]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, GVD_WAVE));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // 0 on the first tile (m_run = -inf)
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      sacc[e] = __builtin_amdgcn_exp2f(sacc[e] - m_new);
      psum += sacc[e];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {        // exact: the rescale is the identity when no query of the wave raised its max
#pragma unroll
      for (int dt = 0; dt < FA_DV / 32; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[dt][e] *= alpha;
    }

    // O^T += V^T P^T : step s contracts key (s&3) + 8*(s>>2) + 4*half, i.e. exactly score register s
    const float* vp = &s_v[buf][col + 4 * half * FA_VLD];
    float v_cur[FA_DV / 32];
#pragma unroll
    for (int dt = 0; dt < FA_DV / 32; ++dt) v_cur[dt] = vp[32 * dt];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float v_nxt[FA_DV / 32];
#pragma unroll
      for (int dt = 0; dt < FA_DV / 32; ++dt) {
        v_nxt[dt] = v_cur[dt];
        if (s + 1 < 16) v_nxt[dt] = vp[(((s + 1) & 3) + 8 * ((s + 1) >> 2)) * FA_VLD + 32 * dt];
      }
#pragma unroll
      for (int dt = 0; dt < FA_DV / 32; ++dt)
        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v_cur[dt], sacc[s], oacc[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < FA_DV / 32; ++dt) v_cur[dt] = v_nxt[dt];
#pragma unroll
      for (int dt = 0; dt < FA_DV / 32; ++dt) {      // MFMA / LDS-read alternation: reads ride in the MFMA shadow
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    if (!GLDS && jt + 1 < ntiles) store_tile(buf ^ 1);
    __syncthreads();     // (GLDS: the compiler drains the DMA with vmcnt(0) ahead of this barrier)
  }

  // ---- normalise and write O[q][d] = O^T[d][q] / l  (lane = query row; d = 32*dt + (e&3) + 8*(e>>2) + 4*half)
  const float l_tot = l_run + __shfl_xor(l_run, 32, GVD_WAVE);
  const float inv = 1.0f / l_tot;
  if (qrow < R) {
    float* orow = ob + (int64_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < FA_DV / 32; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int d = 32 * dt + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (d < dh) orow[d] = oacc[dt][e] * inv;
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Variant with 16x16x4 MFMA tiles: 16 queries per wave (64 per workgroup), ~190 registers -> TWO waves per SIMD
// (two workgroups per CU, single-buffered LDS), so one wave's softmax VALU work and barriers hide under the other
// wave's MFMAs.  v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+reg][col=l&15].
//   S^T sub-tile (16 keys x 16 q): lane (q=l&15, g=l>>4) ends up with keys 4g+reg  -> running max/sum per lane,
//                                  two xor-shuffles (16, 32) combine the four key groups of a query
//   O^T += V^T P^T: step s contracts key 4g+s = score register s; D rows d = 16*dt + 4g + reg, column = own query
// ---------------------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void flash_attn16_kernel(const FaParams p) {
  __shared__ __attribute__((aligned(16))) float s_k[32 * FA_KLD];
  __shared__ float s_v[32 * FA_VLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int c16 = lane & 15, g = lane >> 4;
  // linear workgroup id, XCD-aware: the 16 query tiles of one (sample, head) run on ONE XCD so its K/V slices are
  // fetched into that L2 once
  const unsigned nqt = (unsigned)((p.R + 63) / 64);
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int qt = lid % nqt;
  const int h = (lid / nqt) % p.n_heads;
  const int b = lid / (nqt * p.n_heads);
  const int c0 = p.c0[h], dh = p.dh[h];
  const int R = p.R;
  const int64_t ld = p.ld;
  const float* qb = p.q + (int64_t)b * R * ld + c0;
  const float* kb_ = p.k + (int64_t)b * R * ld + c0;
  const float* vb = p.v + (int64_t)b * R * ld + c0;
  float* ob = p.o + (int64_t)b * R * ld + c0;
  const int qrow = qt * 64 + wave * 16 + c16;

  // Q as B operand: qreg[sb][t] = Q[qrow][16*sb + 4*g + t] (log2 domain)
  f32x4 qreg[FA_DK / 16];
#pragma unroll
  for (int sb = 0; sb < FA_DK / 16; ++sb) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (qrow < R) {
      const float* src = qb + (int64_t)qrow * ld + 16 * sb + 4 * g;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (16 * sb + 4 * g + t < dh) v[t] = src[t] * 1.4426950408889634f;
    }
    qreg[sb] = v;
  }

  float rk[24], rv[24];
  auto load_tile = [&](int key0) {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int key = key0 + wave * 8 + rr;
      const bool ok = key < R;
      const float* kr = kb_ + (int64_t)(ok ? key : 0) * ld + lane;
      const float* vr = vb + (int64_t)(ok ? key : 0) * ld + lane;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const bool in = ok && (lane + 64 * c) < dh;
        rk[rr * 3 + c] = in ? kr[64 * c] : 0.f;
        rv[rr * 3 + c] = in ? vr[64 * c] : 0.f;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float* kd = &s_k[(wave * 8 + rr) * FA_KLD + lane];
      float* vd = &s_v[(wave * 8 + rr) * FA_VLD + lane];
      kd[0] = rk[rr * 3]; kd[64] = rk[rr * 3 + 1];
      if (lane + 128 < FA_DK) kd[128] = rk[rr * 3 + 2];
      vd[0] = rv[rr * 3]; vd[64] = rv[rr * 3 + 1]; vd[128] = rv[rr * 3 + 2];
    }
  };

  constexpr int NDT = FA_DK / 16;      // 11 output row tiles of 16 (176 >= head width)
  f32x4v oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) oacc[dt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int ntiles = (R + 31) / 32;
  load_tile(0);
  store_tile();
  __syncthreads();
#pragma unroll 1
  for (int jt = 0; jt < ntiles; ++jt) {
    const int key0 = jt * 32;
    if (jt + 1 < ntiles) load_tile(key0 + 32);          // flies under this tile's MFMAs

    // S^T for the two 16-key sub-tiles; their accumulator chains are interleaved (16x16x4 has a 40-cycle dependent
    // latency against a 32-cycle issue interval)
    f32x4v sacc[2];
    sacc[0] = f32x4v{0.f, 0.f, 0.f, 0.f};
    sacc[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
    {
      const float* kp0 = &s_k[c16 * FA_KLD + 4 * g];
      const float* kp1 = kp0 + 16 * FA_KLD;
#pragma unroll
      for (int sb = 0; sb < FA_DK / 16; ++sb) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(kp0 + 16 * sb);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(kp1 + 16 * sb);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], qreg[sb][t], sacc[0], 0, 0, 0);
          sacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t], qreg[sb][t], sacc[1], 0, 0, 0);
        }
      }
    }
    // online softmax: this lane holds keys 16u + 4g + reg of its query
    float mt = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (key0 + 16 * u + 4 * g + r >= R) sacc[u][r] = -INFINITY;
        mt = fmaxf(mt, sacc[u][r]);
      }
    mt = fmaxf(mt, __shfl_xor(mt, 16, GVD_WAVE));
    mt = fmaxf(mt, __shfl_xor(mt, 32, GVD_WAVE));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sacc[u][r] = __builtin_amdgcn_exp2f(sacc[u][r] - m_new);
        psum += sacc[u][r];
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) oacc[dt] *= alpha;
    }
    // O^T += V^T P^T
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float* vp = &s_v[(16 * u + 4 * g + s4) * FA_VLD + c16];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
          oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[16 * dt], sacc[u][s4], oacc[dt], 0, 0, 0);
      }
    }
    __syncthreads();                       // every wave finished reading this tile
    if (jt + 1 < ntiles) store_tile();
    __syncthreads();
  }

  float l_tot = l_run + __shfl_xor(l_run, 16, GVD_WAVE);
  l_tot += __shfl_xor(l_tot, 32, GVD_WAVE);
  const float inv = 1.0f / l_tot;
  if (qrow < R) {
    float* orow = ob + (int64_t)qrow * ld;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 16 * dt + 4 * g + r;
        if (d < dh) orow[d] = oacc[dt][r] * inv;
      }
  }
}

}  // namespace

extern "C" int gvd_flash_attn_f32(const float* q, const float* k, const float* v, float* o, int B, int R, int64_t ld,
                                  int n_heads, const int* head_col0, const int* head_width, gvd_stream_t stream) {
  if (!q || !k || !v || !o || B <= 0 || R <= 0 || n_heads <= 0 || n_heads > FA_MAXH || !head_col0 || !head_width)
    return GVD_EINVAL;
  FaParams p = {};
  p.q = q; p.k = k; p.v = v; p.o = o; p.ld = ld; p.B = B; p.R = R; p.n_heads = n_heads;
  for (int h = 0; h < n_heads; ++h) {
    if (head_width[h] <= 0 || head_width[h] > FA_DK || head_col0[h] < 0 || head_col0[h] + head_width[h] > ld)
      return GVD_EINVAL;
    p.c0[h] = head_col0[h]; p.dh[h] = head_width[h];
  }
  dim3 grid((unsigned)((R + 127) / 128), (unsigned)n_heads, (unsigned)B);
  // default: 16x16x4 tiles, 2 waves/SIMD (B=256: 11.3 ms/layer vs 12.2 ms for the 32x32x2 / 1-wave kernel above;
  // GVD_FLASH_V16=0 selects the latter)
  static const int v16 = getenv("GVD_FLASH_V16") ? atoi(getenv("GVD_FLASH_V16")) : 1;
  if (v16) {
    const unsigned nwg = (unsigned)((R + 63) / 64) * n_heads * B;
    hipLaunchKernelGGL(flash_attn16_kernel, dim3(nwg), dim3(256), 0, gvd_s(stream), p);
    GVD_CHECK_LAUNCH();
    return 0;
  }
  static const int glds = getenv("GVD_FLASH_GLDS") ? atoi(getenv("GVD_FLASH_GLDS")) : 0;   // tuning knob
  if (glds) hipLaunchKernelGGL(flash_attn_kernel<true>, grid, dim3(256), 0, gvd_s(stream), p);
  else hipLaunchKernelGGL(flash_attn_kernel<false>, grid, dim3(256), 0, gvd_s(stream), p);
  GVD_CHECK_LAUNCH();
  return 0;
}
