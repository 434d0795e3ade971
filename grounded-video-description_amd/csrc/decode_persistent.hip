// Persistent greedy decoder for decode batches (B <= 4): the WHOLE token loop of AttModel._sample
// (model.py:580-624; TopDownCore.forward AttModel.py:134-164) as ONE cooperative launch.
//
// Why: at B = 4 the multi-kernel loop (decode.hip) is a chain of 7 launches per token whose kernels each sit on a
// ~5 us launch/ramp floor and re-stream 98 MB of LSTM / query / vocabulary weights per token.  Here 256 workgroups
// (one per CU, 512 threads) stay resident for all L tokens:
//   * the weights stay ON CHIP: the workgroup's 16 language-LSTM rows and 4 query rows in registers for the whole
//     call (104 VGPRs per lane), its <= 20 vocabulary rows in LDS (80 KB); the 16 attention-LSTM rows (48 VGPRs) are
//     re-fetched from L2 once per token during the logits phase, which keeps them out of the register-hungry
//     attention phase (holding them as well costs ~50 spilled registers there);
//   * the recurrent state crosses workgroups through small sc1-coherent buffers and the fence-free two-level grid
//     barrier of gvd_common.h (2.1 us), six per token;
//   * the only per-token HBM stream left is the one the roofline is about: the region/temporal features.
//
// Phases per token (B = barrier):
//   P1 att-LSTM   gates = fc_gates + [xt | h_att] W^T (K = 1536)  -> h_att                                    B
//   P2 queries    [q_temporal | q_region] = h_att [W_att ; W_att2]^T + b   (4 rows per workgroup)             B
//   P3 attention  one (sample, chunk) item per workgroup: scores, chunk-local softmax, partial context         B
//   P4 combine    (sample, 16 columns) per workgroup: merge chunk partials of both attentions -> att + att2   B
//   P5 lang-LSTM  gates = [att+att2 | h_att | h_lang] W^T + b (K = 3072) -> h_lang                            B
//   P6 logits     <= 20 vocabulary rows per workgroup -> per-workgroup (max, sum-exp, top-2) statistics       B
//   P7 token rule every workgroup merges the 256 statistics records (redundantly, no barrier): log-softmax,
//                 top-2, UNK -> runner-up (model.py:587-608), next input xt = relu(embed[token]) into its LDS
// Work split inside a workgroup for the products: wave = (unit or row u = w & 3, K-half kh = w >> 2); a lane keeps
// float4 slices k = 256*block + 4*lane of its rows, the LSTM gates packed in pairs so one v_pk_fma_f32 feeds two gate
// accumulators from one broadcast activation (these products are VALU-FMA-bound); 16 (row, sample) partial sums are
// reduced across the 64 lanes with a 17-shuffle transposing butterfly, across the K-halves through LDS.
// Measured (MI355X, B = 4, R = 1000, V = 5000): 41 us per token vs 85 us for the kernel-per-op loop.
#include "gvd_common.h"
#include "top2.h"
#include "decode_persistent.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int PD_H = 1024, PD_E = 512, PD_A = 512;
constexpr int PD_NT = 512;          // threads per workgroup (8 waves, 2 per SIMD -> 256 VGPRs per lane)
constexpr int PD_G = 256;           // workgroups (one per CU)
constexpr int PD_MB = 4;            // sample rows
constexpr int PD_RPW = 20;          // max vocabulary rows per workgroup (V <= 5120)
constexpr int PD_MAXCH = 64;        // max rows of an attention chunk
constexpr int PD_MAXNCT = 512;      // max chunks per sample (both attentions)
constexpr int PD_STAT = 8;          // floats per statistics record

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b, float acc) {
  return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], fmaf(a[0], b[0], acc))));
}

// Sum each of 16 per-lane values over the 64 lanes; returns the total of value (lane & 15).
__device__ __forceinline__ float reduce16(const float (&a)[16], int lane) {
  float r8[8], r4[4], r2[2];
  const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float keep = b0 ? a[2 * i + 1] : a[2 * i], send = b0 ? a[2 * i] : a[2 * i + 1];
    r8[i] = keep + __shfl_xor(send, 1, GVD_WAVE);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float keep = b1 ? r8[2 * i + 1] : r8[2 * i], send = b1 ? r8[2 * i] : r8[2 * i + 1];
    r4[i] = keep + __shfl_xor(send, 2, GVD_WAVE);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float keep = b2 ? r4[2 * i + 1] : r4[2 * i], send = b2 ? r4[2 * i] : r4[2 * i + 1];
    r2[i] = keep + __shfl_xor(send, 4, GVD_WAVE);
  }
  float r = (b3 ? r2[1] : r2[0]) + __shfl_xor(b3 ? r2[0] : r2[1], 8, GVD_WAVE);
  r += __shfl_xor(r, 16, GVD_WAVE);
  r += __shfl_xor(r, 32, GVD_WAVE);
  return r;
}

// Same for 4 values; returns the total of value (lane & 3).
__device__ __forceinline__ float reduce4(const float (&a)[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  float r2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float keep = b0 ? a[2 * i + 1] : a[2 * i], send = b0 ? a[2 * i] : a[2 * i + 1];
    r2[i] = keep + __shfl_xor(send, 1, GVD_WAVE);
  }
  float r = (b1 ? r2[1] : r2[0]) + __shfl_xor(b1 ? r2[0] : r2[1], 2, GVD_WAVE);
#pragma unroll
  for (int off = 4; off < 64; off <<= 1) r += __shfl_xor(r, off, GVD_WAVE);
  return r;
}

__global__ __launch_bounds__(PD_NT, 2) void greedy_persistent_kernel(const PdParams p) {
  // activations of the products, per sample: [att+att2 | h_att | h_lang] (the lang-LSTM input order), and xt
  __shared__ __attribute__((aligned(16))) float s_act[PD_MB][3 * PD_H];
  __shared__ __attribute__((aligned(16))) float s_xt[PD_MB][PD_E];
  __shared__ __attribute__((aligned(16))) float s_wlog[PD_RPW][PD_H];   // this workgroup's vocabulary rows
  __shared__ float s_red[2][4][16];
  __shared__ float s_score[PD_MAXCH];
  __shared__ float s_sc[PD_MAXNCT];
  __shared__ float s_stat[24];
  __shared__ __attribute__((aligned(16))) f32x4 s_half[256];
  __shared__ float s_logit[PD_MB][PD_RPW + 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: row / K-half offsets become SGPR math
  const int u = wave & 3, kh = wave >> 2;
  const int wg = blockIdx.x;
  const int B = p.B, R = p.R, Ft = p.Ft, V = p.V, L = p.L;
  const int nct = p.nch_r + p.nch_t;
  const int rpw = (V + PD_G - 1) / PD_G;
  unsigned round = 0;
  bool dead = false;              // latched barrier timeout (thread 0)

  const __amdgpu_buffer_rsrc_t rs_hatt = gvd_rsrc(p.h_att), rs_hlang = gvd_rsrc(p.h_lang), rs_q = gvd_rsrc(p.q12),
                               rs_sum = gvd_rsrc(p.att_sum), rs_pctx = gvd_rsrc(p.part_ctx),
                               rs_pml = gvd_rsrc(p.part_ml), rs_stats = gvd_rsrc(p.stats);

  // ---------------------------------------------------------------- resident weights
  const int j = 4 * wg + u;                       // hidden unit of this wave (both LSTMs)
  // LSTM weights as gate PAIRS (.x = gate 2gp, .y = gate 2gp+1) so that one v_pk_fma_f32 feeds two gate accumulators
  // from one broadcast activation (the VALU FMA rate, not memory, bounds these products: 16 rows x 3072 x 4 samples).
  // The language LSTM (96 registers) stays resident for the whole launch; the attention LSTM's 48 are re-fetched from
  // L2 every step during P6 (prefetch for the next P1), which keeps them out of the register-hungry attention phase.
  f32x2 w_lang[2][6][4], w_att[2][3][4];
  f32x4 w_q[2];
#pragma unroll
  for (int gp = 0; gp < 2; ++gp) {
    const int64_t row0 = (int64_t)(2 * gp) * PD_H + j, row1 = row0 + PD_H;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int k = 256 * (kh * 6 + i) + 4 * lane;            // index into [att_sum | h_att | h_lang]
      const float* s0 = k < 2 * PD_H ? p.lang_w_ih + row0 * (2 * PD_H) + k : p.lang_w_hh + row0 * PD_H + (k - 2 * PD_H);
      const float* s1 = k < 2 * PD_H ? p.lang_w_ih + row1 * (2 * PD_H) + k : p.lang_w_hh + row1 * PD_H + (k - 2 * PD_H);
      const f32x4 a = *reinterpret_cast<const f32x4*>(s0), b = *reinterpret_cast<const f32x4*>(s1);
#pragma unroll
      for (int c = 0; c < 4; ++c) w_lang[gp][i][c] = f32x2{a[c], b[c]};
    }
  }
  // buffer loads with scalar row offsets: no per-lane 64-bit addresses to keep alive across the token loop
  const __amdgpu_buffer_rsrc_t rs_wih = gvd_rsrc(p.att_w_ih), rs_whh = gvd_rsrc(p.att_w_hh);
  auto load_w_att = [&]() {
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int row0 = (2 * gp) * PD_H + j, row1 = row0 + PD_H;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int k0 = 256 * (kh * 3 + i);                    // block start in [xt | h_att]  (wave-uniform)
        f32x4 a, b;
        if (k0 < PD_E) {
          a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wih, 16 * lane, (row0 * (PD_H + PD_E) + PD_H + k0) * 4, 0));
          b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wih, 16 * lane, (row1 * (PD_H + PD_E) + PD_H + k0) * 4, 0));
        } else {
          a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_whh, 16 * lane, (row0 * PD_H + k0 - PD_E) * 4, 0));
          b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_whh, 16 * lane, (row1 * PD_H + k0 - PD_E) * 4, 0));
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) w_att[gp][i][c] = f32x2{a[c], b[c]};
      }
    }
  };
  load_w_att();
  const int qrow = 4 * wg + u;                    // row of the stacked query projection [W_att ; W_att2]
#pragma unroll
  for (int i = 0; i < 2; ++i)
    w_q[i] = *reinterpret_cast<const f32x4*>(p.q_w + (int64_t)qrow * PD_H + 512 * kh + 256 * i + 4 * lane);
  for (int idx = tid; idx < PD_RPW * (PD_H / 4); idx += PD_NT) {
    const int rl = idx / (PD_H / 4), c4 = idx % (PD_H / 4);
    const int n = wg * rpw + rl;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (rl < rpw && n < V) v = *reinterpret_cast<const f32x4*>(p.logit_w + (int64_t)n * PD_H + 4 * c4);
    *reinterpret_cast<f32x4*>(&s_wlog[rl][4 * c4]) = v;
  }
  float vb = 0.f;                                 // logit bias of the row this lane publishes in P6 (lane < 12)
  {
    const int rl = wave + 8 * (lane >> 2), n = wg * rpw + rl;
    if (lane < 12 && rl < rpw && n < V) vb = p.logit_b[n];
  }
  // epilogue lanes: wave (u, kh = 0), lane m < B owns (sample m, unit j): cell states and constant gate terms
  const bool epi = kh == 0 && lane < B;
  float c_att = 0.f, c_lang = 0.f, fcg[4] = {0.f, 0.f, 0.f, 0.f}, lb[4] = {0.f, 0.f, 0.f, 0.f}, qb = 0.f;
  if (epi) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      fcg[g] = p.fc_gates[(int64_t)lane * 4 * PD_H + g * PD_H + j];
      lb[g] = p.lang_b_ih[g * PD_H + j] + p.lang_b_hh[g * PD_H + j];
    }
    qb = p.q_b[qrow];
  }
  // zero state; BOS token 0 (model.py:588): xt = relu(embed[0])
  for (int idx = tid; idx < PD_MB * 3 * PD_H; idx += PD_NT) (&s_act[0][0])[idx] = 0.f;
  for (int idx = tid; idx < PD_MB * PD_E; idx += PD_NT) (&s_xt[0][0])[idx] = fmaxf(p.embed[idx % PD_E], 0.f);
  __syncthreads();

  int n_stamp = 0;
#define PD_STAMP() do { if (p.trace && wg == 0 && tid == 0) p.trace[n_stamp++] = wall_clock64(); } while (0)
  PD_STAMP();
  for (int t = 0; t < L; ++t) {
    // ============================================================ P1: attention LSTM
    {
      f32x2 acc2[2][PD_MB];
#pragma unroll
      for (int gp = 0; gp < 2; ++gp)
#pragma unroll
        for (int m = 0; m < PD_MB; ++m) acc2[gp][m] = f32x2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int kb = kh * 3 + i;                           // 256-float block of [xt | h_att]
#pragma unroll
        for (int m = 0; m < PD_MB; ++m) {
          const float* ap = kb < 2 ? &s_xt[m][256 * kb + 4 * lane] : &s_act[m][PD_H + 256 * (kb - 2) + 4 * lane];
          const f32x4 a = *reinterpret_cast<const f32x4*>(ap);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp)
              acc2[gp][m] = __builtin_elementwise_fma(w_att[gp][i][c], f32x2{a[c], a[c]}, acc2[gp][m]);
        }
      }
      float acc[16];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int m = 0; m < PD_MB; ++m) acc[g * 4 + m] = acc2[g >> 1][m][g & 1];
      const float r = reduce16(acc, lane);
      if (lane < 16) s_red[kh][u][lane] = r;
      __syncthreads();
      if (epi) {
        const int m = lane;
        float gt[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) gt[g] = s_red[0][u][g * 4 + m] + s_red[1][u][g * 4 + m] + fcg[g];
        const float gi = sigmoid_f(gt[0]), gf = sigmoid_f(gt[1]), gg = tanhf(gt[2]), go = sigmoid_f(gt[3]);
        c_att = gf * c_att + gi * gg;
        st_agent_f32(rs_hatt, (unsigned)(m * PD_H + j) * 4, go * tanhf(c_att));
      }
    }
    grid_barrier_tree(p.sync, round++, PD_G, dead);
    PD_STAMP();

    // ============================================================ P2: attention queries
    {
      // stage h_att (all samples) into its LDS slot: also the lang-LSTM input and the next step's recurrence
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * PD_NT;                      // float4 index in [PD_MB][256]
        const int m = idx >> 8, c4 = idx & 255;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < B) v = ld_agent_x4(rs_hatt, (unsigned)(m * PD_H + 4 * c4) * 4);
        *reinterpret_cast<f32x4*>(&s_act[m][PD_H + 4 * c4]) = v;
      }
      __syncthreads();
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < PD_MB; ++m)
          acc[m] = dot4(w_q[i], *reinterpret_cast<const f32x4*>(&s_act[m][PD_H + 512 * kh + 256 * i + 4 * lane]), acc[m]);
      const float r = reduce4(acc, lane);
      if (lane < 4) s_red[kh][u][lane] = r;
      __syncthreads();
      if (epi) st_agent_f32(rs_q, (unsigned)(lane * 2 * PD_A + qrow) * 4, s_red[0][u][lane] + s_red[1][u][lane] + qb);
    }
    grid_barrier_tree(p.sync, round++, PD_G, dead);
    PD_STAMP();

    // ============================================================ P3: attention partials
    for (int item = wg; item < B * nct; item += PD_G) {
      const int b = item / nct, c = item - b * nct;
      const bool tmp = c >= p.nch_r;                           // temporal attention chunk
      const int cc = tmp ? c - p.nch_r : c;
      const int chunk = tmp ? p.chunk_t : p.chunk_r, N = tmp ? Ft : R;
      const int n0 = cc * chunk;
      const int rows = min(chunk, N - n0);
      const float* pf = (tmp ? p.p_conv : p.p_pool) + ((int64_t)b * N + n0) * PD_A;
      const float* ff = (tmp ? p.conv : p.pool) + ((int64_t)b * N + n0) * PD_H;
      const float* wv = tmp ? p.a1_w : p.a2_w;
      const float ab = tmp ? *p.a1_b : *p.a2_b;
      const unsigned qoff = (unsigned)(b * 2 * PD_A + (tmp ? 0 : PD_A)) * 4;
      // pre-scaled queries / folded weights of tanh_fast (gvd_common.h): same arithmetic as attn_partial_kernel
      const f32x4 q0 = GVD_TWO_LOG2E * ld_agent_x4(rs_q, qoff + 16 * lane);
      const f32x4 q1 = GVD_TWO_LOG2E * ld_agent_x4(rs_q, qoff + 1024 + 16 * lane);
      const AttnLaneW W = attn_lane_w(wv, lane);
      const uint8_t* am = tmp ? nullptr : p.pnt_mask + (int64_t)b * (R + 1) + 1 + n0;
      float* lo = tmp ? nullptr : p.att2_weights + ((int64_t)b * L + t) * R + n0;
      // context phase role: thread = (4 columns of H = 1024, row parity).  Its first 4 feature rows do not depend on
      // the scores: request them now so their HBM latency hides behind the scoring pass
      const int c4 = tid & 255, par = tid >> 8;
      const float* fb = ff + 4 * c4;
      f32x4 v0[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        v0[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (par + 2 * x < rows) v0[x] = *reinterpret_cast<const f32x4*>(fb + (int64_t)(par + 2 * x) * PD_H);
      }
      // scores: 8 waves x 2 rows per pass; the lane owns columns [4 lane, +4) and [256 + 4 lane, +4) of A = 512
      for (int r = wave * 2; r < rows; r += 16) {
        const bool two = (r + 1) < rows;
        const float* p0 = pf + (int64_t)r * PD_A;
        const float* p1 = two ? p0 + PD_A : p0;
        const f32x4 x00 = *reinterpret_cast<const f32x4*>(p0 + 4 * lane);
        const f32x4 x01 = *reinterpret_cast<const f32x4*>(p0 + 256 + 4 * lane);
        const f32x4 x10 = *reinterpret_cast<const f32x4*>(p1 + 4 * lane);
        const f32x4 x11 = *reinterpret_cast<const f32x4*>(p1 + 256 + 4 * lane);
        float s0 = attn_score_lane(x00, x01, q0, q1, W);
        float s1 = attn_score_lane(x10, x11, q0, q1, W);
        s0 = wave_sum(s0) + ab;
        s1 = wave_sum(s1) + ab;
        if (lane == 0) {
          const float e0 = (am && am[r]) ? GVD_MIN_VALUE : s0;
          s_score[r] = e0;
          if (lo) lo[r] = e0;
          if (two) {
            const float e1 = (am && am[r + 1]) ? GVD_MIN_VALUE : s1;
            s_score[r + 1] = e1;
            if (lo) lo[r + 1] = e1;
          }
        }
      }
      __syncthreads();
      // chunk-local softmax numerators (rows <= 64: wave 0 holds them)
      float mloc = -INFINITY;
      if (tid < rows) mloc = s_score[tid];
      mloc = wave_max(mloc);
      if (tid == 0) s_stat[0] = mloc;
      __syncthreads();
      mloc = s_stat[0];
      float pr = 0.f;
      if (tid < rows) pr = expf(s_score[tid] - mloc);
      __syncthreads();
      if (tid < rows) s_score[tid] = pr;
      const float lsum = wave_sum(pr);
      if (tid == 0) {
        st_agent_f32(rs_pml, (unsigned)(b * nct + c) * 8, mloc);
        st_agent_f32(rs_pml, (unsigned)(b * nct + c) * 8 + 4, lsum);
      }
      __syncthreads();
      // partial context
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int r0 = par + 2 * x;
          const float pw = r0 < rows ? s_score[r0] : 0.f;
          acc[0] = fmaf(pw, v0[x][0], acc[0]); acc[1] = fmaf(pw, v0[x][1], acc[1]);
          acc[2] = fmaf(pw, v0[x][2], acc[2]); acc[3] = fmaf(pw, v0[x][3], acc[3]);
        }
        int r = par + 8;
        for (; r + 6 < rows; r += 8) {
          f32x4 v[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) v[x] = *reinterpret_cast<const f32x4*>(fb + (int64_t)(r + 2 * x) * PD_H);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const float pw = s_score[r + 2 * x];
            acc[0] = fmaf(pw, v[x][0], acc[0]); acc[1] = fmaf(pw, v[x][1], acc[1]);
            acc[2] = fmaf(pw, v[x][2], acc[2]); acc[3] = fmaf(pw, v[x][3], acc[3]);
          }
        }
        for (; r < rows; r += 2) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(fb + (int64_t)r * PD_H);
          const float pw = s_score[r];
          acc[0] = fmaf(pw, v[0], acc[0]); acc[1] = fmaf(pw, v[1], acc[1]);
          acc[2] = fmaf(pw, v[2], acc[2]); acc[3] = fmaf(pw, v[3], acc[3]);
        }
        if (par) s_half[c4] = acc;
        __syncthreads();
        if (!par) {
          const f32x4 o = s_half[c4];
          acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
          st_agent_x4(rs_pctx, (unsigned)(((int64_t)b * nct + c) * PD_H + 4 * c4) * 4, acc);
        }
      }
      __syncthreads();   // s_score / s_half are reused by the next item
    }
    grid_barrier_tree(p.sync, round++, PD_G, dead);
    PD_STAMP();

    // ============================================================ P4: combine chunk partials -> att + att2
    {
      const int b = wg >> 6, cb = wg & 63;                     // sample, block of 16 columns
      if (b < B) {
        // per-chunk rescale factors exp(m_c - M_side) / L_side (the side's softmax normaliser folded in)
        // thread = (column quad cq, chunk slot cs): 64-byte pieces of the partial contexts; the first piece is
        // requested before the (m, l) round trip so that the two coherent-load latencies overlap
        const int cq = tid & 3, cs = tid >> 2;
        f32x4 v_first = {0.f, 0.f, 0.f, 0.f};
        if (cs < nct) v_first = ld_agent_x4(rs_pctx, (unsigned)(((int64_t)b * nct + cs) * PD_H + 16 * cb + 4 * cq) * 4);
        float mc = -INFINITY, lc = 0.f;
        const bool has = tid < nct;
        const bool side1 = tid >= p.nch_r;
        if (has) {
          mc = ld_agent_f32(rs_pml, (unsigned)(b * nct + tid) * 8);
          lc = ld_agent_f32(rs_pml, (unsigned)(b * nct + tid) * 8 + 4);
        }
        float m0 = wave_max((has && !side1) ? mc : -INFINITY), m1 = wave_max((has && side1) ? mc : -INFINITY);
        if (lane == 0) { s_stat[wave] = m0; s_stat[8 + wave] = m1; }
        __syncthreads();
        m0 = s_stat[0]; m1 = s_stat[8];
#pragma unroll
        for (int w = 1; w < 8; ++w) { m0 = fmaxf(m0, s_stat[w]); m1 = fmaxf(m1, s_stat[8 + w]); }
        const float e = has ? expf(mc - (side1 ? m1 : m0)) : 0.f;
        float l0 = wave_sum((has && !side1) ? e * lc : 0.f), l1 = wave_sum((has && side1) ? e * lc : 0.f);
        __syncthreads();
        if (lane == 0) { s_stat[wave] = l0; s_stat[8 + wave] = l1; }
        __syncthreads();
        l0 = 0.f; l1 = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { l0 += s_stat[w]; l1 += s_stat[8 + w]; }
        if (has) s_sc[tid] = e / (side1 ? l1 : l0);
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (cs < nct) {
          const float sc = s_sc[cs];
          acc[0] = sc * v_first[0]; acc[1] = sc * v_first[1]; acc[2] = sc * v_first[2]; acc[3] = sc * v_first[3];
        }
        for (int c = cs + PD_NT / 4; c < nct; c += PD_NT / 4) {
          const f32x4 v = ld_agent_x4(rs_pctx, (unsigned)(((int64_t)b * nct + c) * PD_H + 16 * cb + 4 * cq) * 4);
          const float sc = s_sc[c];
          acc[0] = fmaf(sc, v[0], acc[0]); acc[1] = fmaf(sc, v[1], acc[1]);
          acc[2] = fmaf(sc, v[2], acc[2]); acc[3] = fmaf(sc, v[3], acc[3]);
        }
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += __shfl_xor(acc[k], off, GVD_WAVE);
        }
        if (lane < 4) s_half[wave * 4 + lane] = acc;
        __syncthreads();
        if (tid < 4) {
          f32x4 tot = s_half[tid];
#pragma unroll
          for (int w = 1; w < 8; ++w) {
            const f32x4 o = s_half[w * 4 + tid];
            tot[0] += o[0]; tot[1] += o[1]; tot[2] += o[2]; tot[3] += o[3];
          }
          st_agent_x4(rs_sum, (unsigned)(b * PD_H + 16 * cb + 4 * tid) * 4, tot);
        }
      }
    }
    grid_barrier_tree(p.sync, round++, PD_G, dead);
    PD_STAMP();

    // ============================================================ P5: language LSTM
    {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * PD_NT;
        const int m = idx >> 8, c4 = idx & 255;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < B) v = ld_agent_x4(rs_sum, (unsigned)(m * PD_H + 4 * c4) * 4);
        *reinterpret_cast<f32x4*>(&s_act[m][4 * c4]) = v;
      }
      __syncthreads();
      f32x2 acc2[2][PD_MB];
#pragma unroll
      for (int gp = 0; gp < 2; ++gp)
#pragma unroll
        for (int m = 0; m < PD_MB; ++m) acc2[gp][m] = f32x2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int kb = kh * 6 + i;
#pragma unroll
        for (int m = 0; m < PD_MB; ++m) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(&s_act[m][256 * kb + 4 * lane]);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp)
              acc2[gp][m] = __builtin_elementwise_fma(w_lang[gp][i][c], f32x2{a[c], a[c]}, acc2[gp][m]);
        }
        __builtin_amdgcn_sched_barrier(0);   // one K-block at a time: 24 hoisted LDS reads would not fit next to the weights
      }
      float acc[16];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int m = 0; m < PD_MB; ++m) acc[g * 4 + m] = acc2[g >> 1][m][g & 1];
      const float r = reduce16(acc, lane);
      if (lane < 16) s_red[kh][u][lane] = r;
      __syncthreads();
      if (epi) {
        const int m = lane;
        float gt[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) gt[g] = s_red[0][u][g * 4 + m] + s_red[1][u][g * 4 + m] + lb[g];
        const float gi = sigmoid_f(gt[0]), gf = sigmoid_f(gt[1]), gg = tanhf(gt[2]), go = sigmoid_f(gt[3]);
        c_lang = gf * c_lang + gi * gg;
        st_agent_f32(rs_hlang, (unsigned)(m * PD_H + j) * 4, go * tanhf(c_lang));
      }
    }
    grid_barrier_tree(p.sync, round++, PD_G, dead);
    PD_STAMP();

    // ============================================================ P6: vocabulary logits of this workgroup's rows
    {
      f32x4 hv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * PD_NT;
        const int m = idx >> 8, c4 = idx & 255;
        hv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m < B) hv[i] = ld_agent_x4(rs_hlang, (unsigned)(m * PD_H + 4 * c4) * 4);
      }
      // next step's attention-LSTM weights: requested AFTER the state loads (memory returns in order) and in flight
      // during P6/P7
      load_w_att();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * PD_NT;
        *reinterpret_cast<f32x4*>(&s_act[idx >> 8][2 * PD_H + 4 * (idx & 255)]) = hv[i];
      }
      __syncthreads();
      float acc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        f32x4 a[PD_MB];
#pragma unroll
        for (int m = 0; m < PD_MB; ++m) a[m] = *reinterpret_cast<const f32x4*>(&s_act[m][2 * PD_H + 256 * jb + 4 * lane]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int rl = wave + 8 * i;                         // local vocabulary row (wave-uniform)
          if (rl < rpw) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(&s_wlog[rl][256 * jb + 4 * lane]);
#pragma unroll
            for (int m = 0; m < PD_MB; ++m) acc[i * 4 + m] = dot4(w, a[m], acc[i * 4 + m]);
          }
        }
      }
      const float r = reduce16(acc, lane);
      if (lane < 12) {
        const int i = lane >> 2, m = lane & 3, rl = wave + 8 * i, n = wg * rpw + rl;
        if (rl < rpw) s_logit[m][rl] = n < V ? r + vb : -INFINITY;
      }
      __syncthreads();
      if (wave < B) {
        // record of sample `wave`: {max, sum exp(x - max), top-1 value, top-1 id, top-2 value, top-2 id}; lane = row
        const int n = wg * rpw + lane;
        const bool valid = lane < rpw && n < V;
        const float x = valid ? s_logit[wave][lane] : -INFINITY;
        Top2 tp = {x, valid ? n : 0x7fffffff, -INFINITY, 0x7fffffff};
        tp = top2_wave(tp);
        const float se = wave_sum(valid ? expf(x - tp.v1) : 0.f);
        if (lane == 0) {
          const f32x4 a = {tp.v1, se, tp.v1, __int_as_float(tp.i1)};
          const f32x4 c = {tp.v2, __int_as_float(tp.i2), 0.f, 0.f};
          const unsigned off = (unsigned)((wave * PD_G + wg) * PD_STAT) * 4;
          st_agent_x4(rs_stats, off, a);
          st_agent_x4(rs_stats, off + 16, c);
        }
      }
    }
    grid_barrier_tree(p.sync, round++, PD_G, dead);
    PD_STAMP();

    // ============================================================ P7: token rule + next input (no barrier needed)
    if (wave < B) {
      const int m = wave;
      float MX = -INFINITY, SE = 0.f;
      Top2 tp = {-INFINITY, 0x7fffffff, -INFINITY, 0x7fffffff};
#pragma unroll
      for (int k = 0; k < PD_G / 64; ++k) {
        const int w2 = lane + 64 * k;
        if (w2 * rpw < V) {                                    // workgroups past the vocabulary hold no rows
          const unsigned off = (unsigned)((m * PD_G + w2) * PD_STAT) * 4;
          const f32x4 a = ld_agent_x4(rs_stats, off), c = ld_agent_x4(rs_stats, off + 16);
          const float mx = a[0], se = a[1];
          if (mx > MX) { SE = SE * expf(MX - mx) + se; MX = mx; }
          else SE += se * expf(mx - MX);
          top2_insert(tp, a[2], __float_as_int(a[3]));
          top2_insert(tp, c[0], __float_as_int(c[1]));
        }
      }
      const float MXw = wave_max(MX);
      SE = wave_sum(MX == -INFINITY ? 0.f : SE * expf(MX - MXw));
      tp = top2_wave(tp);
      const float lse = logf(SE);
      const bool keep = tp.i1 != p.unk;
      const int it = keep ? tp.i1 : tp.i2;
      if (wg == 0 && lane == 0) {
        p.seq[(int64_t)m * L + t] = it;
        p.seq_lp[(int64_t)m * L + t] = ((keep ? tp.v1 : tp.v2) - MXw) - lse;
      }
      const float* e = p.embed + (int64_t)min(max(it, 0), V - 1) * PD_E;   // (a NaN row must not turn into a wild read)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v = *reinterpret_cast<const f32x4*>(e + 256 * i + 4 * lane);
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
        *reinterpret_cast<f32x4*>(&s_xt[m][256 * i + 4 * lane]) = v;
      }
    }
    __syncthreads();
    PD_STAMP();
  }
#undef PD_STAMP

  // a barrier that timed out anywhere invalidates the results: report it and poison the token ids
  if (wg == 0 && tid == 0) {
    const unsigned err = __hip_atomic_load(p.sync + GVD_SYNC_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p.status) *p.status = (int)err;
    if (err)
      for (int i = 0; i < B * L; ++i) p.seq[i] = -1;
  }
}

}  // namespace

namespace {

size_t pd_carve(PdParams* p, void* workspace, int nct) {
  char* b = reinterpret_cast<char*>(workspace);
  size_t off = 0;
  auto take = [&](size_t bytes) { void* q = b ? b + off : nullptr; off += (bytes + 255) & ~size_t(255); return q; };
  unsigned* sync = (unsigned*)take((size_t)GVD_SYNC_WORDS * sizeof(unsigned));
  float* h_att = (float*)take((size_t)PD_MB * PD_H * 4);
  float* h_lang = (float*)take((size_t)PD_MB * PD_H * 4);
  float* q12 = (float*)take((size_t)PD_MB * 2 * PD_A * 4);
  float* att_sum = (float*)take((size_t)PD_MB * PD_H * 4);
  float* part_ctx = (float*)take((size_t)PD_MB * nct * PD_H * 4);
  float* part_ml = (float*)take((size_t)PD_MB * nct * 2 * 4);
  float* stats = (float*)take((size_t)PD_MB * PD_G * PD_STAT * 4);
  if (p) {
    p->sync = sync; p->h_att = h_att; p->h_lang = h_lang; p->q12 = q12; p->att_sum = att_sum;
    p->part_ctx = part_ctx; p->part_ml = part_ml; p->stats = stats;
  }
  return off;
}

}  // namespace

size_t gvd_pd_workspace_bytes() { return pd_carve(nullptr, nullptr, PD_MAXNCT); }

bool gvd_pd_shape_ok(int B, int H, int A, int E, int V, int R, int Ft) {
  return B >= 1 && B <= PD_MB && H == PD_H && A == PD_A && E == PD_E && V >= 2 && V <= PD_G * PD_RPW && R >= 1 &&
         Ft >= 1;
}

bool gvd_pd_eligible(int B, int H, int A, int E, int V, int R, int Ft) {
  const char* env = getenv("GVD_PERSISTENT");            // read per call: tests A/B the two decode paths in one process
  const int enabled = env ? atoi(env) : 1;
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  return enabled && cus >= PD_G && gvd_pd_shape_ok(B, H, A, E, V, R, Ft);
}

int gvd_pd_launch(PdParams p, void* workspace, hipStream_t st) {
  // attention chunking: about 256 / B work items per sample (one per workgroup), the same number of rows per item for
  // the region and the temporal attention so no workgroup holds the phase up, chunks of at most 64 rows
  const int target = PD_G / p.B;
  int chunk = (p.R + p.Ft + target - 1) / target;
  if (chunk < 1) chunk = 1;
  while (chunk < PD_MAXCH && (p.R + chunk - 1) / chunk + (p.Ft + chunk - 1) / chunk > target) ++chunk;
  if (chunk > PD_MAXCH) chunk = PD_MAXCH;
  p.chunk_r = chunk < p.R ? chunk : p.R;
  p.chunk_t = chunk < p.Ft ? chunk : p.Ft;
  p.nch_r = (p.R + p.chunk_r - 1) / p.chunk_r;
  p.nch_t = (p.Ft + p.chunk_t - 1) / p.chunk_t;
  const int nct = p.nch_r + p.nch_t;
  if (nct > PD_MAXNCT) return GVD_EINVAL;
  pd_carve(&p, workspace, nct);
  hipError_t e = hipMemsetAsync(p.sync, 0, (size_t)GVD_SYNC_WORDS * sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  if (gvd_spin_limit_env()) {      // test aid: forced barrier timeouts (GVD_SPIN_LIMIT)
    e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.sync + GVD_SYNC_LIMIT), (int)gvd_spin_limit_env(), 1, st);
    if (e != hipSuccess) return (int)e;
  }
  void* args[] = {&p};
  // Plain launch after an explicit co-residency check (gvd_grid_fits: what the cooperative launch verifies), because the
  // cooperative path costs a ~12 us dispatch gap on either side of the kernel; GVD_COOP_LAUNCH=1 restores it.  The grid barrier
  // bounds its spins either way (status word -> GvdHipError on the host).
  static const bool coop = getenv("GVD_COOP_LAUNCH") ? atoi(getenv("GVD_COOP_LAUNCH")) != 0 : false;
  const void* fn = reinterpret_cast<const void*>(greedy_persistent_kernel);
  if (coop) {
    e = hipLaunchCooperativeKernel(fn, dim3(PD_G), dim3(PD_NT), args, 0, st);
  } else {
    static const bool fits = gvd_grid_fits(fn, PD_NT, PD_G);
    if (!fits) return GVD_EINVAL;                  // the caller falls back to the kernel-per-op loop
    e = hipLaunchKernel(fn, dim3(PD_G), dim3(PD_NT), args, 0, st);
  }
  return e == hipSuccess ? 0 : (int)e;
}
