// Additive visual attention of the GVD decoder step — the HBM-bound north-star kernel.
//
//   e[n]   = w . tanh(p_feats[b,n,:] + q[b,:]) + alpha_b ;  e[att_mask] = -1e8
//   alpha  = softmax_n(e) ;  ctx[b,:] = sum_n alpha[n] feats[b,n,:]
//   logits_out[b,n] = e[n] with [pnt_mask] = -1e8
//
// Reference: Attention2.forward (region attention, AttModel.py:71-108) and Attention.forward (temporal
// attention, AttModel.py:33-53).  The reference materialises att+q, tanh(.), the scores, the softmax and a
// bmm as separate [B,N,512]-sized ATen ops (about 3x the traffic); here every byte of p_feats [B,N,A] and
// feats [B,N,H] is read from HBM exactly once per step and nothing of size [B,N,*] is written.
//
// Decomposition (MI355X): the N axis is cut into chunks; one 256-thread workgroup per (chunk, sample),
// both attentions (region chunks first, then temporal chunks) in ONE launch so the grid has >> 256
// workgroups.  Phase 1: each wave scores rows — the lane owns 8 fixed columns of the A=512 projection
// (two coalesced float4 loads per row; its 8 q and 8 w values live in registers after one LDS-staged
// broadcast), 64-lane xor-shuffle reduction per row.  Phase 2: chunk-local softmax numerators (online
// softmax partial: max m, sum l), then each thread owns 4 of the H=1024 feature columns and streams the
// chunk's rows with coalesced float4 loads, 8 rows in flight.  A second tiny kernel merges the per-chunk
// (m, l, ctx) partials of both attentions and emits att+att2 for the language LSTM.
// Algorithmic bytes per sample-call: N*(A+H)*4 (+2N mask bytes) — SURVEY.md §8d.
#include "gvd_common.h"
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int ATT_A = 512;
constexpr int ATT_H = 1024;
constexpr int MAX_CHUNK = 64;   // rows per workgroup (LDS score buffer)

struct SideDev {
  const float* feats; const float* p_feats; const float* q; int64_t ldq;
  const float* w; const float* alpha_bias;
  const uint8_t* att_mask; int64_t ld_att_mask;
  const uint8_t* pnt_mask; int64_t ld_pnt_mask;
  float* logits_out; int64_t ld_logits;
  float* scores_out; int64_t ld_scores;
  const int* row_map;    // optional [B,N]: row n of sample b lives at row row_map[b*N+n] of the FLAT feats / p_feats
  int N, chunk, nchunks, group;
};

struct FwdParams {
  SideDev side[2];
  int nside;
  float* part_ctx;   // [B, NCtot, H]
  float* part_ml;    // [B, NCtot, 2]
  int nctot;
};

// workgroup (chunk c of side s, sample b)
// streaming load of data that is read exactly once per launch.  (A run-time `nt ? nontemporal : plain` select folds
// into ONE plain load - the hint is lost - so the choice has to be made at compile time.)
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T* ptr) {
  if constexpr (NT) return __builtin_nontemporal_load(ptr);
  else return *ptr;
}

// the region side's score function (gvd_attn_side.score_mode); the temporal side of the same launch is always additive.
// MODE = GVD_SCORE_ADD compiles to attn_score_lane alone: the README configuration's kernels are unchanged.
template <int MODE>
__device__ __forceinline__ float score_sel(bool addm, f32x4 x0, f32x4 x1, f32x4 q0, f32x4 q1, const AttnLaneW& W) {
  if constexpr (MODE == GVD_SCORE_ADD) return attn_score_lane(x0, x1, q0, q1, W);
  else return addm ? attn_score_lane(x0, x1, q0, q1, W) : attn_score_lane_m<MODE>(x0, x1, q0, q1, W);
}
template <int MODE>
__device__ __forceinline__ AttnLaneW lane_w_sel(bool addm, const float* w, int lane) {
  if constexpr (MODE == GVD_SCORE_DOT) {
    if (!addm) { AttnLaneW o; o.wn0 = f32x4{0.f, 0.f, 0.f, 0.f}; o.wn1 = o.wn0; o.wsum = 0.f; return o; }
  }
  return attn_lane_w(w, lane);
}

template <bool NT, int MODE>
__global__ __launch_bounds__(256, 8) void attn_partial_kernel(const FwdParams p) {   // 8 waves / SIMD (<= 64 VGPRs, no spill)
  __shared__ float s_score[MAX_CHUNK];
  __shared__ float s_red[8];
  __shared__ int s_live[MAX_CHUNK];
  __shared__ int s_src[MAX_CHUNK];    // row-in-chunk -> row offset from pf / fb (identity unless the side has a row map)
  __shared__ int s_n[2];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.y;
  int c = blockIdx.x;
  const int sidx = (p.nside > 1 && c >= p.side[0].nchunks) ? 1 : 0;
  const SideDev& S = p.side[sidx];
  const int cglob = c;
  if (sidx) c -= p.side[0].nchunks;
  const int n0 = c * S.chunk;
  const int rows = min(S.chunk, S.N - n0);

  // ---- phase 1: scores.  lane owns columns [4*lane, 4*lane+4) and [256+4*lane, ...+4) of A = 512
  const float* qb = S.q + (int64_t)b * S.ldq;
  const bool addm = MODE == GVD_SCORE_ADD || sidx == 1;      // (compile-time true for the README configuration)
  const float qscale = (MODE == GVD_SCORE_DOT && !addm) ? 1.0f : GVD_TWO_LOG2E;
  const f32x4 q0 = qscale * *reinterpret_cast<const f32x4*>(qb + 4 * lane);         // pre-scaled: see tanh_fast
  const f32x4 q1 = qscale * *reinterpret_cast<const f32x4*>(qb + 256 + 4 * lane);
  const AttnLaneW W = lane_w_sel<MODE>(addm, S.w, lane);
  const float ab = (MODE == GVD_SCORE_DOT && !addm) ? 0.f : *S.alpha_bias;
  const int fbi = S.group > 1 ? b / S.group : b;   // beams of one sample share its features
  // compacted features (masked-proposal compaction, csrc/compact.hip): the side's rows are looked up through row_map in
  // the flat [rows, .] arrays - the dense [B,N,.] copies are never materialised
  const int* rmap = S.row_map ? S.row_map + (int64_t)fbi * S.N + n0 : nullptr;
  const float* pf = rmap ? S.p_feats : S.p_feats + ((int64_t)fbi * S.N + n0) * ATT_A;
  const uint8_t* am = S.att_mask ? S.att_mask + (int64_t)b * S.ld_att_mask + n0 : nullptr;
  const uint8_t* pm = S.pnt_mask ? S.pnt_mask + (int64_t)b * S.ld_pnt_mask + n0 : nullptr;
  float* lo = S.logits_out ? S.logits_out + (int64_t)b * S.ld_logits + n0 : nullptr;
  float* so = S.scores_out ? S.scores_out + (int64_t)b * S.ld_scores + n0 : nullptr;

  // Rows the attention mask removes (att_mask = 1: proposals below the score threshold / padding) get the score -1e8
  // whatever their features are, and - as soon as the chunk holds one live row - the softmax weight exp(-1e8 - m) = 0
  // exactly: neither their projection row nor their feature row is fetched.  Wave 0 compacts the chunk's live rows into
  // s_live (the chunk has <= 64 rows: one ballot); a chunk without any live row keeps all its rows in the context pass
  // (uniform weights; only matters when the whole sample is masked).
  if (wave == 0) {
    const bool in = lane < rows;
    const bool live = in && !(am && am[lane]);
    const unsigned long long bal = __ballot(live);
    const int nl = __popcll(bal);
    if (live) s_live[__popcll(bal & ((1ull << lane) - 1ull))] = lane;
    if (in) s_src[lane] = rmap ? rmap[lane] : lane;
    if (in && !live) {
      s_score[lane] = GVD_MIN_VALUE;
      if (so) so[lane] = GVD_MIN_VALUE;
      if (lo) lo[lane] = GVD_MIN_VALUE;
    }
    if (nl == 0 && in) s_live[lane] = lane;
    if (lane == 0) { s_n[0] = nl; s_n[1] = nl ? nl : rows; }
  }
  __syncthreads();
  const int nlive = s_n[0];
  for (int i = wave * 2; i < nlive; i += 8) {
    const bool two = (i + 1) < nlive;
    const int r = s_live[i], r2 = two ? s_live[i + 1] : r;
    const float* p0 = pf + (int64_t)s_src[r] * ATT_A;
    const float* p1 = pf + (int64_t)s_src[r2] * ATT_A;
    f32x4 x00 = ld_stream<NT>(reinterpret_cast<const f32x4*>(p0 + 4 * lane));
    f32x4 x01 = ld_stream<NT>(reinterpret_cast<const f32x4*>(p0 + 256 + 4 * lane));
    f32x4 x10 = ld_stream<NT>(reinterpret_cast<const f32x4*>(p1 + 4 * lane));
    f32x4 x11 = ld_stream<NT>(reinterpret_cast<const f32x4*>(p1 + 256 + 4 * lane));
    float s0 = score_sel<MODE>(addm, x00, x01, q0, q1, W);
    float s1 = score_sel<MODE>(addm, x10, x11, q0, q1, W);
    s0 = wave_sum(s0) + ab;
    s1 = wave_sum(s1) + ab;
    if (lane == 0) {
      s_score[r] = s0;
      if (so) so[r] = s0;
      if (lo) lo[r] = (pm && pm[r]) ? GVD_MIN_VALUE : s0;
      if (two) {
        s_score[r2] = s1;
        if (so) so[r2] = s1;
        if (lo) lo[r2] = (pm && pm[r2]) ? GVD_MIN_VALUE : s1;
      }
    }
  }
  __syncthreads();

  // ---- chunk-local softmax numerators
  float m = -INFINITY;
  if (tid < rows) m = s_score[tid];
  m = wave_max(m);   // rows <= 64: only wave 0 holds data
  if (tid == 0) s_red[0] = m;
  __syncthreads();
  m = s_red[0];
  float pr = 0.f;
  if (tid < rows) { pr = expf(s_score[tid] - m); }
  __syncthreads();
  if (tid < rows) s_score[tid] = pr;
  float l = wave_sum(pr);
  if (tid == 0) {
    float* ml = p.part_ml + ((int64_t)b * p.nctot + cglob) * 2;
    ml[0] = m; ml[1] = l;
  }
  __syncthreads();

  // ---- phase 2: partial context.  thread owns columns [4*tid, 4*tid+4) of H = 1024
  const float* fb = (rmap ? S.feats : S.feats + ((int64_t)fbi * S.N + n0) * ATT_H) + 4 * tid;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nctx = s_n[1];                 // live rows (their order is the row order: same sums as over all rows)
  int i = 0;
  for (; i + 8 <= nctx; i += 8) {
    f32x4 v[8];
    int rr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rr[u] = s_live[i + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld_stream<NT>(reinterpret_cast<const f32x4*>(fb + (int64_t)s_src[rr[u]] * ATT_H));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float pw = s_score[rr[u]];
      acc[0] = fmaf(pw, v[u][0], acc[0]); acc[1] = fmaf(pw, v[u][1], acc[1]);
      acc[2] = fmaf(pw, v[u][2], acc[2]); acc[3] = fmaf(pw, v[u][3], acc[3]);
    }
  }
  for (; i < nctx; ++i) {
    const int rr = s_live[i];
    const f32x4 v = ld_stream<NT>(reinterpret_cast<const f32x4*>(fb + (int64_t)s_src[rr] * ATT_H));
    const float pw = s_score[rr];
    acc[0] = fmaf(pw, v[0], acc[0]); acc[1] = fmaf(pw, v[1], acc[1]);
    acc[2] = fmaf(pw, v[2], acc[2]); acc[3] = fmaf(pw, v[3], acc[3]);
  }
  *reinterpret_cast<f32x4*>(p.part_ctx + ((int64_t)b * p.nctot + cglob) * ATT_H + 4 * tid) = acc;
}

// Beam-search variant: ONE workgroup serves the G beam rows of a sample (rows b*G .. b*G+G-1 share its features).
// Every p_feats / feats row of the chunk is loaded once and used for all G queries: G score accumulations per
// projection row, G context accumulators per feature row, so the HBM stream is the sample's bytes, not G x them
// ("shared across beams in a batched redesign").  Partials are written per beam row in the layout
// attn_combine_kernel expects.  grid = (chunks, samples).
template <int G, bool NT, int MODE>
__global__ __launch_bounds__(256, 6) void attn_partial_group_kernel(const FwdParams p) {   // >= 6 waves / SIMD: <= 80 VGPRs
  __shared__ float s_score[G][MAX_CHUNK];
  __shared__ float s_m[G];
  __shared__ int s_live[MAX_CHUNK];
  __shared__ int s_n[2];
  // the G pre-scaled queries live in LDS, not in 8 G registers per lane: with them in registers the kernel needed 105
  // VGPRs at G = 5 (4 waves per SIMD; PMC: 2.9 resident on average, 70 % of the wave cycles parked on memory)
  __shared__ __attribute__((aligned(16))) float s_q[G][ATT_A];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int smp = blockIdx.y;                    // sample; its beam rows are smp*G + g
  int c = blockIdx.x;
  const int sidx = (p.nside > 1 && c >= p.side[0].nchunks) ? 1 : 0;
  const SideDev& S = p.side[sidx];
  const int cglob = c;
  if (sidx) c -= p.side[0].nchunks;
  const int n0 = c * S.chunk;
  const int rows = min(S.chunk, S.N - n0);

  const bool addm = MODE == GVD_SCORE_ADD || sidx == 1;      // (compile-time true for the README configuration)
  const float qscale = (MODE == GVD_SCORE_DOT && !addm) ? 1.0f : GVD_TWO_LOG2E;
  const AttnLaneW W = lane_w_sel<MODE>(addm, S.w, lane);
  for (int i = tid; i < G * (ATT_A / 4); i += 256) {      // pre-scaled queries (tanh_fast, gvd_common.h)
    const int g = i / (ATT_A / 4), a4 = i % (ATT_A / 4);
    *reinterpret_cast<f32x4*>(&s_q[g][4 * a4]) =
        qscale * *reinterpret_cast<const f32x4*>(S.q + (int64_t)(smp * G + g) * S.ldq + 4 * a4);
  }
  const float ab = (MODE == GVD_SCORE_DOT && !addm) ? 0.f : *S.alpha_bias;
  const float* pf = S.p_feats + ((int64_t)smp * S.N + n0) * ATT_A;

  // Rows masked for EVERY beam of the sample are not fetched (score -1e8 whatever the features; weight exactly 0 as
  // long as each beam has a live row in the chunk) - see attn_partial_kernel.  Wave 0 compacts the live rows.
  if (wave == 0) {
    const bool in = lane < rows;
    bool all_masked = in && S.att_mask != nullptr;
    bool each_has_live = true;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const bool mg = in && S.att_mask && S.att_mask[((int64_t)smp * G + g) * S.ld_att_mask + n0 + lane];
      all_masked = all_masked && mg;
      each_has_live = each_has_live && (__ballot(in && !mg) != 0ull);
    }
    const bool live = in && !all_masked;
    const unsigned long long bal = __ballot(live);
    if (live) s_live[__popcll(bal & ((1ull << lane) - 1ull))] = lane;
    if (in && !live) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int64_t row = (int64_t)smp * G + g;
        s_score[g][lane] = GVD_MIN_VALUE;
        if (S.scores_out) S.scores_out[row * S.ld_scores + n0 + lane] = GVD_MIN_VALUE;
        if (S.logits_out) S.logits_out[row * S.ld_logits + n0 + lane] = GVD_MIN_VALUE;
      }
    }
    if (lane == 0) { s_n[0] = __popcll(bal); s_n[1] = each_has_live ? 1 : 0; }
  }
  __syncthreads();
  const int nlive = s_n[0];
  // ---- phase 1: two projection rows per wave per pass (4 x 16 B per lane in flight), G scores from each
  for (int i = wave * 2; i < nlive; i += 8) {
    const bool two = (i + 1) < nlive;
    const int r = s_live[i], r2 = two ? s_live[i + 1] : r;
    const float* p0 = pf + (int64_t)r * ATT_A;
    const float* p1 = pf + (int64_t)r2 * ATT_A;
    const f32x4 x00 = ld_stream<NT>(reinterpret_cast<const f32x4*>(p0 + 4 * lane));
    const f32x4 x01 = ld_stream<NT>(reinterpret_cast<const f32x4*>(p0 + 256 + 4 * lane));
    const f32x4 x10 = ld_stream<NT>(reinterpret_cast<const f32x4*>(p1 + 4 * lane));
    const f32x4 x11 = ld_stream<NT>(reinterpret_cast<const f32x4*>(p1 + 256 + 4 * lane));
    float sc0[G], sc1[G];
    int lz = 0;
    asm volatile("" : "+v"(lz));       // opaque 0: keeps the query reads inside the row loop (loop-invariant code motion
                                       // would put all 8 G of them back into registers)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(&s_q[g][4 * lane + lz]);
      const f32x4 q1 = *reinterpret_cast<const f32x4*>(&s_q[g][256 + 4 * lane + lz]);
      sc0[g] = wave_sum(score_sel<MODE>(addm, x00, x01, q0, q1, W)) + ab;
      sc1[g] = wave_sum(score_sel<MODE>(addm, x10, x11, q0, q1, W)) + ab;
    }
    if (lane < 2 * G) {
      const int g = lane % G, second = lane / G;
      if (!second || two) {
        float e = 0.f;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) if (g == gg) e = second ? sc1[gg] : sc0[gg];
        const int rr = second ? r2 : r;
        const int64_t row = (int64_t)smp * G + g;
        const bool am = S.att_mask && S.att_mask[row * S.ld_att_mask + n0 + rr];
        if (am) e = GVD_MIN_VALUE;
        s_score[g][rr] = e;
        if (S.scores_out) S.scores_out[row * S.ld_scores + n0 + rr] = e;
        if (S.logits_out) {
          const bool pm = S.pnt_mask && S.pnt_mask[row * S.ld_pnt_mask + n0 + rr];
          S.logits_out[row * S.ld_logits + n0 + rr] = pm ? GVD_MIN_VALUE : e;
        }
      }
    }
  }
  __syncthreads();

  // ---- chunk-local softmax numerators: wave g handles beam g (G <= 8 -> two passes over the 4 waves)
  for (int g = wave; g < G; g += 4) {
    float m = lane < rows ? s_score[g][lane] : -INFINITY;     // rows <= 64
    m = wave_max(m);
    const float pr = lane < rows ? expf(s_score[g][lane] - m) : 0.f;
    if (lane < rows) s_score[g][lane] = pr;
    const float l = wave_sum(pr);
    if (lane == 0) {
      float* ml = p.part_ml + ((int64_t)(smp * G + g) * p.nctot + cglob) * 2;
      ml[0] = m; ml[1] = l;
    }
  }
  __syncthreads();

  // ---- phase 2: G partial contexts from one pass over the chunk's feature rows; thread owns 4 columns of H = 1024
  const float* fb = S.feats + ((int64_t)smp * S.N + n0) * ATT_H + 4 * tid;
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool skip = s_n[1] != 0;                 // every beam has a live row: rows masked for all beams weigh exactly 0
  const int nctx = skip ? nlive : rows;
  int i = 0;
  for (; i + 8 <= nctx; i += 8) {                // 8 feature rows (8 x 16 B per lane) in flight
    f32x4 v[8];
    int rr[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) rr[u] = skip ? s_live[i + u] : i + u;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld_stream<NT>(reinterpret_cast<const f32x4*>(fb + (int64_t)rr[u] * ATT_H));
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float pw = s_score[g][rr[u]];
        acc[g][0] = fmaf(pw, v[u][0], acc[g][0]); acc[g][1] = fmaf(pw, v[u][1], acc[g][1]);
        acc[g][2] = fmaf(pw, v[u][2], acc[g][2]); acc[g][3] = fmaf(pw, v[u][3], acc[g][3]);
      }
  }
  for (; i < nctx; ++i) {
    const int rr = skip ? s_live[i] : i;
    const f32x4 v = ld_stream<NT>(reinterpret_cast<const f32x4*>(fb + (int64_t)rr * ATT_H));
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float pw = s_score[g][rr];
      acc[g][0] = fmaf(pw, v[0], acc[g][0]); acc[g][1] = fmaf(pw, v[1], acc[g][1]);
      acc[g][2] = fmaf(pw, v[2], acc[g][2]); acc[g][3] = fmaf(pw, v[3], acc[g][3]);
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g)
    *reinterpret_cast<f32x4*>(p.part_ctx + ((int64_t)(smp * G + g) * p.nctot + cglob) * ATT_H + 4 * tid) = acc[g];
}


struct CombParams {
  const float* part_ctx; const float* part_ml;
  int nc[2]; int nside; int nctot;
  float* out_sum; int64_t ld_out;
  float* ctx_out[2];   // optional, [B,H] each
};

constexpr int MAX_NC = 512;   // chunks per side the combine kernel can merge

// NG groups of 256 threads: each group merges every NG-th chunk (the per-thread chain of dependent-latency partial
// reads is NG x shorter), group 0 adds the groups up through LDS.  NG = 4 for decode batches, where this kernel is
// a latency chain in the token loop; NG = 1 when there are enough rows to fill the chip anyway.
template <int NG>
__global__ __launch_bounds__(256 * NG) void attn_combine_kernel(const CombParams p) {
  constexpr int NW = 4 * NG;
  __shared__ float s_sc[MAX_NC];
  __shared__ float s_red[2 * NW];
  __shared__ f32x4 s_acc[NG > 1 ? (NG - 1) * 256 : 1];
  const int b = blockIdx.x, tid = threadIdx.x, grp = tid >> 8, t = tid & 255, wave = tid >> 6;
  f32x4 total = {0.f, 0.f, 0.f, 0.f};
  int c0 = 0;
  for (int s = 0; s < p.nside; ++s) {
    const int nc = p.nc[s];
    const float* ml = p.part_ml + ((int64_t)b * p.nctot + c0) * 2;
    // all (m, l) pairs in parallel: global max, rescale factors exp(m_c - M) into LDS, normaliser L
    float m_loc = -INFINITY;
    for (int c = tid; c < nc; c += 256 * NG) m_loc = fmaxf(m_loc, ml[2 * c]);
    m_loc = wave_max(m_loc);
    __syncthreads();
    if ((tid & 63) == 0) s_red[wave] = m_loc;
    __syncthreads();
    float M = s_red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) M = fmaxf(M, s_red[w]);
    float l_loc = 0.f;
    for (int c = tid; c < nc; c += 256 * NG) {
      const float sc = expf(ml[2 * c] - M);
      s_sc[c] = sc;
      l_loc = fmaf(sc, ml[2 * c + 1], l_loc);
    }
    l_loc = wave_sum(l_loc);
    if ((tid & 63) == 0) s_red[NW + wave] = l_loc;
    __syncthreads();
    float L = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) L += s_red[NW + w];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* pc = p.part_ctx + ((int64_t)b * p.nctot + c0) * ATT_H + 4 * t;
    int c = grp;
    for (; c + 3 * NG < nc; c += 4 * NG) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(pc + (int64_t)(c + u * NG) * ATT_H);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float sc = s_sc[c + u * NG];
        acc[0] = fmaf(sc, v[u][0], acc[0]); acc[1] = fmaf(sc, v[u][1], acc[1]);
        acc[2] = fmaf(sc, v[u][2], acc[2]); acc[3] = fmaf(sc, v[u][3], acc[3]);
      }
    }
    for (; c < nc; c += NG) {
      const float sc = s_sc[c];
      const f32x4 v = *reinterpret_cast<const f32x4*>(pc + (int64_t)c * ATT_H);
      acc[0] = fmaf(sc, v[0], acc[0]); acc[1] = fmaf(sc, v[1], acc[1]);
      acc[2] = fmaf(sc, v[2], acc[2]); acc[3] = fmaf(sc, v[3], acc[3]);
    }
    if (NG > 1) {
      if (grp > 0) s_acc[(grp - 1) * 256 + t] = acc;
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int g = 0; g < NG - 1; ++g) {
          const f32x4 o = s_acc[g * 256 + t];
          acc[0] += o[0]; acc[1] += o[1]; acc[2] += o[2]; acc[3] += o[3];
        }
      }
    }
    if (grp == 0) {
      const float inv = 1.0f / L;
      acc[0] *= inv; acc[1] *= inv; acc[2] *= inv; acc[3] *= inv;
      if (p.ctx_out[s]) *reinterpret_cast<f32x4*>(p.ctx_out[s] + (int64_t)b * ATT_H + 4 * t) = acc;
      total[0] += acc[0]; total[1] += acc[1]; total[2] += acc[2]; total[3] += acc[3];
    }
    c0 += nc;
    __syncthreads();   // s_sc / s_red / s_acc reused by the next side
  }
  if (grp == 0 && p.out_sum) *reinterpret_cast<f32x4*>(p.out_sum + (int64_t)b * p.ld_out + 4 * t) = total;
}

__global__ void tanh_fast_kernel(const float* x, float* y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = tanh_fast(x[i]);
}

// rows per chunk: 50 (measured best against 32 / 64, profiles/r03/attn_chunk_y.log, beam_chunk_p.log); halved (not below
// 13) while fewer than ~512 workgroups would exist (small batches)
int pick_chunk(int N, int B) {
  int chunk = 50;
  while (chunk > 20 && (long)B * ((N + chunk - 1) / chunk) < 512) chunk = (chunk + 1) / 2;
  if (chunk > MAX_CHUNK) chunk = MAX_CHUNK;
  if (chunk > N) chunk = N;
  if (chunk < 1) chunk = 1;
  return chunk;
}

int nchunks_of(int N, int B) { int c = pick_chunk(N, B); return (N + c - 1) / c; }

bool side_ok(const gvd_attn_side* s) {
  if (!s || s->score_mode < GVD_SCORE_ADD || s->score_mode > GVD_SCORE_DOT) return false;
  const bool has_alpha = s->score_mode != GVD_SCORE_DOT;      // 'dp': no alpha_net in the module (AttModel.py:63-66,92-95)
  return s->feats && s->p_feats && s->q && (!has_alpha || (s->w && s->alpha_bias && gvd_aligned16(s->w))) && s->N > 0 &&
         gvd_aligned16(s->feats) && gvd_aligned16(s->p_feats) && gvd_aligned16(s->q) && (s->ldq % 4) == 0;
}

void fill_side(SideDev& d, const gvd_attn_side* s, int B) {
  d.feats = s->feats; d.p_feats = s->p_feats; d.q = s->q; d.ldq = s->ldq; d.w = s->w;
  d.alpha_bias = s->alpha_bias; d.att_mask = s->att_mask; d.ld_att_mask = s->ld_att_mask;
  d.pnt_mask = s->pnt_mask; d.ld_pnt_mask = s->ld_pnt_mask; d.logits_out = s->logits_out;
  d.ld_logits = s->ld_logits; d.scores_out = s->scores_out; d.ld_scores = s->ld_scores;
  d.row_map = s->row_map;
  d.N = s->N; d.group = s->group; d.chunk = pick_chunk(s->N, B);
  d.nchunks = (s->N + d.chunk - 1) / d.chunk;
}

}  // namespace

extern "C" int gvd_tanh_fast_f32(const float* x, float* y, int64_t n, gvd_stream_t stream) {
  if (!x || !y || n <= 0) return GVD_EINVAL;
  hipLaunchKernelGGL(tanh_fast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, gvd_s(stream), x, y, n);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t gvd_attn_workspace_bytes(int B, int n_region, int n_temporal, int H) {
  long nc = nchunks_of(n_region, B) + (n_temporal > 0 ? nchunks_of(n_temporal, B) : 0);
  return (size_t)B * nc * (H + 4) * sizeof(float);
}

extern "C" int gvd_attn_fwd_prof(const gvd_attn_side* region, const gvd_attn_side* temporal, int B, int A, int H,
                                 float* out_sum, int64_t ld_out, float* ctx_region, float* ctx_temporal,
                                 void* workspace, gvd_prof* prof, gvd_stream_t stream) {
  if (A != ATT_A || H != ATT_H || B <= 0 || !workspace || !gvd_aligned16(workspace)) return GVD_EINVAL;
  if (!side_ok(region) || (temporal && (!side_ok(temporal) || temporal->score_mode != GVD_SCORE_ADD))) return GVD_EINVAL;
  if (out_sum && (!gvd_aligned16(out_sum) || (ld_out % 4) != 0)) return GVD_EINVAL;
  const int mode = region->score_mode;
  FwdParams p = {};
  fill_side(p.side[0], region, B);
  p.nside = 1;
  if (temporal) { fill_side(p.side[1], temporal, B); p.nside = 2; }
  p.nctot = p.side[0].nchunks + (temporal ? p.side[1].nchunks : 0);
  if (p.side[0].nchunks > MAX_NC || (temporal && p.side[1].nchunks > MAX_NC)) return GVD_EINVAL;
  p.part_ctx = reinterpret_cast<float*>(workspace);
  p.part_ml = p.part_ctx + (int64_t)B * p.nctot * ATT_H;
  hipStream_t st = gvd_s(stream);
  gvd_prof_begin(prof, st);
  // Streaming hint: when one launch reads more than the 256 MB Infinity Cache can hold, nothing it reads survives to
  // the next token anyway, and nontemporal loads stream measurably faster (tools/stream_read_micro.hip: 7.1 vs 6.3 TB/s
  // read-only; this kernel 311 -> 285 us at B = 256).  Smaller launches keep plain loads so the features stay cached
  // across tokens.
  const double launch_bytes = (double)B / (region->group > 1 ? region->group : 1) *
                              ((double)region->N + (temporal ? temporal->N : 0)) * (ATT_A + ATT_H) * 4.0;
  const bool nt = launch_bytes > 192.0 * 1024 * 1024;
  // beam search: both attentions share features within groups of G rows -> one workgroup per (chunk, sample)
  const int G = region->group;
  const bool grouped = G >= 2 && G <= 5 && B % G == 0 && (!temporal || temporal->group == G);
  if (grouped && (region->row_map || (temporal && temporal->row_map))) return GVD_EINVAL;   // (row kernel only)
  if (grouped) {
    const dim3 grid((unsigned)p.nctot, (unsigned)(B / G));
#define GVD_LAUNCH_GROUP_M(GG, MM)                                                                             \
    if (nt) hipLaunchKernelGGL((attn_partial_group_kernel<GG, true, MM>), grid, dim3(256), 0, st, p);          \
    else hipLaunchKernelGGL((attn_partial_group_kernel<GG, false, MM>), grid, dim3(256), 0, st, p)
#define GVD_LAUNCH_GROUP(GG)                                                                                   \
    if (mode == GVD_SCORE_ADD) { GVD_LAUNCH_GROUP_M(GG, GVD_SCORE_ADD); }                                      \
    else if (mode == GVD_SCORE_MUL) { GVD_LAUNCH_GROUP_M(GG, GVD_SCORE_MUL); }                                 \
    else { GVD_LAUNCH_GROUP_M(GG, GVD_SCORE_DOT); }
    switch (G) {
      case 2: GVD_LAUNCH_GROUP(2); break;
      case 3: GVD_LAUNCH_GROUP(3); break;
      case 4: GVD_LAUNCH_GROUP(4); break;
      default: GVD_LAUNCH_GROUP(5); break;
    }
#undef GVD_LAUNCH_GROUP
#undef GVD_LAUNCH_GROUP_M
  } else {
    const dim3 grid((unsigned)p.nctot, (unsigned)B);
#define GVD_LAUNCH_ROW(MM)                                                                                     \
    if (nt) hipLaunchKernelGGL((attn_partial_kernel<true, MM>), grid, dim3(256), 0, st, p);                    \
    else hipLaunchKernelGGL((attn_partial_kernel<false, MM>), grid, dim3(256), 0, st, p)
    if (mode == GVD_SCORE_ADD) { GVD_LAUNCH_ROW(GVD_SCORE_ADD); }
    else if (mode == GVD_SCORE_MUL) { GVD_LAUNCH_ROW(GVD_SCORE_MUL); }
    else { GVD_LAUNCH_ROW(GVD_SCORE_DOT); }
#undef GVD_LAUNCH_ROW
  }
  gvd_prof_end(prof, st);
  GVD_CHECK_LAUNCH();
  CombParams c = {};
  c.part_ctx = p.part_ctx; c.part_ml = p.part_ml; c.nside = p.nside; c.nctot = p.nctot;
  c.nc[0] = p.side[0].nchunks; c.nc[1] = temporal ? p.side[1].nchunks : 0;
  c.out_sum = out_sum; c.ld_out = ld_out; c.ctx_out[0] = ctx_region; c.ctx_out[1] = ctx_temporal;
  if (B <= 64)
    hipLaunchKernelGGL(attn_combine_kernel<4>, dim3((unsigned)B), dim3(1024), 0, st, c);
  else
    hipLaunchKernelGGL(attn_combine_kernel<1>, dim3((unsigned)B), dim3(256), 0, st, c);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_attn_fwd(const gvd_attn_side* region, const gvd_attn_side* temporal, int B, int A, int H,
                            float* out_sum, int64_t ld_out, float* ctx_region, float* ctx_temporal,
                            void* workspace, gvd_stream_t stream) {
  return gvd_attn_fwd_prof(region, temporal, B, A, H, out_sum, ld_out, ctx_region, ctx_temporal, workspace,
                           nullptr, stream);
}
