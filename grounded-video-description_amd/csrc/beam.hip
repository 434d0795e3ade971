// Beam-search bookkeeping of one decode step for all samples in ONE launch (CaptionModelBU.py:49-96,154-166 with the
// repair of SURVEY.md §3.4): candidate merge of the K x K (word rank c, parent beam q) expansions, history fork, finished-
// beam record.  The reference does this per sample on the host (a device->host copy + a Python sort per step); the
// batched torch formulation it replaces (beam.py, GVD_BEAM_FUSED=0) is ~45 small launches per step.  One wave per
// sample: K <= 8, so a sample's candidates, forks and records are a few hundred scalar operations.
//
// Semantics reproduced bit for bit:
//   * candidate j = c*K + q (t = 0: only beam 0 expands, j = c) with score sums[q] + ys[q][c] (one fp32 add);
//   * stable sort by descending score, first K kept -> (q_sel, c_sel); equal scores keep the lower j first;
//   * histories beam_seq / beam_lps / beam_att[0..t-1] fork from parent q_sel; beam_att[t] = att2_ind[q_sel];
//   * a beam whose word is 0 (or every beam at t = L-1) is finished: the best finished beam of the step (first maximum)
//     replaces the record when strictly better; finished beams continue with sums = -1000.
#include "gvd_common.h"
#include <math.h>
#include <stdint.h>

namespace {

constexpr int BEAM_MAXK = 8;

struct BeamParams {
  const float* ys; const int64_t* ix;
  float* sums; const int64_t* att2_ind;
  int64_t* beam_seq; float* beam_lps; int64_t* beam_att;
  float* best_p; int64_t* best_seq; float* best_lps; int64_t* best_vix;
  int64_t* parent; int64_t* word;
  int B, K, L, t;
};

// One 64-lane wave per sample: every lane repeats the (tiny) candidate selection from the same inputs, lane s then forks
// history row s and moves its element of the finished-beam record, so the per-row loads / stores are independent instead of
// one thread's dependent chain over L rows.
__global__ __launch_bounds__(64) void beam_step_kernel(const BeamParams p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int K = p.K, L = p.L, t = p.t;
  const int64_t BK = (int64_t)p.B * K;
  const float* ys = p.ys + (int64_t)b * K * K;
  const int64_t* ix = p.ix + (int64_t)b * K * K;
  float* sums = p.sums + (int64_t)b * K;
  float cand[BEAM_MAXK * BEAM_MAXK];
  const int ncand = t == 0 ? K : K * K;
  for (int j = 0; j < ncand; ++j) {
    const int q = t == 0 ? 0 : j % K, c = t == 0 ? j : j / K;
    cand[j] = sums[q] + ys[q * K + c];
  }
  const float best_before = p.best_p[b];
  int qs[BEAM_MAXK], cs[BEAM_MAXK];
  float np[BEAM_MAXK];
  // stable descending selection of the first K: repeatedly the first maximum among the candidates not taken yet
  unsigned long long taken = 0ull;
  for (int k = 0; k < K; ++k) {
    int bj = -1;
    float bv = 0.f;
    for (int j = 0; j < ncand; ++j) {
      if ((taken >> j) & 1ull) continue;
      if (bj < 0 || cand[j] > bv) { bj = j; bv = cand[j]; }
    }
    taken |= 1ull << bj;
    qs[k] = t == 0 ? 0 : bj % K;
    cs[k] = t == 0 ? bj : bj / K;
    np[k] = bv;
  }
  int64_t wd[BEAM_MAXK];
  float wl[BEAM_MAXK];
  for (int k = 0; k < K; ++k) { wd[k] = ix[qs[k] * K + cs[k]]; wl[k] = ys[qs[k] * K + cs[k]]; }
  // finished beams: the step's best (first maximum) against the record
  float step_best = -INFINITY;
  int first = 0;
  bool any = false;
  for (int k = 0; k < K; ++k) {
    const bool fin = t == L - 1 || wd[k] == 0;
    if (fin && (!any || np[k] > step_best)) { step_best = np[k]; first = k; any = true; }
  }
  const bool better = any && step_best > best_before;
  __syncthreads();                      // every lane has read sums / best_p before anyone rewrites them
  for (int s = lane; s < L; s += 64) {
    int64_t* bs = p.beam_seq + (int64_t)s * BK + (int64_t)b * K;
    float* bl = p.beam_lps + (int64_t)s * BK + (int64_t)b * K;
    int64_t* ba = p.beam_att + (int64_t)s * BK + (int64_t)b * K;
    int64_t rec_seq;
    float rec_lps;
    if (s < t) {                        // fork the history row from the parent beams
      int64_t ts[BEAM_MAXK], ta[BEAM_MAXK];
      float tl[BEAM_MAXK];
      for (int k = 0; k < K; ++k) { ts[k] = bs[k]; tl[k] = bl[k]; ta[k] = ba[k]; }
      for (int k = 0; k < K; ++k) { bs[k] = ts[qs[k]]; bl[k] = tl[qs[k]]; ba[k] = ta[qs[k]]; }
      rec_seq = ts[qs[first]]; rec_lps = tl[qs[first]];
    } else if (s == t) {
      for (int k = 0; k < K; ++k) {
        bs[k] = wd[k];
        bl[k] = wl[k];
        if (t >= 1) ba[k] = p.att2_ind[(int64_t)b * K + qs[k]];
      }
      rec_seq = wd[first]; rec_lps = wl[first];
    } else {
      rec_seq = bs[first]; rec_lps = bl[first];
    }
    if (better) {
      p.best_seq[(int64_t)b * L + s] = rec_seq;
      p.best_lps[(int64_t)b * L + s] = rec_lps;
    }
  }
  if (lane < K) {
    p.parent[(int64_t)b * K + lane] = (int64_t)b * K + qs[lane];
    p.word[(int64_t)b * K + lane] = wd[lane];
    sums[lane] = (t == L - 1 || wd[lane] == 0) ? -1000.0f : np[lane];
  }
  if (lane == 0 && better) { p.best_p[b] = step_best; p.best_vix[b] = first; }
}

}  // namespace

extern "C" int gvd_beam_step(const gvd_beam_step_args* a, gvd_stream_t stream) {
  if (!a || !a->ys || !a->ix || !a->sums || !a->att2_ind || !a->beam_seq || !a->beam_lps || !a->beam_att || !a->best_p ||
      !a->best_seq || !a->best_lps || !a->best_vix || !a->parent || !a->word || a->B <= 0 || a->K <= 0 ||
      a->K > BEAM_MAXK || a->L <= 0 || a->t < 0 || a->t >= a->L)
    return GVD_EINVAL;
  BeamParams p;
  p.ys = a->ys; p.ix = a->ix; p.sums = a->sums; p.att2_ind = a->att2_ind;
  p.beam_seq = a->beam_seq; p.beam_lps = a->beam_lps; p.beam_att = a->beam_att;
  p.best_p = a->best_p; p.best_seq = a->best_seq; p.best_lps = a->best_lps; p.best_vix = a->best_vix;
  p.parent = a->parent; p.word = a->word;
  p.B = a->B; p.K = a->K; p.L = a->L; p.t = a->t;
  hipLaunchKernelGGL(beam_step_kernel, dim3((unsigned)a->B), dim3(64), 0, gvd_s(stream), p);
  GVD_CHECK_LAUNCH();
  return 0;
}
