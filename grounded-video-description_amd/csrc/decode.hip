// Greedy decode driver: the whole token loop of AttModel._sample (model.py:580-624; sample_max=1,
// beam_size=1) as one asynchronous launch sequence on one stream — no host round trip per token, no
// per-step tensor allocation, no concatenations (the LSTM kernels read their input blocks in place).
//
// Per step (TopDownCore.forward, AttModel.py:134-164):
//   1 att-LSTM   gates = [fc | xt] W_ih^T + h_att W_hh^T + b    (fc part hoisted out of the loop)
//   2 q          [q_temporal | q_region] = h_att [W_att ; W_att2]^T + b      (one GEMM, stacked weights)
//   3 attention  temporal + region partials in one streaming pass, 4 combine -> att + att2
//   5 lang-LSTM  gates = [att+att2 | h_att] W_ih^T + h_lang W_hh^T + b
//   6 logits     h_lang W_logit^T + b ;  7 token rule: log-softmax, top-2, UNK -> runner-up, embed next
#include "gvd_common.h"
#include "decode_persistent.h"

namespace {

struct Ws {
  float *fc_gates, *h_att[2], *c_att[2], *h_lang[2], *c_lang[2], *q12, *att_sum, *logits, *xt, *w_stack,
      *b_stack;
  int64_t* it0;
  void* attn_ws;
  void* pd_ws;
  size_t total;
};

size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

Ws carve(void* base, int B, int Ft, int R, int H, int A, int E, int V) {
  Ws w;
  size_t off = 0;
  char* b = reinterpret_cast<char*>(base);
  auto take = [&](size_t bytes) { void* p = b ? b + off : nullptr; off += align_up(bytes); return p; };
  const size_t f = sizeof(float);
  w.fc_gates = (float*)take((size_t)B * 4 * H * f);
  for (int i = 0; i < 2; ++i) {
    w.h_att[i] = (float*)take((size_t)B * H * f); w.c_att[i] = (float*)take((size_t)B * H * f);
    w.h_lang[i] = (float*)take((size_t)B * H * f); w.c_lang[i] = (float*)take((size_t)B * H * f);
  }
  w.q12 = (float*)take((size_t)B * 2 * A * f);
  w.att_sum = (float*)take((size_t)B * H * f);
  w.logits = (float*)take((size_t)B * V * f);
  w.xt = (float*)take((size_t)B * E * f);
  w.w_stack = (float*)take((size_t)2 * A * H * f);
  w.b_stack = (float*)take((size_t)2 * A * f);
  w.it0 = (int64_t*)take((size_t)B * sizeof(int64_t));
  w.attn_ws = take(gvd_attn_workspace_bytes(B, R, Ft, H));
  w.pd_ws = take(gvd_pd_shape_ok(B, H, A, E, V, R, Ft) ? gvd_pd_workspace_bytes() : 0);
  w.total = off;
  return w;
}

// One launch instead of five memsets + four device copies (each a runtime fill / copy kernel with its own dispatch gap:
// ~80 us in front of the 3.2 ms of a batch_size = 4 call): zero state of both LSTMs, BOS token ids, stacked h2att weights
// [W_att ; W_att2] and biases.
struct InitParams {
  float* z[4]; int64_t nz;             // four [B,H] state arrays to clear
  int64_t* it0; int B;
  float* w_stack; const float* w1; const float* w2; int64_t nw;      // A*H floats each
  float* b_stack; const float* b1; const float* b2; int nb;          // A floats each
};
__global__ __launch_bounds__(256) void greedy_init_kernel(const InitParams p) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const gvd_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  if (i4 < p.nz) {
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<gvd_f32x4*>(p.z[k] + i4) = zero;
  }
  if (i4 < p.nw) {
    *reinterpret_cast<gvd_f32x4*>(p.w_stack + i4) = *reinterpret_cast<const gvd_f32x4*>(p.w1 + i4);
    *reinterpret_cast<gvd_f32x4*>(p.w_stack + p.nw + i4) = *reinterpret_cast<const gvd_f32x4*>(p.w2 + i4);
  }
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < p.nb) { p.b_stack[i] = p.b1[i]; p.b_stack[p.nb + i] = p.b2[i]; }
  if (i < p.B) p.it0[i] = 0;
}

#define GVD_TRY(x) do { int rc__ = (x); if (rc__ != 0) return rc__; } while (0)
#define GVD_HIP(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return (int)e__; } while (0)

}  // namespace

extern "C" size_t gvd_greedy_workspace_bytes(int B, int Ft, int R, int H, int A, int E, int V) {
  return carve(nullptr, B, Ft, R, H, A, E, V).total;
}

extern "C" int gvd_greedy_decode(const gvd_greedy_args* a, gvd_stream_t stream) {
  if (!a || !a->workspace || !gvd_aligned16(a->workspace) || a->B <= 0 || a->L <= 0) return GVD_EINVAL;
  const int mode = a->att_input_mode;
  if (mode != GVD_ATT_INPUT_BOTH && mode != GVD_ATT_INPUT_FEATMAP && mode != GVD_ATT_INPUT_REGION) return GVD_EINVAL;
  const bool use_temporal = mode != GVD_ATT_INPUT_REGION;          // AttModel.py:140-141
  const int score_mode = a->region_attn_mode;                      // AttModel.py:82-95
  if (score_mode < GVD_SCORE_ADD || score_mode > GVD_SCORE_DOT) return GVD_EINVAL;
  if (score_mode != GVD_SCORE_DOT && (!a->att2_alpha_w || !a->att2_alpha_b)) return GVD_EINVAL;
  const int B = a->B, H = a->H, A = a->A, E = a->E, V = a->V, R = a->R, Ft = use_temporal ? a->Ft : 0, L = a->L;
  hipStream_t st = gvd_s(stream);
  Ws w = carve(a->workspace, B, Ft, R, H, A, E, V);

  // zero state, BOS token ids, stacked h2att weights [W_att ; W_att2]: one launch (greedy_init_kernel)
  if ((((int64_t)B * H) & 3) || (((int64_t)A * H) & 3) || !gvd_aligned16(a->att1_h2att_w) || !gvd_aligned16(a->att2_h2att_w))
    return GVD_EINVAL;
  {
    InitParams ip = {};
    ip.z[0] = w.h_att[0]; ip.z[1] = w.c_att[0]; ip.z[2] = w.h_lang[0]; ip.z[3] = w.c_lang[0]; ip.nz = (int64_t)B * H;
    ip.it0 = w.it0; ip.B = B;
    ip.w_stack = w.w_stack; ip.w1 = a->att1_h2att_w; ip.w2 = a->att2_h2att_w; ip.nw = (int64_t)A * H;
    ip.b_stack = w.b_stack; ip.b1 = a->att1_h2att_b; ip.b2 = a->att2_h2att_b; ip.nb = A;
    int64_t n = ip.nz > ip.nw ? ip.nz : ip.nw;
    n = (n + 3) / 4;
    if (n < A) n = A;
    if (n < B) n = B;
    hipLaunchKernelGGL(greedy_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ip);
    GVD_CHECK_LAUNCH();
  }

  // loop-invariant part of the att-LSTM gates: fc W_ih[:, :H]^T + b_ih + b_hh   (AttModel.py:138)
  {
    gvd_gemm_args g = {};
    g.nseg = 1;
    g.seg[0] = {a->fc, H, 0, a->att_w_ih, (int64_t)(E + H), 0, H};
    g.nbias = a->att_b_ih; g.nbias2 = a->att_b_hh;
    g.C = w.fc_gates; g.ldc = 4 * H; g.M = B; g.N = 4 * H; g.batch = 1;
    GVD_TRY(gvd_gemm_nt_f32(&g, stream));
  }
  const bool persistent = !a->no_persistent && mode == GVD_ATT_INPUT_BOTH && score_mode == GVD_SCORE_ADD &&
                          gvd_pd_eligible(B, H, A, E, V, R, Ft);
  if (a->pool_row_map && persistent) return GVD_EINVAL;   // persistent kernel: dense layout
  if (persistent) {
    // decode batch: the whole token loop as ONE persistent cooperative launch (decode_persistent.hip).  The event
    // timer `prof` has no per-step attention kernel to bracket on this path and records nothing.
    PdParams q = {};
    q.fc_gates = w.fc_gates; q.conv = a->conv; q.p_conv = a->p_conv; q.pool = a->pool; q.p_pool = a->p_pool;
    q.pnt_mask = a->pnt_mask; q.embed = a->embed; q.att_w_ih = a->att_w_ih; q.att_w_hh = a->att_w_hh;
    q.lang_w_ih = a->lang_w_ih; q.lang_w_hh = a->lang_w_hh; q.lang_b_ih = a->lang_b_ih; q.lang_b_hh = a->lang_b_hh;
    q.q_w = w.w_stack; q.q_b = w.b_stack;
    q.a1_w = a->att1_alpha_w; q.a1_b = a->att1_alpha_b; q.a2_w = a->att2_alpha_w; q.a2_b = a->att2_alpha_b;
    q.logit_w = a->logit_w; q.logit_b = a->logit_b;
    q.B = B; q.Ft = Ft; q.R = R; q.V = V; q.L = L; q.unk = a->unk_idx;
    q.seq = a->seq; q.seq_lp = a->seq_logprobs; q.att2_weights = a->att2_weights; q.status = a->status;
    q.trace = reinterpret_cast<unsigned long long*>(a->trace);
    const int rc = gvd_pd_launch(q, w.pd_ws, st);
    if (rc == 0) return 0;
    (void)hipGetLastError();   // e.g. not all 256 workgroups can be co-resident here: use the multi-kernel loop
  }
  if (a->status) GVD_HIP(hipMemsetAsync(a->status, 0, sizeof(int), st));
  GVD_TRY(gvd_embed_relu(w.it0, 1, B, a->embed, E, w.xt, E, stream));   // BOS = token 0 (model.py:588)

  int cur = 0;
  for (int t = 0; t < L; ++t) {
    const int nxt = cur ^ 1;
    {  // attention LSTM
      gvd_lstm_args l = {};
      l.nseg = 2;
      l.seg[0] = {w.xt, E, 0, a->att_w_ih + H, (int64_t)(E + H), 0, E};
      l.seg[1] = {w.h_att[cur], H, 0, a->att_w_hh, H, 0, H};
      l.rowbias = w.fc_gates; l.rowbias_ld = 4 * H;
      l.c_prev = w.c_att[cur]; l.ldc_prev = H;
      l.h_out = w.h_att[nxt]; l.ldh = H; l.c_out = w.c_att[nxt]; l.ldc_out = H;
      l.B = B; l.H = H;
      GVD_TRY(gvd_lstm_cell_fwd(&l, stream));
    }
    {  // both attention queries
      gvd_gemm_args g = {};
      g.nseg = 1;
      g.seg[0] = {w.h_att[nxt], H, 0, w.w_stack, H, 0, H};
      g.nbias = w.b_stack;
      g.C = w.q12; g.ldc = 2 * A; g.M = B; g.N = 2 * A; g.batch = 1;
      GVD_TRY(gvd_gemm_nt_f32(&g, stream));
    }
    {  // temporal + region attention -> att + att2
      gvd_attn_side reg = {};
      reg.feats = a->pool; reg.p_feats = a->p_pool; reg.q = w.q12 + A; reg.ldq = 2 * A;
      reg.w = a->att2_alpha_w; reg.alpha_bias = a->att2_alpha_b;
      reg.att_mask = a->pnt_mask + 1; reg.ld_att_mask = R + 1;
      reg.pnt_mask = a->pnt_mask + 1; reg.ld_pnt_mask = R + 1;
      reg.logits_out = a->att2_weights + (int64_t)t * R; reg.ld_logits = (int64_t)L * R;
      reg.N = R; reg.row_map = a->pool_row_map; reg.score_mode = score_mode;
      gvd_attn_side tmp = {};
      tmp.feats = a->conv; tmp.p_feats = a->p_conv; tmp.q = w.q12; tmp.ldq = 2 * A;
      tmp.w = a->att1_alpha_w; tmp.alpha_bias = a->att1_alpha_b; tmp.N = Ft;
      // what reaches the language LSTM (AttModel.py:147-152): att + att2, att alone (the temporal context written straight
      // into att_sum, the region side still produces its logits), or att2 alone (no temporal side in the launch)
      if (mode == GVD_ATT_INPUT_BOTH)
        GVD_TRY(gvd_attn_fwd_prof(&reg, &tmp, B, A, H, w.att_sum, H, nullptr, nullptr, w.attn_ws, a->prof, stream));
      else if (mode == GVD_ATT_INPUT_FEATMAP)
        GVD_TRY(gvd_attn_fwd_prof(&reg, &tmp, B, A, H, nullptr, 0, nullptr, w.att_sum, w.attn_ws, a->prof, stream));
      else
        GVD_TRY(gvd_attn_fwd_prof(&reg, nullptr, B, A, H, w.att_sum, H, nullptr, nullptr, w.attn_ws, a->prof, stream));
    }
    {  // language LSTM
      gvd_lstm_args l = {};
      l.nseg = 3;
      l.seg[0] = {w.att_sum, H, 0, a->lang_w_ih, (int64_t)(2 * H), 0, H};
      l.seg[1] = {w.h_att[nxt], H, 0, a->lang_w_ih + H, (int64_t)(2 * H), 0, H};
      l.seg[2] = {w.h_lang[cur], H, 0, a->lang_w_hh, H, 0, H};
      l.b_ih = a->lang_b_ih; l.b_hh = a->lang_b_hh;
      l.c_prev = w.c_lang[cur]; l.ldc_prev = H;
      l.h_out = w.h_lang[nxt]; l.ldh = H; l.c_out = w.c_lang[nxt]; l.ldc_out = H;
      l.B = B; l.H = H;
      GVD_TRY(gvd_lstm_cell_fwd(&l, stream));
    }
    {  // vocabulary logits
      gvd_gemm_args g = {};
      g.nseg = 1;
      g.seg[0] = {w.h_lang[nxt], H, 0, a->logit_w, H, 0, H};
      g.nbias = a->logit_b;
      g.C = w.logits; g.ldc = V; g.M = B; g.N = V; g.batch = 1;
      GVD_TRY(gvd_gemm_nt_f32(&g, stream));
    }
    GVD_TRY(gvd_logsoftmax_top2_embed(w.logits, V, B, V, a->unk_idx, a->seq + t, L, a->seq_logprobs + t, L,
                                      a->embed, E, w.xt, E, stream));
    cur = nxt;
  }
  return 0;
}
