// Internal interface between decode.hip (C-ABI entry point gvd_greedy_decode) and decode_persistent.hip.
#pragma once
#include "gvd_common.h"

struct PdParams {
  // inputs (read-only for the whole launch)
  const float* fc_gates;                     // [B, 4H] loop-invariant att-LSTM gate terms incl. both biases
  const float* conv; const float* p_conv;    // temporal features [B,Ft,H] and their projections [B,Ft,A]
  const float* pool; const float* p_pool;    // region features [B,R,H] / [B,R,A]
  const uint8_t* pnt_mask;                   // [B, R+1]
  const float* embed;                        // [V, E]
  const float* att_w_ih; const float* att_w_hh;                      // [4H, H+E], [4H, H]
  const float* lang_w_ih; const float* lang_w_hh;                    // [4H, 2H], [4H, H]
  const float* lang_b_ih; const float* lang_b_hh;
  const float* q_w; const float* q_b;                                // stacked [W_att ; W_att2]: [2A, H], [2A]
  const float* a1_w; const float* a1_b; const float* a2_w; const float* a2_b;   // alpha nets (temporal, region)
  const float* logit_w; const float* logit_b;
  int B, Ft, R, V, L, unk;
  int chunk_r, nch_r, chunk_t, nch_t;        // attention chunking (filled by gvd_pd_launch)
  // exchange buffers, agent-coherent accesses only (carved from the workspace by gvd_pd_launch)
  float* h_att; float* h_lang; float* q12; float* att_sum; float* part_ctx; float* part_ml; float* stats;
  unsigned* sync;
  // outputs
  int64_t* seq; float* seq_lp; float* att2_weights; int* status;
  unsigned long long* trace;                 // optional: 1 + 7 L phase-boundary time stamps of workgroup 0
};

bool gvd_pd_shape_ok(int B, int H, int A, int E, int V, int R, int Ft);   // pure: sizes the workspace
bool gvd_pd_eligible(int B, int H, int A, int E, int V, int R, int Ft);   // shape + GVD_PERSISTENT + >= 256 CUs
size_t gvd_pd_workspace_bytes();
int gvd_pd_launch(PdParams p, void* workspace, hipStream_t st);
