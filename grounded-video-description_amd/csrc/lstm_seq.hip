// Persistent bidirectional LSTM layer: the frame-wise context encoder under `--t_attn_mode bilstm` (opts.py:60; model.py:145-
// 149,399: nn.LSTM(1024, 512, 2 layers, bidirectional, batch_first); gate order i,f,g,o).
//
// Same decomposition as the bi-GRU layer (csrc/gru.hip): the input projections gi = X W_ih^T + b_ih of both directions are
// one MFMA GEMM done beforehand (N = 2 * 4 * Hh); this kernel runs the sequential part of one layer for both directions
// concurrently as ONE launch of co-resident workgroups.  Workgroup = (direction, 8 hidden units): its 4 x 8 = 32 rows of
// W_hh are exactly ONE 32-column MFMA tile (no padding columns, against 24 of 32 for the GRU's three gates) and stay in
// registers for the whole sequence; per step every wave multiplies the 32-row batch tiles of h_{t-1} (read from the layer
// output, agent-coherent) with its K-quarter, the partial tiles are summed through LDS, 256 threads apply the gate math
//   i = sig(.), f = sig(.), g = tanh(.), o = sig(.) ; c' = f c + i g ; h' = o tanh(c')
// and a fence-free grid barrier publishes h_t.  The cell state of (row, unit) is read and written by the SAME thread at every
// step (plain accesses to a [B,2,Hh] scratch array).  Training (gates_seq / c_seq given): the post-activation gates and the
// cell states of every step are kept for the BPTT (lstm_fn.py).
#include "gvd_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int L_HH = 512;              // hidden size per direction
constexpr int L_HU = 8;                // hidden units per workgroup: 4 gates x 8 units = one MFMA column tile
constexpr int L_NW = L_HH / L_HU;      // workgroups per direction
constexpr int L_MAX_TILES = 8;         // batch tiles of 32 rows per launch
constexpr int L_LDA = L_HH + 4;        // padded LDS row of an h tile
constexpr int L_LDP = 4 * L_HU + 1;    // s_part row stride (odd: conflict-free column reads)

struct LstmParams {
  const float* gi;       // [B, T, 2, 4*Hh]  input projections incl. b_ih
  const float* w_hh[2];  // [4*Hh, Hh] per direction
  const float* b_hh[2];  // [4*Hh]
  float* out;            // [B, T, 2*Hh]
  float* c_state;        // [B, 2, Hh] scratch: the running cell state
  float* gates_seq;      // optional [B, T, 2, 4*Hh]: post-activation gates of every step
  float* c_seq;          // optional [B, T, 2, Hh]: cell state after every step
  unsigned* sync;        // GVD_SYNC_WORDS words of grid_barrier_tree state (zeroed by the host before the launch)
  int B, T;
};

__global__ __launch_bounds__(256, 1) void lstm_layer_kernel(const LstmParams p) {
  __shared__ __attribute__((aligned(16))) float s_a[2][32 * L_LDA];   // double-buffered h_{t-1} batch tiles
  __shared__ float s_part[4][32][L_LDP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, half = lane >> 5;
  // grid = nparts x (2 directions x L_NW unit slices): the batch tiles are dealt to the nparts groups
  const int lid = blockIdx.x % (2 * L_NW), part = blockIdx.x / (2 * L_NW), nparts = gridDim.x / (2 * L_NW);
  const int dir = lid / L_NW;
  const int j0 = (lid % L_NW) * L_HU;
  const int B = p.B, T = p.T;
  const int nt_all = (B + 31) / 32;
  const int nt_per = (nt_all + nparts - 1) / nparts;
  const int tile0 = part * nt_per;
  const int ntiles = max(0, min(nt_per, nt_all - tile0));
  const int64_t ld_out = (int64_t)T * 2 * L_HH;     // batch stride of out
  const int64_t ld_gi = (int64_t)T * 8 * L_HH;
  const unsigned nwg = gridDim.x;
  const __amdgpu_buffer_rsrc_t out_rs = gvd_rsrc(p.out);   // h_t is exchanged between workgroups inside this launch (sc1)

  // this lane's slice of W_hh: column col of the workgroup's 32 = gate col / 8, unit j0 + col % 8; K-quarter of the wave
  f32x4 wreg[16];
  {
    const int wrow = (col / L_HU) * L_HH + j0 + (col % L_HU);
    const float* wp = p.w_hh[dir] + (int64_t)wrow * L_HH + wave * 128 + half * 4;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) wreg[kb] = *reinterpret_cast<const f32x4*>(wp + kb * 8);
  }
  // gate-phase role: thread = (row tid / 8 of the batch tile, unit j0 + tid % 8)
  const int g_row = tid >> 3, g_jj = tid & 7;
  float bh[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bh[q] = p.b_hh[dir][q * L_HH + j0 + g_jj];

  bool dead = false;   // latched barrier timeout (thread 0)
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    const int64_t off_t = (int64_t)t * 2 * L_HH + dir * L_HH;
    const int64_t off_tp = (int64_t)tp * 2 * L_HH + dir * L_HH;

    f32x4 ra[16];
    auto load_tile = [&](int mt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int idx = tid + i * 256;
        const int b = (tile0 + mt) * 32 + (idx >> 7);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b < B) v = ld_agent_x4(out_rs, (unsigned)(((int64_t)b * ld_out + off_tp + (idx & 127) * 4) * 4));
        ra[i] = v;
      }
    };
    auto store_tile = [&](float* buf) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int idx = tid + i * 256;
        *reinterpret_cast<f32x4*>(&buf[(idx >> 7) * L_LDA + (idx & 127) * 4]) = ra[i];
      }
    };
    // gate-phase inputs (the four gi gates, the thread's own c_{t-1}) are fetched one batch tile ahead
    float cur[5], nxt[5];
    auto load_gate_inputs = [&](int mt, float* g_) {
      const int b = (tile0 + mt) * 32 + g_row;
#pragma unroll
      for (int q = 0; q < 5; ++q) g_[q] = 0.f;
      if (b < B) {
        const float* gip = p.gi + (int64_t)b * ld_gi + (int64_t)t * 8 * L_HH + dir * 4 * L_HH + j0 + g_jj;
#pragma unroll
        for (int q = 0; q < 4; ++q) g_[q] = gip[q * L_HH];
        if (step > 0) g_[4] = p.c_state[((int64_t)b * 2 + dir) * L_HH + j0 + g_jj];
      }
    };
#pragma unroll
    for (int q = 0; q < 5; ++q) cur[q] = nxt[q] = 0.f;
    if (ntiles > 0) {
      load_gate_inputs(0, cur);
      if (step > 0) load_tile(0);
    }

#pragma unroll 1
    for (int mt = 0; mt < ntiles; ++mt) {
      if (step > 0) {
        float* abuf = s_a[mt & 1];
        store_tile(abuf);
        __syncthreads();
        if (mt + 1 < ntiles) load_tile(mt + 1);            // next tile's loads fly during this tile's MFMAs
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float* ap = abuf + col * L_LDA + wave * 128 + half * 4;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kb * 8);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wreg[kb][s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
          s_part[wave][row][col] = acc[e];
        }
      }
      if (mt + 1 < ntiles) load_gate_inputs(mt + 1, nxt);
      __syncthreads();
      const int b = (tile0 + mt) * 32 + g_row;
      if (b < B) {
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float gh = bh[q];
          if (step > 0)
            gh += s_part[0][g_row][q * L_HU + g_jj] + s_part[1][g_row][q * L_HU + g_jj] + s_part[2][g_row][q * L_HU + g_jj] +
                  s_part[3][g_row][q * L_HU + g_jj];
          pre[q] = cur[q] + gh;
        }
        const float gi_ = sigmoid_f(pre[0]), gf = sigmoid_f(pre[1]), gg = tanhf(pre[2]), go = sigmoid_f(pre[3]);
        const float c_new = __builtin_fmaf(gf, cur[4], __fmul_rn(gi_, gg));
        const float h_new = __fmul_rn(go, tanhf(c_new));
        st_agent_f32(out_rs, (unsigned)(((int64_t)b * ld_out + off_t + j0 + g_jj) * 4), h_new);
        p.c_state[((int64_t)b * 2 + dir) * L_HH + j0 + g_jj] = c_new;
        if (p.gates_seq) {
          float* gs = p.gates_seq + (int64_t)b * ld_gi + (int64_t)t * 8 * L_HH + dir * 4 * L_HH + j0 + g_jj;
          gs[0] = gi_; gs[L_HH] = gf; gs[2 * L_HH] = gg; gs[3 * L_HH] = go;
        }
        if (p.c_seq) p.c_seq[(int64_t)b * ld_out + off_t + j0 + g_jj] = c_new;
      }
#pragma unroll
      for (int q = 0; q < 5; ++q) cur[q] = nxt[q];
      // (s_part is rewritten only after the next tile's staging barrier or the grid barrier; s_a[mt & 1] two tiles later)
    }
    grid_barrier_tree(p.sync, (unsigned)step, nwg, dead);     // publish h_t before step t + 1 reads it
  }
}

int lstm_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  return cus;
}

}  // namespace

extern "C" int gvd_lstm_bidir_layer(const float* gi, const float* w_hh_fw, const float* b_hh_fw, const float* w_hh_bw,
                                    const float* b_hh_bw, float* out, float* c_state, float* gates_seq, float* c_seq, int B,
                                    int T, int Hh, void* sync_ws, gvd_stream_t stream) {
  if (!gi || !w_hh_fw || !b_hh_fw || !w_hh_bw || !b_hh_bw || !out || !c_state || !sync_ws || B <= 0 || T <= 0 || Hh != L_HH)
    return GVD_EINVAL;
  if (!gvd_aligned16(gi) || !gvd_aligned16(w_hh_fw) || !gvd_aligned16(w_hh_bw) || !gvd_aligned16(out)) return GVD_EINVAL;
  hipStream_t st = gvd_s(stream);
  const int per = L_MAX_TILES * 32;
  const int nslices = (B + per - 1) / per;
  // batches beyond 256 rows run as consecutive launches over batch slices (samples are independent)
  for (int si = 0; si < nslices; ++si) {
    const int b0 = si * per;
    const int nb = (B - b0 < per) ? (B - b0) : per;
    LstmParams p;
    p.gi = gi + (int64_t)b0 * T * 8 * L_HH;
    p.w_hh[0] = w_hh_fw; p.w_hh[1] = w_hh_bw; p.b_hh[0] = b_hh_fw; p.b_hh[1] = b_hh_bw;
    p.out = out + (int64_t)b0 * T * 2 * L_HH;
    p.c_state = c_state + (int64_t)b0 * 2 * L_HH;
    p.gates_seq = gates_seq ? gates_seq + (int64_t)b0 * T * 8 * L_HH : nullptr;
    p.c_seq = c_seq ? c_seq + (int64_t)b0 * T * 2 * L_HH : nullptr;
    p.B = nb; p.T = T;
    p.sync = reinterpret_cast<unsigned*>(sync_ws) + (size_t)GVD_SYNC_WORDS * si;
    if ((int64_t)nb * T * 2 * L_HH * 4 >= (int64_t)0x7fffffff) return GVD_EINVAL;   // 32-bit buffer offsets
    void* args[] = {&p};
    const void* fn = reinterpret_cast<const void*>(lstm_layer_kernel);
    // plain launch after an explicit co-residency check (the hand-rolled barrier needs every workgroup resident)
    int nparts = (nb > 32 && lstm_cus() >= 4 * L_NW) ? 2 : 1;
    if (nparts == 2 && !gvd_grid_fits(fn, 256, 4 * L_NW)) nparts = 1;
    if (!gvd_grid_fits(fn, 256, nparts * 2 * L_NW)) return (int)hipErrorCooperativeLaunchTooLarge;
    const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)(nparts * 2 * L_NW)), dim3(256), args, 0, st);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
