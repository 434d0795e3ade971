// Persistent bidirectional GRU layer (frame-wise context encoder, model.py:150-154,399: nn.GRU(1024, 512,
// 2 layers, bidirectional, batch_first); gate order r,z,n).
//
// The input projections  gi = X W_ih^T + b_ih  of BOTH directions are one big MFMA GEMM done beforehand
// (gvd_gemm_nt_f32, N = 2*3*Hh).  This kernel runs the sequential part of one layer — for every time step
//   gh = h W_hh^T + b_hh ; r = sig(gi_r+gh_r) ; z = sig(gi_z+gh_z) ; n = tanh(gi_n + r*gh_n) ; h' = (1-z) n + z h
// — for the forward and the backward direction concurrently, as ONE cooperative launch: the library RNN the
// reference relies on issues ~6 small kernels per (step, direction, layer), i.e. ~5.8k launches for Ft=480,
// which is what makes the reference-default Ft=480 configuration latency-bound.
//
// Decomposition (MI355X): workgroup = (direction, slice of HU=8 hidden units) -> 2*64 = 128 workgroups, one per
// CU, all co-resident (cooperative launch); batches of more than 32 rows use two such groups (256 CUs), each taking
// half of the 32-row batch tiles.  The workgroup's 3*HU = 24 rows of W_hh are loaded ONCE and stay in
// registers for the whole sequence: wave w owns K-quarter [128w, 128w+128), lane (col = l&31, half = l>>5) holds
// the 64 values it feeds to the 32x32x2 fp32 MFMA as the B operand.  Per step every wave multiplies all 32-row
// batch tiles of h_{t-1} (read straight from the layer output tensor, the only state) with its K-quarter,
// partial tiles are summed through LDS, 256 threads apply the gate math to the (32 rows x 8 units) tile and
// write h_t into the output; a barrier (gvd_common.h: fence-free, over sc1-coherent state accesses) publishes h_t for the
// next step - grid-wide (two-level tree, 2.1 us vs 10.6 us for a release/acquire-fence counter) for the 8-unit form, and
// for the 16- / 32-unit forms one-level among the 32 / 16 workgroups of ONE (direction, batch-tile group): nobody else
// reads that h_t (1.05 us per step less at B = 256, and no waiting for the slowest workgroup of the whole grid).
#include "gvd_common.h"
#include <hip/hip_cooperative_groups.h>
#include <stdlib.h>

namespace cg = cooperative_groups;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GRU_HH = 512;     // hidden size per direction
constexpr int GRU_HU = 8;       // hidden units per workgroup (batches of at most 64 rows; 16 above: see gru_layer_kernel)
constexpr int GRU_NW = GRU_HH / GRU_HU;   // workgroups per direction at HU = 8
constexpr int MAX_TILES = 8;    // batch tiles of 32 rows per launch (B <= 256 per launch)
constexpr int LDA = GRU_HH + 4; // padded LDS row of an h tile: conflict-free ds_read_b128 over 32 rows

struct GruParams {
  const float* gi;       // [B, T, 2, 3*Hh]  input projections incl. b_ih (row stride = 6*Hh)
  const float* w_hh[2];  // [3*Hh, Hh] per direction
  const float* b_hh[2];  // [3*Hh]
  float* out;            // [B, T, 2*Hh]
  unsigned* sync;        // GVD_SYNC_WORDS words of grid_barrier_tree state (zeroed by the host before the launch)
  int B, T;
};

// HU = hidden units per workgroup.  HU = 8: 2 x 64 workgroups per group (24 of the 32 MFMA columns used), up to two groups
// sharing the batch tiles.  HU = 16 (batches of more than 64 rows): 2 x 32 workgroups per group, 48 columns = two MFMA column
// tiles per batch tile, up to FOUR groups - every workgroup then stages (and waits at a barrier pair for) half as many 64 KB
// h_{t-1} tiles per time step, which is what the step cost at B = 256 was made of (19.6 us: 4 tiles x (staging from L2 +
// two barriers) per workgroup against 6.8 us of MFMAs).  HU = 32 (more than 128 rows): 2 x 16 workgroups per group, 96
// columns = three FULL column tiles, up to EIGHT groups: one batch tile per workgroup and step at B = 256 - one staged tile
// and 192 MFMAs per wave and step instead of two tiles and 256 (a quarter of HU = 16's MFMA columns are padding).  Its
// h_{t-1} buffer is single (66 KB + the 50 KB of partial sums; a second tile, when fewer groups are co-resident, is staged
// after the barrier that ends the previous tile's MFMA phase).  Same k order per output -> bitwise equal results for
// every HU.
template <bool CG_SYNC, int HU>
__global__ __launch_bounds__(256, 1) void gru_layer_kernel(const GruParams p) {
  constexpr int NWG = GRU_HH / HU;                 // workgroups per direction and group
  constexpr int NCT = (3 * HU + 31) / 32;          // MFMA column tiles (1 or 2)
  constexpr int NU = HU / 8;                       // hidden units per thread in the gate phase
  constexpr int LDP = 3 * HU + 1;                  // s_part row stride (odd: conflict-free column reads)
  constexpr int NBUF = HU >= 32 ? 1 : 2;           // h_{t-1} tile buffers
  __shared__ __attribute__((aligned(16))) float s_a[NBUF][32 * LDA];   // (double-buffered) h_{t-1} batch tiles
  __shared__ float s_part[4][32][LDP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, half = lane >> 5;
  // grid = nparts x (2 directions x NWG unit slices): the batch tiles are dealt to the nparts groups
  const int lid = blockIdx.x % (2 * NWG), part = blockIdx.x / (2 * NWG), nparts = gridDim.x / (2 * NWG);
  const int dir = lid / NWG;
  const int j0 = (lid % NWG) * HU;
  const int B = p.B, T = p.T;
  const int nt_all = (B + 31) / 32;
  const int nt_per = (nt_all + nparts - 1) / nparts;
  const int tile0 = part * nt_per;
  const int ntiles = max(0, min(nt_per, nt_all - tile0));
  const int64_t ld_out = (int64_t)T * 2 * GRU_HH;     // batch stride of out
  const int64_t ld_gi = (int64_t)T * 6 * GRU_HH;
  const unsigned nwg = gridDim.x;
  // h_t is exchanged between workgroups inside this launch: all accesses of `out` are agent-coherent (sc1)
  const __amdgpu_buffer_rsrc_t out_rs = gvd_rsrc(p.out);

  // ---- this lane's slice of W_hh: column c = 32 ct + col of the workgroup's 3 HU columns = gate c / HU, unit
  //      j0 + c % HU (columns >= 3 HU unused); register-resident for the whole sequence
  f32x4 wreg[NCT][16];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int c = 32 * ct + col;
    const bool used = c < 3 * HU;
    const int wrow = used ? (c / HU) * GRU_HH + j0 + (c % HU) : 0;
    const float* wp = p.w_hh[dir] + (int64_t)wrow * GRU_HH + wave * 128 + half * 4;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (used) v = *reinterpret_cast<const f32x4*>(wp + kb * 8);
      wreg[ct][kb] = v;
    }
  }
  // gate-phase role: thread = (row r = tid/8 of the batch tile, the NU CONSECUTIVE units jj = NU (tid%8) + u, u < NU): its gi
  // gates, its own h_{t-1} and its h_t are one NU-wide vector access each (16 bytes at HU = 32)
  const int g_row = tid >> 3, g_jj = tid & 7;
  float bh_r[NU], bh_z[NU], bh_n[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    bh_r[u] = p.b_hh[dir][j0 + NU * g_jj + u];
    bh_z[u] = p.b_hh[dir][GRU_HH + j0 + NU * g_jj + u];
    bh_n[u] = p.b_hh[dir][2 * GRU_HH + j0 + NU * g_jj + u];
  }
  // staging role: 16 x 16-byte pieces per thread per tile; piece i -> tile row (tid + 256 i) / 128, float4 column % 128
  // (a wave-load covers 1 KiB contiguous of one sample's h_{t-1}: fully coalesced)

  bool dead = false;   // latched barrier timeout (thread 0)
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    const int64_t off_t = (int64_t)t * 2 * GRU_HH + dir * GRU_HH;
    const int64_t off_tp = (int64_t)tp * 2 * GRU_HH + dir * GRU_HH;

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
        *reinterpret_cast<f32x4*>(&buf[(idx >> 7) * LDA + (idx & 127) * 4]) = ra[i];
      }
    };
    // gate-phase inputs (gi_r, gi_z, gi_n, own h_{t-1}) are fetched one batch tile ahead so their latency hides
    // behind the MFMA phase of the current tile
    float cur_r[NU], cur_z[NU], cur_n[NU], cur_h[NU], nxt_r[NU], nxt_z[NU], nxt_n[NU], nxt_h[NU];
    auto load_gate_inputs = [&](int mt, float* r_, float* z_, float* n_, float* h_) {
      const int b = (tile0 + mt) * 32 + g_row;
#pragma unroll
      for (int u = 0; u < NU; ++u) r_[u] = z_[u] = n_[u] = h_[u] = 0.f;
      if (b < B) {
        typedef float vecu __attribute__((ext_vector_type(NU)));
        const float* gip = p.gi + (int64_t)b * ld_gi + (int64_t)t * 6 * GRU_HH + dir * 3 * GRU_HH + j0 + NU * g_jj;
        const unsigned hoff = (unsigned)(((int64_t)b * ld_out + off_tp + j0 + NU * g_jj) * 4);
        if constexpr (NU == 1) {
          r_[0] = gip[0]; z_[0] = gip[GRU_HH]; n_[0] = gip[2 * GRU_HH];
          if (step > 0) h_[0] = ld_agent_f32(out_rs, hoff);
        } else {
          const vecu vr = *reinterpret_cast<const vecu*>(gip), vz = *reinterpret_cast<const vecu*>(gip + GRU_HH),
                     vn = *reinterpret_cast<const vecu*>(gip + 2 * GRU_HH);
#pragma unroll
          for (int u = 0; u < NU; ++u) { r_[u] = vr[u]; z_[u] = vz[u]; n_[u] = vn[u]; }
          if (step > 0) {
            if constexpr (NU == 4) {
              const f32x4 vh = ld_agent_x4(out_rs, hoff);
#pragma unroll
              for (int u = 0; u < NU; ++u) h_[u] = vh[u];
            } else {
              const gvd_f32x2 vh = ld_agent_x2(out_rs, hoff);
              h_[0] = vh[0]; h_[1] = vh[1];
            }
          }
        }
      }
    };
#pragma unroll
    for (int u = 0; u < NU; ++u) cur_r[u] = cur_z[u] = cur_n[u] = cur_h[u] = nxt_r[u] = nxt_z[u] = nxt_n[u] = nxt_h[u] = 0.f;
    if (ntiles > 0) {
      load_gate_inputs(0, cur_r, cur_z, cur_n, cur_h);
      if (step > 0) load_tile(0);
    }

    // per batch tile: stage h_{t-1} through LDS, gh partials on the MFMA (lane supplies A[i = col][k = 128*wave +
    // 8*kb + 4*half + s]), K-quarter partials summed through LDS, gate math, write h_t
#pragma unroll 1
    for (int mt = 0; mt < ntiles; ++mt) {
      if (step > 0) {
        float* abuf = s_a[NBUF == 2 ? (mt & 1) : 0];
        store_tile(abuf);
        __syncthreads();
        if (mt + 1 < ntiles) load_tile(mt + 1);            // next tile's loads fly during this tile's MFMAs
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
        const float* ap = abuf + col * LDA + wave * 128 + half * 4;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kb * 8);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
              acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wreg[ct][kb][s], acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          if (32 * ct + col < 3 * HU) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
              s_part[wave][row][32 * ct + col] = acc[ct][e];
            }
          }
        }
      }
      if (mt + 1 < ntiles) load_gate_inputs(mt + 1, nxt_r, nxt_z, nxt_n, nxt_h);
      __syncthreads();
      const int b = (tile0 + mt) * 32 + g_row;
      if (b < B) {
        float hn[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int jj = NU * g_jj + u;
          float gr = bh_r[u], gz = bh_z[u], gn = bh_n[u];
          if (step > 0) {
            gr += s_part[0][g_row][jj] + s_part[1][g_row][jj] + s_part[2][g_row][jj] + s_part[3][g_row][jj];
            gz += s_part[0][g_row][HU + jj] + s_part[1][g_row][HU + jj] + s_part[2][g_row][HU + jj] +
                  s_part[3][g_row][HU + jj];
            gn += s_part[0][g_row][2 * HU + jj] + s_part[1][g_row][2 * HU + jj] +
                  s_part[2][g_row][2 * HU + jj] + s_part[3][g_row][2 * HU + jj];
          }
          // (explicit fused multiply-adds: the contraction the compiler picks must not depend on the code shape around it
          // - the unit forms HU = 8 / 16 / 32 differ in exactly that - or results would depend on the batch size)
          const float r = sigmoid_f(cur_r[u] + gr);
          const float z = sigmoid_f(cur_z[u] + gz);
          const float n = tanhf(__builtin_fmaf(r, gn, cur_n[u]));
          hn[u] = __builtin_fmaf(z, cur_h[u], __fmul_rn(1.f - z, n));
        }
        const unsigned ooff = (unsigned)(((int64_t)b * ld_out + off_t + j0 + NU * g_jj) * 4);
        if constexpr (NU == 4) {
          const f32x4 v = {hn[0], hn[1], hn[2], hn[3]};
          st_agent_x4(out_rs, ooff, v);
        } else if constexpr (NU == 2) {
          const gvd_f32x2 v = {hn[0], hn[1]};
          st_agent_x2(out_rs, ooff, v);
        } else {
          st_agent_f32(out_rs, ooff, hn[0]);
        }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) { cur_r[u] = nxt_r[u]; cur_z[u] = nxt_z[u]; cur_n[u] = nxt_n[u]; cur_h[u] = nxt_h[u]; }
      // s_part is rewritten only after the next tile's staging barrier (or the grid barrier); s_a[mt&1] is rewritten
      // two tiles later, i.e. after two more barriers (the single buffer of HU = 32: by the next tile's staging, which
      // every wave reaches after the barrier above, i.e. after all MFMA reads of this tile): no extra barrier needed here.
      // For step == 0 (no staging barrier) s_part is not used at all.
    }
    // (3) publish h_t grid-wide before step t+1 reads it
    if (CG_SYNC) {
      // every storing wave drains its own stores: the library sync's leader lane issues the agent-scope release
      // (buffer_wbl2), which only covers stores that already reached L2
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      cg::this_grid().sync();
    } else if (NWG <= 32) {
      // h_t of (direction, batch-tile group) is produced and consumed by that sub-grid's NWG workgroups only
      subgrid_barrier(p.sync, (unsigned)step, (unsigned)(2 * part + dir), NWG, dead);
    } else {
      grid_barrier_tree(p.sync, (unsigned)step, nwg, dead);
    }
  }
}

int gru_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  return cus;
}

// One reverse step of the layer's BPTT for BOTH directions (training; nn.GRU backward).  Direction d handles time
// t_d (forward direction walks T-1..0, backward direction 0..T-1).  With the gates recomputed from gi (saved) and
// gh = h_{t-1} W_hh^T + b_hh (one GEMM over all steps, the hidden sequence IS the layer output):
//   dh  = dout[t] + carry_mm + carry_z        (carry_mm = d_gh[t'] W_hh of the step before, carry_z = dh' z')
//   dn  = dh (1-z) (1-n^2) ; dz = dh (h_prev - n) z (1-z) ; dr = dn gh_n r (1-r)
//   d_gi[t] = [dr, dz, dn] ; d_gh[t] = [dr, dz, dn r] ; carry_z = dh z
// One thread per (direction, row, hidden unit); latency-trivial (2*B*Hh threads), exists to replace ~15 library
// launches per (step, direction).
__global__ __launch_bounds__(256) void gru_bwd_step_kernel(const float* __restrict__ dout, const float* __restrict__ gi,
                                                           const float* __restrict__ gh, const float* __restrict__ out,
                                                           const float* __restrict__ carry_mm, float* __restrict__ carry_z,
                                                           float* __restrict__ d_gi, float* __restrict__ d_gh, int B, int T,
                                                           int Hh, int t_fw, int t_bw, int first) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)2 * B * Hh) return;
  const int j = (int)(i % Hh);
  const int b = (int)((i / Hh) % B);
  const int d = (int)(i / ((int64_t)Hh * B));
  const int t = d ? t_bw : t_fw;
  const int tp = d ? t + 1 : t - 1;
  const int64_t g0 = (((int64_t)b * T + t) * 2 + d) * 3 * Hh + j;
  const float ghn = gh[g0 + 2 * Hh];
  const float r = 1.f / (1.f + expf(-(gi[g0] + gh[g0])));
  const float z = 1.f / (1.f + expf(-(gi[g0 + Hh] + gh[g0 + Hh])));
  const float n = tanhf(fmaf(r, ghn, gi[g0 + 2 * Hh]));
  const float hp = (tp >= 0 && tp < T) ? out[((int64_t)b * T + tp) * 2 * Hh + d * Hh + j] : 0.f;
  const int64_t ci = ((int64_t)d * B + b) * Hh + j;
  float dh = dout[((int64_t)b * T + t) * 2 * Hh + d * Hh + j];
  if (!first) dh += carry_mm[ci] + carry_z[ci];
  const float dn = dh * (1.f - z) * (1.f - n * n);
  const float dz = dh * (hp - n) * z * (1.f - z);
  const float dr = dn * ghn * r * (1.f - r);
  d_gi[g0] = dr; d_gi[g0 + Hh] = dz; d_gi[g0 + 2 * Hh] = dn;
  d_gh[g0] = dr; d_gh[g0 + Hh] = dz; d_gh[g0 + 2 * Hh] = dn * r;
  carry_z[ci] = dh * z;
}

}  // namespace

extern "C" int gvd_gru_bwd_step(const float* dout, const float* gi, const float* gh, const float* out,
                                const float* carry_mm, float* carry_z, float* d_gi, float* d_gh, int B, int T, int Hh,
                                int t_fw, int t_bw, int first, gvd_stream_t stream) {
  if (!dout || !gi || !gh || !out || !carry_z || !d_gi || !d_gh || (!first && !carry_mm) || B <= 0 || T <= 0 || Hh <= 0 ||
      t_fw < 0 || t_fw >= T || t_bw < 0 || t_bw >= T)
    return GVD_EINVAL;
  const int64_t n = (int64_t)2 * B * Hh;
  hipLaunchKernelGGL(gru_bwd_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, gvd_s(stream), dout, gi, gh, out,
                     carry_mm, carry_z, d_gi, d_gh, B, T, Hh, t_fw, t_bw, first);
  GVD_CHECK_LAUNCH();
  return 0;
}

extern "C" int gvd_gru_bidir_layer(const float* gi, const float* w_hh_fw, const float* b_hh_fw, const float* w_hh_bw,
                                   const float* b_hh_bw, float* out, int B, int T, int Hh, void* sync_ws,
                                   gvd_stream_t stream) {
  if (!gi || !w_hh_fw || !b_hh_fw || !w_hh_bw || !b_hh_bw || !out || B <= 0 || T <= 0 || Hh != GRU_HH) return GVD_EINVAL;
  if (!gvd_aligned16(gi) || !gvd_aligned16(w_hh_fw) || !gvd_aligned16(w_hh_bw) || !gvd_aligned16(out)) return GVD_EINVAL;
  hipStream_t st = gvd_s(stream);
  const int nslices = (B + MAX_TILES * 32 - 1) / (MAX_TILES * 32);
  // batches beyond MAX_TILES*32 rows run as consecutive launches over batch slices (samples are independent)
  for (int si = 0; si < nslices; ++si) {
    const int b0 = si * MAX_TILES * 32;
    GruParams p;
    const int nb = (B - b0 < MAX_TILES * 32) ? (B - b0) : MAX_TILES * 32;
    p.gi = gi + (int64_t)b0 * T * 6 * GRU_HH;
    p.w_hh[0] = w_hh_fw; p.w_hh[1] = w_hh_bw; p.b_hh[0] = b_hh_fw; p.b_hh[1] = b_hh_bw;
    p.out = out + (int64_t)b0 * T * 2 * GRU_HH;
    p.B = nb; p.T = T;
    p.sync = sync_ws ? reinterpret_cast<unsigned*>(sync_ws) + (size_t)GVD_SYNC_WORDS * si : nullptr;
    if ((int64_t)nb * T * 2 * GRU_HH * 4 >= (int64_t)0x7fffffff) return GVD_EINVAL;   // 32-bit buffer offsets
    void* args[] = {&p};
    // cooperative launch in both modes: it validates that all workgroups are co-resident.
    // More than 64 rows (3+ batch tiles): 16 hidden units per workgroup, the tiles dealt to up to 4 groups of 64
    // workgroups; otherwise 8 units per workgroup and up to 2 groups of 128.
    // The hand-rolled barrier form is launched PLAINLY after an explicit co-residency check (gvd_grid_fits; the cooperative
    // path costs a ~12 us dispatch gap on either side of the kernel; GVD_COOP_LAUNCH=1 restores it); the library grid sync
    // needs the cooperative launch.
    static const bool coop_env = getenv("GVD_COOP_LAUNCH") ? atoi(getenv("GVD_COOP_LAUNCH")) != 0 : false;
    const bool coop = coop_env || !sync_ws;
    auto launch = [&](const void* f, unsigned nwg) {
      if (coop) return hipLaunchCooperativeKernel(f, dim3(nwg), dim3(256), args, 0, st);
      if (!gvd_grid_fits(f, 256, (int)nwg)) return hipErrorCooperativeLaunchTooLarge;
      return hipLaunchKernel(f, dim3(nwg), dim3(256), args, 0, st);
    };
    const bool wide = nb > 64;
    const int ntiles = (nb + 31) / 32;
    hipError_t e = hipErrorUnknown;
    bool done = false;
    if (ntiles > 4 && gru_cus() >= ntiles * 2 * (GRU_HH / 32)) {
      // more than 128 rows: 32 hidden units per workgroup, ONE batch tile per workgroup and step (ntiles groups of 32)
      const void* fn = sync_ws ? reinterpret_cast<const void*>(gru_layer_kernel<false, 32>)
                               : reinterpret_cast<const void*>(gru_layer_kernel<true, 32>);
      e = launch(fn, (unsigned)(ntiles * 2 * (GRU_HH / 32)));
      done = e == hipSuccess;
      if (!done) (void)hipGetLastError();      // not co-resident here: the 16-unit form below, with fewer groups
    }
    if (done) {
    } else if (wide) {
      const void* fn = sync_ws ? reinterpret_cast<const void*>(gru_layer_kernel<false, 16>)
                               : reinterpret_cast<const void*>(gru_layer_kernel<true, 16>);
      const int per = 2 * (GRU_HH / 16);
      int nparts = gru_cus() / per;
      if (nparts > ntiles) nparts = ntiles;
      if (nparts > 4) nparts = 4;
      if (nparts < 1) nparts = 1;
      e = launch(fn, (unsigned)(nparts * per));
      while (e != hipSuccess && nparts > 1) {   // not co-resident here: fewer groups
        (void)hipGetLastError();
        nparts /= 2;
        e = launch(fn, (unsigned)(nparts * per));
      }
    } else {
      const void* fn = sync_ws ? reinterpret_cast<const void*>(gru_layer_kernel<false, 8>)
                               : reinterpret_cast<const void*>(gru_layer_kernel<true, 8>);
      const int nparts = (nb > 32 && gru_cus() >= 4 * GRU_NW) ? 2 : 1;
      e = launch(fn, (unsigned)(nparts * 2 * GRU_NW));
      if (e != hipSuccess && nparts == 2) {   // 256 workgroups not co-resident here: one group of 128
        (void)hipGetLastError();
        e = launch(fn, (unsigned)(2 * GRU_NW));
      }
    }
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

extern "C" int gvd_grid_sync_words(void) { return GVD_SYNC_WORDS; }
