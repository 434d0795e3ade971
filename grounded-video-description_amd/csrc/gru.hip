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
// CU, all co-resident (cooperative launch).  The workgroup's 3*HU = 24 rows of W_hh are loaded ONCE and stay in
// registers for the whole sequence: wave w owns K-quarter [128w, 128w+128), lane (col = l&31, half = l>>5) holds
// the 64 values it feeds to the 32x32x2 fp32 MFMA as the B operand.  Per step every wave multiplies all 32-row
// batch tiles of h_{t-1} (read straight from the layer output tensor, the only state) with its K-quarter,
// partial tiles are summed through LDS, 256 threads apply the gate math to the (32 rows x 8 units) tile and
// write h_t into the output; a grid-wide sync publishes h_t for the next step.
#include "gvd_common.h"
#include <hip/hip_cooperative_groups.h>

namespace cg = cooperative_groups;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GRU_HH = 512;     // hidden size per direction
constexpr int GRU_HU = 8;       // hidden units per workgroup
constexpr int GRU_NW = GRU_HH / GRU_HU;   // workgroups per direction
constexpr int MAX_TILES = 8;    // batch tiles of 32 rows per pass (B <= 256 per launch)

struct GruParams {
  const float* gi;       // [B, T, 2, 3*Hh]  input projections incl. b_ih (row stride = 6*Hh)
  const float* w_hh[2];  // [3*Hh, Hh] per direction
  const float* b_hh[2];  // [3*Hh]
  float* out;            // [B, T, 2*Hh]
  int B, T;
};

__global__ __launch_bounds__(256, 1) void gru_layer_kernel(const GruParams p) {
  __shared__ float s_part[4][32][33];
  cg::grid_group grid = cg::this_grid();
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int col = lane & 31, half = lane >> 5;
  const int dir = blockIdx.x / GRU_NW;
  const int j0 = (blockIdx.x % GRU_NW) * GRU_HU;
  const int B = p.B, T = p.T;
  const int ntiles = (B + 31) / 32;
  const int64_t ld_out = (int64_t)T * 2 * GRU_HH;     // batch stride of out
  const int64_t ld_gi = (int64_t)T * 6 * GRU_HH;

  // ---- this lane's slice of W_hh: column `col` of the tile = gate col/HU, unit j0 + col%HU (cols >= 24 unused)
  f32x4 wreg[16];
  {
    const bool used = col < 3 * GRU_HU;
    const int wrow = used ? (col / GRU_HU) * GRU_HH + j0 + (col % GRU_HU) : 0;
    const float* wp = p.w_hh[dir] + (int64_t)wrow * GRU_HH + wave * 128 + half * 4;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (used) v = *reinterpret_cast<const f32x4*>(wp + kb * 8);
      wreg[kb] = v;
    }
  }
  // gate-phase role: thread = (row r = tid/8 of the batch tile, unit jj = tid%8)
  const int g_row = tid >> 3, g_jj = tid & 7;
  const float bh_r = p.b_hh[dir][j0 + g_jj];
  const float bh_z = p.b_hh[dir][GRU_HH + j0 + g_jj];
  const float bh_n = p.b_hh[dir][2 * GRU_HH + j0 + g_jj];

  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    const int64_t off_t = (int64_t)t * 2 * GRU_HH + dir * GRU_HH;
    const int64_t off_tp = (int64_t)tp * 2 * GRU_HH + dir * GRU_HH;

    // (1) issue the gate-phase inputs of every batch tile now (gi_r, gi_z, gi_n, own h_{t-1}): their latency
    //     hides behind the MFMA phase instead of being paid once per tile after each LDS reduction
    float pre_r[MAX_TILES], pre_z[MAX_TILES], pre_n[MAX_TILES], pre_h[MAX_TILES];
#pragma unroll
    for (int mt = 0; mt < MAX_TILES; ++mt) {
      pre_r[mt] = pre_z[mt] = pre_n[mt] = pre_h[mt] = 0.f;
      const int b = mt * 32 + g_row;
      if (mt < ntiles && b < B) {
        const float* gip = p.gi + (int64_t)b * ld_gi + (int64_t)t * 6 * GRU_HH + dir * 3 * GRU_HH + j0 + g_jj;
        pre_r[mt] = gip[0]; pre_z[mt] = gip[GRU_HH]; pre_n[mt] = gip[2 * GRU_HH];
        if (step > 0) pre_h[mt] = p.out[(int64_t)b * ld_out + off_tp + j0 + g_jj];
      }
    }

    // (2) gh partials: h_{t-1}[b, k] lives in out[b, tp, dir*Hh + k]; lane supplies A[i = col][k = 128*wave + 8*kb +
    //     4*half + s].  The 16 x 16-byte loads of tile mt+1 are in flight while the 64 MFMAs of tile mt issue.
    f32x16 acc[MAX_TILES];
#pragma unroll
    for (int mt = 0; mt < MAX_TILES; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;
    if (step > 0) {
      f32x4 abuf[2][16];
      auto load_a = [&](f32x4* dst, int mt) {
        const int b = mt * 32 + col;
        const bool ok = b < B;
        const float* hp = p.out + (int64_t)(ok ? b : 0) * ld_out + off_tp + wave * 128 + half * 4;
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
          if (ok) a = *reinterpret_cast<const f32x4*>(hp + kb * 8);
          dst[kb] = a;
        }
      };
      load_a(abuf[0], 0);
#pragma unroll
      for (int mt = 0; mt < MAX_TILES; ++mt) {
        if (mt < ntiles) {
          if (mt + 1 < ntiles) load_a(abuf[(mt + 1) & 1], mt + 1);
#pragma unroll
          for (int kb = 0; kb < 16; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(abuf[mt & 1][kb][s], wreg[kb][s], acc[mt], 0, 0, 0);
        }
      }
    }

    // (3) per batch tile: sum the 4 K-quarter partials through LDS, gate math, write h_t
#pragma unroll
    for (int mt = 0; mt < MAX_TILES; ++mt) {
      if (mt < ntiles) {
        if (step > 0) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
            s_part[wave][row][col] = acc[mt][e];
          }
        }
        __syncthreads();
        const int b = mt * 32 + g_row;
        if (b < B) {
          float gr = bh_r, gz = bh_z, gn = bh_n;
          if (step > 0) {
            gr += s_part[0][g_row][g_jj] + s_part[1][g_row][g_jj] + s_part[2][g_row][g_jj] + s_part[3][g_row][g_jj];
            gz += s_part[0][g_row][GRU_HU + g_jj] + s_part[1][g_row][GRU_HU + g_jj] + s_part[2][g_row][GRU_HU + g_jj] +
                  s_part[3][g_row][GRU_HU + g_jj];
            gn += s_part[0][g_row][2 * GRU_HU + g_jj] + s_part[1][g_row][2 * GRU_HU + g_jj] +
                  s_part[2][g_row][2 * GRU_HU + g_jj] + s_part[3][g_row][2 * GRU_HU + g_jj];
          }
          const float r = sigmoid_f(pre_r[mt] + gr);
          const float z = sigmoid_f(pre_z[mt] + gz);
          const float n = tanhf(pre_n[mt] + r * gn);
          p.out[(int64_t)b * ld_out + off_t + j0 + g_jj] = (1.f - z) * n + z * pre_h[mt];
        }
        __syncthreads();
      }
    }
    // every storing wave drains its own stores before the barrier inside grid.sync(): the sync's leader lane
    // issues the agent-scope release (buffer_wbl2), which only covers stores that already reached L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    grid.sync();   // h_t of every slice visible to every workgroup before step t+1 reads it
  }
}

}  // namespace

extern "C" int gvd_gru_bidir_layer(const float* gi, const float* w_hh_fw, const float* b_hh_fw, const float* w_hh_bw,
                                   const float* b_hh_bw, float* out, int B, int T, int Hh, gvd_stream_t stream) {
  if (!gi || !w_hh_fw || !b_hh_fw || !w_hh_bw || !b_hh_bw || !out || B <= 0 || T <= 0 || Hh != GRU_HH) return GVD_EINVAL;
  if (!gvd_aligned16(gi) || !gvd_aligned16(w_hh_fw) || !gvd_aligned16(w_hh_bw) || !gvd_aligned16(out)) return GVD_EINVAL;
  hipStream_t st = gvd_s(stream);
  // batches beyond MAX_TILES*32 rows run as consecutive launches over batch slices (samples are independent)
  for (int b0 = 0; b0 < B; b0 += MAX_TILES * 32) {
    GruParams p;
    const int nb = (B - b0 < MAX_TILES * 32) ? (B - b0) : MAX_TILES * 32;
    p.gi = gi + (int64_t)b0 * T * 6 * GRU_HH;
    p.w_hh[0] = w_hh_fw; p.w_hh[1] = w_hh_bw; p.b_hh[0] = b_hh_fw; p.b_hh[1] = b_hh_bw;
    p.out = out + (int64_t)b0 * T * 2 * GRU_HH;
    p.B = nb; p.T = T;
    void* args[] = {&p};
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(gru_layer_kernel), dim3(2 * GRU_NW), dim3(256),
                                              args, 0, st);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}
