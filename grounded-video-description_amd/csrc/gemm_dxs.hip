// Small-M "dX" products of the token loop's BPTT on the fp32 matrix cores:  out[M, n] = A[M, Kred] . W[Kred, n]  for
// M = one decode batch (32 .. 64 rows; also the [Lc B, .] post-loop products) against a weight matrix consumed IN PLACE
// with the contraction index as its slow axis (W = weight_ih / weight_hh / the stacked h2att weights, [4H or 2A rows, *]):
//     d[att+att2 | h_att] = dgates_lang . W_ih(lang)        d h_lang(t-1) = dgates_lang . W_hh(lang)
//     d h_att(t-1)        = dgates_att . W_hh(att)          d h_att      += dq12 . [W_h2att(temporal); W_h2att(region)]
// (autograd of nn.LSTMCell / nn.Linear: AttModel.py:139,160,39,77).  Up to GVD_DX_MAX_GROUPS products with different A / W /
// output - e.g. the three gate products of one BPTT step - run as ONE launch.
//
// Shape of the problem: 1.6 GFLOP against 50 MB of weights that stream from L2 / Infinity Cache, 64 output rows: neither
// a tile grid (24 .. 32 column blocks) nor a K loop alone fills 256 CUs.  Here
//   * a wave owns 64 rows x 128 columns (8 accumulator tiles) and a K slice; the weight operand goes global -> registers
//     with one 16-byte load per lane and k pair (4 interleaved column tiles: 512 contiguous bytes per half wave, no LDS),
//     the A operand through a wave-PRIVATE transposed LDS tile (no workgroup barrier in the K loop);
//   * the 4 waves of a workgroup split the workgroup's K slice and reduce through LDS; `split` workgroups split K further
//     and leave their partial tile in a workspace; the LAST one to arrive (one relaxed agent-scope counter per tile; partials
//     written / read with sc1 accesses: gvd_common.h) adds the `split` partials in slice order - deterministic, no second
//     launch - adds the optional addend and writes the output.
#include "gvd_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int DX_KC = 32;                 // k rows per chunk
constexpr int DX_PITCH = 65;              // floats per k row of the transposed A tile (bank = (4 (l & 7) + (l >> 3)) mod 32)

struct DxGroupDev {
  const float* A; int64_t lda;
  const float* W; int64_t ldw;
  float* out; int64_t ldo;
  const float* addend; int64_t ld_add;
  int Kred, cb0;                          // contraction length; first 128-column block of this group in the launch
};

struct DxParams {
  DxGroupDev g[GVD_DX_MAX_GROUPS];
  int ngroups, M, split, ncb;             // ncb: 128-column blocks of all groups
  float* part;                            // [split][Mpad][ncb * 128]
  unsigned* counters;                     // [m blocks][ncb], zero between launches
  int64_t part_slab;                      // Mpad * ncb * 128
};

template <int MT>
__global__ __launch_bounds__(256, 1) void gemm_dxs_kernel(const DxParams p) {
  // A tiles 4 waves x 2 buffers x 32 k x 65 floats (66,560 B), reused by the reduction [wave][MT x 16][lane][4] (65,536 B x MT)
  __shared__ __attribute__((aligned(16))) float smem[MT == 1 ? 4 * 2 * DX_KC * DX_PITCH : 4 * 2 * 16 * 64 * 4];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, half = lane >> 5;
  const int cb = blockIdx.x % p.ncb, sl = blockIdx.x / p.ncb;
  const int m0 = blockIdx.y * (32 * MT);
  int gi = 0;
#pragma unroll
  for (int i = 1; i < GVD_DX_MAX_GROUPS; ++i)
    if (i < p.ngroups && cb >= p.g[i].cb0) gi = i;
  const DxGroupDev& G = p.g[gi];
  const int n0 = (cb - G.cb0) * 128;
  const int kl = G.Kred / (p.split * 4);                  // this wave's K slice
  const int kbeg = (sl * 4 + wave) * kl;
  const int nch = kl / DX_KC;

  float* As = smem + wave * (2 * DX_KC * DX_PITCH);
  // staging role: 8 lanes per A row (32 k = 128 B), 8 rows per instruction
  const int srow = lane >> 3, skc = 4 * (lane & 7);
  const float* Ap[4 * MT];
#pragma unroll
  for (int it = 0; it < 4 * MT; ++it) Ap[it] = G.A + (int64_t)min(m0 + 8 * it + srow, p.M - 1) * G.lda + skc;
  const float* Wp = G.W + n0 + 4 * c + (int64_t)half * G.ldw;

  f32x4 ga[4 * MT];
  f32x4 w[2][16];
  auto fetch = [&](const int set, int ch) {
    const int k0 = kbeg + ch * DX_KC;
#pragma unroll
    for (int it = 0; it < 4 * MT; ++it) ga[it] = *reinterpret_cast<const f32x4*>(Ap[it] + k0);
#pragma unroll
    for (int j = 0; j < 16; ++j) w[set][j] = *reinterpret_cast<const f32x4*>(Wp + (int64_t)(k0 + 2 * j) * G.ldw);
  };
  auto stage = [&](const int buf) {
    float* T = As + buf * (DX_KC * DX_PITCH);
#pragma unroll
    for (int it = 0; it < 4 * MT; ++it)
#pragma unroll
      for (int j = 0; j < 4; ++j) T[(skc + j) * DX_PITCH + 8 * it + srow] = ga[it][j];
  };
  f32x16 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto compute = [&](const int set) {
    const float* T = As + set * (DX_KC * DX_PITCH) + half * DX_PITCH + c;
    // all 16 x MT fragment reads of the chunk are issued first (one wave per SIMD: nothing else hides a read's latency in
    // front of every k pair), the MFMAs then wait on them one k pair at a time
    float a[16][MT];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) a[j][i] = T[2 * j * DX_PITCH + 32 * i];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][i], w[set][j][jn], acc[i][jn], 0, 0, 0);
  };
  // chunk ch lives in register set / LDS buffer ch & 1; its successor is fetched before its MFMAs and staged after them
  fetch(0, 0);
  stage(0);
  auto chunk = [&](const int cur, int ch) {
    const bool more = ch + 1 < nch;
    if (more) fetch(cur ^ 1, ch + 1);
    compute(cur);
    if (more) stage(cur ^ 1);
  };
  int ch = 0;
#pragma unroll 1
  for (; ch + 1 < nch; ch += 2) {
    chunk(0, ch);
    chunk(1, ch + 1);
  }
  if (ch < nch) chunk(0, ch);

  // ---- reduction over the 4 waves (K slices) of the workgroup
  __syncthreads();                                        // every wave is done with its private A tiles
  float* red = smem;                                      // [wave][MT * 16][lane][4]
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const f32x4 v = {acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]};
      *reinterpret_cast<f32x4*>(&red[((wave * (MT * 16) + i * 16 + e) * 64 + lane) * 4]) = v;
    }
  __syncthreads();
  constexpr int PER = MT * 4;                             // (m tile, register) pairs per wave
  f32x4 sum[PER];
  int rows[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int idx = wave * PER + k;                       // = i * 16 + e
    f32x4 v = *reinterpret_cast<const f32x4*>(&red[((0 * (MT * 16) + idx) * 64 + lane) * 4]);
#pragma unroll
    for (int wv = 1; wv < 4; ++wv) v += *reinterpret_cast<const f32x4*>(&red[((wv * (MT * 16) + idx) * 64 + lane) * 4]);
    sum[k] = v;
    const int e = idx & 15;
    rows[k] = m0 + 32 * (idx >> 4) + (e & 3) + 8 * (e >> 2) + 4 * half;
  }
  const int64_t colg = (int64_t)cb * 128 + 4 * c;         // column in the launch-wide partial layout
  const int64_t ldp = (int64_t)p.ncb * 128;
  if (p.split > 1) {
    const __amdgpu_buffer_rsrc_t rp = gvd_rsrc(p.part + (int64_t)sl * p.part_slab);
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (rows[k] < p.M) st_agent_x4(rp, (unsigned)(((int64_t)rows[k] * ldp + colg) * 4), sum[k]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      unsigned* cnt = p.counters + (int64_t)blockIdx.y * p.ncb + cb;
      const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (unsigned)p.split - 1u;
      if (s_last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    // slice order: the same sum whichever workgroup came last.  The sc1 loads bypass the caches (0.5 - 1 us each): all
    // PER x 4 loads of a group of four slices are issued before the first add, not one dependent load per add
    unsigned off[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      off[k] = (unsigned)(((int64_t)min(rows[k], p.M - 1) * ldp + colg) * 4);
      sum[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int s2 = 0; s2 < p.split; s2 += 4) {
      f32x4 t[PER][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const __amdgpu_buffer_rsrc_t rq = gvd_rsrc(p.part + (int64_t)min(s2 + u, p.split - 1) * p.part_slab);
#pragma unroll
        for (int k = 0; k < PER; ++k) t[k][u] = ld_agent_x4(rq, off[k]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s2 + u < p.split) {
#pragma unroll
          for (int k = 0; k < PER; ++k) sum[k] += t[k][u];
        }
    }
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (rows[k] >= p.M) continue;
    f32x4 v = sum[k];
    if (G.addend) v += *reinterpret_cast<const f32x4*>(G.addend + (int64_t)rows[k] * G.ld_add + n0 + 4 * c);
    *reinterpret_cast<f32x4*>(G.out + (int64_t)rows[k] * G.ldo + n0 + 4 * c) = v;
  }
}

int pick_split(const gvd_dx_group* g, int ngroups, int M, int ncb) {
  const int mblk = (M + 63) / 64;
  int split = 1;
  for (;;) {
    const int next = split * 2;
    bool ok = (long)ncb * mblk * next <= 320;             // about one workgroup per CU
    for (int i = 0; i < ngroups && ok; ++i) ok = (g[i].Kred % (next * 4 * DX_KC)) == 0;
    if (!ok || next > 16) break;
    split = next;
  }
  return split;
}

}  // namespace

// The tile counters live in a FIXED region at the head of the workspace, whatever the launch's shape: a workspace is reused by
// launches of different shapes (the M = 64 products of every BPTT step, then the [Lc B, .] post-loop ones), and a counter region
// sized per launch would let one launch's counters fall onto bytes an earlier launch used for partial sums - non-zero counters,
// a wrong "last workgroup", a wrong sum.  (Found by the full GPU suite in round 5: the gradient of `embed` at B = 64.)
constexpr size_t DX_COUNTER_BYTES = 65536;

static bool dx_counters_fit(int M, int ncb) {
  return (size_t)ncb * (size_t)((M + 63) / 64) * sizeof(unsigned) <= DX_COUNTER_BYTES;
}

extern "C" size_t gvd_gemm_dx_small_workspace_bytes(int M, int total_cols) {
  if (M <= 0 || total_cols <= 0) return 0;
  const int ncb = (total_cols + 127) / 128;
  // one counter per output tile (zero before the FIRST launch; every launch leaves them zero), then up to 16 partial slabs
  return DX_COUNTER_BYTES + (size_t)16 * ((size_t)(M + 63) / 64 * 64) * (size_t)ncb * 128 * sizeof(float);
}

extern "C" int gvd_gemm_dx_small_f32(const gvd_dx_group* groups, int ngroups, int M, void* workspace, size_t workspace_bytes,
                                     gvd_stream_t stream) {
  if (!groups || ngroups < 1 || ngroups > GVD_DX_MAX_GROUPS || M <= 0 || !workspace || !gvd_aligned16(workspace))
    return GVD_EINVAL;
  DxParams p = {};
  int ncb = 0;
  for (int i = 0; i < ngroups; ++i) {
    const gvd_dx_group& g = groups[i];
    if (!g.A || !g.W || !g.out || g.Kred <= 0 || (g.Kred % (4 * DX_KC)) || g.ncols <= 0 || (g.ncols % 128)) return GVD_EINVAL;
    if (!gvd_aligned16(g.A) || !gvd_aligned16(g.W) || !gvd_aligned16(g.out) || (g.lda % 4) || (g.ldw % 4) || (g.ldo % 4))
      return GVD_EINVAL;
    if (g.addend && (!gvd_aligned16(g.addend) || (g.ld_add % 4))) return GVD_EINVAL;
    p.g[i] = {g.A, g.lda, g.W, g.ldw, g.out, g.ldo, g.addend, g.ld_add, g.Kred, ncb};
    ncb += g.ncols / 128;
  }
  p.ngroups = ngroups; p.M = M; p.ncb = ncb;
  p.split = pick_split(groups, ngroups, M, ncb);
  const int mblk64 = (M + 63) / 64;
  if (!dx_counters_fit(M, ncb)) return GVD_EINVAL;
  const size_t cbytes = DX_COUNTER_BYTES;
  p.part_slab = (int64_t)mblk64 * 64 * ncb * 128;
  if (cbytes + (p.split > 1 ? (size_t)p.split * p.part_slab * sizeof(float) : 0) > workspace_bytes) return GVD_EINVAL;
  if ((int64_t)p.part_slab * 4 >= (1ll << 31)) return GVD_EINVAL;              // 32-bit buffer offsets inside one slab
  p.counters = reinterpret_cast<unsigned*>(workspace);
  p.part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + cbytes);
  hipStream_t st = gvd_s(stream);
  if (M <= 32) hipLaunchKernelGGL(gemm_dxs_kernel<1>, dim3((unsigned)(ncb * p.split), 1u), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(gemm_dxs_kernel<2>, dim3((unsigned)(ncb * p.split), (unsigned)mblk64), dim3(256), 0, st, p);
  GVD_CHECK_LAUNCH();
  return 0;
}
