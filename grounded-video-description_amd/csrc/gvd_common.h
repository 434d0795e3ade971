// Shared device/host helpers for libgvd_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/gvd_hip.h"

#define GVD_WAVE 64

#define GVD_CHECK_LAUNCH()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

static inline hipStream_t gvd_s(gvd_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool gvd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, GVD_WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, GVD_WAVE));
  return v;
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------------------------
// tanh of the additive-attention scores  e[n] = sum_k w_k tanh(p_feats[n,k] + q_k)   (AttModel.py:39-45, 84-90).
// ocml's tanhf is 29 VALU instructions (two of them quarter-rate); with 512 evaluations per streamed 6 KB row - and G
// times that in the beam-grouped kernel - the score pass was VALU-bound, not HBM-bound.  Here
//     tanh(s) = 1 - 2 r,   r = 1 / (1 + 2^(2 log2(e) s))
// on the hardware v_exp_f32 / v_rcp_f32 (1 ulp each): |error| <= 2.5e-7 absolute over the whole real line (measured
// on the device by tests/test_gpu_kernels.py::test_tanh_fast_error_bound; 1.9e-7 in an fp32 emulation), saturating
// exactly to +-1 (2^y -> inf / 0), NaN-propagating.  The score kernels fold the constants into per-lane registers:
//     sum_k w_k tanh(x_k + q_k) = sum_k w_k + sum_k (-2 w_k) r_k,   r_k = rcp(1 + exp2(fma(x_k, C, C q_k)))
// = fma + v_exp + add + v_rcp + fma per element (11 issue slots instead of 31).  Every forward score kernel
// (attn_partial_kernel, attn_partial_group_kernel, the persistent decoder) uses attn_score_lane in the same k order,
// so they stay bitwise interchangeable.
// ---------------------------------------------------------------------------------------------------------------
constexpr float GVD_TWO_LOG2E = 2.8853900817779268f;
// (Round 4: a Newton step on the reciprocal - r + r (1 - d r), rcp error 1 ulp -> 1/2 ulp - was built and measured: the noise
// of the attention logits against the CPU oracle stayed at 1.1-1.2e-6 (it is the fp32 summation order of the 512-term dot
// product, not the tanh), the greedy kernel took +0.7 %, the VALU-bound beam kernel +14 %; removed.  DESIGN.md section 5.)
__device__ __forceinline__ float gvd_rcp_1p(float e) { return __builtin_amdgcn_rcpf(1.0f + e); }      // 1 / (1 + e)
__device__ __forceinline__ float tanh_fast(float s) {
  const float e = __builtin_amdgcn_exp2f(s * GVD_TWO_LOG2E);
  return fmaf(-2.0f, gvd_rcp_1p(e), 1.0f);
}
// acc + (-2 w) / (1 + exp2(x C + qs)),  qs = C q,  wn = -2 w
__device__ __forceinline__ float attn_score_fma(float x, float qs, float wn, float acc) {
  const float e = __builtin_amdgcn_exp2f(fmaf(x, GVD_TWO_LOG2E, qs));
  return fmaf(wn, gvd_rcp_1p(e), acc);
}
typedef float gvd_score_f32x4 __attribute__((ext_vector_type(4)));
// per-lane constants of a score pass: the lane owns columns [4 lane, +4) and [256 + 4 lane, +4) of A = 512
struct AttnLaneW {
  gvd_score_f32x4 wn0, wn1;   // -2 w
  float wsum;                 // sum of the lane's 8 w values (the "1" of every 1 - 2 r)
};
__device__ __forceinline__ AttnLaneW attn_lane_w(const float* w, int lane) {
  const gvd_score_f32x4 w0 = *reinterpret_cast<const gvd_score_f32x4*>(w + 4 * lane);
  const gvd_score_f32x4 w1 = *reinterpret_cast<const gvd_score_f32x4*>(w + 256 + 4 * lane);
  AttnLaneW o;
  o.wn0 = -2.0f * w0; o.wn1 = -2.0f * w1;
  o.wsum = ((w0[0] + w1[0]) + (w0[1] + w1[1])) + ((w0[2] + w1[2]) + (w0[3] + w1[3]));
  return o;
}
// one row's lane-partial score (before the wave reduction): x0 / x1 = the lane's two 16-byte slices of the projection row
__device__ __forceinline__ float attn_score_lane(gvd_score_f32x4 x0, gvd_score_f32x4 x1, gvd_score_f32x4 qs0,
                                                 gvd_score_f32x4 qs1, const AttnLaneW& W) {
  float s = W.wsum;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s = attn_score_fma(x0[k], qs0[k], W.wn0[k], s);
    s = attn_score_fma(x1[k], qs1[k], W.wn1[k], s);
  }
  return s;
}

// The other two score functions of the region attention (opts.py:63 `--region_attn_mode`; gvd_attn_side.score_mode):
//   GVD_SCORE_MUL ('mix_mul', AttModel.py:82-83)   sum_k w_k tanh(x_k q_k): the same five issue slots per element with the
//                 exponent x_k (C q_k) in place of x_k C + C q_k
//   GVD_SCORE_DOT ('dp', AttModel.py:92-95)        sum_k x_k q_k: one fma per element, q unscaled, no alpha_net
// MODE = GVD_SCORE_ADD is attn_score_lane above, instruction for instruction.  qs0 / qs1: C q for ADD / MUL, q for DOT.
template <int MODE>
__device__ __forceinline__ float attn_score_lane_m(gvd_score_f32x4 x0, gvd_score_f32x4 x1, gvd_score_f32x4 qs0,
                                                   gvd_score_f32x4 qs1, const AttnLaneW& W) {
  if constexpr (MODE == GVD_SCORE_ADD) return attn_score_lane(x0, x1, qs0, qs1, W);
  float s = MODE == GVD_SCORE_MUL ? W.wsum : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if constexpr (MODE == GVD_SCORE_MUL) {
      s = fmaf(W.wn0[k], gvd_rcp_1p(__builtin_amdgcn_exp2f(x0[k] * qs0[k])), s);
      s = fmaf(W.wn1[k], gvd_rcp_1p(__builtin_amdgcn_exp2f(x1[k] * qs1[k])), s);
    } else {
      s = fmaf(x0[k], qs0[k], s);
      s = fmaf(x1[k], qs1[k], s);
    }
  }
  return s;
}

// XCD-aware bijective remap of a linear workgroup id (cdna_hip_programming.md T1): hardware round-robins
// consecutive ids over the 8 XCDs; after the remap each XCD walks a contiguous chunk of logical ids so
// neighbouring tiles (which share an operand panel) hit the same private L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  unsigned q = nwg / nx, r = nwg % nx;
  unsigned xcd = bid % nx, idx = bid / nx;
  unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent-kernel toolkit (gru.hip, decode_persistent.hip): data exchanged between workgroups inside ONE launch
// goes through agent-scope coherent accesses (the `sc1` cache policy: what a relaxed agent-scope atomic compiles
// to on gfx950, here on 16-byte buffer loads/stores), so the grid barrier needs no cache write-back/invalidate
// fence at all.  Measured on MI355X, 256 workgroups (tools/barrier_micro.hip): release/acquire-fence barrier on one
// counter 10.6 us; same counter without fences 4.9 us (256 serialized atomics + 256 pollers on one line);
// two-level tree below 2.1 us.
// ---------------------------------------------------------------------------------------------------------------
typedef float gvd_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned gvd_u32x4 __attribute__((ext_vector_type(4)));

constexpr int GVD_SYNC_GROUPS = 16;
constexpr int GVD_SYNC_WORDS = 64 + 2 * 32 * GVD_SYNC_GROUPS;   // uint32 words of one barrier object (zeroed by the host)
constexpr int GVD_SYNC_ERR = 32;                                // word raised when a bounded spin ran out
constexpr int GVD_SYNC_LIMIT = 33;                              // optional spin-limit override (0 = GVD_SPIN_LIMIT); test aid:
                                                                // the host writes it when the environment sets GVD_SPIN_LIMIT
constexpr unsigned GVD_SPIN_LIMIT = 4000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gvd_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ gvd_f32x4 ld_agent_x4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(gvd_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ float ld_agent_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16));
}
typedef float gvd_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gvd_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gvd_f32x2 ld_agent_x2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(gvd_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 16));
}
__device__ __forceinline__ void st_agent_x2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, gvd_f32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(gvd_u32x2, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ void st_agent_x4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, gvd_f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(gvd_u32x4, v), r, byte_off, 0, 16);
}
__device__ __forceinline__ void st_agent_f32(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, 16);
}

// Grid-wide barrier number `round` (0,1,2,... per launch) of a cooperative launch with nwg workgroups (nwg a
// multiple of 16).  Every wave drains its own stores, then one lane arrives on its group's counter (group = id & 15),
// the last of a group arrives on the top counter, the last group stores the 16 per-group release words every
// workgroup polls (16 pollers per 128-byte line).  Counters are monotonic over rounds.  Only sc1-coherent data is
// ordered by this barrier.  Spins are bounded: on timeout the error word is raised and the kernel runs on (wrong
// values, reported by the host) instead of hanging the GPU.
// `dead` (a per-thread flag, used by thread 0 only) latches a timeout: the workgroup then stops waiting at later
// barriers, so a broken launch costs one spin limit, not one per barrier.
__device__ __forceinline__ void grid_barrier_tree(unsigned* sync, unsigned round, unsigned nwg, bool& dead) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && !dead) {
    const unsigned g = blockIdx.x & (GVD_SYNC_GROUPS - 1);
    unsigned* rel = sync + 64 + 32 * GVD_SYNC_GROUPS + 32 * g;
    const unsigned per = nwg / GVD_SYNC_GROUPS;
    if (__hip_atomic_fetch_add(sync + 64 + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
        (round + 1) * per - 1) {
      if (__hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
          (round + 1) * GVD_SYNC_GROUPS - 1) {
#pragma unroll
        for (int k = 0; k < GVD_SYNC_GROUPS; ++k)
          __hip_atomic_store(sync + 64 + 32 * GVD_SYNC_GROUPS + 32 * k, round + 1, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    unsigned spins = 0, limit = GVD_SPIN_LIMIT;
    while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
      if (spins == 0) {          // (slow path only: a workgroup that has to wait reads the override once per barrier)
        const unsigned o = __hip_atomic_load(sync + GVD_SYNC_LIMIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o) limit = o;
      }
      __builtin_amdgcn_s_sleep(1);
      if (++spins > limit) {
        __hip_atomic_store(sync + GVD_SYNC_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        break;
      }
    }
  }
  __syncthreads();
}

// Barrier number `round` among the `per` workgroups of ONE of up to 16 independent sub-grids of a launch (sub-grid `sg`;
// per <= 32), on the same barrier object: one arrival counter and one release word per sub-grid (the group slots of the
// tree above), one level.  For recurrences that only couple a known subset of the workgroups (the bi-GRU: the workgroups of
// one direction and one batch-tile group) - they neither pay the second level nor wait for the slowest workgroup of the
// whole grid.  Same timeout handling as the tree.
__device__ __forceinline__ void subgrid_barrier(unsigned* sync, unsigned round, unsigned sg, unsigned per, bool& dead) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && !dead) {
    unsigned* rel = sync + 64 + 32 * GVD_SYNC_GROUPS + 32 * sg;
    if (__hip_atomic_fetch_add(sync + 64 + 32 * sg, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (round + 1) * per - 1)
      __hip_atomic_store(rel, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0, limit = GVD_SPIN_LIMIT;
    while (__hip_atomic_load(rel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
      if (spins == 0) {
        const unsigned o = __hip_atomic_load(sync + GVD_SYNC_LIMIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (o) limit = o;
      }
      __builtin_amdgcn_s_sleep(1);
      if (++spins > limit) {
        __hip_atomic_store(sync + GVD_SYNC_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        dead = true;
        break;
      }
    }
  }
  __syncthreads();
}

// Host: can `grid` workgroups of `block` threads of kernel `fn` be resident at the same time on the current device?  (The check
// a cooperative launch makes; the persistent kernels are launched PLAINLY after it - the cooperative path costs a ~12 us
// dispatch gap before and after every such kernel, 0.1 ms of a 3 ms batch_size = 4 call - and bound their barrier spins.)
static inline bool gvd_grid_fits(const void* fn, int block, int grid) {
  // (per translation unit: a few entries, answers never change for a (kernel, block) pair on one device type; concurrent
  // first calls - two host threads driving two streams - are serialised by the mutex)
  struct Entry { const void* fn; int block; long slots; };
  static Entry cache[8] = {};
  static int used = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < used; ++i)
    if (cache[i].fn == fn && cache[i].block == block) return cache[i].slots >= grid;
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, block, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
  const long slots = (long)per * cus;
  if (used < 8) { cache[used].fn = fn; cache[used].block = block; cache[used].slots = slots; ++used; }
  return slots >= grid;
}

// GVD_SPIN_LIMIT (environment, test aid): spin-limit override the host writes into word GVD_SYNC_LIMIT of a barrier object
// (0 = unset).  tests force 1 to exercise the timeout -> status word -> fallback path without a shared GPU.
static inline unsigned gvd_spin_limit_env() {
  static const unsigned v = [] {
    const char* e = getenv("GVD_SPIN_LIMIT");
    const long x = e ? atol(e) : 0;
    return x > 0 ? (unsigned)x : 0u;
  }();
  return v;
}

// event-pair recorder (prof.hip); no-ops when p == nullptr
void gvd_prof_begin(gvd_prof* p, hipStream_t st);
void gvd_prof_end(gvd_prof* p, hipStream_t st);
constexpr int GVD_PROF_TAG_WORDS = 8;
int gvd_prof_next(gvd_prof* p);
void gvd_prof_tag(gvd_prof* p, const int64_t (&words)[GVD_PROF_TAG_WORDS]);
