// Feature ingest, device side.  The reference's dataloader builds every padded input on the
// CPU: np.zeros((1000, 2048)) + row copy + masked_fill_ per segment (dataloader_anet.py:317-344) = three passes over
// 8 MB per sample on host cores.  Here the host only copies the VALID raw rows of the feature files into pinned
// staging and ships them (ingest.py); this kernel establishes the padding/masking contract on the GPU, in place:
// every row whose mask byte is set (proposal below prop_thresh / beyond num_pps, frame beyond num_frm) is zero-filled.
// One 64-lane wave per row, 16-byte stores when the row allows it; rows that are kept are not touched at all.
#include "gvd_common.h"

namespace {

__global__ __launch_bounds__(256) void zero_masked_rows_kernel(float* __restrict__ x, int64_t rows, int D,
                                                               const uint8_t* __restrict__ mask,
                                                               int64_t rows_per_batch, int64_t mask_ld,
                                                               int64_t mask_off) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  if (!mask[(row / rows_per_batch) * mask_ld + mask_off + row % rows_per_batch]) return;
  float* xr = x + row * D;
  if ((D & 3) == 0 && (reinterpret_cast<uintptr_t>(xr) & 15u) == 0) {
    const gvd_f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < D / 4; i += 64) reinterpret_cast<gvd_f32x4*>(xr)[i] = z;
  } else {
    for (int i = lane; i < D; i += 64) xr[i] = 0.f;
  }
}

}  // namespace

extern "C" int gvd_zero_masked_rows(float* x, int64_t rows, int D, const uint8_t* mask, int64_t rows_per_batch,
                                    int64_t mask_ld, int64_t mask_off, gvd_stream_t stream) {
  if (!x || !mask || rows <= 0 || D <= 0 || rows_per_batch <= 0 || mask_ld < rows_per_batch + mask_off || mask_off < 0)
    return GVD_EINVAL;
  hipLaunchKernelGGL(zero_masked_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, gvd_s(stream), x, rows, D,
                     mask, rows_per_batch, mask_ld, mask_off);
  GVD_CHECK_LAUNCH();
  return 0;
}
