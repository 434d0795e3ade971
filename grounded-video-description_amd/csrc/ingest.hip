// Feature ingest, device side.  The reference's dataloader builds every padded input on the
// CPU: np.zeros((1000, 2048)) + row copy + masked_fill_ per segment (dataloader_anet.py:317-344) = three passes over
// 8 MB per sample on host cores.  Here the host only copies the VALID raw rows of the feature files into pinned
// staging and ships them (ingest.py); this kernel establishes the padding/masking contract on the GPU, in place:
// every row whose mask byte is set (proposal below prop_thresh / beyond num_pps, frame beyond num_frm) is zero-filled.
// One 64-lane wave per row, 16-byte stores when the row allows it; rows that are kept are not touched at all.
#include "gvd_common.h"
#include <errno.h>
#include <sys/uio.h>
#include <unistd.h>

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

// Host side of the ingest (no GPU work): read `rows` rows of `row_bytes` bytes that lie back to back in a file (a .npy
// payload) into destination rows that are `dst_stride` bytes apart - the frame-feature files are column blocks of the wider
// segs_feat rows (dataloader_anet.py:198-206).  One pread when the destination is contiguous, otherwise scatter reads
// (preadv, up to 1024 rows per call) straight from the page cache into the pinned staging rows: no contiguous scratch copy,
// no Python object per row, and - called through ctypes - no GIL while it runs, so the reader threads scale.
extern "C" int64_t gvd_pread_rows(int fd, int64_t file_off, void* dst, int64_t rows, int64_t row_bytes, int64_t dst_stride) {
  if (fd < 0 || file_off < 0 || !dst || rows < 0 || row_bytes <= 0 || dst_stride < row_bytes) return -EINVAL;
  char* d = static_cast<char*>(dst);
  int64_t done = 0;
  if (dst_stride == row_bytes) {
    const int64_t want = rows * row_bytes;
    while (done < want) {
      const ssize_t n = pread(fd, d + done, (size_t)(want - done), (off_t)(file_off + done));
      if (n < 0) { if (errno == EINTR) continue; return -errno; }
      if (n == 0) break;
      done += n;
    }
    return done;
  }
  constexpr int IOV = 1024;
  struct iovec iov[IOV];
  int64_t r = 0;
  while (r < rows) {
    const int n = (int)(rows - r < IOV ? rows - r : IOV);
    for (int i = 0; i < n; ++i) { iov[i].iov_base = d + (r + i) * dst_stride; iov[i].iov_len = (size_t)row_bytes; }
    const int64_t want = (int64_t)n * row_bytes;
    const ssize_t got = preadv(fd, iov, n, (off_t)(file_off + done));
    if (got < 0) { if (errno == EINTR) continue; return -errno; }
    done += got;
    if (got != want) break;          // short read (end of file): the caller compares with rows * row_bytes
    r += n;
  }
  return done;
}
