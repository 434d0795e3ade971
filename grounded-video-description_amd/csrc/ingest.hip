// Feature ingest, device side.  The reference's dataloader builds every padded input on the
// CPU: np.zeros((1000, 2048)) + row copy + masked_fill_ per segment (dataloader_anet.py:317-344) = three passes over
// 8 MB per sample on host cores.  Here the host only copies the VALID raw rows of the feature files into pinned
// staging and ships them (ingest.py); this kernel establishes the padding/masking contract on the GPU, in place:
// every row whose mask byte is set (proposal below prop_thresh / beyond num_pps, frame beyond num_frm) is zero-filled.
// One 64-lane wave per row, 16-byte stores when the row allows it; rows that are kept are not touched at all.
#include "gvd_common.h"
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <pthread.h>
#include <thread>
#include <vector>

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

// Open a float32 C-ordered .npy file, parse its header (format 1.0 / 2.0 / 3.0: magic, version, header length, the Python
// dict literal with 'descr', 'fortran_order', 'shape'), and read its first min(rows, max_rows) rows - rows = product of all
// but the last dimension, which must equal D - into destination rows `dst_stride` bytes apart (gvd_pread_rows).  The whole
// per-file work of the ingest as ONE native call: no Python header parsing (ast.literal_eval), no file object, no GIL.
// Returns the rows in the file (>= 0; *rows_read = rows copied) or a negative code: -errno, or -1000 - k for a malformed /
// unsupported header (k: 1 magic, 2 header length, 3 dtype, 4 order, 5 shape, 6 last dimension, 7 short read).
// How the payload travels from the page cache into the (pinned, GPU-mapped) destination rows:
//   GVD_READ_MAPPED  the file is mapped (no MAP_POPULATE: fault-around maps 16 resident pages per fault under the per-VMA
//                    lock) and the rows are copied in user space, then the mapping is dropped.  Default of the ingest.
//   GVD_READ_PREAD   pread() straight into the destination rows.
// Both were timed under the running GPU pipeline (files -> captions, profiles/r05/files_read_path.txt): the read() system
// call is the part that slows down 2.5-3.5x while the decode is enqueued next to it, and every few batches all readers
// stall in it for ~50 ms; the user-space copy out of a mapping stays at its idle speed.  A file that is truncated by
// another process WHILE it is mapped raises SIGBUS instead of a short read - callers that read feature files being
// rewritten select GVD_READ_PREAD (ingest.py: GVD_INGEST_READ=pread).


static void copy_rows(char* d, const char* src, int64_t rows, int64_t row_bytes, int64_t dst_stride) {
  if (dst_stride == row_bytes) { memcpy(d, src, (size_t)(rows * row_bytes)); return; }
  for (int64_t r = 0; r < rows; ++r) memcpy(d + r * dst_stride, src + r * row_bytes, (size_t)row_bytes);
}

static int64_t npy_read_rows(const char* path, void* dst, int64_t max_rows, int64_t D, int64_t dst_stride, int64_t* rows_read,
                             int mode) {
  if (!path || !dst || max_rows < 0 || D <= 0 || dst_stride < D * 4 || !rows_read) return -EINVAL;
  *rows_read = 0;
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) return -errno;
  char buf[4096];
  const ssize_t got = pread(fd, buf, sizeof(buf) - 1, 0);
  int64_t rc = 0, hlen = 0, hoff = 0;
  if (got < 10 || memcmp(buf, "\x93NUMPY", 6) != 0) rc = -1001;
  if (!rc) {
    const unsigned char* u = reinterpret_cast<const unsigned char*>(buf);
    if (u[6] == 1) { hlen = u[8] | (u[9] << 8); hoff = 10; }
    else if (u[6] == 2 || u[6] == 3) { hlen = (int64_t)u[8] | ((int64_t)u[9] << 8) | ((int64_t)u[10] << 16) | ((int64_t)u[11] << 24); hoff = 12; }
    else rc = -1001;
    if (!rc && (hlen <= 0 || hoff + hlen > got)) rc = -1002;
  }
  int64_t rows_file = 1, last = -1;
  if (!rc) {
    buf[hoff + hlen] = 0;
    const char* h = buf + hoff;
    const char* d = strstr(h, "'descr'");
    if (!d || !(strstr(d, "'<f4'") || strstr(d, "'=f4'") || strstr(d, "'|f4'"))) rc = -1003;
    const char* f = strstr(h, "'fortran_order'");
    if (!rc && (!f || !strstr(f, "False") || (strstr(f, "True") && strstr(f, "True") < strstr(f, "False")))) rc = -1004;
    const char* sh = strstr(h, "'shape'");
    const char* q = sh ? strchr(sh, '(') : nullptr;
    if (!rc && !q) rc = -1005;
    if (!rc) {
      ++q;
      int nd = 0;
      int64_t prod = 1;
      while (*q && *q != ')') {
        while (*q == ' ' || *q == ',') ++q;
        if (*q == ')' || !*q) break;
        char* e = nullptr;
        const long long v = strtoll(q, &e, 10);
        if (e == q || v < 0) { rc = -1005; break; }
        if (last >= 0) prod *= last;
        last = v;
        ++nd;
        q = e;
      }
      if (!rc && nd == 0) rc = -1005;
      rows_file = prod;
      if (!rc && last != D) rc = -1006;
    }
  }
  if (!rc) {
    const int64_t rows = rows_file < max_rows ? rows_file : max_rows;
    bool copied = false;
    if (rows > 0 && mode == GVD_READ_MAPPED) {
      const int64_t off = hoff + hlen, len = off + rows * D * 4;
      struct stat st;
      if (fstat(fd, &st) != 0) rc = -errno;
      else if ((int64_t)st.st_size < len) rc = -1007;          // the header promises more rows than the file holds
      else {
        void* m = mmap(nullptr, (size_t)len, PROT_READ, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {            // (a file system that cannot map - ENODEV - is read with pread below)
          copy_rows(static_cast<char*>(dst), static_cast<const char*>(m) + off, rows, D * 4, dst_stride);
          munmap(m, (size_t)len);
          copied = true;
        }
      }
    }
    if (rows > 0 && !rc && !copied) {
      const int64_t n = gvd_pread_rows(fd, hoff + hlen, dst, rows, D * 4, dst_stride);
      if (n < 0) rc = n;
      else if (n != rows * D * 4) rc = -1007;
    }
    if (!rc) { *rows_read = rows; rc = rows_file; }
  }
  close(fd);
  return rc;
}

extern "C" int64_t gvd_npy_read_rows_f32(const char* path, void* dst, int64_t max_rows, int64_t D, int64_t dst_stride,
                                         int64_t* rows_read) {
  return npy_read_rows(path, dst, max_rows, D, dst_stride, rows_read, GVD_READ_PREAD);
}

// Persistent reader threads (host side only).  Spawning the readers per batch - clone + an 8 MB stack mapping + its teardown,
// 31 times per batch - put every batch's first reads behind the process's address-space lock while the GPU pipeline was
// running (files -> captions timeline, profiles/r05): the threads are created once, sleep on a condition variable between
// batches and inherit the CPU affinity of the thread that makes the FIRST call (ingest.py's pinned staging thread).
namespace {
class ReaderPool {
 public:
  template <typename F> void run(int helpers, F& work) {
    std::unique_lock<std::mutex> call(call_mu_);              // one batch at a time
    if (helpers < 0) helpers = 0;                             // (done_ counts up from 0: a negative target would never be met)
    if (helpers > (int)threads_.size()) grow(helpers);
    {
      std::lock_guard<std::mutex> g(mu_);
      fn_ = [&work]() { work(); };
      want_ = helpers; done_ = 0; ++gen_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> g(mu_);
    cv_done_.wait(g, [&] { return done_ == want_; });
    fn_ = nullptr;
  }
  ~ReaderPool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
 private:
  void grow(int n) {
    while ((int)threads_.size() < n) {
      const int id = (int)threads_.size();
      threads_.emplace_back([this, id]() {
        uint64_t seen = 0;
        for (;;) {
          std::function<void()> f;
          {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return stop_ || (gen_ != seen && id < want_); });
            if (stop_) return;
            seen = gen_;
            f = fn_;
          }
          f();
          { std::lock_guard<std::mutex> g(mu_); ++done_; }
          cv_done_.notify_one();
        }
      });
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, cv_done_;
  std::vector<std::thread> threads_;
  std::function<void()> fn_;
  int want_ = 0, done_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};
// One pool per process, never destroyed (no join at exit).  A fork()ed child inherits the object but none of its threads: the
// atfork handler drops the pointer, and the child's first batch builds a pool of its own (the parent's object is leaked there).
std::atomic<ReaderPool*> g_pool{nullptr};
std::once_flag g_atfork_once;
ReaderPool& reader_pool() {
  std::call_once(g_atfork_once, [] { pthread_atfork(nullptr, nullptr, [] { g_pool.store(nullptr, std::memory_order_relaxed); }); });
  ReaderPool* p = g_pool.load(std::memory_order_acquire);
  if (!p) {
    ReaderPool* fresh = new ReaderPool();
    if (g_pool.compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) p = fresh;
    else delete fresh;                  // another caller installed one first (that pool has no threads yet either way)
  }
  return *p;
}
}  // namespace

// All feature files of ONE batch in one native call: n (path, destination) jobs handed to `n_threads` native threads that
// pull job indices from an atomic counter (the files differ in size: 8 MB region features, 4 + 2 MB frame features).  The
// Python side makes ONE GIL-free call per batch instead of three per segment from a pool of Python threads: those threads
// re-acquired the GIL after every file while the main thread was busy enqueueing the previous batch's ~400 launches - the
// composed files -> captions rate sat at 0.66 of its slower stage (profiles/r04).  The threads inherit
// the calling thread's CPU affinity (ingest.py pins that thread to the staging buffers' NUMA node).
// rows_file[i] receives gvd_npy_read_rows_f32's result for job i (rows in the file, or a negative code), rows_read[i] the
// rows copied.  Returns the number of failed jobs.
extern "C" int gvd_npy_read_batch_f32(const char* const* paths, void* const* dsts, const int64_t* max_rows, const int64_t* D,
                                      const int64_t* dst_stride, int n, int n_threads, int mode, int64_t* rows_read,
                                      int64_t* rows_file, int64_t* job_ns) {
  if (!paths || !dsts || !max_rows || !D || !dst_stride || !rows_read || !rows_file || n < 0 || (mode != GVD_READ_PREAD && mode != GVD_READ_MAPPED)) return -EINVAL;
  if (n == 0) return 0;                     // empty batch: nothing to read (and no pooled thread to wait for)
  std::atomic<int> next(0), failed(0);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= n) break;
      const auto t0 = std::chrono::steady_clock::now();
      rows_file[i] = npy_read_rows(paths[i], dsts[i], max_rows[i], D[i], dst_stride[i], &rows_read[i], mode);
      if (job_ns) job_ns[i] = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
      if (rows_file[i] < 0) failed.fetch_add(1, std::memory_order_relaxed);
    }
  };
  int nt = n_threads < 1 ? 1 : n_threads;
  if (nt > n) nt = n;
  reader_pool().run(nt - 1, work);          // nt - 1 pooled threads + the caller
  return failed.load();
}
