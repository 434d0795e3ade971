// Event-pair recorder: times selected kernels on the stream they are launched on (bench.py `roofline`).
#include "gvd_common.h"
#include <vector>

struct gvd_prof {
  std::vector<hipEvent_t> start, stop;
  int used = 0;
  bool open = false;
};

extern "C" gvd_prof* gvd_prof_create(int max_pairs) {
  if (max_pairs <= 0) return nullptr;
  gvd_prof* p = new gvd_prof();
  p->start.resize(max_pairs);
  p->stop.resize(max_pairs);
  for (int i = 0; i < max_pairs; ++i) {
    if (hipEventCreate(&p->start[i]) != hipSuccess || hipEventCreate(&p->stop[i]) != hipSuccess) {
      delete p;
      return nullptr;
    }
  }
  return p;
}

extern "C" void gvd_prof_destroy(gvd_prof* p) {
  if (!p) return;
  for (size_t i = 0; i < p->start.size(); ++i) { (void)hipEventDestroy(p->start[i]); (void)hipEventDestroy(p->stop[i]); }
  delete p;
}

extern "C" void gvd_prof_reset(gvd_prof* p) { if (p) { p->used = 0; p->open = false; } }

void gvd_prof_begin(gvd_prof* p, hipStream_t st) {
  if (!p || p->used >= (int)p->start.size()) return;
  (void)hipEventRecord(p->start[p->used], st);
  p->open = true;
}

void gvd_prof_end(gvd_prof* p, hipStream_t st) {
  if (!p || !p->open) return;
  (void)hipEventRecord(p->stop[p->used], st);
  p->used++;
  p->open = false;
}

extern "C" int gvd_prof_read(gvd_prof* p, float* total_ms, int* count) {
  if (!p || !total_ms || !count) return GVD_EINVAL;
  float tot = 0.f;
  for (int i = 0; i < p->used; ++i) {
    float ms = 0.f;
    hipError_t e = hipEventElapsedTime(&ms, p->start[i], p->stop[i]);
    if (e != hipSuccess) return (int)e;
    tot += ms;
  }
  *total_ms = tot;
  *count = p->used;
  return 0;
}
