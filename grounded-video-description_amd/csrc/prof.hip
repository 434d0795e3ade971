// Event-pair recorder: times selected kernels on the stream they are launched on (bench.py `roofline`).
#include "gvd_common.h"
#include <vector>

struct gvd_prof {
  std::vector<hipEvent_t> start, stop;
  std::vector<int64_t> tag;            // GVD_PROF_TAG_WORDS host-side words per pair (what was launched: the caller's business)
  int used = 0;
  bool open = false;
};

extern "C" gvd_prof* gvd_prof_create(int max_pairs) {
  if (max_pairs <= 0) return nullptr;
  gvd_prof* p = new gvd_prof();
  p->start.resize(max_pairs);
  p->stop.resize(max_pairs);
  p->tag.assign((size_t)max_pairs * GVD_PROF_TAG_WORDS, 0);
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

// Per-pair read-out (bench.py's per-shape table of the fp32-MFMA products): ms[i] = elapsed ms of pair i, tags[i * 8 ..] the
// words the launcher attached to it (gvd_prof_tag).  Returns the number of pairs written (<= max_pairs) or a negative hipError_t.
extern "C" int gvd_prof_read_pairs(gvd_prof* p, float* ms, int64_t* tags, int max_pairs) {
  if (!p || !ms || max_pairs < 0) return GVD_EINVAL;
  const int n = p->used < max_pairs ? p->used : max_pairs;
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipEventElapsedTime(&ms[i], p->start[i], p->stop[i]);
    if (e != hipSuccess) return -(int)e;
    if (tags)
      for (int k = 0; k < GVD_PROF_TAG_WORDS; ++k) tags[(size_t)i * GVD_PROF_TAG_WORDS + k] = p->tag[(size_t)i * GVD_PROF_TAG_WORDS + k];
  }
  return n;
}

// the pair the next gvd_prof_begin opens: index (-1 when the recorder is full) and its tag words
int gvd_prof_next(gvd_prof* p) { return (!p || p->used >= (int)p->start.size()) ? -1 : p->used; }
void gvd_prof_tag(gvd_prof* p, const int64_t (&words)[GVD_PROF_TAG_WORDS]) {
  if (!p || p->used >= (int)p->start.size()) return;
  for (int k = 0; k < GVD_PROF_TAG_WORDS; ++k) p->tag[(size_t)p->used * GVD_PROF_TAG_WORDS + k] = words[k];
}
