// Event-pair timing of kernel families (see prof.h).  iq_prof_collect() synchronises, sums the
// elapsed time of every recorded pair per family and recycles the events.
#include <vector>

#include "iqvit.h"
#include "prof.h"

int g_iq_prof_on = 0;

namespace {
struct Pair { hipEvent_t a, b; int fam; };
std::vector<Pair> g_pairs;
std::vector<hipEvent_t> g_free;
hipEvent_t g_open[IQ_FAM_COUNT];

hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void iq_prof_mark(int fam, hipStream_t st, bool begin) {
  if (fam < 0 || fam >= IQ_FAM_COUNT) return;
  hipEvent_t e = get_event();
  (void)hipEventRecord(e, st);
  if (begin) {
    g_open[fam] = e;
  } else {
    g_pairs.push_back(Pair{g_open[fam], e, fam});
  }
}

extern "C" int iq_prof_enable(int on) {
  g_iq_prof_on = on ? 1 : 0;
  return 0;
}

extern "C" int iq_prof_collect(double* ms, long long* count) {
  if (!ms || !count) return 1;
  for (int i = 0; i < IQ_FAM_COUNT; ++i) { ms[i] = 0.0; count[i] = 0; }
  if (!g_pairs.empty()) (void)hipEventSynchronize(g_pairs.back().b);
  for (auto& p : g_pairs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { ms[p.fam] += t; count[p.fam] += 1; }
    g_free.push_back(p.a);
    g_free.push_back(p.b);
  }
  g_pairs.clear();
  return 0;
}
