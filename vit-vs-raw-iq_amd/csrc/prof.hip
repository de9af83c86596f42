// Event-pair timing of kernel families (see prof.h).  iq_prof_collect() synchronises, sums the
// elapsed time of every recorded pair per family and recycles the events; pairs that carry a kernel
// name are also summed per name (iq_prof_kernels).
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "iqvit.h"
#include "prof.h"

int g_iq_prof_on = 0;

namespace {
struct Pair { hipEvent_t a, b; int fam; int kid; double bytes, flops; };
struct KernelSum { double ms = 0, bytes = 0, flops = 0; long long count = 0; int fam = 0; };
std::vector<Pair> g_pairs;
std::vector<hipEvent_t> g_free;
hipEvent_t g_open[IQ_FAM_COUNT];
int g_open_kid[IQ_FAM_COUNT];
double g_open_bytes[IQ_FAM_COUNT], g_open_flops[IQ_FAM_COUNT];
std::vector<std::string> g_names;            // kernel id -> name
std::map<std::string, int> g_ids;
std::map<int, KernelSum> g_sums;             // since the last iq_prof_kernels(reset)

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
    g_open_kid[fam] = -1;
    g_open_bytes[fam] = g_open_flops[fam] = 0.0;
  } else {
    g_pairs.push_back(Pair{g_open[fam], e, fam, g_open_kid[fam], g_open_bytes[fam], g_open_flops[fam]});
  }
}

void iq_prof_kernel(int fam, const char* name, double bytes, double flops) {
  if (fam < 0 || fam >= IQ_FAM_COUNT || !name) return;
  auto it = g_ids.find(name);
  int id;
  if (it == g_ids.end()) { id = (int)g_names.size(); g_names.push_back(name); g_ids[name] = id; }
  else id = it->second;
  g_open_kid[fam] = id;
  g_open_bytes[fam] = bytes;
  g_open_flops[fam] = flops;
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
    if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
      ms[p.fam] += t; count[p.fam] += 1;
      if (p.kid >= 0) {
        KernelSum& k = g_sums[p.kid];
        k.ms += t; k.bytes += p.bytes; k.flops += p.flops; k.count += 1; k.fam = p.fam;
      }
    }
    g_free.push_back(p.a);
    g_free.push_back(p.b);
  }
  g_pairs.clear();
  return 0;
}

// Per-kernel sums of the pairs collected so far, one line per kernel name:
//   name \t family \t launches \t total ms \t total algorithmic bytes \t total flops \n
// Returns the number of bytes the full text needs (excluding the terminating 0); writes at most cap - 1 of them.
extern "C" size_t iq_prof_kernels(char* out, size_t cap, int reset) {
  std::string s;
  char line[512];
  for (auto& kv : g_sums) {
    const KernelSum& k = kv.second;
    snprintf(line, sizeof(line), "%s\t%d\t%lld\t%.6f\t%.0f\t%.0f\n", g_names[kv.first].c_str(), k.fam, k.count, k.ms, k.bytes, k.flops);
    s += line;
  }
  if (out && cap > 0) {
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(out, s.data(), n);
    out[n] = 0;
  }
  if (reset) g_sums.clear();
  return s.size();
}
