// Optional per-kernel-family timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: zero cost beyond one predictable branch per entry point.
// A launch site may also name the KERNEL it dispatched and the algorithmic bytes / flops of that launch
// (IqProfScope::kernel): bench.py's roofline.kernels lists the top kernels by time from these records, under the same
// kernel names rocprofv3 --kernel-trace --stats prints (profiles/*_kernel_stats_*.csv).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

enum IqProfFamily { IQ_FAM_GEMM_NT = 0, IQ_FAM_WGRAD = 1, IQ_FAM_ATTN_FWD = 2, IQ_FAM_ATTN_BWD = 3,
                    IQ_FAM_LN_FWD = 4, IQ_FAM_LN_BWD = 5, IQ_FAM_MISC = 6, IQ_FAM_OPT = 7, IQ_FAM_COUNT = 8 };

extern int g_iq_prof_on;
void iq_prof_mark(int fam, hipStream_t st, bool begin);
void iq_prof_kernel(int fam, const char* name, double bytes, double flops);   // detail of the OPEN scope of `fam`

struct IqProfScope {
  int fam; hipStream_t st; bool on;
  IqProfScope(int f, hipStream_t s) : fam(f), st(s), on(g_iq_prof_on != 0) { if (on) iq_prof_mark(fam, st, true); }
  ~IqProfScope() { if (on) iq_prof_mark(fam, st, false); }
  // name: printf-style kernel name with its template arguments, as rocprofv3 lists it (without the namespace)
  void kernel(const char* name, double bytes, double flops) const { if (on) iq_prof_kernel(fam, name, bytes, flops); }
};
#define IQ_PROF(fam, stream) IqProfScope iq_prof_scope_((fam), (hipStream_t)(stream))
// IQ_PROF_K(bytes, flops, "kernel_name<%d, %d>", a, b): names the kernel of the enclosing IQ_PROF scope
#define IQ_PROF_K(bytes_, flops_, ...)                                     \
  do {                                                                     \
    if (iq_prof_scope_.on) {                                               \
      char iq_prof_name_[192];                                             \
      snprintf(iq_prof_name_, sizeof(iq_prof_name_), __VA_ARGS__);         \
      iq_prof_scope_.kernel(iq_prof_name_, (double)(bytes_), (double)(flops_)); \
    }                                                                      \
  } while (0)
