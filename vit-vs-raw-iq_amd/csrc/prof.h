// Optional per-kernel-family timing with HIP events on the launch stream (bench.py's roofline leg).
// Disabled by default: zero cost beyond one predictable branch per entry point.
#pragma once
#include <hip/hip_runtime.h>

enum IqProfFamily { IQ_FAM_GEMM_NT = 0, IQ_FAM_WGRAD = 1, IQ_FAM_ATTN_FWD = 2, IQ_FAM_ATTN_BWD = 3,
                    IQ_FAM_LN_FWD = 4, IQ_FAM_LN_BWD = 5, IQ_FAM_MISC = 6, IQ_FAM_OPT = 7, IQ_FAM_COUNT = 8 };

extern int g_iq_prof_on;
void iq_prof_mark(int fam, hipStream_t st, bool begin);

struct IqProfScope {
  int fam; hipStream_t st; bool on;
  IqProfScope(int f, hipStream_t s) : fam(f), st(s), on(g_iq_prof_on != 0) { if (on) iq_prof_mark(fam, st, true); }
  ~IqProfScope() { if (on) iq_prof_mark(fam, st, false); }
};
#define IQ_PROF(fam, stream) IqProfScope iq_prof_scope_((fam), (hipStream_t)(stream))
