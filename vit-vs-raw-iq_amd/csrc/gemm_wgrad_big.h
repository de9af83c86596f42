// gemm_wgrad_big.hip: LDS-shared 256-row output tiles for grouped weight gradients (declared for gemm_wgrad.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "iqvit.h"

struct WbPlan { int tk, ntile, splits, rows_per_split; size_t floats; bool transposed[4]; };

// false when the group does not fit that kernel (M % 64, operand alignment, no common column tile 128 | 192 | 256)
bool wgrad_big_plan(const iq_wgrad_problem_t* pr, int nprob, int M, WbPlan* out);
// launches the partial-tile kernel; slab[i] / bslab[i]: `plan.splits` rows of pad4(N*K) resp. pad4(N) floats per problem
void wgrad_big_launch(const iq_wgrad_problem_t* pr, int nprob, int M, const WbPlan& plan, float* ws, float** slab, float** bslab,
                      hipStream_t st);
