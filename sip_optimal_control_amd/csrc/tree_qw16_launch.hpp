// tree_qw16_launch.hpp -- table entry of the fused tree kernels (tree_qw16.hip) for the tree C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "tree_qw16.hpp"

namespace sipamd {

struct TreeClass {
  int n, m; // padded state / control dimension of the class
  const char *name;
  hipError_t (*launch)(const TreeSchedule &ts, const double *input, double *output, double *work, double *pgains,
                       double *spill, int32_t *status, long batch, hipStream_t s);
  // the same sweep that also writes every LQR::Workspace field into the work arena (sip_lqr_tree_factor_solve_workspace)
  hipError_t (*launch_export)(const TreeSchedule &ts, const double *input, double *output, double *work, double *pgains,
                              double *spill, int32_t *status, long batch, hipStream_t s);
};

// Smallest size class that holds a tree whose largest state / control dimensions are max_n / max_m;
// nullptr: none (max_n > 15 or max_m > 8): the general engine serves it.
const TreeClass *find_tree_class(int max_n, int max_m);

} // namespace sipamd
