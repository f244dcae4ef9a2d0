// Versioned internal interface between libribca_hip.so and libribca_hip_test.so.
//
// Both libraries are built with -fvisibility=hidden: the product library exports the C entry points of include/ribca_hip.h and nothing else --
// no C++ launcher, no kernel stub, no helper.  The kernel-level hooks of tests/ and tools/ (include/ribca_hip_test.h) still have to drive the
// SAME launchers and kernels the product's forward uses (and share its state: the A/B variant switch, the thread-local error record), so the
// product library hands them out as ONE table of function pointers, through the one entry point ribca_internal_table(version).  The table is
// not a stable ABI: its layout belongs to one build, the version is bumped whenever a launcher's signature or the list below changes, and a
// mismatch is a NULL (the hooks then return a status).
#pragma once
#include <stdint.h>

#include "ribca_kernels.h"
#include "ribca_status.h"

#define RIBCA_INTERNAL_VERSION 6001

// every function of namespace ribca that csrc/ribca_test_api.hip calls (none of them overloaded)
#define RIBCA_INTERNAL_FUNCS(X)                                                                                                              \
  X(api_finish) X(api_fail) X(mx_wh_bytes) X(mx_wx_bytes) X(gemm_padded_n) X(gemm_resid_bn) X(gemm_resid_tiles) X(gemm_resid_part_rows)      \
  X(gemm_set_variant) X(gemm_set_stamp_buffer) X(gemm_mx_supported) X(cell_attention_supported) X(make_attn_geom) X(launch_pack_wf)          \
  X(launch_pack_weight) X(launch_pack_weight_fold) X(launch_mx_pack_w) X(launch_mx_pack_act) X(launch_layernorm_ps) X(launch_row_stats_ps)   \
  X(launch_ln_finalize) X(launch_gemm_resid) X(launch_gemm_gelu) X(launch_gemm_qkv) X(launch_gemm_qkv_ln) X(launch_gemm_gelu_ln)             \
  X(launch_gemm_gelu_mx) X(launch_gemm_resid_ps) X(launch_gemm_mx_resid) X(launch_gemm_mx_qkv_ln) X(launch_gemm_mx_gelu) X(launch_attention) \
  X(launch_cell_qkv_attention)

namespace ribca {
struct InternalTable {
  int32_t version;
  int32_t n_funcs;
#define RIBCA_X(name) decltype(&ribca::name) name;
  RIBCA_INTERNAL_FUNCS(RIBCA_X)
#undef RIBCA_X
};
}  // namespace ribca
