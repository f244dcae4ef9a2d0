// Status plumbing of the translation units that export C entry points (ribca_api.hip: the product ABI of include/ribca_hip.h;
// ribca_test_api.hip: the kernel-level hooks of include/ribca_hip_test.h, a library of their own).  The record is thread-local and lives in
// libribca_hip.so; ribca_last_error() returns it whichever library the failing entry point came from.
#pragma once
#include <hip/hip_runtime.h>

namespace ribca {
int api_fail(const char* msg);                       // records msg, returns 1
int api_hip_fail(hipError_t e, const char* what);    // "<what>: <runtime's text>", returns 1
// tail of every entry point that enqueues work: a launcher that refused (ribca_common.h launch_error) or a launch the runtime rejected
// becomes the non-zero status the headers promise; 0 otherwise
int api_finish();
const char* api_last_error();
}  // namespace ribca

#define HIP_TRY(expr)                                               \
  do {                                                              \
    hipError_t e_ = (expr);                                         \
    if (e_ != hipSuccess) return ribca::api_hip_fail(e_, #expr);    \
  } while (0)
#define RIBCA_FINISH()                                   \
  do {                                                   \
    if (ribca::api_finish() != 0) return 1;              \
  } while (0)
