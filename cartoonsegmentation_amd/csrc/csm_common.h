// csm_common.h -- shared host-side helpers of libcsm355 (error state, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "csm355.h"

namespace csm {
void set_error(const char *fmt, ...);
inline int fail_arg(const char *what) { set_error("invalid argument: %s", what); return CSM_ERR_ARG; }
inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e)); return CSM_ERR_HIP; }
    return CSM_OK;
}
inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }
}  // namespace csm
#define CSM_REQUIRE(cond) do { if (!(cond)) return csm::fail_arg(#cond); } while (0)
#define CSM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { csm::set_error("%s: %s", #call, hipGetErrorString(e_)); return CSM_ERR_HIP; } } while (0)
