// api_common.h -- what the translation units behind include/xclip.h share: the thread-local error string and the argument-check
// helpers.  (Two units because the attention kernels are compiled with -amdgpu-mfma-vgpr-form and the GEMM-shaped ones are not.)
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/xclip.h"

namespace xcapi {

int fail(const char* fn, const char* what);                  // records "<fn>: <what>" for xclip_last_error(), returns 1
int check_launch(const char* fn);                            // hipGetLastError() -> 0, or 2 with the HIP error string recorded

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int vec_of(int dtype) { return dtype == XCLIP_BF16 ? 8 : 4; }
inline int esize(int dtype) { return dtype == XCLIP_BF16 ? 2 : 4; }
inline bool dtype_ok(int dtype) { return dtype == XCLIP_F32 || dtype == XCLIP_BF16; }

}  // namespace xcapi

#define XC_REQUIRE(cond, msg) \
    do {                      \
        if (!(cond)) return xcapi::fail(__func__, msg); \
    } while (0)
