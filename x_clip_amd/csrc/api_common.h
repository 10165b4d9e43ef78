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

// Measurement switches (kernel generation A/B, ablation masks whose results are garbage, policy overrides) exist only in the
// measurement build (`python -m x_clip_amd.build --measure` -> libxclip_hip_measure.so, compiled with -DXCLIP_MEASURE and loaded
// explicitly by tools/).  libxclip_hip.so -- the product -- reads no environment variable: a stray XCLIP_* setting cannot change what
// a training step computes (tests/test_abi_exports.py checks that the library does not even import getenv).
#ifdef XCLIP_MEASURE
inline int measure_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
constexpr int measure_env(const char*, int dflt) { return dflt; }
#endif

}  // namespace xcapi

#define XC_REQUIRE(cond, msg) \
    do {                      \
        if (!(cond)) return xcapi::fail(__func__, msg); \
    } while (0)
