// Shared host-side helpers for libpromptcache_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/promptcache_hip.h"

#define PC_EXPORT extern "C" __attribute__((visibility("default")))

void pc_set_error(const char* fmt, ...);

#define PC_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) {               \
            pc_set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

// Launch-error check without synchronising (keeps every entry point graph-capturable).
static inline int pc_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pc_set_error("%s: %s", what, hipGetErrorString(e));
        return PC_ERR_HIP(e);
    }
    return PC_OK;
}

static inline int pc_ceil_div(int a, int b) { return (a + b - 1) / b; }

// PC_FORMAL_HANDOFF=1 (environment, read at every launch that has an in-launch hand-off; captured graphs keep what their capture
// read): the arrival of gemm_skinny_ks_kernel / the fused split-KV merge as an acq_rel fetch-add -- the C++-memory-model form --
// instead of relaxed + s_waitcnt vmcnt(0).  Default off: +1.7 us per launch (buffer_wbl2 + buffer_inv on the critical path).
#include <stdlib.h>
static inline int pc_formal_handoff(void) {
    const char* e = getenv("PC_FORMAL_HANDOFF");
    return (e && e[0] == '1') ? 1 : 0;
}

#ifdef __HIPCC__
// fp32 -> split-precision fp16 pair (hi = fp16(s), lo = fp16(s - hi)).  The value is pinned in a register first:
// under -ffp-contract=fast hipcc may fuse the multiply that produced s into the conversion (v_fma_mix*_f16) for one
// of the two uses and not the other, and the pair then describes two different roundings of the product (observed in
// the GELU epilogue: hi one fp16 ulp off with lo of the wrong sign, at values within an fp32 ulp of an fp16 tie).
__device__ __forceinline__ void pc_split(float s, _Float16& hi, _Float16& lo) {
    asm volatile("" : "+v"(s));
    hi = (_Float16)s;
    lo = (_Float16)(s - (float)hi);
}
#endif
