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
