// Shared helpers for libdeft_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/deft_hip.h"

void deft_set_error(const char* fmt, ...);

#define DEFT_CHECK(cond, code, ...)                 \
    do {                                            \
        if (!(cond)) {                              \
            deft_set_error(__VA_ARGS__);            \
            return (code);                          \
        }                                           \
    } while (0)

#define DEFT_CHECK_LAUNCH(name)                                              \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            deft_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return -100;                                                     \
        }                                                                    \
    } while (0)

static inline int deft_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
