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

// Dynamic LDS, 16-byte aligned base (ds_read/write_b128).  Kernels that use it declare no
// static __shared__ objects, so the dynamic region starts at offset 0 (guide G17).
#ifndef DEFT_DYN_LDS      /* the unit-test SIMT emulator pre-defines this hook */
#define DEFT_DYN_LDS(type, var) extern __shared__ __attribute__((aligned(16))) type var[]
#endif

static inline int deft_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
