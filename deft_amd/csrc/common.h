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

// native 4-vector for all staging traffic: HIP's float4 is a struct whose plain copies lower to
// memcpy through a private (scratch) alloca that SROA does not split
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Raw buffer loads (SRD in SGPRs + 32-bit byte offset in a VGPR).  An offset >= num_records
// returns 0 from the hardware: im2col zero padding / invalid tile rows cost one v_cndmask on
// the OFFSET instead of a select on the loaded data (which would pull the s_waitcnt vmcnt in
// front of the MFMAs).  DEFT_OOB + any in-tile offset stays >= 2^31 without wrapping.
#define DEFT_OOB 0x80000000u
#ifndef DEFT_BUFFER_HOOKS      /* the unit-test SIMT emulator pre-defines these hooks */
typedef __amdgpu_buffer_rsrc_t deft_rsrc_t;
__device__ __forceinline__ deft_rsrc_t deft_make_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ f32x4 deft_buffer_load_x4(deft_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
// LDS-DMA (`buffer_load_dwordx4 ... lds`): lane l of the wave deposits its 16 bytes at
// lds_wave_base + 16*l (wave-uniform base in M0, lane-linear image); out-of-range lanes deposit
// zeros (verified on gfx950: tools/probe/lds_dma_oob.hip).  Completion is counted by vmcnt.
__device__ __forceinline__ void deft_buffer_load_lds_x4(deft_rsrc_t r, float* lds_wave_base, unsigned byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)byte_off, 0, 0, 0);
}
#endif

// Cross-workgroup hand-over of split-K partial tiles.  The workgroups of a tile may sit on different XCDs, whose
// L2s are not coherent with each other: partials are written and read with agent-scope accesses (sc1: write-through /
// miss in the non-coherent levels) instead of paying a whole-L2 write-back + invalidate (`__threadfence()`) per
// workgroup; deft_ws_publish() waits until this wave's stores have been acknowledged before the ticket is taken.
#ifndef DEFT_WS_HOOKS          /* the unit-test SIMT emulator pre-defines these hooks */
__device__ __forceinline__ void deft_ws_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float deft_ws_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void deft_ws_publish() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);                 // gfx9 encoding: vmcnt(0) expcnt(0) lgkmcnt(0); vmcnt counts stores too
}
__device__ __forceinline__ int deft_ws_ticket(int* p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void deft_ws_reset(int* p) { __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

static inline int deft_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
