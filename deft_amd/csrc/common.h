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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// x = hi + mid + lo with three bf16 pieces (8 + 8 + 8 mantissa bits, round-to-nearest-even each: exact for fp32).
__device__ __forceinline__ void split3(const f32x4 v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = (__bf16)v[e];
        const float r1 = v[e] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        h[e] = hh; m[e] = mm; l[e] = (__bf16)(r1 - (float)mm);
    }
}

// ---- the operand pieces of the split arithmetic (DeftGemmDesc.prec = 1) -------------------------------------------------------------
// DEFT_PIECES = 3: x = hi + mid + lo as three bf16 pieces (exact), six v_mfma_f32_32x32x16_bf16 products per fp32 product (rounds 1-4).
// DEFT_PIECES = 2: x ~ h1 + h2 as two FP16 pieces (11 + 11 significant bits, round-to-nearest each: |x - h1 - h2| <= 2^-24 |x|, half an
//   fp32 ulp, while h2 is a normal fp16 number; below that 2^-25 absolute in the scaled domain), THREE v_mfma_f32_32x32x16_f16 products per fp32
//   product (h1 g1, h1 g2, h2 g1; the dropped h2 g2 is <= 2^-24 relative), fp32 accumulation.  fp16 has 5 exponent bits, so the operands are
//   brought into range by exact powers of two: activations are multiplied by DEFT_ASCALE = 2^4 where they are split (any |x| < 4094 is
//   representable; a residual keeps full precision down to |x| = 2^-6, and 2^-29 absolute below) and every kernel that does so multiplies its
//   per-column epilogue scale by 2^-4; weight rows are scaled by the HOST (engine.py: 2^k per output channel so that the row's largest
//   entry lands in [2^12, 2^13), folded into DeftGemmDesc.scale).  An activation beyond the range becomes +-inf in h1 and NaN downstream --
//   loud, never silently wrong.  Half the matrix-core work and 2/3 of the piece bytes of the three-piece form.
#ifndef DEFT_PIECES
#define DEFT_PIECES 2
#endif
constexpr int DEFT_NP = DEFT_PIECES;
constexpr int DEFT_NPROD = DEFT_NP == 3 ? 6 : 3;
#if DEFT_PIECES == 3
typedef __bf16 deft_piece_t;
#define DEFT_ASCALE 1.f
#define DEFT_ASCALE_INV 1.f
#elif DEFT_PIECES == 2
typedef _Float16 deft_piece_t;
#define DEFT_ASCALE 16.f
#define DEFT_ASCALE_INV 0.0625f
#else
#error "DEFT_PIECES is 2 (fp16 pieces) or 3 (bf16 pieces)"
#endif
typedef deft_piece_t pcx8 __attribute__((ext_vector_type(8)));
typedef deft_piece_t pcx4 __attribute__((ext_vector_type(4)));
// products of an fp32 product, smallest terms first: (piece of A, piece of B)
__device__ __forceinline__ constexpr int deft_qa(int q) { return DEFT_NP == 3 ? (q == 0 ? 1 : q == 1 ? 2 : q == 2 ? 0 : q == 3 ? 1 : 0) : (q == 0 ? 1 : 0); }
__device__ __forceinline__ constexpr int deft_qb(int q) { return DEFT_NP == 3 ? (q == 0 ? 1 : q == 1 ? 0 : q == 2 ? 2 : q == 3 ? 0 : q == 4 ? 1 : 0) : (q == 1 ? 1 : 0); }
typedef float deft_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ deft_f32x16 deft_mfma_pc(const pcx8 a, const pcx8 b, const deft_f32x16 c) {
#if DEFT_PIECES == 3
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
#if DEFT_PIECES == 2
// Two fp16 pieces of a PAIR of values in 3 (4 with a scale) VALU instructions: h = (fp16(x0), fp16(x1)) packed, m = (fp16(x0 - h.lo),
// fp16(x1 - h.hi)) -- v_fma_mix{lo,hi}_f16 computes x - h in fp32 (exact: the residual of a round-to-nearest fp16 conversion is an fp32
// number) and rounds ONCE to fp16 into one half of the destination: the same bits as `(_Float16)(x - (float)(_Float16)x)`.  hipcc's own code
// for that expression is 9-10 instructions per pair inside the kernels (separate conversions, v_cvt_f32_f16 + v_sub, canonicalising v_max);
// tools/probe/f16_split_asm.hip checks these sequences against the C++ expression on the hardware, bit for bit.  `sc` (a power of two) is
// folded into the same instructions.  (Outputs feed LDS / global stores or, a K step later, matrix instructions: no asm-to-MFMA adjacency --
// and that is a REQUIREMENT, not a nicety: v_fma_mixhi_f16 writes half a register, and a matrix instruction that reads the register within a
// few instructions sees the stale half; the compiler inserts no wait states for inline asm.  Found in round 6 in csrc/pairmlp.hip, which
// therefore splits with the C++ expression; profiles/r6_asm_split_mfma_hazard.md.)
#ifndef DEFT_F16_SPLIT_HOOK    /* the unit-test SIMT emulator pre-defines this hook with the C++ expression */
__device__ __forceinline__ void deft_split2_pair(float x0, float x1, unsigned& h, unsigned& m) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(m) : "v"(h), "v"(x0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(m) : "v"(h), "v"(x1));
#else
    h = m = 0; (void)x0; (void)x1;
#endif
}
__device__ __forceinline__ void deft_split2_pair_scaled(float x0, float x1, float sc, unsigned& h, unsigned& m) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "s"(sc));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(x0), "s"(sc), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(x1), "s"(sc), "v"(h));
#else
    h = m = 0; (void)x0; (void)x1; (void)sc;
#endif
}
#endif
#endif
// the 16 x 16 x 32 form (direct.hip: 16 pixels x 16 output channels)
__device__ __forceinline__ f32x4 deft_mfma16_pc(const pcx8 a, const pcx8 b, const f32x4 c) {
#if DEFT_PIECES == 3
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}
// the pieces of four values (A operands: pass scale = DEFT_ASCALE; weights arrive scaled by the host: scale = 1)
__device__ __forceinline__ void deft_split(const f32x4 v, pcx4 (&pc)[DEFT_NP], const float scale = 1.f) {
#if DEFT_PIECES == 2
    typedef unsigned deft_u32x2 __attribute__((ext_vector_type(2)));
    unsigned h0, m0, h1, m1;
    if (scale == 1.f) {
        deft_split2_pair(v[0], v[1], h0, m0);
        deft_split2_pair(v[2], v[3], h1, m1);
    } else {
        deft_split2_pair_scaled(v[0], v[1], scale, h0, m0);
        deft_split2_pair_scaled(v[2], v[3], scale, h1, m1);
    }
    const deft_u32x2 h = {h0, h1}, m = {m0, m1};
    pc[0] = __builtin_bit_cast(pcx4, h);
    pc[1] = __builtin_bit_cast(pcx4, m);
#else
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float r = v[e];
#pragma unroll
        for (int q = 0; q < DEFT_NP; ++q) {
            const deft_piece_t h = (deft_piece_t)r;
            pc[q][e] = h;
            r -= (float)h;
        }
    }
    (void)scale;
#endif
}

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
// the same with a wave-uniform byte offset in an SGPR (soffset: not part of the hardware's range check, so the
// out-of-range trick stays on voff alone) -- a chunk cursor then costs no VALU
__device__ __forceinline__ void deft_buffer_load_lds_x4s(deft_rsrc_t r, void* lds_wave_base, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}
// Pipeline barrier of the LDS-DMA loops: wait until at most N of THIS wave's DMA pieces are still in flight, then
// rendezvous the workgroup (raw s_barrier: __syncthreads() would drain vmcnt to 0 while a DMA is outstanding).  The
// empty asm statements keep the compiler from moving LDS accesses across it.
#define DEFT_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// lgkmcnt(0): this wave's LDS reads have RETURNED before it arrives.  The compiler may sink MFMAs (not memory operations) below
// the barrier together with the wait for their ds_read operands; another wave's DMA into the stage those reads target would
// then race with them (seen on the hardware as rare wrong tiles in the one-stage loop).
#define DEFT_PIPE_BARRIER_ONLY()                                   \
    do {                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
        __builtin_amdgcn_s_barrier();                              \
        asm volatile("" ::: "memory");                             \
    } while (0)
#define DEFT_PIPE_BARRIER(N)                                            \
    do {                                                                \
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); \
        __builtin_amdgcn_s_barrier();                                   \
        asm volatile("" ::: "memory");                                  \
    } while (0)
#endif

// Cross-workgroup hand-over of split-K partial tiles.  The workgroups of a tile may sit on different XCDs, whose
// L2s are not coherent with each other: partials are written and read with agent-scope accesses (sc1: write-through /
// miss in the non-coherent levels) instead of paying a whole-L2 write-back + invalidate (`__threadfence()`) per
// workgroup; deft_ws_publish() waits until this wave's stores have been acknowledged before the ticket is taken.
#ifndef DEFT_WS_HOOKS          /* the unit-test SIMT emulator pre-defines these hooks */
__device__ __forceinline__ void deft_ws_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float deft_ws_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the split-K hand-over relies on gfx950 sc1 (write-through / L2-served) accesses and gfx9 vmcnt semantics"
#endif
// The protocol is the hardware one of MI355X_MICROARCH.md (Workgroup dispatch ...: 'sc1 stores AND sc1 loads'): every
// partial is an sc1 (write-through) store, each wave waits until its stores are acknowledged (vmcnt counts stores on
// gfx9; inline asm so that no compiler pass can drop or move the wait), the workgroup rendezvous, ONE lane takes the
// ticket with a relaxed agent-scope atomic, and the last arriver reads every partial with sc1 (L2-served) loads -- no
// L2 write-back / L1 invalidate per workgroup.  Agent-scope release/acquire fences instead were measured 1.3-4x slower
// on these 19x34 / 38x68 launches.
__device__ __forceinline__ void deft_ws_publish() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ int deft_ws_ticket(int* p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void deft_ws_reset(int* p) { __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// the value may have changed, as far as the optimiser knows (no instruction is emitted)
#ifndef DEFT_OPAQUE            /* the unit-test SIMT emulator pre-defines this hook */
#define DEFT_OPAQUE(v) asm volatile("" : "+v"(v))
#endif
// the same without pinning the statement's place among the other asm statements (the scheduler may still move it)
#ifndef DEFT_OPAQUE_NV
#define DEFT_OPAQUE_NV(v) asm("" : "+v"(v))
#endif

// 1 / x to one ulp (v_rcp_f32) instead of the IEEE division sequence
#ifndef DEFT_FAST_RCP          /* the unit-test SIMT emulator pre-defines this hook */
#define DEFT_FAST_RCP(x) __builtin_amdgcn_rcpf(x)
#endif

// round-half-even double -> int (cv2's saturate_cast<int>(double) = lrint)
#ifndef DEFT_RINT_HOOK        /* the unit-test SIMT emulator pre-defines this hook */
__device__ __forceinline__ int deft_rint(double v) { return __double2int_rn(v); }
#endif

static inline int deft_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ReLU that PROPAGATES NaN (as torch.relu does; fmaxf(NaN, 0) = 0 would launder it): an operand beyond the fp16-piece range turns its
// products into inf - inf = NaN, and that NaN has to reach the heat map, where the host checks for it (deft_amd/detector.py _check_finite) --
// with fmaxf an out-of-range activation became a silent zero.  Same bits as fmaxf for every non-NaN input but -0 (kept; fmaxf gives +0).
__device__ __forceinline__ float deft_relu(const float v) { return v < 0.f ? 0.f : v; }

// ---- LDS-transposed epilogue shared by the MFMA kernels ----------------------------------------------------------
// Phase 1: every wave parks its TM x TN grid of 32x32 accumulators, scaled and shifted, in the LDS tile T[BM][LDT]
// (D reg r of lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), col l&31: a half-wave writes 32 consecutive floats).
// `post`: what the kernel's own operand scaling left in the accumulators (DEFT_ASCALE_INV for the kernels that split their A operand).
template <int TM, int TN>
__device__ __forceinline__ void deft_epilogue_stage(float* T, int LDT, const deft_f32x16 (&acc)[TM][TN], int wm, int wn, int lane,
                                                    const DeftGemmDesc& p, int n0, const float post = DEFT_ASCALE_INV) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cl = (wn * TN + j) * 32 + (lane & 31);
        const int co = n0 + cl;
        const int coc = co < p.Cout ? co : p.Cout - 1;
        const float sc = (p.scale ? p.scale[coc] : 1.f) * post;
        const float sh = p.shift ? p.shift[coc] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float* tp = T + ((wm * TM + i) * 32 + 4 * (lane >> 5)) * LDT + cl;
#pragma unroll
            for (int r = 0; r < 16; ++r) tp[((r & 3) + 8 * (r >> 2)) * LDT] = acc[i][j][r] * sc + sh;
        }
    }
}
// Phase 2 (after a barrier): every thread owns 8 consecutive channels of an output row: residual (16-byte loads), ReLU,
// fp32 store (p.y, nullable) and the operand pieces (p.y3, nullable; DEFT_PIECES of them) as 16-byte stores.  row_to_m(row) -> the output
// row (pixel index) of tile row `row`, or -1.
template <int BM, int BN, int NT, typename RowFn>
__device__ __forceinline__ void deft_epilogue_rows(const float* T, const DeftGemmDesc& p, int n0, int tid, RowFn row_to_m) {
    constexpr int LDT = BN + 4, G = BN / 8, NI = (BM * G + NT - 1) / NT;
    static_assert(G <= 64 && (G & (G - 1)) == 0, "the threads of a row are an aligned power-of-two lane group");
    const bool has_res = p.res != nullptr;
    if (p.fold_y != nullptr) {
        // folded 1x1 conv (DeftGemmDesc.fold_w): every lane stays active through the shuffles; invalid items contribute zeros
        for (int it = 0; it < NI; ++it) {
            const int item = it * NT + tid;
            const int row = item / G, c8 = item - row * G;
            const int co = n0 + c8 * 8;
            const bool cok = row < BM && co < p.Cout;
            const long long ml = row < BM ? row_to_m(row) : -1;
            const float* tp = T + (row < BM ? row : 0) * LDT + c8 * 8;
            f32x4 v0 = *(const f32x4*)tp, v1 = *(const f32x4*)(tp + 4);
            if (has_res && cok && ml >= 0) {
                const float* rp = p.res + (size_t)ml * p.ldr + co;
                v0 += *(const f32x4*)rp;
                v1 += *(const f32x4*)(rp + 4);
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = deft_relu(v0[e]); v1[e] = deft_relu(v1[e]); }
            }
            if (!cok) { v0 = f32x4{0.f, 0.f, 0.f, 0.f}; v1 = v0; }
            if (p.y != nullptr && cok && ml >= 0) {
                float* yp = p.y + (size_t)ml * p.ldy + co;
                *(f32x4*)yp = v0;
                *(f32x4*)(yp + 4) = v1;
            }
            const int coc = cok ? co : 0;
            for (int c2 = 0; c2 < p.fold_n; ++c2) {
                const float* fw = p.fold_w + (size_t)c2 * p.Cout + coc;
                const f32x4 w0 = *(const f32x4*)fw, w1 = *(const f32x4*)(fw + 4);
                float s = v0[0] * w0[0];
                s = fmaf(v0[1], w0[1], s); s = fmaf(v0[2], w0[2], s); s = fmaf(v0[3], w0[3], s);
                s = fmaf(v1[0], w1[0], s); s = fmaf(v1[1], w1[1], s); s = fmaf(v1[2], w1[2], s); s = fmaf(v1[3], w1[3], s);
#pragma unroll
                for (int o = 1; o < G; o <<= 1) s += __shfl_xor(s, o);
                if (c8 == 0 && ml >= 0) p.fold_y[((size_t)(n0 / BN) * (size_t)p.M + (size_t)ml) * p.fold_ld + c2] = s;
            }
        }
        return;
    }
#pragma unroll 2
    for (int it = 0; it < NI; ++it) {
        const int item = it * NT + tid;
        const int row = item / G, c8 = item - row * G;
        const int co = n0 + c8 * 8;
        if (row >= BM || co >= p.Cout) continue;
        const long long ml = row_to_m(row);
        if (ml < 0) continue;
        const size_t m = (size_t)ml;
        const float* tp = T + row * LDT + c8 * 8;
        f32x4 v0 = *(const f32x4*)tp, v1 = *(const f32x4*)(tp + 4);
        if (has_res) {
            const float* rp = p.res + m * p.ldr + co;
            v0 += *(const f32x4*)rp;
            v1 += *(const f32x4*)(rp + 4);
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = deft_relu(v0[e]); v1[e] = deft_relu(v1[e]); }
        }
        if (p.y != nullptr) {
            float* yp = p.y + m * p.ldy + co;
            *(f32x4*)yp = v0;
            *(f32x4*)(yp + 4) = v1;
        }
        if (p.y3 != nullptr) {
            pcx4 c0[DEFT_NP], c1[DEFT_NP];
            deft_split(v0, c0, DEFT_ASCALE);
            deft_split(v1, c1, DEFT_ASCALE);
            deft_piece_t* yp = (deft_piece_t*)p.y3 + m * p.ldy3 * DEFT_NP + (co >> 5) * (32 * DEFT_NP) + (co & 31);
#pragma unroll
            for (int q = 0; q < DEFT_NP; ++q) *(pcx8*)(yp + 32 * q) = __builtin_shufflevector(c0[q], c1[q], 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
}

// ---- LDS images of the pre-split operands: where the 16-byte slot (piece q, k-slot ks) of row r sits ------------------------------------
// im2col chunk rows (32 k): DEFT_NP * 4 slots = DEFT_NP * 64 B per row.  A ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31}, + 32) over 64 banks of 4 B; a fragment read has one row per lane, fixed (q, ks).
//   3 pieces, 192-byte rows: physical slot q*4 + (ks ^ ((r >> 2) & 3)) (rounds 2-4).
//   2 pieces, 128-byte rows: a row covers half the banks, so the 16 rows of a lane group must spread over all 8 slots of both halves:
//   physical slot (q*4 + ks) ^ ((r >> 1) & 7) -- the eight even (odd) rows of either lane group have eight different (r >> 1) & 7.
__device__ __forceinline__ constexpr int deft_p3_phys(int q, int ks, int r) {
    return DEFT_NP == 3 ? q * 4 + (ks ^ ((r >> 2) & 3)) : (q * 4 + ks) ^ ((r >> 1) & 7);
}
// the inverse: which (piece, k-slot) the DMA deposits at physical slot ps of row r
__device__ __forceinline__ void deft_p3_logical(int ps, int r, int& q, int& ks) {
    if (DEFT_NP == 3) { q = ps >> 2; ks = (ps & 3) ^ ((r >> 2) & 3); }
    else { const int lg = ps ^ ((r >> 1) & 7); q = lg >> 2; ks = lg & 3; }
}

// igemm3.hip (pre-split operands, DeftGemmDesc.x3): validation, tile choice and launch, called from deft_conv2d_nhwc
int deft_p3_check(const DeftGemmDesc* d, const char* who);
void deft_p3_pick_tile(const DeftGemmDesc* d, int* bm, int* bn);
int deft_p3_dispatch(const DeftGemmDesc* d, hipStream_t s);
int deft_p3h_dispatch(const DeftGemmDesc* d, hipStream_t s);
// dcn.hip (patch form of the DCN, DeftGemmDesc.p3_kernel = 2), called from deft_dcn_v2_nhwc
int deft_dcnp_dispatch(const DeftGemmDesc* d, hipStream_t s);
int deft_conv3p_dispatch(const DeftGemmDesc* d, hipStream_t s);      // plain 3x3 conv, <= 32 columns, fp32 input (DeftGemmDesc.p3_kernel = 3)
