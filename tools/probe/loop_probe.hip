// Probe: which ingredient of the implicit-GEMM K loop costs MFMA issue rate?  A 128x128x32-chunk loop is built up
// step by step (template flags); every variant executes the SAME 64 v_mfma_f32_32x32x2_f32 per wave per chunk.
//   F=0 MFMA only (operands in registers)        F=1 + 16 ds_read_b128 fragment reads per chunk
//   F=2 + 8 ds_write_b128 + 2 barriers per chunk   F=3 + 8 global loads per chunk (register staged, waited at the store)
//   F=4 + ~60 address VALU per chunk
// hipcc --offload-arch=gfx950 -O3 tools/probe/loop_probe.hip -o tools/probe/loop_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LD 36
template <int F>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* out, int chunks, int W, int ldx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem; float* Bs = smem + 128 * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int g = tid & 7, rbase = tid >> 3, frow = lane & 31, fk = (lane >> 5) * 16;
    for (int i = tid; i < 256 * LD; i += 256) smem[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 97 - 48);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a[2][16], b[2][16];
    for (int i = 0; i < 2; ++i) for (int q = 0; q < 16; ++q) { a[i][q] = 0.01f * (lane + q + i); b[i][q] = 0.02f * (lane - q + i); }
    f32x4 va[4], vb[4];
    for (int i = 0; i < 4; ++i) { va[i] = f32x4{0.1f, 0.2f, 0.3f, 0.4f}; vb[i] = va[i]; }
    const float* base = src + (size_t)(blockIdx.x % 512) * 32768;
    int r0 = rbase, r1 = g;
    for (int kt = 0; kt < chunks; ++kt) {
        if (F >= 2) {
            for (int i = 0; i < 4; ++i) *(f32x4*)&As[(rbase + 32 * i) * LD + g * 4] = va[i];
            for (int i = 0; i < 4; ++i) *(f32x4*)&Bs[(rbase + 32 * i) * LD + g * 4] = vb[i];
            __syncthreads();
        }
        if (F >= 3) {
            for (int i = 0; i < 4; ++i) {
                unsigned off;
                if (F >= 4) {     // im2col-like address arithmetic
                    const int iy = r0 + (kt % 3) + i, ix = r1 + (kt % 5);
                    const bool ok = (unsigned)iy < 4096u && (unsigned)ix < (unsigned)W;
                    off = ok ? (unsigned)(((iy * W + ix) * ldx + g * 4) & 32767) : 0u;
                } else off = (unsigned)((((kt * 4 + i) * 256 + tid) * 4) & 32767);
                va[i] = *(const f32x4*)(base + off);
                vb[i] = *(const f32x4*)(base + ((off + 8192) & 32767));
            }
        }
        if (F >= 1) {
            for (int i = 0; i < 2; ++i) for (int q = 0; q < 4; ++q) {
                const f32x4 t = *(const f32x4*)&As[((wm * 2 + i) * 32 + frow) * LD + fk + 4 * q];
                a[i][4 * q] = t.x; a[i][4 * q + 1] = t.y; a[i][4 * q + 2] = t.z; a[i][4 * q + 3] = t.w;
            }
            for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) {
                const f32x4 t = *(const f32x4*)&Bs[((wn * 2 + j) * 32 + frow) * LD + fk + 4 * q];
                b[j][4 * q] = t.x; b[j][4 * q + 1] = t.y; b[j][4 * q + 2] = t.z; b[j][4 * q + 3] = t.w;
            }
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (F >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s + va[0].x + vb[3].w;
}
// prefetch distance 2: loads of chunk kt+2 are issued while chunk kt computes (two staging register sets)
__global__ __launch_bounds__(256) void k2(const float* __restrict__ src, float* out, int chunks, int W, int ldx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem; float* Bs = smem + 128 * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int g = tid & 7, rbase = tid >> 3, frow = lane & 31, fk = (lane >> 5) * 16;
    for (int i = tid; i < 256 * LD; i += 256) smem[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 97 - 48);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a[2][16], b[2][16];
    f32x4 va0[4], vb0[4], va1[4], vb1[4];
    const float* base = src + (size_t)(blockIdx.x % 512) * 32768;
    int r0 = rbase, r1 = g;
    auto load = [&](int kt, f32x4 (&va)[4], f32x4 (&vb)[4]) {
        for (int i = 0; i < 4; ++i) {
            const int iy = r0 + (kt % 3) + i, ix = r1 + (kt % 5);
            const bool ok = (unsigned)iy < 4096u && (unsigned)ix < (unsigned)W;
            const unsigned off = ok ? (unsigned)(((iy * W + ix) * ldx + g * 4) & 32767) : 0u;
            va[i] = *(const f32x4*)(base + off);
            vb[i] = *(const f32x4*)(base + ((off + 8192) & 32767));
        }
    };
    auto body = [&](f32x4 (&va)[4], f32x4 (&vb)[4], int kt_next2) {
        for (int i = 0; i < 4; ++i) *(f32x4*)&As[(rbase + 32 * i) * LD + g * 4] = va[i];
        for (int i = 0; i < 4; ++i) *(f32x4*)&Bs[(rbase + 32 * i) * LD + g * 4] = vb[i];
        __syncthreads();
        load(kt_next2, va, vb);                 // refill the set just stored: chunk kt+2
        for (int i = 0; i < 2; ++i) for (int q = 0; q < 4; ++q) {
            const f32x4 t = *(const f32x4*)&As[((wm * 2 + i) * 32 + frow) * LD + fk + 4 * q];
            a[i][4 * q] = t.x; a[i][4 * q + 1] = t.y; a[i][4 * q + 2] = t.z; a[i][4 * q + 3] = t.w;
        }
        for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) {
            const f32x4 t = *(const f32x4*)&Bs[((wn * 2 + j) * 32 + frow) * LD + fk + 4 * q];
            b[j][4 * q] = t.x; b[j][4 * q + 1] = t.y; b[j][4 * q + 2] = t.z; b[j][4 * q + 3] = t.w;
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    load(0, va0, vb0); load(1, va1, vb1);
    for (int kt = 0; kt < chunks; kt += 2) {
        body(va0, vb0, kt + 2);
        body(va1, vb1, kt + 3);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s + va0[0].x + vb1[3].w;
}
void run2(const float* src, float* out, int wgs) {
    const int grid = 256 * wgs, chunks = 600;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k2, dim3(grid), dim3(256), 256 * LD * 4, 0, src, out, 20, 272, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k2, dim3(grid), dim3(256), 256 * LD * 4, 0, src, out, chunks, 272, 64);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("F=4 prefetch-2 wg/CU=%d %-40s %.1f TFLOP/s\n", wgs, "(two staging register sets)", (double)grid * 4 * chunks * 64.0 * 4096.0 / ms / 1e9);
}
// wave specialisation: waves 0-3 only read fragments + MFMA, wave 4 issues every LDS-DMA of the chunk
// (2 LDS stages, unpadded swizzled image, one barrier per chunk)
template <bool SPECIAL>
__global__ __launch_bounds__(320) void k3(const float* __restrict__ src, float* out, int chunks, int W, int ldx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 256 * 32; i += blockDim.x) smem[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 97 - 48);
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x % 512) * 32768), 0, 0x7FFFFFFF, 0x00020000);
    const int nload = SPECIAL ? 1 : 4;              // waves that issue DMA
    const bool loader = SPECIAL ? wave == 4 : wave < 4;
    const bool consumer = wave < 4;
    const int lw = SPECIAL ? 0 : wave;              // loader index
    auto dma = [&](int kt, int stage) {             // 32 KB = 32 wave-instructions per chunk, split over the loader waves
        for (int j = lw; j < 32; j += nload) {
            const int row = j * 8 + (lane >> 3), g = lane & 7;
            const int iy = row + (kt % 3), ix = g + (kt % 5);
            const bool ok = (unsigned)iy < 4096u && (unsigned)ix < (unsigned)W;
            const unsigned off = ok ? (unsigned)((((iy * W + ix) * ldx + (g ^ ((row >> 1) & 7)) * 4) & 32767) * 4) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + stage * 8192 + j * 256), 16, off, 0, 0, 0);
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float a[2][16], b[2][16];
    const int wm = (wave >> 1) & 1, wn = wave & 1, frow = lane & 31, fk = (lane >> 5) * 16, fsw = (frow >> 1) & 7;
    if (loader) dma(0, 0);
    __syncthreads();
    for (int kt = 0; kt < chunks; ++kt) {
        const int cs = kt & 1;
        if (loader) dma(kt + 1, cs ^ 1);
        if (consumer) {
            const float* As = smem + cs * 8192; const float* Bs = As + 4096;
            for (int i = 0; i < 2; ++i) for (int q = 0; q < 4; ++q) {
                const f32x4 t = *(const f32x4*)&As[((wm * 2 + i) * 32 + frow) * 32 + (((fk >> 2) + q) ^ fsw) * 4];
                a[i][4 * q] = t.x; a[i][4 * q + 1] = t.y; a[i][4 * q + 2] = t.z; a[i][4 * q + 3] = t.w;
            }
            for (int j = 0; j < 2; ++j) for (int q = 0; q < 4; ++q) {
                const f32x4 t = *(const f32x4*)&Bs[((wn * 2 + j) * 32 + frow) * 32 + (((fk >> 2) + q) ^ fsw) * 4];
                b[j][4 * q] = t.x; b[j][4 * q + 1] = t.y; b[j][4 * q + 2] = t.z; b[j][4 * q + 3] = t.w;
            }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (consumer) out[blockIdx.x * 256 + tid] = s;
}
template <bool SPECIAL> void run3(const float* src, float* out, int wgs) {
    const int grid = 256 * wgs, chunks = 600, nthr = SPECIAL ? 320 : 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k3<SPECIAL>, dim3(grid), dim3(nthr), 65536, 0, src, out, 20, 272, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k3<SPECIAL>, dim3(grid), dim3(nthr), 65536, 0, src, out, chunks, 272, 64);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("LDS-DMA 2-stage, %s, wg/CU=%d: %.1f TFLOP/s\n", SPECIAL ? "1 loader wave + 4 MFMA waves" : "every wave loads (as shipped)", wgs,
           (double)grid * 4 * chunks * 64.0 * 4096.0 / ms / 1e9);
}
template <int F> void run(const float* src, float* out, const char* what, int wgs = 3) {
    const int grid = 256 * wgs, chunks = 600;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<F>, dim3(grid), dim3(256), 256 * LD * 4, 0, src, out, 20, 272, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<F>, dim3(grid), dim3(256), 256 * LD * 4, 0, src, out, chunks, 272, 64);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("F=%d wg/CU=%d %-58s %.1f TFLOP/s\n", F, wgs, what, (double)grid * 4 * chunks * 64.0 * 4096.0 / ms / 1e9);
}
int main() {
    float *src, *out; hipMalloc(&src, 512 * 32768 * 4); hipMalloc(&out, 768 * 256 * 4);
    hipMemset(src, 0, 512 * 32768 * 4);
    run<2>(src, out, "warm-up", 3); run<2>(src, out, "warm-up", 3);
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(src, out, "+ loads + address VALU", 2);
        run3<false>(src, out, 2);
        run3<true>(src, out, 2);
    }
    run<0>(src, out, "MFMA only");
    run<1>(src, out, "+ 16 ds_read_b128 fragment reads / chunk");
    run<2>(src, out, "+ 8 ds_write_b128 + 2 barriers / chunk");
    run<3>(src, out, "+ 8 global loads / chunk (waited at the LDS store)");
    run<4>(src, out, "+ im2col-like address VALU");
    return 0;
}
