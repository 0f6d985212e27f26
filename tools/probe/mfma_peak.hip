// Probe: attainable v_mfma_f32_32x32x2_f32 rate on this box (random-ish vs zero operands, 1-3 waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_peak.hip -o tools/probe/mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    float a = seed * (float)(threadIdx.x % 61 + 1) * 0.013f, b = seed * (float)(threadIdx.x % 53 + 3) * 0.007f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc3, 0, 0, 0);
        }
        a = -a; b = -b;      // keep the accumulators bounded, operands toggling
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (float seed : {1.0f, 0.0f})
        for (int wg_per_cu : {1, 2, 3, 4}) {
            const int grid = 256 * wg_per_cu, iters = 20000;
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, 200, seed);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, seed);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)grid * 4 /*waves*/ * iters * 32.0 * 4096.0;
            printf("operands %s  %d wave(s)/SIMD: %.1f TFLOP/s (%.2f ms)\n", seed ? "varying" : "zero", wg_per_cu, fl / ms / 1e9, ms);
        }
    return 0;
}
