// Does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs on gfx950, and does v_cvt_pk f32 -> f16 produce them?  (Decides whether a
// two-piece fp16 operand split keeps its second piece for small residuals.)  Plain HIP: hipcc --offload-arch=gfx950 -O2 f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float tiny, float big) {
    h8 a, b;
    const _Float16 ta = (_Float16)tiny, tb = (_Float16)big;      // tiny = 2^-20: subnormal in f16
    for (int e = 0; e < 8; ++e) { a[e] = e == 0 ? ta : (_Float16)0.f; b[e] = e == 0 ? tb : (_Float16)0.f; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)ta; out[2] = (float)tb; }
}
int main() {
    float* d; float h[3];
    hipMalloc(&d, 12);
    const float tiny = 9.5367431640625e-07f /* 2^-20 */, big = 1024.f;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, tiny, big);
    hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    printf("f16(2^-20) = %g (subnormal kept by the conversion: %s); mfma(2^-20 [f16 subnormal] x 1024) = %g, expected %g: %s\n", h[1], h[1] == tiny ? "yes" : "NO",
           h[0], tiny * big * 2 /* lanes 0 and 32 both hold k = 0 of their groups */, (h[0] == tiny * big || h[0] == 2 * tiny * big) ? "subnormal inputs HONOURED" : "FLUSHED or unexpected");
    return 0;
}
