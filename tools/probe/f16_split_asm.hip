#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& h, unsigned& m) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(m) : "v"(h), "v"(x0));
    asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(m) : "v"(h), "v"(x1));
#endif
}
__device__ __forceinline__ void split2_pair_scaled(float x0, float x1, float sc, unsigned& h, unsigned& m) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "s"(sc));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(m) : "v"(x0), "s"(sc), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(m) : "v"(x1), "s"(sc), "v"(h));
#endif
}
__global__ void k(const f4* x, unsigned* o, float sc) {
    f4 v = x[threadIdx.x];
    unsigned h0, m0, h1, m1;
    split2_pair(v[0], v[1], h0, m0);
    split2_pair_scaled(v[2], v[3], __builtin_amdgcn_readfirstlane(sc), h1, m1);
    o[threadIdx.x*4+0] = h0; o[threadIdx.x*4+1] = m0; o[threadIdx.x*4+2] = h1; o[threadIdx.x*4+3] = m1;
}
int main() {
    const int N = 256;
    float hx[N*4]; unsigned ho[N*4];
    for (int i = 0; i < N*4; ++i) hx[i] = (i%7==0 ? 1e-6f : 1.f) * (float)(i*0.37123f - 40.f) * (i%5==0 ? 0.001f : 1.f);
    f4* dx; unsigned* dq;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dq, sizeof(ho));
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(N), 0, 0, dx, dq, 16.f);
    hipMemcpy(ho, dq, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < N; ++t) {
        for (int half = 0; half < 2; ++half) {
            const float sc = half ? 16.f : 1.f;
            for (int e = 0; e < 2; ++e) {
                const float x = hx[t*4 + half*2 + e] * sc;
                const _Float16 h = (_Float16)x; const _Float16 m = (_Float16)(x - (float)h);
                unsigned short eh, em; __builtin_memcpy(&eh, &h, 2); __builtin_memcpy(&em, &m, 2);
                const unsigned gh = (ho[t*4 + half*2] >> (16*e)) & 0xffff, gm = (ho[t*4 + half*2 + 1] >> (16*e)) & 0xffff;
                if (gh != eh || gm != em) { if (bad < 5) printf("mismatch t=%d half=%d e=%d x=%g got %04x %04x want %04x %04x\n", t, half, e, x, gh, gm, eh, em); ++bad; }
            }
        }
    }
    printf("f16 split asm check: %d mismatches of %d\n", bad, N*4);
    return bad != 0;
}
