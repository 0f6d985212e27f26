// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS to LDS for out-of-range lanes (offset >= num_records)?
// hipcc --offload-arch=gfx950 -O2 tools/probe/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* x, float* y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int i = threadIdx.x; i < 1024; i += 256) smem[i] = 7.0f;      // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x7FFFFFFF, 0x00020000);
    unsigned off = threadIdx.x * 16;
    if (threadIdx.x & 1) off = 0x80000000u;
    const int wave = threadIdx.x >> 6;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + wave * 256), 16, off, 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) y[i] = smem[i];
}
int main() {
    float *x, *y, hx[1024], hy[1024];
    for (int i = 0; i < 1024; ++i) hx[i] = 100.f + i;
    hipMalloc(&x, 4096); hipMalloc(&y, 4096);
    hipMemcpy(x, hx, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, x, y);
    hipMemcpy(hy, y, 4096, hipMemcpyDeviceToHost);
    int bad_valid = 0, zero_oob = 0, stale_oob = 0;
    for (int t = 0; t < 256; ++t)
        for (int j = 0; j < 4; ++j) {
            const float v = hy[t * 4 + j];
            if (t & 1) { zero_oob += v == 0.f; stale_oob += v == 7.f; }
            else bad_valid += v != hx[t * 4 + j];
        }
    printf("valid lanes wrong: %d   OOB lanes: zero %d  stale-sentinel %d (of 512)\n", bad_valid, zero_oob, stale_oob);
    return 0;
}
