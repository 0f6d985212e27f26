// Probe for the "fp32 through the bf16 matrix cores" lever: C = A.B^T with every fp32 operand split into
// three bf16 pieces (hi, mid, lo) and six v_mfma_f32_32x32x16_bf16 products per fp32 product
// (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid), fp32 accumulation.
//   part 1: one 128x128xK tile against an fp64 reference -- operand layout check + the numerical error next to a
//           plain fp32 chain;
//   part 2: the K loop of a 128x128x32-chunk tile (global loads -> split -> LDS planes -> fragments -> 48 MFMAs per wave
//           per chunk) over many workgroups: fp32-EQUIVALENT TFLOP/s (2*M*N*K / t), to set next to the 128-130 TFLOP/s
//           asymptote of the v_mfma_f32_32x32x2_f32 loop.
// hipcc --offload-arch=gfx950 -O3 tools/probe/split_bf16_loop.hip -o tools/probe/split_bf16_loop.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define LDB 40                      // bf16 elements per LDS row (32 + 8 pad: 80-byte stride)
#define PLANE (128 * LDB)

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

// A [128][K], B [128][K] fp32 (row-major, K % 32 == 0); one workgroup; C [128][128]
template <bool STORE>
__global__ __launch_bounds__(256) void tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Cout,
                                                    int K, int chunks, int wrap) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
    __bf16* As = smem;                 // [3 planes][128][LDB]
    __bf16* Bs = smem + 3 * PLANE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int g = tid & 7, rbase = tid >> 3, frow = lane & 31, fkg = (lane >> 5) * 8;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* Ab = A + (size_t)(blockIdx.x % wrap) * 128 * K;
    const float* Bb = B + (size_t)(blockIdx.x % wrap) * 128 * K;
    f32x4 va[4], vb[4];
    auto issue = [&](int kt) {
        const int k0 = (kt * 32) % K;
        for (int i = 0; i < 4; ++i) {
            va[i] = *(const f32x4*)(Ab + (size_t)(rbase + 32 * i) * K + k0 + g * 4);
            vb[i] = *(const f32x4*)(Bb + (size_t)(rbase + 32 * i) * K + k0 + g * 4);
        }
    };
    auto store = [&]() {
        for (int i = 0; i < 4; ++i) {
            bf16x4 h, m, l;
            for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(va[i][e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
            __bf16* p = As + (rbase + 32 * i) * LDB + g * 4;
            *(bf16x4*)p = h; *(bf16x4*)(p + PLANE) = m; *(bf16x4*)(p + 2 * PLANE) = l;
            for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(vb[i][e], a, b, c); h[e] = a; m[e] = b; l[e] = c; }
            p = Bs + (rbase + 32 * i) * LDB + g * 4;
            *(bf16x4*)p = h; *(bf16x4*)(p + PLANE) = m; *(bf16x4*)(p + 2 * PLANE) = l;
        }
    };
    issue(0);
    for (int kt = 0; kt < chunks; ++kt) {
        store();
        __syncthreads();
        if (kt + 1 < chunks) issue(kt + 1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {                       // two K=16 halves of the chunk
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *(const bf16x8*)(As + p * PLANE + ((wm * 2 + i) * 32 + frow) * LDB + kh * 16 + fkg);
                    b[i][p] = *(const bf16x8*)(Bs + p * PLANE + ((wn * 2 + i) * 32 + frow) * LDB + kh * 16 + fkg);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);    // mid.mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);    // lo.hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);    // hi.lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);    // mid.hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);    // hi.mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);    // hi.hi
                    acc[i][j] = c;
                }
        }
        __syncthreads();
    }
    if (STORE) {
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) {
            const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int col = (wn * 2 + j) * 32 + (lane & 31);
            Cout[row * 128 + col] = acc[i][j][r];
        }
    } else {
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 12345.678f) Cout[tid] = s;
    }
}

int main(int argc, char** argv) {
    const int K = 512, lds = 6 * PLANE * 2;
    std::vector<float> hA(128 * K), hB(128 * K), hC(128 * 128);
    srand(1);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f + 0.05f * ((&v - hB.data()) % 7);    // asymmetric
    float *dA, *dB, *dC;
    const int wrap = 64;
    hipMalloc(&dA, (size_t)wrap * 128 * K * 4); hipMalloc(&dB, (size_t)wrap * 128 * K * 4); hipMalloc(&dC, 128 * 128 * 4);
    for (int w = 0; w < wrap; ++w) {
        hipMemcpy(dA + (size_t)w * 128 * K, hA.data(), 128 * K * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB + (size_t)w * 128 * K, hB.data(), 128 * K * 4, hipMemcpyHostToDevice);
    }
    hipFuncSetAttribute((const void*)tile_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)tile_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(tile_kernel<true>, dim3(1), dim3(256), lds, 0, dA, dB, dC, K, K / 32, wrap);
    hipMemcpy(hC.data(), dC, 128 * 128 * 4, hipMemcpyDeviceToHost);
    double e6 = 0, e32 = 0, scale = 0;
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
            double ref = 0; float ch = 0.f;
            for (int k = 0; k < K; ++k) { ref += (double)hA[i * K + k] * (double)hB[j * K + k]; ch = fmaf(hA[i * K + k], hB[j * K + k], ch); }
            e6 = fmax(e6, fabs(hC[i * 128 + j] - ref)); e32 = fmax(e32, fabs((double)ch - ref)); scale = fmax(scale, fabs(ref));
        }
    printf("part 1  K=%d  max|C| %.3f   6-term bf16 max abs err %.3e   fp32 fmaf chain max abs err %.3e\n", K, scale, e6, e32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nwg : {256 * 2, 256 * 8, 256 * 32}) {
        const int chunks = 288;
        hipLaunchKernelGGL(tile_kernel<false>, dim3(nwg), dim3(256), lds, 0, dA, dB, dC, K, chunks, wrap);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(tile_kernel<false>, dim3(nwg), dim3(256), lds, 0, dA, dB, dC, K, chunks, wrap);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("part 2  %6d workgroups x %d chunks: %.3f ms  %.1f fp32-equivalent TFLOP/s\n", nwg, chunks, ms,
               2.0 * 128 * 128 * 32 * chunks * nwg / (ms * 1e-3) / 1e12);
    }
    return 0;
}
