// Probe: how does v_mfma_f32_32x32x16_f16 round its fp32 accumulation on gfx950?  (Round 6: the embeddings sit 9e-4 from the oracle with BOTH
// split arithmetics while the oracle itself is 1e-4 from fp64 -- is the matrix core's accumulate round-to-nearest or truncation, once per
// instruction or once per product?)
//   1. one product added to a large accumulator: c + x, x a fraction of ulp(c)            -> nearest or toward zero?
//   2. sixteen products of 0.3 ulp each                                                  -> summed exactly and rounded once, or one by one?
//   3. a chain of 144 instructions (K = 2304) on random operands, against fp64 and against two host models of the chain
//      (exact 16-term dot per instruction, then one fp32 add rounded to nearest / truncated toward zero).
// hipcc --offload-arch=gfx950 -O2 tools/probe/mfma_rounding.hip -o /tmp/mfma_rounding && /tmp/mfma_rounding
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void chain(const h8* A, const h8* B, const float* C, float* D, int steps) {     // one wave; A/B: [steps][64 lanes] fragments
    const int l = threadIdx.x;
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = C[l * 16 + i];
    for (int s = 0; s < steps; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[s * 64 + l], B[s * 64 + l], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[l * 16 + i] = acc[i];
}

static int drow(int i, int l) { return (i & 3) + 8 * (i >> 2) + 4 * (l >> 5); }
static float rtz_add(float a, double b) {          // a + b in double (exact enough: 53 bits), truncated toward zero to fp32
    const double s = (double)a + b;
    float r = (float)s;
    if (std::fabs((double)r) > std::fabs(s)) r = std::nextafterf(r, 0.0f);
    return r;
}

struct Run {
    std::vector<_Float16> a, b;      // [steps][32][16], [steps][16][32]
    std::vector<float> c, d;         // [32][32]
    int steps;
};
static void run(Run& R) {
    const int S = R.steps;
    std::vector<h8> fa(S * 64), fb(S * 64);
    std::vector<float> fc(64 * 16), fd(64 * 16);
    for (int s = 0; s < S; ++s)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                fa[s * 64 + l][e] = R.a[(s * 32 + (l & 31)) * 16 + 8 * (l >> 5) + e];
                fb[s * 64 + l][e] = R.b[(s * 16 + 8 * (l >> 5) + e) * 32 + (l & 31)];
            }
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) fc[l * 16 + i] = R.c[drow(i, l) * 32 + (l & 31)];
    h8 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, S * 64 * sizeof(h8)); hipMalloc(&dB, S * 64 * sizeof(h8)); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    hipMemcpy(dA, fa.data(), S * 64 * sizeof(h8), hipMemcpyHostToDevice); hipMemcpy(dB, fb.data(), S * 64 * sizeof(h8), hipMemcpyHostToDevice);
    hipMemcpy(dC, fc.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, S);
    hipMemcpy(fd.data(), dD, 4096, hipMemcpyDeviceToHost);
    R.d.assign(1024, 0.f);
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 16; ++i) R.d[drow(i, l) * 32 + (l & 31)] = fd[l * 16 + i];
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
}

int main() {
    {   // 1. one product
        Run R; R.steps = 1; R.a.assign(32 * 16, (_Float16)0); R.b.assign(16 * 32, (_Float16)0); R.c.assign(1024, 1024.0f);
        const float ms[8] = {1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.5f, 3.f, 3.5f};
        for (int r = 0; r < 32; ++r) R.a[r * 16] = (_Float16)(r & 1 ? -1.f : 1.f);
        for (int c = 0; c < 32; ++c) R.b[c] = (_Float16)(ms[c & 7] * 6.103515625e-05f);      // m * 2^-14; ulp(1024) = 2^-13
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) R.c[r * 32 + c] = r & 1 ? -1024.f : 1024.f;
        run(R);
        printf("1. c = +-1024 (ulp 2^-13), one product x = f * ulp:  result - c in ulps (nearest-even would give 0 1 1 1 1 1 2 2; truncation 0 0 0 0 1 1 1 1)\n");
        for (int r = 0; r < 2; ++r) {
            printf("   sign %c:", r ? '-' : '+');
            for (int c = 0; c < 8; ++c) printf("  f=%.3f -> %+g", ms[c] * 0.5, (R.d[r * 32 + c] - R.c[r * 32 + c]) * 8192.0);
            printf("\n");
        }
    }
    {   // 2. sixteen products of 0.3 ulp
        Run R; R.steps = 1; R.a.assign(32 * 16, (_Float16)1.f); R.b.assign(16 * 32, (_Float16)0); R.c.assign(1024, 16384.0f);     // ulp 2^-9
        const _Float16 x = (_Float16)(0.3f * 0.001953125f);
        for (int k = 0; k < 16; ++k) for (int c = 0; c < 32; ++c) R.b[k * 32 + c] = c < 16 ? x : (k < (c - 15) ? x : (_Float16)0);
        run(R);
        printf("2. c = 16384 (ulp 2^-9), n products of %.4f ulp each: result - c in ulps (exact sum in brackets)\n  ", (double)x * 512.0);
        for (int c = 15; c < 32; ++c) { const int n = c < 16 ? 16 : c - 15; printf(" n=%d: %g [%.2f]", n, (R.d[c] - 16384.0) * 512.0, n * (double)x * 512.0); }
        printf("\n");
    }
    {   // 3. the chain
        const int S = 144;
        Run R; R.steps = S; R.a.resize(S * 32 * 16); R.b.resize(S * 16 * 32); R.c.assign(1024, 0.f);
        srand(7);
        auto rnd = [] { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return u - 6.0; };
        for (auto& v : R.a) v = (_Float16)(float)rnd();
        for (auto& v : R.b) v = (_Float16)(float)(rnd() + 0.25);              // a small mean: sums that grow, as after a ReLU
        for (auto& v : R.a) v = (_Float16)(float)std::fabs((double)v);
        run(R);
        double se_dev = 0, se_rne = 0, se_rtz = 0, b_dev = 0, b_rne = 0, b_rtz = 0; int eq_rne = 0, eq_rtz = 0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double exact = 0; float rne = 0.f, rtz = 0.f;
                for (int s = 0; s < S; ++s) {
                    double dot = 0;
                    for (int k = 0; k < 16; ++k) dot += (double)R.a[(s * 32 + r) * 16 + k] * (double)R.b[(s * 16 + k) * 32 + c];
                    exact += dot;
                    rne = (float)((double)rne + dot);
                    rtz = rtz_add(rtz, dot);
                }
                const double ulp = std::ldexp(1.0, std::ilogb(exact) - 23);
                const double ed = (R.d[r * 32 + c] - exact) / ulp, en = (rne - exact) / ulp, ez = (rtz - exact) / ulp;
                se_dev += ed * ed; se_rne += en * en; se_rtz += ez * ez; b_dev += ed; b_rne += en; b_rtz += ez;
                eq_rne += R.d[r * 32 + c] == rne; eq_rtz += R.d[r * 32 + c] == rtz;
            }
        printf("3. chain of %d instructions (K = %d), 1024 outputs, error vs the exact sum in ulps of the result:\n", S, S * 16);
        printf("   device              mean %+8.2f  rms %8.2f\n", b_dev / 1024, std::sqrt(se_dev / 1024));
        printf("   host model nearest  mean %+8.2f  rms %8.2f   bit-equal to the device: %d / 1024\n", b_rne / 1024, std::sqrt(se_rne / 1024), eq_rne);
        printf("   host model truncate mean %+8.2f  rms %8.2f   bit-equal to the device: %d / 1024\n", b_rtz / 1024, std::sqrt(se_rtz / 1024), eq_rtz);
    }
    return 0;
}
