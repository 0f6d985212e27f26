// The pair MLP of the affinity estimator kept ON CHIP (round 6): AFE_module.forward_stacker2 + forward_final
// (AFE.py:190-213; the nets of add_final, AFE.py:331-347) for every (history object t, current object j) pair,
//     z[t][j] = relu(w5 . relu(W4 relu(s3 (W3 relu(s2 (W2 relu(U'[t] + V'[j])) + t2)) + t3) + t4) + b5),
// from the separable first layer (U' = x_hist Ua^T, V' = x_cur Vb^T + cb: deft_conv2d_nhwc) to the relu'd logit at the pair's final place
// in the affinity block, in ONE launch -- the 256 + 128 + 64 floats per pair that the four-launch chain (deft_pair_layer, two
// deft_conv2d_nhwc, deft_affinity_finish phase 1) wrote to and re-read from HBM (5.7 GB per 32-frame step at 100 x 500) never leave registers.
//
// How.  The roles of the matrix instruction's operands are SWAPPED against the conv kernels: the weights are the A operand (32 output
// channels x 16 k per v_mfma_f32_32x32x16), the activations the B operand (16 k x 32 PAIR ROWS).  The accumulator of such a product holds, in
// lane l, pair row l & 31 and output channels (r & 3) + 8 (r >> 2) + 4 (l >> 5) in register r -- and the B operand of the NEXT layer wants,
// in lane l, pair row l & 31 and eight k values of k group l >> 5: the same lane owns what it needs.  Register r = 8 s + i of a finished
// 32-channel tile becomes element i of the B operand of k step s, i.e. k step s, k group g, element i IS channel
//     c(s, g, i) = (i & 3) + 8 (2 s + (i >> 2)) + 4 g
// of the tile -- a fixed permutation of the contraction index, which the host applies to the columns of W3 / W4 when it builds the weight
// image (deft_amd/engine.py pair_mlp_image).  No LDS transposition, no shuffle between the layers: scale, shift, ReLU and the operand split
// run on the accumulator registers, the result feeds the next matrix instruction.
//
// Organisation.  A workgroup is 8 waves = 256 pair rows (32 per wave); it is PERSISTENT (grid = compute units) and walks over row tiles.
// Every wave needs every weight: they arrive as a stream of 21 chunks per tile (16 of layer 2: two k steps x 256 channels; 4 of layer 3: four
// k steps x 128; 1 of layer 4: eight k steps x 64 -- 16 x NP KB each, 48 matrix instructions per wave and chunk with two fp16 pieces) by LDS-DMA
// into a three-stage ring, lane-linear = conflict-free for the fragment reads; chunk ci + 2 is issued behind barrier ci, so a chunk has two
// steps to land; the stream runs on across tiles.  U' / V' rows are read straight from L2 (the 32 rows of a wave share one U' row most of the
// time), one chunk ahead.  Arithmetic: the library's split arithmetic (common.h: NP pieces per operand, NPROD products, fp32 accumulation);
// layer 5 (64 -> 1) is an fp32 dot product on the registers of layer 4.
#include <cstdlib>

#include "common.h"

typedef deft_f32x16 f32x16;

#define PM_FRAG 1024                                   // one A fragment: 64 lanes x 16 B (8 halves of one piece)
#define PM_NFRAG (16 * DEFT_NP)                        // fragments per chunk: (k steps x channel tiles = 16) x pieces
#define PM_CHUNK (PM_NFRAG * PM_FRAG)                  // 32 KB with two pieces, 48 KB with three
#define PM_NCHUNK 21                                   // 16 (layer 2) + 4 (layer 3) + 1 (layer 4)
#define PM_CONST 960                                   // s2 t2 [256] | s3 t3 [128] | s4 t4 [64] | w5 [64]
#define PM_NPW (2 * DEFT_NP)                           // DMA pieces per wave and chunk (8 waves)

constexpr int pairmlp_lds_bytes() { return 3 * PM_CHUNK + PM_CONST * 4; }

// the operand exists HERE (no instruction is emitted): the loads of the next chunk's U' / V' values may then take the registers it was made from
#if defined(__HIP_DEVICE_COMPILE__)
#define PM_PIN(v) asm volatile("" : "+v"(v))
#else
#define PM_PIN(v) ((void)(v))
#endif

// The operand split of THIS kernel is the plain C++ expression (compiler-generated conversions), not common.h's hand-written
// v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16 sequences: here a split result feeds a matrix instruction a few instructions later, and on the MI355X
// that adjacency is a hazard the compiler does not cover for inline asm -- v_fma_mixhi_f16 writes HALF a register (op_sel), the MFMA that reads
// the register as its B operand too soon sees the old half: results changed from run to run by ~1e-6 (the second piece is the 2^-11 residual).
// Bisected on the hardware (profiles/r6_asm_split_mfma_hazard.md: the asm split + 16 wait states, or this C++ form, are bit-stable; vmcnt(0) at
// every barrier is not).  Same bits as the asm form (tools/probe/f16_split_asm.hip); ~6 more VALU per pair of values, invisible next to the
// 24 matrix instructions they feed.
__device__ __forceinline__ void pm_split(const f32x4 v, pcx4 (&pc)[DEFT_NP]) {
#if DEFT_PIECES == 2 && !defined(DEFT_F16_SPLIT_HOOK)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e] * DEFT_ASCALE;
        const _Float16 h = (_Float16)x;
        pc[0][e] = h;
        pc[1][e] = (_Float16)(x - (float)h);
    }
#else
    deft_split(v, pc, DEFT_ASCALE);
#endif
}

// the two NP-piece B operands (k steps 0 and 1 of a 32-channel tile) out of a finished accumulator tile: v = relu(acc * sc + sh)
__device__ __forceinline__ void pm_tile_to_operands(const f32x16& acc, const float* cs, const float* ct, int ch0, int g, pcx8 (&pb)[2][DEFT_NP]) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                      // registers 4 j .. 4 j + 3 = channels ch0 + 8 j + 4 g + (0 .. 3)
        const f32x4 sc = *(const f32x4*)(cs + ch0 + 8 * j + 4 * g);
        const f32x4 sh = *(const f32x4*)(ct + ch0 + 8 * j + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = deft_relu(acc[4 * j + e] * (sc[e] * DEFT_ASCALE_INV) + sh[e]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        pcx4 c0[DEFT_NP], c1[DEFT_NP];
        pm_split(v[2 * s], c0);
        pm_split(v[2 * s + 1], c1);
#pragma unroll
        for (int q = 0; q < DEFT_NP; ++q) pb[s][q] = __builtin_shufflevector(c0[q], c1[q], 0, 1, 2, 3, 4, 5, 6, 7);
    }
}

__global__ __launch_bounds__(512, 2) void pair_mlp_kernel(DeftPairMlp p, int ntiles) {
    DEFT_DYN_LDS(char, smem);
    char* const Wst = smem;                            // [3 stages][PM_CHUNK]
    float* const cst = (float*)(smem + 3 * PM_CHUNK);
    float* const cs2 = cst, * const ct2 = cst + 256, * const cs3 = cst + 512, * const ct3 = cst + 640, * const cs4 = cst + 768, * const ct4 = cst + 832,
               * const cw5 = cst + 896;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5;

    for (int i = tid; i < PM_CONST; i += 512) {
        float v;
        if (i < 256) v = p.s2[i];
        else if (i < 512) v = p.t2[i - 256];
        else if (i < 640) v = p.s3[i - 512];
        else if (i < 768) v = p.t3[i - 640];
        else if (i < 832) v = p.s4[i - 768];
        else if (i < 896) v = p.t4[i - 832];
        else v = p.w5[i - 896];
        cst[i] = v;
    }

    const deft_rsrc_t rw = deft_make_rsrc(p.wimg);
    const deft_rsrc_t ru = deft_make_rsrc(p.U);
    const deft_rsrc_t rv = deft_make_rsrc(p.V);
    auto issue_chunk = [&](int ci, int stage) {        // the PM_NPW pieces of this wave: fragments wave, wave + 8, ...
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < PM_NPW; ++i) {
            const int f = wave + 8 * i;
            deft_buffer_load_lds_x4s(rw, Wst + stage * PM_CHUNK + f * PM_FRAG, (unsigned)(lane * 16), (unsigned)(ci * PM_CHUNK + f * PM_FRAG));
        }
        asm volatile("" ::: "memory");
    };
    const char* const fbase = Wst + lane * 16;         // this lane's 16 bytes of a fragment
    auto frag = [&](int stage, int f) -> pcx8 { return *(const pcx8*)(fbase + stage * PM_CHUNK + f * PM_FRAG); };

    if ((int)blockIdx.x < ntiles) {
        issue_chunk(0, 0);
        issue_chunk(1, 1);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- this lane's pair row: byte offsets of its U' / V' rows (DEFT_OOB beyond M: the loads return 0) ----
        const int m = tile * 256 + wave * 32 + (lane & 31);
        unsigned ro_u = DEFT_OOB, ro_v = DEFT_OOB;
        int orow = -1;
        if (m < p.M) {
            int u = m / p.Q;
            int j = m - u * p.Q;
            orow = u * (p.Q + 1) + j;
            if (p.Tper > 0) {                          // batched: (current frame c, history row t, object j), as deft_pair_layer
                const int c = u / p.Tper;
                u = p.u0 + c * p.du + (u - c * p.Tper);
                j = p.v0 + c * p.dv + j;
            }
            ro_u = (unsigned)(u * p.ldu * 4 + g * 32);
            ro_v = (unsigned)(j * p.ldu * 4 + g * 32);
        }
        f32x16 acc2[8];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
        // U' / V' values of a chunk (two k steps x 8 k of this lane's k group): [k step][U lo, U hi, V lo, V hi]
        f32x4 uv[2][4];
        auto load_uv = [&](int ci) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned kb = (unsigned)((ci * 2 + s) * 64);
                uv[s][0] = deft_buffer_load_x4(ru, ro_u + kb);
                uv[s][1] = deft_buffer_load_x4(ru, ro_u + kb + 16u);
                uv[s][2] = deft_buffer_load_x4(rv, ro_v + kb);
                uv[s][3] = deft_buffer_load_x4(rv, ro_v + kb + 16u);
            }
        };
        load_uv(0);

        // ================================ layer 2: 512 -> 256, A = relu(U' + V') generated here ================================
#pragma unroll
        for (int ci = 0; ci < 16; ++ci) {
            const int st = ci % 3;
            // the B operands of both k steps first: the registers of the loaded values then take the NEXT chunk's loads
            pcx8 pb[2][DEFT_NP];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 a0, a1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] = deft_relu(uv[s][0][e] + uv[s][2][e]);
                    a1[e] = deft_relu(uv[s][1][e] + uv[s][3][e]);
                }
                pcx4 c0[DEFT_NP], c1[DEFT_NP];
                pm_split(a0, c0);
                pm_split(a1, c1);
#pragma unroll
                for (int q = 0; q < DEFT_NP; ++q) pb[s][q] = __builtin_shufflevector(c0[q], c1[q], 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < DEFT_NP; ++q) PM_PIN(pb[s][q]);
            // chunk ci has landed: VMEM loads return in order, the U' / V' loads consumed above were issued after its pieces, and of what this
            // wave issued since only chunk ci + 1's pieces may still fly
            DEFT_PIPE_BARRIER(PM_NPW);
            if (ci + 1 < 16) load_uv(ci + 1);
            issue_chunk(ci + 2, (ci + 2) % 3);                        // (ci + 2 <= 17 < 21)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int t0 = 0; t0 < 8; t0 += 4) {                   // four channel tiles at a time: consecutive MFMAs hit different accumulators
                    pcx8 pa[4][DEFT_NP];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int q = 0; q < DEFT_NP; ++q) pa[t][q] = frag(st, (s * 8 + t0 + t) * DEFT_NP + q);
#pragma unroll
                    for (int q = 0; q < DEFT_NPROD; ++q)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc2[t0 + t] = deft_mfma_pc(pa[t][deft_qa(q)], pb[s][deft_qb(q)], acc2[t0 + t]);
                }
            }
        }
        // ================================ layer 3: 256 -> 128 (four chunks of two 32-channel k tiles) ================================
        f32x16 acc3[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[t][r] = 0.f;
#pragma unroll
        for (int c3 = 0; c3 < 4; ++c3) {
            const int ci = 16 + c3, st = ci % 3;
            DEFT_PIPE_BARRIER(PM_NPW);
            issue_chunk((ci + 2) % PM_NCHUNK, (ci + 2) % 3);          // 18, 19, 20, then chunk 0 of the NEXT tile (the stream runs on; a last
                                                                      // tile's surplus chunks land in a stage nobody reads again)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                pcx8 pb[2][DEFT_NP];
                pm_tile_to_operands(acc2[2 * c3 + kk], cs2, ct2, 32 * (2 * c3 + kk), g, pb);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    pcx8 pa[4][DEFT_NP];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int q = 0; q < DEFT_NP; ++q) pa[t][q] = frag(st, ((kk * 2 + s) * 4 + t) * DEFT_NP + q);
#pragma unroll
                    for (int q = 0; q < DEFT_NPROD; ++q)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc3[t] = deft_mfma_pc(pa[t][deft_qa(q)], pb[s][deft_qb(q)], acc3[t]);
                }
            }
        }
        // ================================ layer 4: 128 -> 64 (one chunk: four k tiles) ================================
        f32x16 acc4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc4[t][r] = 0.f;
        {
            const int ci = 20, st = ci % 3;
            DEFT_PIPE_BARRIER(PM_NPW);
            issue_chunk((ci + 2) % PM_NCHUNK, (ci + 2) % 3);          // chunk 1 of the next tile
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                pcx8 pb[2][DEFT_NP];
                pm_tile_to_operands(acc3[kt], cs3, ct3, 32 * kt, g, pb);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    pcx8 pa[2][DEFT_NP];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < DEFT_NP; ++q) pa[t][q] = frag(st, ((kt * 2 + s) * 2 + t) * DEFT_NP + q);
#pragma unroll
                    for (int q = 0; q < DEFT_NPROD; ++q)
#pragma unroll
                        for (int t = 0; t < 2; ++t) acc4[t] = deft_mfma_pc(pa[t][deft_qa(q)], pb[s][deft_qb(q)], acc4[t]);
                }
            }
        }
        // ================================ layer 5: 64 -> 1 on the registers, + b5, ReLU ================================
        float z = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = 32 * t + 8 * j + 4 * g;
                const f32x4 sc = *(const f32x4*)(cs4 + ch), sh = *(const f32x4*)(ct4 + ch), w5 = *(const f32x4*)(cw5 + ch);
#pragma unroll
                for (int e = 0; e < 4; ++e) z = fmaf(deft_relu(acc4[t][4 * j + e] * (sc[e] * DEFT_ASCALE_INV) + sh[e]), w5[e], z);
            }
        z += __shfl_xor(z, 32);                                       // the other k group's 32 channels
        if (g == 0 && orow >= 0) p.out[orow] = deft_relu(z + p.b5);
    }
    DEFT_WAIT_VM(0);                                                  // (the surplus chunks of the last tile: no DMA may outlive the workgroup)
}

extern "C" int deft_pair_mlp(const DeftPairMlp* d, void* stream) {
    DEFT_CHECK(d && d->U && d->V && d->wimg && d->s2 && d->t2 && d->s3 && d->t3 && d->s4 && d->t4 && d->w5 && d->out, -110, "deft_pair_mlp: null pointer");
    DEFT_CHECK(d->M > 0 && d->Q > 0 && d->ldu >= 512 && (d->ldu & 3) == 0 && ((((size_t)d->U | (size_t)d->V | (size_t)d->wimg)) & 15) == 0, -111,
               "deft_pair_mlp: M=%d Q=%d ldu=%d (U' / V' rows of 512 floats, 16-byte aligned)", d->M, d->Q, d->ldu);
    DEFT_CHECK(d->Tper >= 0 && (d->Tper == 0 || (d->M % (d->Tper * d->Q)) == 0), -112, "deft_pair_mlp: M=%d is not a whole number of (Tper=%d x Q=%d) frames", d->M,
               d->Tper, d->Q);
    DEFT_CHECK((long long)(d->M / d->Q + 1) * (d->Q + 1) < (1ll << 31), -113, "deft_pair_mlp: output index overflows");
    constexpr int lds = pairmlp_lds_bytes();
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)pair_mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        DEFT_CHECK(e == hipSuccess, -101, "deft_pair_mlp: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
        done = true;
    }
    const int ntiles = deft_cdiv(d->M, 256);
    static const int cap = [] { const char* e = getenv("DEFT_PAIR_MLP_GRID"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    const int grid = ntiles < cap ? ntiles : cap;                     // persistent: one workgroup per compute unit (the variable: tests of the tile walk)
    hipLaunchKernelGGL(pair_mlp_kernel, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, *d, ntiles);
    DEFT_CHECK_LAUNCH("pair_mlp");
    return 0;
}

extern "C" int deft_pair_mlp_image_bytes(void) { return PM_NCHUNK * PM_CHUNK; }
