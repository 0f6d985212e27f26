// FP32-MFMA implicit GEMM for gfx950 with three A-operand loaders:
//   MODE_CONV : NHWC im2col (any KHxKW / stride / pad; concat inputs via ld)
//   MODE_DCN  : DCNv2 modulated deformable gather (bilinear, zero pad, * sigmoid(mask))
//   MODE_PAIR : affinity pair grid, A[(i,j)][k] = relu(U'[i][k] + V'[j][k])
// and one epilogue: y = acc*scale[co] + shift[co] (+ residual) (ReLU).
//
// Tile: BM x BN outputs per 256-thread workgroup (4 wavefronts of 64), K consumed
// in chunks of 32.  The A/B chunks are staged global -> VGPR -> LDS ([rows][32+4]
// floats, 144-B row stride: conflict-free for ds_read_b128), with the next
// chunk's global loads in flight while the current chunk's MFMAs issue.
// v_mfma_f32_32x32x2_f32 contracts two k per instruction; the k index a lane
// half h feeds at step kk is k = 16*h + kk, so every lane reads its 16 k-values
// as four ds_read_b128 (the k permutation is the same for A and B, so the
// contraction is unchanged).  Exact fp32: the parity bar (bit-exact top-k,
// 1e-3 on boxes) rules out bf16/fp8 MFMA; peak is 157.3 TFLOP/s.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_CONV = 0, MODE_DCN = 1, MODE_PAIR = 2 };

#define LDS_STRIDE 36
#define ROW_INVALID (-(1 << 28))

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void igemm_kernel(DeftGemmDesc p, int mtiles, int ntiles) {
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int GA = BM / 32;  // float4 groups per thread, A tile
    constexpr int GB = BN / 32;  // float4 groups per thread, B tile
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) float As[BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_STRIDE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, bijective workgroup remap: block b runs on XCD b%8 (observed), so give
    // every XCD one contiguous run of tiles -- neighbouring n-tiles of an m-tile then
    // share their A rows in that XCD's private L2.
    int bid = blockIdx.x;
    {
        const int nwg = mtiles * ntiles;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / ntiles, nt = bid - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int g = tid & 7;       // which float4 of the 32-wide k chunk this thread stages
    const int rbase = tid >> 3;  // 0..31: staged rows are rbase + 32*i

    // per-row loader state, fixed for the whole K loop
    int r0[GA], r1[GA], r2[GA];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int m = m0 + rbase + 32 * i;
        r0[i] = ROW_INVALID; r1[i] = 0; r2[i] = 0;
        if (m < p.M) {
            if (MODE == MODE_PAIR) {
                int u = m / p.Q;
                int j = m - u * p.Q;
                if (p.Tper > 0) {       // batched: (current frame c, history row t, object j)
                    const int c = u / p.Tper;
                    u = p.u0 + c * p.du + (u - c * p.Tper);
                    j = p.v0 + c * p.dv + j;
                }
                r0[i] = u * p.ldx;
                r1[i] = j * p.ldx;
            } else {
                const int ohw = p.OH * p.OW;
                const int n = m / ohw;
                const int rem = m - n * ohw;
                const int oy = rem / p.OW;
                const int ox = rem - oy * p.OW;
                if (MODE == MODE_CONV) {
                    r0[i] = oy * p.stride - p.pad;
                    r1[i] = ox * p.stride - p.pad;
                } else {
                    r0[i] = oy;
                    r1[i] = ox;
                }
                r2[i] = n * p.H * p.W;
            }
        }
    }

    auto load_a = [&](int kc, float4* va) {
        const int kflat = kc + g * 4;
        if (MODE == MODE_CONV) {
            int c = kflat, r = 0, s = 0;
            if (p.KH * p.KW > 1) {
                c = kflat & (p.Cin - 1);
                const int tap = kflat >> p.cin_log2;
                r = tap / p.KW;
                s = tap - r * p.KW;
            }
            const bool kok = kflat < p.Ktot;
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const int iy = r0[i] + r, ix = r1[i] + s;
                const bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                // unconditional load from a clamped (always valid) address + select: no branch per load
                const int pix = ok ? r2[i] + iy * p.W + ix : 0;
                float4 v = *(const float4*)(p.x + (size_t)pix * p.ldx + (ok ? c : 0));
                if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                va[i] = v;
            }
        } else if (MODE == MODE_DCN) {
            // upstream DCNv2 modulated_deformable_im2col + dmcn_im2col_bilinear semantics
            const int c = kflat & (p.Cin - 1);
            const int tap = kflat >> p.cin_log2;  // 0..8 (Kpad == Ktot since Cin % 32 == 0)
            const int r = tap / 3, s = tap - 3 * r;
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r0[i] != ROW_INVALID) {
                    const float* om = p.x2 + (size_t)(m0 + rbase + 32 * i) * p.ldom;
                    const float dy = om[2 * tap], dx = om[2 * tap + 1], ml = om[18 + tap];
                    const float h_im = (float)(r0[i] - 1 + r) + dy;
                    const float w_im = (float)(r1[i] - 1 + s) + dx;
                    if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                        const float hl = floorf(h_im), wl = floorf(w_im);
                        const float lh = h_im - hl, lw = w_im - wl;
                        const float hh = 1.f - lh, hw = 1.f - lw;
                        const int h_low = (int)hl, w_low = (int)wl;
                        const int h_high = h_low + 1, w_high = w_low + 1;
                        const float mask = 1.f / (1.f + expf(-ml));
                        const float* base = p.x + (size_t)r2[i] * p.ldx + c;
                        float4 v1 = v, v2 = v, v3 = v, v4 = v;
                        if (h_low >= 0 && w_low >= 0)
                            v1 = *(const float4*)(base + (size_t)(h_low * p.W + w_low) * p.ldx);
                        if (h_low >= 0 && w_high <= p.W - 1)
                            v2 = *(const float4*)(base + (size_t)(h_low * p.W + w_high) * p.ldx);
                        if (h_high <= p.H - 1 && w_low >= 0)
                            v3 = *(const float4*)(base + (size_t)(h_high * p.W + w_low) * p.ldx);
                        if (h_high <= p.H - 1 && w_high <= p.W - 1)
                            v4 = *(const float4*)(base + (size_t)(h_high * p.W + w_high) * p.ldx);
                        const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                        v.x = (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x) * mask;
                        v.y = (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y) * mask;
                        v.z = (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z) * mask;
                        v.w = (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w) * mask;
                    }
                }
                va[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r0[i] != ROW_INVALID) {
                    const float4 u = *(const float4*)(p.x + r0[i] + kflat);
                    const float4 t = *(const float4*)(p.x2 + r1[i] + kflat);
                    v.x = fmaxf(u.x + t.x, 0.f);
                    v.y = fmaxf(u.y + t.y, 0.f);
                    v.z = fmaxf(u.z + t.z, 0.f);
                    v.w = fmaxf(u.w + t.w, 0.f);
                }
                va[i] = v;
            }
        }
    };
    auto load_b = [&](int kc, float4* vb) {
#pragma unroll
        for (int i = 0; i < GB; ++i)
            vb[i] = *(const float4*)(p.w + (size_t)(n0 + rbase + 32 * i) * p.Kpad + kc + g * 4);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 va[GA], vb[GB];
    const int nk = p.Kpad >> 5;
    load_a(0, va);
    load_b(0, vb);

    const int frow = lane & 31;        // row of the 32-row MFMA tile this lane feeds
    const int fk = (lane >> 5) * 16;   // first of this lane's 16 k-values in the chunk

    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // all waves finished reading the previous chunk
#pragma unroll
        for (int i = 0; i < GA; ++i) *(float4*)&As[(rbase + 32 * i) * LDS_STRIDE + g * 4] = va[i];
#pragma unroll
        for (int i = 0; i < GB; ++i) *(float4*)&Bs[(rbase + 32 * i) * LDS_STRIDE + g * 4] = vb[i];
        __syncthreads();
        if (kt + 1 < nk) {  // next chunk's global loads stay in flight under the MFMAs
            load_a((kt + 1) << 5, va);
            load_b((kt + 1) << 5, vb);
        }
        float a[TM][16], b[TN][16];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float4* ap = (const float4*)&As[((wm * TM + i) * 32 + frow) * LDS_STRIDE + fk];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = ap[q];
                a[i][4 * q + 0] = t.x; a[i][4 * q + 1] = t.y; a[i][4 * q + 2] = t.z; a[i][4 * q + 3] = t.w;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float4* bp = (const float4*)&Bs[((wn * TN + j) * 32 + frow) * LDS_STRIDE + fk];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = bp[q];
                b[j][4 * q + 0] = t.x; b[j][4 * q + 1] = t.y; b[j][4 * q + 2] = t.z; b[j][4 * q + 3] = t.w;
            }
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
    }

    // epilogue: D reg r of lane l is (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31);
    // the 32 lanes of a half-wave write 32 consecutive channels of one pixel (128 B).
    // Residual loads are hoisted out of the per-element path (one uniform branch, 16
    // loads in flight) -- a per-element `if (res)` makes hipcc wait vmcnt(0) per load.
    const bool has_res = p.res != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int co = n0 + (wn * TN + j) * 32 + (lane & 31);
        const bool cok = co < p.Cout;
        const int coc = cok ? co : p.Cout - 1;
        const float sc = p.scale ? p.scale[coc] : 1.f;
        const float sh = p.shift ? p.shift[coc] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int q = 0; q < 4; ++q) {       // 4 rows at a time keeps the epilogue's VGPR footprint small
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (has_res) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        int m = mb + t + 8 * q;
                        m = m < p.M ? m : p.M - 1;
                        rv[t] = p.res[(size_t)m * p.ldr + coc];
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = mb + t + 8 * q;
                    float v = acc[i][j][4 * q + t] * sc + sh + rv[t];
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (cok && m < p.M) p.y[(size_t)m * p.ldy + co] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int MODE>
static int launch_igemm(const DeftGemmDesc& d, hipStream_t s) {
    const int mtiles = deft_cdiv(d.M, BM), ntiles = deft_cdiv(d.Cout, BN);
    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, MODE>), dim3(mtiles * ntiles), dim3(256), 0, s, d,
                       mtiles, ntiles);
    DEFT_CHECK_LAUNCH("igemm");
    return 0;
}

static int check_common(const DeftGemmDesc* d, const char* who) {
    DEFT_CHECK(d != nullptr, -1, "%s: null descriptor", who);
    DEFT_CHECK(d->x && d->w && d->y, -2, "%s: null x/w/y pointer", who);
    DEFT_CHECK(d->M > 0 && d->Cout > 0, -3, "%s: empty problem M=%d Cout=%d", who, d->M, d->Cout);
    DEFT_CHECK(d->Kpad > 0 && (d->Kpad & 31) == 0 && d->Ktot <= d->Kpad, -4, "%s: Kpad=%d must be a multiple of 32 >= Ktot=%d", who, d->Kpad, d->Ktot);
    DEFT_CHECK((d->ldx & 3) == 0 && (((size_t)d->x) & 15) == 0 && (((size_t)d->w) & 15) == 0, -5, "%s: x/w must be 16-byte aligned, ldx %% 4 == 0", who);
    DEFT_CHECK(d->ldy >= d->Cout, -6, "%s: ldy=%d < Cout=%d", who, d->ldy, d->Cout);
    DEFT_CHECK(!d->res || d->ldr >= d->Cout, -7, "%s: ldr=%d < Cout=%d", who, d->ldr, d->Cout);
    return 0;
}

// blocks needed before a bigger tile is worth it: 2 workgroups on each of 256 CUs
#define FILL_BLOCKS 512

extern "C" int deft_conv2d_nhwc(const DeftGemmDesc* d, void* stream) {
    if (int e = check_common(d, "deft_conv2d_nhwc")) return e;
    DEFT_CHECK((d->Cin & 3) == 0, -10, "deft_conv2d_nhwc: Cin=%d must be a multiple of 4 (pad channels)", d->Cin);
    DEFT_CHECK(d->Ktot == d->KH * d->KW * d->Cin, -11, "deft_conv2d_nhwc: Ktot mismatch");
    DEFT_CHECK(d->KH * d->KW == 1 || ((d->Cin & (d->Cin - 1)) == 0 && (1 << d->cin_log2) == d->Cin), -12,
               "deft_conv2d_nhwc: Cin=%d must be a power of two for KHxKW>1", d->Cin);
    DEFT_CHECK(d->M == d->N * d->OH * d->OW, -13, "deft_conv2d_nhwc: M != N*OH*OW");
    DEFT_CHECK(d->ldx >= d->Cin, -14, "deft_conv2d_nhwc: ldx < Cin");
    hipStream_t s = (hipStream_t)stream;
    int bm = d->tile >> 16, bn = d->tile & 0xffff;
    if (d->tile == 0) {
        const long long m128 = deft_cdiv(d->M, 128);
        if (d->Cout <= 32) { bm = 128; bn = 32; }
        else if (d->Cout > 64 && m128 * deft_cdiv(d->Cout, 128) >= FILL_BLOCKS) { bm = 128; bn = 128; }
        else if (m128 * deft_cdiv(d->Cout, 64) >= FILL_BLOCKS) { bm = 128; bn = 64; }
        else { bm = 64; bn = 64; }
    }
    if (bm == 128 && bn == 128) return launch_igemm<128, 128, 2, 2, MODE_CONV>(*d, s);
    if (bm == 128 && bn == 64) return launch_igemm<128, 64, 2, 2, MODE_CONV>(*d, s);
    if (bm == 128 && bn == 32) return launch_igemm<128, 32, 4, 1, MODE_CONV>(*d, s);
    if (bm == 64 && bn == 64) return launch_igemm<64, 64, 2, 2, MODE_CONV>(*d, s);
    if (bm == 64 && bn == 128) return launch_igemm<64, 128, 2, 2, MODE_CONV>(*d, s);
    DEFT_CHECK(false, -15, "deft_conv2d_nhwc: unsupported tile %dx%d", bm, bn);
    return -15;
}

extern "C" int deft_dcn_v2_nhwc(const DeftGemmDesc* d, void* stream) {
    if (int e = check_common(d, "deft_dcn_v2_nhwc")) return e;
    DEFT_CHECK(d->x2 != nullptr && d->ldom >= 27, -20, "deft_dcn_v2_nhwc: offset/mask map missing or ldom < 27");
    DEFT_CHECK(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, -21, "deft_dcn_v2_nhwc: only 3x3/s1/p1 (dla.py:652-660)");
    DEFT_CHECK((d->Cin & 31) == 0 && (d->Cin & (d->Cin - 1)) == 0 && (1 << d->cin_log2) == d->Cin, -22,
               "deft_dcn_v2_nhwc: Cin=%d must be a power of two >= 32", d->Cin);
    DEFT_CHECK(d->Ktot == 9 * d->Cin && d->Kpad == d->Ktot, -23, "deft_dcn_v2_nhwc: Ktot/Kpad mismatch");
    DEFT_CHECK(d->OH == d->H && d->OW == d->W && d->M == d->N * d->H * d->W, -24, "deft_dcn_v2_nhwc: geometry mismatch");
    hipStream_t s = (hipStream_t)stream;
    int bm = d->tile >> 16, bn = d->tile & 0xffff;
    if (d->tile == 0) {
        bm = 64;
        bn = (d->Cout > 64 && (long long)deft_cdiv(d->M, 64) * deft_cdiv(d->Cout, 128) >= FILL_BLOCKS) ? 128 : 64;
    }
    if (bm == 64 && bn == 64) return launch_igemm<64, 64, 2, 2, MODE_DCN>(*d, s);
    if (bm == 64 && bn == 128) return launch_igemm<64, 128, 2, 2, MODE_DCN>(*d, s);
    DEFT_CHECK(false, -25, "deft_dcn_v2_nhwc: unsupported tile %dx%d", bm, bn);
    return -25;
}

extern "C" int deft_pair_layer(const DeftGemmDesc* d, void* stream) {
    if (int e = check_common(d, "deft_pair_layer")) return e;
    DEFT_CHECK(d->x2 != nullptr && d->Q > 0 && d->M % d->Q == 0, -30, "deft_pair_layer: need V' and M %% Q == 0");
    DEFT_CHECK(d->Kpad == d->Ktot && d->ldx >= d->Ktot && (((size_t)d->x2) & 15) == 0, -31, "deft_pair_layer: K must be a multiple of 32 and <= ldx");
    hipStream_t s = (hipStream_t)stream;
    int bm = d->tile >> 16, bn = d->tile & 0xffff;
    if (d->tile == 0) {
        bm = 128;
        bn = (d->Cout > 64 && (long long)deft_cdiv(d->M, 128) * deft_cdiv(d->Cout, 128) >= FILL_BLOCKS) ? 128 : 64;
    }
    if (bm == 128 && bn == 128) return launch_igemm<128, 128, 2, 2, MODE_PAIR>(*d, s);
    if (bm == 128 && bn == 64) return launch_igemm<128, 64, 2, 2, MODE_PAIR>(*d, s);
    DEFT_CHECK(false, -32, "deft_pair_layer: unsupported tile %dx%d", bm, bn);
    return -32;
}
