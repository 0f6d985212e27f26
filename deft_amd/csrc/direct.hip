// Direct (patch-in-LDS) convolution for the layers of DLA-34 that READ A FULL-RESOLUTION MAP (gfx950 only):
//   base_layer   7x7, 3(+1 pad) -> 16, stride 1, pad 3 @ H x W      (dla.py:301-306)
//   level0       3x3, 16 -> 16,     stride 1, pad 1 @ H x W      (dla.py:307-308, _make_conv_level :331-345)
//   level1       3x3, 16 -> 32,     stride 2, pad 1 @ H x W      (dla.py:309-310)
// the first two together 6.2 GFLOP per 1088x608 frame on 661 504 pixels -- 16 channels wide, so an implicit GEMM has 16 (32 as pixel
// pairs) columns and a 147 / 144-deep contraction: on igemm.hip they run on v_mfma_f32_32x32x2_f32 at 61-70 TFLOP/s with the
// matrix pipe 62 % busy (profiles/r2_summary.md), 1/3 of it on padding.  Here a workgroup stages the (8 + KH - 1) x (32 + KW - 1)
// fp32 input patch of an 8 x 32 output tile ONCE, splits it into the operand pieces of the split arithmetic (common.h DEFT_PIECES: two fp16
// pieces in the product build, three bf16 pieces in libdeft_bf16x3.so) on the way into LDS -- one split per input element instead of one per
// (output pixel, tap) -- and takes every tap out of the patch as a shifted fragment read feeding v_mfma_f32_16x16x32_{f16,bf16} (16 pixels x
// 16 output channels x 32 k): three (six) products per fp32 product, fp32 accumulation, the same arithmetic as DeftGemmDesc.prec = 1.
//
// K order inside one MFMA (lane l: row/col l & 15, k-group g = l >> 4 holding 8 consecutive k):
//   Cin = 16:  step t covers taps 2t and 2t+1:  tap = 2t + (g >> 1), channels 8 (g & 1) .. +7     (3x3: 5 steps, tap 9 = zero weights)
//   Cin = 4:   step t is window row t, 8 pixels x 4 channels: pixel 2g + (e >> 2), channel e & 3  (7x7: 7 steps, pixel 7 = zero weights)
// The weight image w3 is [Cout/16 column blocks][steps][3 pieces][64 lanes][8 bf16]: the B fragments, loaded lane-linearly into registers once.
// Stride 2 (level1): a 4 x 32 output tile from a 9 x 65 patch; wave w owns column block w & 1 (16 of the 32 output channels) and
// output rows 2 (w >> 1), +1 -- the same four m-tiles and register budget per wave as the stride-1 form.
#include "common.h"

typedef float dc_f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int DC_TW = 32;

template <int KH, int KW, int CIN, int STRIDE>
struct DcCfg {
    static_assert(CIN == 16 || CIN == 4, "16 channels, or the 4-channel (padded RGB) image");
    static_assert(CIN == 16 || KW <= 8, "the 4-channel form covers one window row of up to 8 pixels per MFMA");
    static_assert(STRIDE == 1 || (STRIDE == 2 && CIN == 16), "stride 2 for the 16-channel form only");
    static constexpr int NB = STRIDE;                                  // 16-channel output column blocks: 1 (Cout <= 16) or 2 (Cout <= 32)
    static constexpr int TH = 8 / NB;                                  // output rows per tile: a wave owns two rows and one column block
    static constexpr int PH = (TH - 1) * STRIDE + KH;
    static constexpr int PW = (DC_TW - 1) * STRIDE + (CIN == 16 ? KW : 9);   // Cin = 4: an 8-pixel K step reaches pixel x + 7
    // bytes per patch pixel and piece.  Cin = 16: 32 B of data in a 48-byte slot -- 16 lanes reading 16 B at a 48-byte stride
    // touch every LDS bank once (at 32 bytes two lanes would share each bank); stride 2: 40-byte slots (lanes 80 B apart: every
    // bank once per 16 lanes), read as two 8-byte halves
    static constexpr int PXB = CIN == 16 ? (STRIDE == 1 ? 48 : 40) : 8;
    static constexpr int PLANE = PH * PW * PXB;
    static constexpr int STEPS = CIN == 16 ? (KH * KW + 1) / 2 : KH;
    static constexpr int LDS = DEFT_NP * PLANE;
    static constexpr int ITEMS = CIN == 16 ? PH * PW * 4 : PH * PW;      // 16-byte global loads per patch
    static constexpr int NI = (ITEMS + 255) / 256;
};

template <int KH, int KW, int CIN, int STRIDE>
__global__ __launch_bounds__(256, STRIDE == 1 ? 3 : 2) void direct_conv_kernel(const DeftGemmDesc p, int tiles_x, int tiles_y) {
    using C = DcCfg<KH, KW, CIN, STRIDE>;
    DEFT_DYN_LDS(char, smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int bid = blockIdx.x;
    {   // XCD-aware, bijective workgroup remap (see igemm.hip): neighbouring tiles share their halo in one XCD's L2
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int txi = bid % tiles_x; bid /= tiles_x;
    const int tyi = bid % tiles_y;
    const int n = bid / tiles_y;
    const int oy0 = tyi * C::TH, ox0 = txi * DC_TW;
    const int cb = C::NB == 1 ? 0 : wave & 1;          // this wave's 16-channel output column block
    const int wrow = C::NB == 1 ? 2 * wave : 2 * (wave >> 1);   // ... and its first output row of the tile

    // ---- weights: this lane's B fragments of every step (L2-resident, lane-linear) ----
    const pcx8* const wf = (const pcx8*)p.w3 + cb * C::STEPS * (DEFT_NP * 64);
    pcx8 Bf[C::STEPS][DEFT_NP];
#pragma unroll
    for (int t = 0; t < C::STEPS; ++t)
#pragma unroll
        for (int q = 0; q < DEFT_NP; ++q) Bf[t][q] = wf[(t * DEFT_NP + q) * 64 + lane];

    const int co = cb * 16 + (lane & 15);          // epilogue constants, fetched ahead of everything that waits
    const float sc = (p.scale ? p.scale[co < p.Cout ? co : 0] : 1.f) * DEFT_ASCALE_INV, sh = p.shift ? p.shift[co < p.Cout ? co : 0] : 0.f;     // (the patch is split with DEFT_ASCALE)

    // ---- stage the patch: fp32 NHWC -> three bf16 piece planes in LDS (zero outside the image) ----
    {
        const deft_rsrc_t rx = deft_make_rsrc(p.x);
        const bool planar = CIN == 4 && (p.tile & DEFT_TILE_PLANAR) != 0;
        dc_f32x4 v[C::NI];
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
            const int it = i * 256 + tid;
            const int px = CIN == 16 ? it >> 2 : it, cq = CIN == 16 ? it & 3 : 0;
            const int py = px / C::PW, pxx = px - py * C::PW;
            const int iy = oy0 * STRIDE - p.pad + py, ix = ox0 * STRIDE - p.pad + pxx;
            const bool ok = it < C::ITEMS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            if (CIN == 4 && planar) {
                // the image as the reference hands it over, [N, 3, H, W] planes (detector.py:150): three 4-byte reads per patch pixel,
                // consecutive lanes on consecutive x -- what deft_nchw_to_nhwc would have written as (r, g, b, 0), without the pass
                const size_t hw = (size_t)p.H * p.W;
                const float* const xp = p.x + ((size_t)n * 3 * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0);
                const float c0 = xp[0], c1 = xp[hw], c2 = xp[2 * hw];
                v[i] = ok ? dc_f32x4{c0, c1, c2, 0.f} : dc_f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const unsigned off = ok ? ((unsigned)((n * p.H + iy) * p.W + ix) * (unsigned)p.ldx + (unsigned)cq * 4u) * 4u : DEFT_OOB;
            v[i] = deft_buffer_load_x4(rx, off);
        }
#pragma unroll
        for (int i = 0; i < C::NI; ++i) {
            const int it = i * 256 + tid;
            if (it < C::ITEMS) {
                const int px = CIN == 16 ? it >> 2 : it, cq = CIN == 16 ? it & 3 : 0;
                pcx4 pc[DEFT_NP];
                deft_split(v[i], pc, DEFT_ASCALE);
                char* const dst = smem + px * C::PXB + cq * 8;
#pragma unroll
                for (int q = 0; q < DEFT_NP; ++q) *(pcx4*)(dst + q * C::PLANE) = pc[q];
            }
        }
    }
    __syncthreads();

    // ---- contraction: wave w owns output rows 2w, 2w+1 of the tile = four 16-pixel m-tiles ----
    dc_f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = dc_f32x4{0.f, 0.f, 0.f, 0.f};
    const int prow = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < C::STEPS; ++t) {
        int a0;                                     // byte offset of this lane's 16 bytes for m-tile (row 0, x 0)
        if (CIN == 16) {
            int tap = 2 * t + (g >> 1);
            tap = tap < KH * KW ? tap : KH * KW - 1;                // the padding tap of the last step: zero weights, any finite data
            const int tr = tap / KW, ts = tap - tr * KW;
            a0 = ((wrow * STRIDE + tr) * C::PW + prow * STRIDE + ts) * C::PXB + (g & 1) * 16;
        } else {
            a0 = ((wrow + t) * C::PW + prow + 2 * g) * C::PXB;
        }
        pcx8 Af[4][DEFT_NP];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const char* const ap = smem + a0 + ((mt >> 1) * C::PW + (mt & 1) * 16) * STRIDE * C::PXB;
#pragma unroll
            for (int q = 0; q < DEFT_NP; ++q) {
                if (CIN == 16 && STRIDE == 1) {
                    Af[mt][q] = *(const pcx8*)(ap + q * C::PLANE);
                } else {                                // 8-byte aligned only: two 8-byte reads (Cin = 4: two pixels)
                    const pcx4 u0 = *(const pcx4*)(ap + q * C::PLANE), u1 = *(const pcx4*)(ap + q * C::PLANE + 8);
                    Af[mt][q] = __builtin_shufflevector(u0, u1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
        }
        // the products of the split, smallest terms first (as the other split kernels); the four m-tiles interleaved so that no MFMA
        // waits on the one before it
#pragma unroll
        for (int q = 0; q < DEFT_NPROD; ++q)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = deft_mfma16_pc(Af[mt][deft_qa(q)], Bf[t][deft_qb(q)], acc[mt]);
    }

    // ---- epilogue: D reg i of lane l is pixel 4 (l >> 4) + i of the m-tile, output channel l & 15 ----
    if (co < p.Cout) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int oy = oy0 + wrow + (mt >> 1);
            if (oy >= p.OH) continue;
            float* const yr = p.y + (size_t)((n * p.OH + oy) * (size_t)p.OW) * p.ldy + co;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = ox0 + (mt & 1) * 16 + 4 * g + i;
                if (ox < p.OW) {
                    float r = acc[mt][i] * sc + sh;
                    if (p.relu) r = deft_relu(r);
                    yr[(size_t)ox * p.ldy] = r;
                }
            }
        }
    }
}

template <int KH, int KW, int CIN, int STRIDE>
int launch_direct(const DeftGemmDesc& d, hipStream_t s) {
    using C = DcCfg<KH, KW, CIN, STRIDE>;
    const int tiles_x = deft_cdiv(d.OW, DC_TW), tiles_y = deft_cdiv(d.OH, C::TH);
    const long long grid = (long long)d.N * tiles_x * tiles_y;
    DEFT_CHECK(grid < (1ll << 31), -70, "deft_conv_direct: too many tiles");
    if (C::LDS > 64 * 1024) {
        static bool done = false;
        if (!done) {
            hipError_t e = hipFuncSetAttribute((const void*)direct_conv_kernel<KH, KW, CIN, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
            DEFT_CHECK(e == hipSuccess, -101, "deft_conv_direct: hipFuncSetAttribute(%d B LDS) failed: %s", C::LDS, hipGetErrorString(e));
            done = true;
        }
    }
    hipLaunchKernelGGL((direct_conv_kernel<KH, KW, CIN, STRIDE>), dim3((unsigned)grid), dim3(256), C::LDS, s, d, tiles_x, tiles_y);
    DEFT_CHECK_LAUNCH("deft_conv_direct");
    return 0;
}

// one thread per 16-byte slot of the fragment image [Cout/16 column blocks][steps][DEFT_NP pieces][64 lanes]
__global__ __launch_bounds__(256) void split_weights_direct_kernel(const float* __restrict__ w, deft_piece_t* __restrict__ w3, int Cout, int Kpad, int KH, int KW,
                                                                   int Cin, int steps) {
    constexpr int PER = DEFT_NP * 64;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nb = (Cout + 15) / 16;
    if (i >= nb * steps * PER) return;
    const int lane = i & 63, q = (i >> 6) % DEFT_NP, t = (i / PER) % steps, cb = i / (PER * steps);
    const int co = cb * 16 + (lane & 15), g = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        int k = -1;                                   // index into the packed row, k = (r*KW + s)*Cin + c
        if (Cin == 16) {
            const int tap = 2 * t + (g >> 1);
            if (tap < KH * KW) k = tap * 16 + (g & 1) * 8 + e;
        } else {
            const int s = 2 * g + (e >> 2);
            if (s < KW) k = (t * KW + s) * 4 + (e & 3);
        }
        v[e] = (k >= 0 && co < Cout) ? w[(size_t)co * Kpad + k] : 0.f;
    }
    pcx4 pc[2][DEFT_NP];
    deft_split(f32x4{v[0], v[1], v[2], v[3]}, pc[0]);
    deft_split(f32x4{v[4], v[5], v[6], v[7]}, pc[1]);
    *(pcx8*)(w3 + (size_t)i * 8) = __builtin_shufflevector(pc[0][q], pc[1][q], 0, 1, 2, 3, 4, 5, 6, 7);
}

int direct_steps(int KH, int KW, int Cin) { return Cin == 16 ? (KH * KW + 1) / 2 : KH; }

}  // namespace

extern "C" int deft_conv_direct(const DeftGemmDesc* d, void* stream) {
    DEFT_CHECK(d != nullptr && d->x && d->w3 && d->y, -1, "deft_conv_direct: null descriptor / x / w3 / y");
    DEFT_CHECK((d->stride == 1 || d->stride == 2) && (d->stride_w == 0 || d->stride_w == d->stride) && d->KH == d->KW && d->pad == d->KH / 2 &&
                   d->OH == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->OW == (d->W + 2 * d->pad - d->KW) / d->stride + 1, -72,
               "deft_conv_direct: stride 1 or 2, pad = KH / 2 (KH=%d KW=%d stride=%d pad=%d OH=%d OW=%d)", d->KH, d->KW, d->stride, d->pad, d->OH, d->OW);
    DEFT_CHECK(d->Cout >= 1 && d->Cout <= 16 * d->stride && d->ldy >= d->Cout, -73, "deft_conv_direct: Cout=%d (at most 16 output channels per unit of stride), ldy=%d", d->Cout, d->ldy);
    DEFT_CHECK(d->res == nullptr && d->rowmap == nullptr && d->splitk <= 1 && d->y3 == nullptr && d->x3 == nullptr, -74,
               "deft_conv_direct: no residual / rowmap / split-K / P3 operands");
    const bool planar = (d->tile & DEFT_TILE_PLANAR) != 0;
    DEFT_CHECK(!planar || (d->Cin == 4 && d->KH == 7), -75, "deft_conv_direct: the planar ([N, 3, H, W]) input form is the 7x7 image layer's (Cin = 4)");
    DEFT_CHECK(planar ? (((size_t)d->x & 3) == 0 && ((size_t)d->w3 & 15) == 0)
                      : ((d->ldx & 3) == 0 && d->ldx >= d->Cin && (((size_t)d->x | (size_t)d->w3) & 15) == 0), -75, "deft_conv_direct: ldx %% 4, 16-byte aligned x / w3");
    DEFT_CHECK(d->M == d->N * d->OH * d->OW && (long long)d->N * d->H * d->W * d->ldx < (1ll << 29), -76, "deft_conv_direct: M mismatch or input exceeds 2 GiB");
    hipStream_t s = (hipStream_t)stream;
    if (d->KH == 3 && d->Cin == 16 && d->stride == 1) return launch_direct<3, 3, 16, 1>(*d, s);
    if (d->KH == 3 && d->Cin == 16 && d->stride == 2) return launch_direct<3, 3, 16, 2>(*d, s);
    if (d->KH == 7 && d->Cin == 4 && d->stride == 1) return launch_direct<7, 7, 4, 1>(*d, s);
    DEFT_CHECK(false, -77, "deft_conv_direct: built for 3x3 x 16 channels (stride 1, 2) and 7x7 x 4 channels (KH=%d Cin=%d stride=%d)", d->KH, d->Cin, d->stride);
    return -77;
}

extern "C" long long deft_direct_weight_bytes(int KH, int KW, int Cin, int Cout) {
    if (!((Cin == 16 && KH * KW >= 1) || (Cin == 4 && KW <= 8)) || KH < 1 || KW < 1 || Cout < 1 || Cout > 32) return -1;
    return (long long)((Cout + 15) / 16) * direct_steps(KH, KW, Cin) * DEFT_NP * 64 * 16;
}

extern "C" int deft_split_weights_direct(const float* w, void* w3, int Cout, int Kpad, int KH, int KW, int Cin, void* stream) {
    DEFT_CHECK(w && w3 && Cout >= 1 && Cout <= 32 && (Cin == 16 || (Cin == 4 && KW <= 8)) && KH >= 1 && KW >= 1 && Kpad >= KH * KW * Cin, -1,
               "deft_split_weights_direct: need Cout <= 32, Cin 16 (or 4 with KW <= 8), Kpad >= KH*KW*Cin (%d %d %d %d %d)", Cout, Kpad, KH, KW, Cin);
    const int steps = direct_steps(KH, KW, Cin);
    hipLaunchKernelGGL(split_weights_direct_kernel, dim3(deft_cdiv(((Cout + 15) / 16) * steps * DEFT_NP * 64, 256)), dim3(256), 0, (hipStream_t)stream, w, (deft_piece_t*)w3, Cout, Kpad, KH,
                       KW, Cin, steps);
    DEFT_CHECK_LAUNCH("split_weights_direct");
    return 0;
}
