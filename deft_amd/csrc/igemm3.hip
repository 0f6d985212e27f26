// Implicit-GEMM convolution on PRE-SPLIT operands for gfx950 ("P3" format, DeftGemmDesc.x3 / w3 / y3).
//
// Same arithmetic as igemm.hip's prec = 1 path -- every fp32 operand as three bf16 pieces, every fp32 product as six
// v_mfma_f32_32x32x16_bf16 products, fp32 accumulation, same k order: results are bit-identical -- but the operand
// split is no longer in the K loop.  Producers write their output as the three pieces (this kernel's epilogue,
// deft_split_planes, the pooling / upsample kernels), weights are split once at load time (deft_split_weights), and
// the loop is:   buffer_load ... lds (global -> LDS, no VGPRs, no VALU, no ds_write)  |  ds_read_b128  |  MFMA
// with NS LDS stages, the DMA of chunk k+NS-1 in flight under the MFMAs of chunk k, and ONE barrier per chunk.
//
// LDS image of a chunk (both operands): [rows][12 slots of 16 B] = 192 B per row (3 pieces x 32 k), slot
// (piece q, k-slot s) of row r at physical slot q*4 + (s ^ ((r >> 2) & 3)): the ds_read_b128 fragment reads
// (lane l -> row l & 31, k-slot kh*2 + (l >> 5)) are conflict-free for all four lane groups.  The DMA image is
// lane-linear (lane l of piece j deposits 16 B at base + 1024*j + 16*l), so the permutation is applied on the SOURCE
// address: activations per lane (im2col gather), weights pre-permuted in memory (w3 is the LDS image verbatim).
//
// Epilogue: the accumulators go through LDS (the staging region is free after the K loop) so that every thread
// owns 8 consecutive channels of one pixel: 16-byte residual loads, 16-byte fp32 stores and 16-byte stores of the
// three bf16 pieces (the operand split now costs 1/9 of what it cost inside a 3x3 conv's K loop, where every
// activation was split once per tap and per n-tile).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define P3_ROW 192            // bytes per staged row: 3 pieces x 32 bf16
#define P3_WBLK 12288         // 64 weight rows of one chunk

template <int BM, int BN, int NS>
constexpr int p3_lds_bytes() {
    constexpr int stage = NS * (BM + BN) * P3_ROW;
    constexpr int tile = BM * (BN + 4) * 4;
    return stage > tile ? stage : tile;
}

// WM x WN wavefronts, each a TM x TN grid of 32x32 accumulators.  NS = LDS stages (2 or 3).
// SPLIT: cross-workgroup split-K (DeftGemmDesc.splitk), same hand-over as igemm.hip.
template <int BM, int BN, int WM, int WN, int NS, bool SPLIT>
__global__ __launch_bounds__(WM* WN * 64) void igemm3_kernel(DeftGemmDesc p, int mtiles, int ntiles) {
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int NA = BM * 12 / NT;             // A slots (16 B) per thread and chunk
    constexpr int NBI = BN * 12 / 64;            // B DMA pieces (1 KB) per chunk
    constexpr int NB = (NBI + NW - 1) / NW;      // ... per wave
    constexpr int STAGE = (BM + BN) * P3_ROW;
    constexpr int PER = NA + NB;                 // DMA pieces a wave issues per chunk (vmcnt bookkeeping)
    static_assert(BM * 12 % NT == 0 && TM >= 1 && TN >= 1 && BN % 64 == 0, "tile shape");
    static_assert(NS == 2 || (NS == 3 && NBI % NW == 0), "3 stages need the same DMA count in every wave");

    DEFT_DYN_LDS(char, smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware, bijective workgroup remap (see igemm.hip)
    int bid = blockIdx.x;
    const int S = SPLIT ? p.splitk : 1;
    {
        const int nwg = mtiles * ntiles * S;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid % S;
    bid /= S;
    const int mt = bid / ntiles, nt = bid - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const deft_rsrc_t rx = deft_make_rsrc(p.x3);
    const deft_rsrc_t rw = deft_make_rsrc(p.w3);
    const unsigned pb = (unsigned)p.ldx3 * 6u;       // bytes per pixel: 3 pieces x ld channels x 2
    const int taps = p.KH * p.KW;

    // ---- per-slot loader state, fixed for the whole K loop: byte offset of (window top-left pixel, this slot's
    // piece and permuted k-slot) in wrap-around arithmetic, and one validity bit per tap ----
    unsigned rb[NA], vm[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int P = i * NT + tid;
        const int row = P / 12, ps = P - row * 12;
        const int q = ps >> 2, sl = (ps & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        rb[i] = 0; vm[i] = 0;
        if (m < p.M) {
            const int ohw = p.OH * p.OW;
            const int n = m / ohw;
            const int rem = m - n * ohw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            rb[i] = ((unsigned)(n * p.H * p.W) + (unsigned)iy0 * (unsigned)p.W + (unsigned)ix0) * pb + (unsigned)(q * 64 + sl * 16);
            unsigned v = 0;
            for (int t = 0; t < taps; ++t) {
                const int r = t / p.KW, s = t - r * p.KW;
                if ((unsigned)(iy0 + r) < (unsigned)p.H && (unsigned)(ix0 + s) < (unsigned)p.W) v |= 1u << t;
            }
            vm[i] = v;
        }
    }
    // weights: piece j of the chunk image = 1 KB of the 64-row block j / 12
    const int nk_all = p.Kpad >> 5;
    unsigned vB[NB];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
        const int j = wave + jb * NW;
        vB[jb] = (unsigned)(((n0 >> 6) + j / 12) * nk_all * 12 + j % 12) * 1024u + (unsigned)lane * 16u;
    }

    // this workgroup's share of the K chunks
    const int k_lo = (int)((long long)nk_all * split / S);
    const int nk = (int)((long long)nk_all * (split + 1) / S) - k_lo;
    int cr = 0, cs = 0, cc = 0;                       // cursor of the next chunk to load: tap (cr, cs), first channel cc
    if (SPLIT && taps > 1) {
        int tap;
        if (p.korder == 1) { tap = k_lo % taps; cc = (k_lo / taps) * 32; }
        else { tap = (k_lo * 32) >> p.cin_log2; cc = (k_lo * 32) & (p.Cin - 1); }
        cr = tap / p.KW; cs = tap - cr * p.KW;
    } else if (SPLIT) {
        cc = k_lo * 32;
    }
    int kload = k_lo;

    auto issue = [&](int stage) {
        char* const as = smem + stage * STAGE;
        char* const bs = as + BM * P3_ROW;
        const unsigned toff = (unsigned)(cr * p.W + cs) * pb + (unsigned)(cc >> 5) * 192u;     // scalar unit
        const unsigned tbit = 1u << (cr * p.KW + cs);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            deft_buffer_load_lds_x4s(rx, as + (i * NT + wave * 64) * 16, (vm[i] & tbit) ? rb[i] + toff : DEFT_OOB, 0u);
        const unsigned soff = (unsigned)kload * (unsigned)P3_WBLK;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
            const int j = wave + jb * NW;
            if (NBI % NW == 0 || j < NBI) deft_buffer_load_lds_x4s(rw, bs + j * 1024, vB[jb], soff);
        }
        ++kload;
        if (p.korder == 1) {                  // (32-channel block, tap): the nine shifted reads of a block follow each other,
            if (++cs == p.KW) {               // so all but the first come out of the XCD's L2 (tap-major order re-reads the
                cs = 0;                       // whole tile's input once per tap -- from HBM when Cin is large)
                if (++cr == p.KH) { cr = 0; cc += 32; }
            }
        } else {
            cc += 32;
            if (taps > 1 && cc >= p.Cin) {
                cc = 0;
                if (++cs == p.KW) { cs = 0; ++cr; }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    auto compute = [&](int stage) {
        const char* const as = smem + stage * STAGE + (wm * TM * 32 + frow) * P3_ROW;
        const char* const bs = smem + stage * STAGE + BM * P3_ROW + (wn * TN * 32 + frow) * P3_ROW;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int so = ((kh * 2 + (lane >> 5)) ^ fsw) * 16;
            bf16x8 pa[TM][3], pb_[TN][3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) pa[i][q] = *(const bf16x8*)(as + i * 32 * P3_ROW + q * 64 + so);
#pragma unroll
                for (int j = 0; j < TN; ++j) pb_[j][q] = *(const bf16x8*)(bs + j * 32 * P3_ROW + q * 64 + so);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];                                           // smallest terms first (as igemm.hip)
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][1], pb_[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][2], pb_[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][0], pb_[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][1], pb_[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][0], pb_[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][0], pb_[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    };

    // ---- K loop: NS stages, one barrier per chunk.  Iteration kt: wait until chunk kt has landed (this wave's DMA
    // pieces, then the barrier for everybody else's), refill the stage that was read in iteration kt-1 (everybody
    // has passed the barrier, so nobody reads it any more) with chunk kt+NS-1, then fragments + MFMAs of chunk kt.
    if (NS == 2) {
        issue(0);
        for (int kt = 0; kt < nk; ++kt) {
            DEFT_PIPE_BARRIER(0);
            if (kt + 1 < nk) issue((kt + 1) & 1);
            compute(kt & 1);
        }
    } else {
        issue(0);
        if (nk > 1) issue(1);
        int st = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) DEFT_PIPE_BARRIER(PER);           // chunk kt+1 may still be in flight
            else DEFT_PIPE_BARRIER(0);
            if (kt + 2 < nk) issue(st == 0 ? 2 : st - 1);      // (kt + 2) % 3
            compute(st);
            st = st == 2 ? 0 : st + 1;
        }
    }
    __syncthreads();                                           // staging region free (no DMA in flight: the last wait was 0)

    if (SPLIT && S > 1) {
        // cross-workgroup split-K hand-over, as igemm.hip: park the partial tile, take a ticket, the last arriver adds
        // the S partials in split order and runs the epilogue
        const int tile_id = mt * ntiles + nt;
        const size_t tsz = (size_t)BM * BN;
        const int woff = (wave * TM * TN) * 1024 + lane;
        float* mine = p.ws + ((size_t)tile_id * S + split) * tsz + woff;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) deft_ws_store(&mine[(i * TN + j) * 1024 + r * 64], acc[i][j][r]);
        deft_ws_publish();
        __syncthreads();
        int* ticket = (int*)smem;
        if (tid == 0) ticket[0] = deft_ws_ticket(&p.ws_cnt[tile_id]);
        __syncthreads();
        const int tk = ticket[0];
        __syncthreads();                                       // (the ticket word is inside the epilogue tile)
        if (tk != S - 1) return;
        if (tid == 0) deft_ws_reset(&p.ws_cnt[tile_id]);
        const float* part = p.ws + (size_t)tile_id * S * tsz + woff;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = deft_ws_load(&part[(i * TN + j) * 1024 + r * 64]);
        for (int sp = 1; sp < S; ++sp) {
            part += tsz;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += deft_ws_load(&part[(i * TN + j) * 1024 + r * 64]);
        }
    }

    // ---- epilogue, phase 1: acc*scale + shift into the LDS tile T[BM][BN + 4] (D reg r of lane l is row (r&3) +
    // 8*(r>>2) + 4*(l>>5), col l&31: a half-wave writes 32 consecutive floats) ----
    constexpr int LDT = BN + 4;
    float* const T = (float*)smem;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cl = (wn * TN + j) * 32 + (lane & 31);
        const int co = n0 + cl;
        const int coc = co < p.Cout ? co : p.Cout - 1;
        const float sc = p.scale ? p.scale[coc] : 1.f;
        const float sh = p.shift ? p.shift[coc] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float* tp = T + ((wm * TM + i) * 32 + 4 * (lane >> 5)) * LDT + cl;
#pragma unroll
            for (int r = 0; r < 16; ++r) tp[((r & 3) + 8 * (r >> 2)) * LDT] = acc[i][j][r] * sc + sh;
        }
    }
    __syncthreads();
    // ---- phase 2: every thread owns 8 consecutive channels of a pixel ----
    constexpr int G = BN / 8;                      // channel groups per tile row
    constexpr int NI = BM * G / NT;
    const bool has_res = p.res != nullptr;
#pragma unroll 2
    for (int it = 0; it < NI; ++it) {
        const int item = it * NT + tid;
        const int row = item / G, c8 = item - row * G;
        const int m = m0 + row, co = n0 + c8 * 8;
        if (m >= p.M || co >= p.Cout) continue;
        const float* tp = T + row * LDT + c8 * 8;
        f32x4 v0 = *(const f32x4*)tp, v1 = *(const f32x4*)(tp + 4);
        if (has_res) {
            const float* rp = p.res + (size_t)m * p.ldr + co;
            v0 += *(const f32x4*)rp;
            v1 += *(const f32x4*)(rp + 4);
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
        }
        if (p.y != nullptr) {
            float* yp = p.y + (size_t)m * p.ldy + co;
            *(f32x4*)yp = v0;
            *(f32x4*)(yp + 4) = v1;
        }
        if (p.y3 != nullptr) {
            bf16x4 h0, m0_, l0, h1, m1, l1;
            split3(v0, h0, m0_, l0);
            split3(v1, h1, m1, l1);
            __bf16* yp = (__bf16*)p.y3 + (size_t)m * p.ldy3 * 3 + (co >> 5) * 96 + (co & 31);
            *(bf16x8*)yp = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            *(bf16x8*)(yp + 32) = __builtin_shufflevector(m0_, m1, 0, 1, 2, 3, 4, 5, 6, 7);
            *(bf16x8*)(yp + 64) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
}

template <auto KERNEL>
static int p3_set_lds_attr(int lds_bytes) {
    static bool done = false;
    if (lds_bytes <= 64 * 1024 || done) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    DEFT_CHECK(e == hipSuccess, -101, "igemm3: hipFuncSetAttribute(%d B LDS) failed: %s", lds_bytes, hipGetErrorString(e));
    done = true;
    return 0;
}

template <int BM, int BN, int WM, int WN, int NS>
static int launch_p3(const DeftGemmDesc& d, hipStream_t s) {
    constexpr int lds = p3_lds_bytes<BM, BN, NS>();
    const int mtiles = deft_cdiv(d.M, BM), ntiles = deft_cdiv(d.Cout, BN);
    const int S = d.splitk > 1 ? d.splitk : 1;
    DEFT_CHECK(S == 1 || (d.ws != nullptr && d.ws_cnt != nullptr && S <= 32 && (d.Kpad >> 5) >= S), -102,
               "igemm3: splitk=%d needs ws and ws_cnt, S <= 32 and at least S K chunks (%d)", S, d.Kpad >> 5);
    if (S > 1) {
        if (int e = p3_set_lds_attr<igemm3_kernel<BM, BN, WM, WN, NS, true>>(lds)) return e;
        hipLaunchKernelGGL((igemm3_kernel<BM, BN, WM, WN, NS, true>), dim3(mtiles * ntiles * S), dim3(WM * WN * 64), lds, s, d, mtiles, ntiles);
    } else {
        if (int e = p3_set_lds_attr<igemm3_kernel<BM, BN, WM, WN, NS, false>>(lds)) return e;
        hipLaunchKernelGGL((igemm3_kernel<BM, BN, WM, WN, NS, false>), dim3(mtiles * ntiles), dim3(WM * WN * 64), lds, s, d, mtiles, ntiles);
    }
    DEFT_CHECK_LAUNCH("igemm3");
    return 0;
}

// automatic tile of the P3 kernel: the 8-wave 256x128 tile (2 waves per SIMD, 170 bf16 FLOP per staged byte) when it
// still fills the chip, else the 4-wave tiles
void deft_p3_pick_tile(const DeftGemmDesc* d, int* bm, int* bn) {
    const long long m256 = deft_cdiv(d->M, 256), m128 = deft_cdiv(d->M, 128);
    if (d->Cout > 64) {
        *bn = 128;
        *bm = m256 * deft_cdiv(d->Cout, 128) >= 256 ? 256 : 128;
    } else {
        *bn = 64;
        *bm = m256 >= 256 ? 256 : 128;
    }
    (void)m128;
}

int deft_p3_check(const DeftGemmDesc* d, const char* who) {
    DEFT_CHECK(d->prec == 1, -60, "%s: the pre-split (x3) path is the prec = 1 arithmetic", who);
    DEFT_CHECK(d->w3 != nullptr && (((size_t)d->x3 | (size_t)d->w3 | (size_t)d->y3) & 15) == 0, -61, "%s: x3 needs w3; x3/w3/y3 16-byte aligned", who);
    DEFT_CHECK((d->Cin & 31) == 0 && (d->ldx3 & 31) == 0 && d->ldx3 >= d->Cin, -62, "%s: x3 needs Cin %% 32 == 0 and ldx3 %% 32 == 0 (Cin=%d ldx3=%d)", who, d->Cin, d->ldx3);
    DEFT_CHECK(d->Kpad == d->Ktot && (d->korder == 0 || d->korder == 1) && d->KH * d->KW <= 32 && d->rowmap == nullptr && d->stride_w == 0, -63,
               "%s: x3 needs Kpad == Ktot, at most 32 taps, no rowmap, no stride_w", who);
    DEFT_CHECK((d->Cout & 7) == 0 && (d->ldy & 3) == 0 && (!d->res || (d->ldr & 3) == 0) && (((size_t)d->y | (size_t)d->res) & 15) == 0, -64,
               "%s: x3 needs Cout %% 8 == 0, ldy/ldr %% 4 == 0, y/res 16-byte aligned", who);
    DEFT_CHECK(d->y3 == nullptr || ((d->ldy3 & 31) == 0 && d->ldy3 >= d->Cout && (d->Cout & 31) == 0), -65, "%s: y3 needs Cout %% 32 == 0 and ldy3 %% 32 == 0", who);
    DEFT_CHECK(d->y != nullptr || d->y3 != nullptr, -66, "%s: no output", who);
    DEFT_CHECK((long long)d->N * d->H * d->W * d->ldx3 * 6 < (1ll << 31), -67, "%s: x3 map exceeds 2 GiB (split the batch)", who);
    DEFT_CHECK((long long)deft_cdiv(d->Cout, 128) * 128 * d->Kpad * 6 < (1ll << 31), -68, "%s: w3 exceeds 2 GiB", who);
    return 0;
}

// `tile`: (BM << 16) | BN as igemm.hip; bit 29 selects 3 LDS stages where the tile has them.
int deft_p3_dispatch(const DeftGemmDesc* d, hipStream_t s) {
    int bm = (d->tile >> 16) & 0x1fff, bn = d->tile & 0xffff;
    const bool three = (d->tile >> 29) & 1;
    if (bm == 0) deft_p3_pick_tile(d, &bm, &bn);
#define P3_TILE(BM_, BN_, WM_, WN_, NS_) \
    if (bm == BM_ && bn == BN_ && three == (NS_ == 3)) return launch_p3<BM_, BN_, WM_, WN_, NS_>(*d, s);
    P3_TILE(256, 128, 4, 2, 2)
    P3_TILE(128, 256, 2, 4, 2)
    P3_TILE(128, 128, 2, 2, 2)
    P3_TILE(128, 128, 2, 2, 3)
    P3_TILE(128, 64, 2, 2, 2)
    P3_TILE(128, 64, 2, 2, 3)
    P3_TILE(256, 64, 4, 2, 2)
    P3_TILE(64, 64, 2, 2, 2)
    P3_TILE(64, 64, 2, 2, 3)
#undef P3_TILE
    DEFT_CHECK(false, -15, "igemm3: unsupported tile %dx%d%s", bm, bn, three ? " (3 stages)" : "");
    return -15;
}

// ---------------------------------------------------------------------------
// fp32 -> P3 converters
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, __bf16* __restrict__ y3, long long rows, int C, int ldx, int ldy3) {
    const int g = C >> 3;                                      // 8-channel groups per row
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * g) return;
    const long long row = i / g;
    const int c = (int)(i - row * g) * 8;
    const float* xp = x + row * ldx + c;
    const f32x4 v0 = *(const f32x4*)xp, v1 = *(const f32x4*)(xp + 4);
    bf16x4 h0, m0, l0, h1, m1, l1;
    split3(v0, h0, m0, l0);
    split3(v1, h1, m1, l1);
    __bf16* yp = y3 + row * ldy3 * 3 + (c >> 5) * 96 + (c & 31);
    *(bf16x8*)yp = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *(bf16x8*)(yp + 32) = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
    *(bf16x8*)(yp + 64) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}

extern "C" int deft_split_planes(const float* x, void* y3, long long rows, int C, int ldx, int ldy3, void* stream) {
    DEFT_CHECK(x && y3 && rows > 0 && C > 0 && (C & 31) == 0 && (ldx & 3) == 0 && ldx >= C && (ldy3 & 31) == 0 && ldy3 >= C, -1,
               "deft_split_planes: need C %% 32 == 0, ldx %% 4 == 0, ldy3 %% 32 == 0 (C=%d ldx=%d ldy3=%d)", C, ldx, ldy3);
    DEFT_CHECK((((size_t)x | (size_t)y3) & 15) == 0, -2, "deft_split_planes: x / y3 must be 16-byte aligned");
    const long long tot = rows * (C >> 3);
    DEFT_CHECK(tot < (1ll << 39), -3, "deft_split_planes: too many rows");
    hipLaunchKernelGGL(split_planes_kernel, dim3(deft_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)y3, rows, C, ldx, ldy3);
    DEFT_CHECK_LAUNCH("split_planes");
    return 0;
}

// one thread per 16-byte slot of the weight image
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ w3, int CoutPad, int Kpad) {
    const int nk = Kpad >> 5;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // (block, chunk, row, physical slot)
    if (i >= (long long)CoutPad * nk * 12) return;
    const int ps = (int)(i % 12);
    const long long t = i / 12;
    const int r = (int)(t & 63);
    const long long bc = t >> 6;
    const int chunk = (int)(bc % nk), blk = (int)(bc / nk);
    const int q = ps >> 2, s = (ps & 3) ^ ((r >> 2) & 3);
    const float* wp = w + (size_t)(blk * 64 + r) * Kpad + chunk * 32 + s * 8;
    bf16x4 pc[2][3];
    split3(*(const f32x4*)wp, pc[0][0], pc[0][1], pc[0][2]);
    split3(*(const f32x4*)(wp + 4), pc[1][0], pc[1][1], pc[1][2]);
    *(bf16x8*)(w3 + i * 8) = __builtin_shufflevector(pc[0][q], pc[1][q], 0, 1, 2, 3, 4, 5, 6, 7);
}

extern "C" int deft_split_weights(const float* w, void* w3, int CoutPad, int Kpad, void* stream) {
    DEFT_CHECK(w && w3 && CoutPad > 0 && (CoutPad & 63) == 0 && Kpad > 0 && (Kpad & 31) == 0, -1,
               "deft_split_weights: need CoutPad %% 64 == 0 and Kpad %% 32 == 0 (%d, %d)", CoutPad, Kpad);
    const long long tot = (long long)CoutPad * (Kpad >> 5) * 12;
    hipLaunchKernelGGL(split_weights_kernel, dim3(deft_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)w3, CoutPad, Kpad);
    DEFT_CHECK_LAUNCH("split_weights");
    return 0;
}
