// FP32-MFMA implicit GEMM for gfx950 with three A-operand loaders:
//   MODE_CONV : NHWC im2col (any KHxKW / stride / pad; concat inputs via ld)
//   MODE_DCN  : DCNv2 modulated deformable gather (bilinear, zero pad, * sigmoid(mask)); K order
//               (32-channel block, tap, channel) -- weights packed to match (engine.pack_dcn_weight)
//   MODE_PAIR : affinity pair grid, A[(i,j)][k] = relu(U'[i][k] + V'[j][k])
// and one epilogue: y = acc*scale[co] + shift[co] (+ residual) (ReLU).
//
// Tile: BM x BN outputs per 256-thread workgroup (4 wavefronts of 64), K consumed
// in chunks of 32.  The A/B chunks are staged global -> VGPR -> LDS ([rows][32+4]
// floats, 144-B row stride: conflict-free for ds_read_b128 / ds_write_b128), with the
// next chunk's global loads in flight while the current chunk's MFMAs issue (the
// staging registers are plain locals: nothing may live in scratch, a scratch round
// trip puts an s_waitcnt vmcnt in front of the MFMAs and serialises load and math).
// v_mfma_f32_32x32x2_f32 contracts two k per instruction; the k index a lane
// half h feeds at step kk is k = 16*h + kk, so every lane reads its 16 k-values
// as four ds_read_b128 (the k permutation is the same for A and B, so the
// contraction is unchanged).  The parity bar (bit-exact top-k, 1e-3 on boxes)
// rules out plain bf16/fp8 MFMA; fp32 accuracy comes either from the fp32 MFMA
// (prec 0, peak 157.3 TFLOP/s) or from six bf16 MFMA products per fp32 product
// with fp32 accumulation (prec 1, see igemm_body).
#include <cstdlib>

#include "common.h"

typedef deft_f32x16 f32x16;

enum { MODE_CONV = 0, MODE_DCN = 1, MODE_PAIR = 2 };


#define LDS_STRIDE 36
#define ROW_INVALID (-(1 << 28))

// dynamic-LDS layout (floats): As[NSTAGE][BM*36] | Bs[NSTAGE][BN*36] | DCN sampling records [9][BM][8] + masks [9][BM];
// with intra-workgroup split-K (WK > 1) the same region is reused after the K loop for the
// partial accumulators of the wk > 0 waves: [(WK-1)][32x32 tiles of the block][16][64].
// PREC = 1 (fp32 through the bf16 matrix cores): the staged chunk is three bf16 planes (hi, mid, lo) of
// [rows][32 + 8 pad] instead of one fp32 image.
#define LDB 40
// PREC = 2: as 1, with the weights pre-split (DeftGemmDesc.w3): their chunk image goes global -> LDS by DMA into one of two
// [BN][192 B] stages (layout and swizzle of igemm3.hip) -- no staging registers, no split arithmetic, no ds_write for B.
template <int BM, int BN, int WK, int NSTAGE, int MODE, int PREC = 0>
constexpr int igemm_lds_floats() {
    constexpr int ld = (MODE == MODE_CONV && NSTAGE == 2) ? 32 : LDS_STRIDE;      // LDS-DMA image is unpadded
    constexpr int stage = (PREC >= 2 ? (DEFT_NP * BM * LDB * 2 + 2 * BN * (DEFT_NP * 64)) / 4 : PREC ? DEFT_NP * (BM + BN) * LDB / 2 : NSTAGE * (BM + BN) * ld) + (MODE == MODE_DCN ? 9 * BM * 5 : 0);
    constexpr int red = (WK - 1) * (BM / 32) * (BN / 32) * 1024;
    return stage > red ? stage : red;
}

// Position of the 32-wide K chunk being loaded: tap (r,s) and first channel c0.  When a chunk
// lies inside one tap (1x1, or Cin >= 32) these are workgroup-uniform and live in SGPRs.
struct KCursor {
    int r, s, c0;
};

// WM x WN x WK = 4 wavefronts: (wm, wn) pick the wave's TM x TN grid of 32x32 output tiles, wk
// its share of every chunk's 16 MFMA k-steps (intra-workgroup split-K for problems too small to
// fill 256 CUs with bigger tiles: all four waves still stage the chunk, partial sums are
// combined through LDS in wave order, so the result is deterministic).
// SPLIT: cross-workgroup split-K (DeftGemmDesc.splitk > 1).  A separate instantiation: its partial-tile hand-over
// must not cost the plain kernels a register (128x128: 104 VGPRs = 3 waves/SIMD, 136 with the hand-over = 2).
// PREC: 0 = v_mfma_f32_32x32x2_f32 (a k-ordered fp32 fmaf chain); 1 = every fp32 product as six
// v_mfma_f32_32x32x16_bf16 products of the operands' bf16 pieces (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid:
// everything down to 2^-24 relative), fp32 accumulation -- the error of an fp32 chain (tools/probe/split_bf16_loop.hip:
// 5.9e-6 vs 6.8e-6 max abs at K = 512) at 16/6 of the fp32 MFMA rate.  WK = 1 tiles, 1-stage loop.
template <int BM, int BN, int WM, int WN, int WK, int MODE, int NSTAGE, bool SPLIT, int PREC = 0>
__device__ __forceinline__ void igemm_body(const DeftGemmDesc& p, int mtiles, int ntiles, int bid) {
    static_assert(PREC == 0 || (WK == 1 && NSTAGE == 1), "split-bf16 path: WK = 1 tiles, 1-stage loop");
    static_assert(PREC < 2 || BN % 64 == 0, "pre-split weights come in 64-row blocks");
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int GA = BM / 32;  // f32x4 groups per thread, A tile
    constexpr int GB = BN / 32;  // f32x4 groups per thread, B tile
    constexpr int KK = 16 / WK;  // MFMA k-steps per chunk per wave
    static_assert(WM * WN * WK == 4, "4 wavefronts per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile too small");

    // CONV with NSTAGE == 2 is the LDS-DMA form: A and B chunks go global -> LDS directly
    // (buffer_load ... lds, zero fill for the im2col padding by out-of-range offsets), no staging
    // VGPRs and no ds_write pass.  The DMA image is lane-linear, so rows are 128 B unpadded and
    // the 16-byte k-slot c of row R sits at physical slot c ^ ((R >> 1) & 7): conflict-free for
    // the ds_read_b128 fragment reads, applied on the SOURCE address of the DMA.
    constexpr bool DMA = (MODE == MODE_CONV && NSTAGE == 2);
    constexpr int LD = DMA ? 32 : LDS_STRIDE;
    DEFT_DYN_LDS(float, smem);
    float* const As = smem;
    float* const Bs = smem + NSTAGE * BM * LD;
    constexpr bool BDMA = PREC >= 2;
    static_assert(!(BDMA && MODE == MODE_DCN), "the DCN reads pre-split weights on its patch form only (dcn.hip)");
    constexpr int NBS = 2;                               // weight stages of the DMA form
    constexpr int BROW = DEFT_NP * 64;                   // bytes per row of the pre-split weight image (igemm3.hip P3_ROW)
    float* const prm = BDMA ? smem + (DEFT_NP * BM * LDB * 2 + NBS * BN * BROW) / 4 : PREC ? smem + DEFT_NP * (BM + BN) * LDB / 2 : Bs + NSTAGE * BN * LD;   // DCN only
    deft_piece_t* const Ap = (deft_piece_t*)smem;   // PREC >= 1: A planes [NP][BM][LDB], then B planes [NP][BN][LDB] (PREC 2: two DMA stages [BN][BROW B])
    deft_piece_t* const Bp = Ap + DEFT_NP * BM * LDB;
    char* const Bd = (char*)(Ap + DEFT_NP * BM * LDB);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = wave / (WM * WN);
    const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;

    // XCD-aware, bijective workgroup remap: block b runs on XCD b%8 (observed), so give
    // every XCD one contiguous run of tiles -- neighbouring n-tiles of an m-tile then
    // share their A rows in that XCD's private L2.
    const int S = SPLIT ? p.splitk : 1;
    {
        const int nwg = mtiles * ntiles * S;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int split = bid % S;      // cross-workgroup split-K: the S workgroups of a tile are neighbours (same XCD)
    bid /= S;
    const int mt = bid / ntiles, nt = bid - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int g = tid & 7;       // which f32x4 of the 32-wide k chunk this thread stages
    const int rbase = tid >> 3;  // 0..31: staged rows are rbase + 32*i

    const deft_rsrc_t rx = deft_make_rsrc(p.x);
    const deft_rsrc_t rx2 = deft_make_rsrc(MODE == MODE_PAIR ? p.x2 : (DMA ? p.w : p.x));   // DMA: the weight matrix
    const int gs = DMA ? (g ^ ((rbase >> 1) & 7)) : g;    // k-slot this thread fetches (DMA: swizzled)

    // per-row loader state, fixed for the whole K loop.
    //   CONV: r0/r1 = top-left input coordinate of the window (ROW_INVALID fails every bounds
    //         test), r2 = n*H*W.   DCN: r2 = n*H*W.   PAIR: r0/r1 = BYTE offsets of the U'/V' rows
    //         (DEFT_OOB for rows beyond M: the buffer loads return 0 and relu(0+0) = 0).
    int r0[GA], r1[GA], r2[GA];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int m = m0 + rbase + 32 * i;
        r0[i] = MODE == MODE_PAIR ? (int)DEFT_OOB : ROW_INVALID;
        r1[i] = MODE == MODE_PAIR ? (int)DEFT_OOB : 0;
        r2[i] = 0;
        if (m < p.M) {
            if (MODE == MODE_PAIR) {
                int u = m / p.Q;
                int j = m - u * p.Q;
                if (p.Tper > 0) {       // batched: (current frame c, history row t, object j)
                    const int c = u / p.Tper;
                    u = p.u0 + c * p.du + (u - c * p.Tper);
                    j = p.v0 + c * p.dv + j;
                }
                r0[i] = u * p.ldx * 4;
                r1[i] = j * p.ldx * 4;
            } else if (MODE == MODE_CONV && p.rowmap != nullptr) {
                // sparse output rows: rowmap[m] = {n*H*W, (y << 16) | x} of the row's output pixel, or {_, -1}
                const int e0 = p.rowmap[2 * m], e1 = p.rowmap[2 * m + 1];
                if (e1 >= 0) {
                    r0[i] = (e1 >> 16) * p.stride - p.pad;
                    r1[i] = (e1 & 0xffff) * (p.stride_w > 0 ? p.stride_w : p.stride) - p.pad;
                    r2[i] = e0;
                }
            } else {
                const int ohw = p.OH * p.OW;
                const int n = m / ohw;
                const int rem = m - n * ohw;
                const int oy = rem / p.OW;
                const int ox = rem - oy * p.OW;
                if (MODE == MODE_CONV) {
                    r0[i] = oy * p.stride - p.pad;
                    r1[i] = ox * (p.stride_w > 0 ? p.stride_w : p.stride) - p.pad;
                } else {
                    r0[i] = 0;
                }
                r2[i] = n * p.H * p.W;
            }
        }
    }

    // byte offset of (window top-left pixel | image origin for DCN, this thread's k-slot): the per-chunk
    // address is then ONE add of a workgroup-uniform tap/channel term -- no integer multiplies in the K loop
    int rb[GA];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        if (MODE == MODE_CONV)         // unsigned wrap-around arithmetic: padding rows give 'negative' bases, invalid rows garbage (never used)
            rb[i] = (int)((((unsigned)r2[i] + (unsigned)r0[i] * (unsigned)p.W + (unsigned)r1[i]) * (unsigned)p.ldx + (unsigned)(gs * 4)) * 4u);
        else if (MODE == MODE_DCN) rb[i] = (r2[i] * p.ldx + g * 4) * 4;
        else rb[i] = 0;
    }

    // DCN: the bilinear sampling records of all (tile row, tap) pairs are computed ONCE, before the
    // K loop -- upstream DCNv2 modulated_deformable_im2col + dmcn_im2col_bilinear semantics -- into
    // prm[tap][row] = the four corner weights, each already multiplied by sigmoid(mask), and pof[tap][row] = the byte offset of
    // the (clamped) top-left corner pixel with two flags in its low bits: bit 0 = the right-hand corners are one pixel further,
    // bit 1 = the lower corners are one line further (both set away from the map border; a corner outside the map has weight
    // 0 and reads some pixel inside it).  20 bytes per record: 11.5 KB at BM = 64 -- four 64x64 / three 64x128 workgroups per CU.
    // The K order of the DCN contraction is (32-channel block, tap, channel): all nine taps of one
    // channel block are consumed back to back, so the 4x4-pixel neighbourhood lines of that block
    // stay in the CU's 32 KB L1 across the 36 corner reads that touch them.
    int* const pof = (int*)(prm + 9 * BM * 4);
    if (MODE == MODE_DCN) {
        for (int idx = tid; idx < 9 * BM; idx += 256) {
            const int tap = idx / BM, row = idx - tap * BM;
            const int m = m0 + row;
            int o1 = 0;
            float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
            if (m < p.M) {
                const int hw = p.H * p.W;
                const int rem = m - (m / hw) * hw;
                const int oy = rem / p.W, ox = rem - oy * p.W;
                const float* om = p.x2 + (size_t)m * p.ldom;
                const float dy = om[2 * tap], dx = om[2 * tap + 1], ml = om[18 + tap];
                const int r = tap / 3, s = tap - 3 * r;
                const float h_im = (float)(oy - 1 + r) + dy;
                const float w_im = (float)(ox - 1 + s) + dx;
                if (h_im > -1.f && w_im > -1.f && h_im < (float)p.H && w_im < (float)p.W) {
                    const float hl = floorf(h_im), wl = floorf(w_im);
                    const float lh = h_im - hl, lw = w_im - wl;
                    const float hh = 1.f - lh, hw_ = 1.f - lw;
                    const int h_low = (int)hl, w_low = (int)wl;
                    const int h_high = h_low + 1, w_high = w_low + 1;
                    const float mask = 1.f / (1.f + expf(-ml));
                    if (h_low >= 0 && w_low >= 0) w1 = hh * hw_ * mask;
                    if (h_low >= 0 && w_high <= p.W - 1) w2 = hh * lw * mask;
                    if (h_high <= p.H - 1 && w_low >= 0) w3 = lh * hw_ * mask;
                    if (h_high <= p.H - 1 && w_high <= p.W - 1) w4 = lh * lw * mask;
                    // h_low in [-1, H-1], w_low in [-1, W-1] here: clamp the four corners into the map
                    const int hl_c = h_low < 0 ? 0 : h_low, wl_c = w_low < 0 ? 0 : w_low;
                    const int hh_c = h_high > p.H - 1 ? p.H - 1 : h_high, wh_c = w_high > p.W - 1 ? p.W - 1 : w_high;
                    o1 = (hl_c * p.W + wl_c) * (p.ldx * 4) | (wh_c - wl_c) | ((hh_c - hl_c) << 1);       // pixel stride is a multiple of 16 bytes
                }
            }
            *(f32x4*)(prm + idx * 4) = f32x4{w1, w2, w3, w4};
            pof[idx] = o1;
        }
        __syncthreads();
    }

    // ---- staging registers.  issue_loads() only ISSUES memory operations (no arithmetic on the
    // returned data), finish_store() does the per-mode arithmetic and the LDS stores: the MFMAs of
    // the current chunk sit between the two, so the loads fly under them. ----
    const bool uniform_tap = MODE != MODE_CONV || p.KH * p.KW == 1 || p.Cin >= 32;
    const unsigned inv_kw = (65536u + (unsigned)p.KW - 1u) / (unsigned)p.KW;
    // this workgroup's share of the K chunks: [k_lo, k_lo + nk) of Kpad/32 (everything when S == 1)
    const int nk_all = p.Kpad >> 5;
    const int k_lo = (int)((long long)nk_all * split / S);
    const int nk = (int)((long long)nk_all * (split + 1) / S) - k_lo;
    KCursor cur = {0, 0, 0};
    if (SPLIT) {                                     // position the (tap, channel) cursor on chunk k_lo
        if (MODE == MODE_DCN) {
            cur.c0 = (k_lo / 9) * 32; cur.s = k_lo % 9;
        } else if (MODE == MODE_CONV && uniform_tap) {
            const int taps = p.KH * p.KW;
            int tap;
            if (taps == 1) { tap = 0; cur.c0 = k_lo * 32; }
            else if (p.korder == 1) { tap = k_lo % taps; cur.c0 = (k_lo / taps) * 32; }
            else { tap = (k_lo * 32) >> p.cin_log2; cur.c0 = (k_lo * 32) & (p.Cin - 1); }
            cur.r = tap / p.KW; cur.s = tap - cur.r * p.KW;
        }
    }
    int kload = k_lo * 32;                           // k offset of the next chunk to load
    f32x4 s0[GA], s1[GA], s2[GA], s3[GA], sw[GA], vb[GB];

    // PREC 2: this wave's pieces of the weight chunk image (1 KB each; piece j of the tile = 64-row block j / 12)
    constexpr int BSLOTS = DEFT_NP * 4;                  // 16-byte slots per weight row
    constexpr int NBP = BDMA ? BN * DEFT_NP / 64 : 1;
    const deft_rsrc_t rw3 = deft_make_rsrc(BDMA ? p.w3 : (const void*)p.w);
    unsigned vB3[NBP];
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        const int j = wave + i * 4;
        vB3[i] = (unsigned)(((n0 >> 6) + j / BSLOTS) * (p.Kpad >> 5) * BSLOTS + j % BSLOTS) * 1024u + (unsigned)lane * 16u;
    }
    int bdma_stage = 0;                              // PREC 2: the B stage the next issue_loads() fills
    int dma_stage = 0;                               // DMA form: LDS stage the next issue_loads() fills
    auto issue_loads = [&]() {
        // Weight DMA FIRST (PREC 2).  LDS-DMA loads and ordinary (VGPR) loads share vmcnt but need not complete in issue order
        // relative to each other; the compiler's partial waits (s_waitcnt vmcnt(N > 0) in front of finish_store()'s first use of a
        // staged register) assume in-order return.  With the DMAs OLDER than the register loads those waits stay sufficient (N counts
        // only younger register loads, and completions within one kind are in order); with the DMAs younger, a DMA that lands early
        // could satisfy the count while the register load is still in flight.
        if (BDMA) {
            const unsigned soff = (unsigned)(kload >> 5) * (unsigned)(64 * BROW);
#pragma unroll
            for (int i = 0; i < NBP; ++i)
                deft_buffer_load_lds_x4s(rw3, Bd + bdma_stage * BN * BROW + (wave + i * 4) * 1024, vB3[i], soff);
            bdma_stage ^= 1;
        }
        if (MODE == MODE_CONV) {
            int r, s, toff;                                  // tap of this lane's k-slot, byte offset of (tap, channel)
            bool kok;
            if (uniform_tap) {
                r = cur.r; s = cur.s;
                toff = ((r * p.W + s) * p.ldx + cur.c0) * 4;                     // scalar unit
                kok = cur.c0 + gs * 4 < p.Cin;   // 1x1: c is the flat k -> masks the K padding; k>1: always true
            } else {                         // Cin in {4, 8, 16}: a chunk spans several taps -> per-lane tap
                const int kflat = kload + gs * 4;
                const int tap = kflat >> p.cin_log2;
                r = (int)(((unsigned)tap * inv_kw) >> 16);      // tap / KW for tap < 64 (host checks the range)
                s = tap - r * p.KW;
                toff = ((r * p.W + s) * p.ldx + (kflat & (p.Cin - 1)) - gs * 4) * 4;   // rb already holds the gs*16 term
                kok = kflat < p.Ktot;
            }
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const int iy = r0[i] + r, ix = r1[i] + s;
                const bool ok = kok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const unsigned off = (unsigned)(rb[i] + toff);
                if (DMA) deft_buffer_load_lds_x4(rx, As + dma_stage * BM * 32 + (wave * 8 + 32 * i) * 32, ok ? off : DEFT_OOB);
                else s0[i] = deft_buffer_load_x4(rx, ok ? off : DEFT_OOB);
            }
        } else if (MODE == MODE_DCN) {
            const int tap = cur.s;                       // chunk j = (channel block cur.c0/32, tap cur.s)
            const int cb4 = cur.c0 * 4;
            const int dcn_dx = p.ldx * 4, dcn_dy = p.W * p.ldx * 4;
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const int rec = tap * BM + rbase + 32 * i;
                sw[i] = *(const f32x4*)(prm + rec * 4);
                const int po = pof[rec];
                const int o1 = rb[i] + cb4 + (po & ~3);
                const int o2 = o1 + ((po & 1) ? dcn_dx : 0);
                const int dy = (po & 2) ? dcn_dy : 0;
                s0[i] = deft_buffer_load_x4(rx, (unsigned)o1);
                s1[i] = deft_buffer_load_x4(rx, (unsigned)o2);
                s2[i] = deft_buffer_load_x4(rx, (unsigned)(o1 + dy));
                s3[i] = deft_buffer_load_x4(rx, (unsigned)(o2 + dy));
            }
        } else {
            const unsigned kb = (unsigned)((kload + g * 4) * 4);
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                s0[i] = deft_buffer_load_x4(rx, (unsigned)r0[i] + kb);
                s1[i] = deft_buffer_load_x4(rx2, (unsigned)r1[i] + kb);
            }
        }
        if (BDMA) {
            // (issued at the top of this function)
        } else if (DMA) {
            const unsigned wo = (unsigned)(((n0 + rbase) * p.Kpad + kload + gs * 4) * 4);
#pragma unroll
            for (int i = 0; i < GB; ++i)
                deft_buffer_load_lds_x4(rx2, Bs + dma_stage * BN * 32 + (wave * 8 + 32 * i) * 32, wo + (unsigned)(32 * i * p.Kpad * 4));
        } else {
            const float* wp = p.w + (unsigned)((n0 + rbase) * p.Kpad + kload + g * 4);
#pragma unroll
            for (int i = 0; i < GB; ++i) vb[i] = *(const f32x4*)(wp + (unsigned)(32 * i * p.Kpad));
        }
        kload += 32;
        if (MODE == MODE_CONV) {                     // advance the (tap, channel) cursor: scalar unit only
            if (p.korder == 1) {                     // (32-channel block, r, s, c): taps fastest, see deft_hip.h
                if (++cur.s == p.KW) {
                    cur.s = 0;
                    if (++cur.r == p.KH) { cur.r = 0; cur.c0 += 32; }
                }
            } else {
                cur.c0 += 32;
                if (p.KH * p.KW > 1 && cur.c0 >= p.Cin) {
                    cur.c0 = 0;
                    if (++cur.s == p.KW) { cur.s = 0; ++cur.r; }
                }
            }
        } else if (MODE == MODE_DCN) {               // (channel block, tap) order: tap fastest
            if (++cur.s == 9) { cur.s = 0; cur.c0 += 32; }
        }
    };
    auto finish_store = [&](int stage) {
        float* as = As + stage * BM * LDS_STRIDE + rbase * LDS_STRIDE + g * 4;
        float* bs = Bs + stage * BN * LDS_STRIDE + rbase * LDS_STRIDE + g * 4;
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            f32x4 v;
            if (MODE == MODE_CONV) {
                v = s0[i];
            } else if (MODE == MODE_DCN) {      // (with PREC the operand scale DEFT_ASCALE is applied by deft_split below)
                v = sw[i].x * s0[i] + sw[i].y * s1[i] + sw[i].z * s2[i] + sw[i].w * s3[i];
            } else {
                const f32x4 t = s0[i] + s1[i];
                v = f32x4{deft_relu(t.x), deft_relu(t.y), deft_relu(t.z), deft_relu(t.w)};
            }
            if (PREC) {
                pcx4 pc[DEFT_NP];
                deft_split(v, pc, DEFT_ASCALE);
                deft_piece_t* ap = Ap + (rbase + 32 * i) * LDB + g * 4;
#pragma unroll
                for (int q = 0; q < DEFT_NP; ++q) *(pcx4*)(ap + q * BM * LDB) = pc[q];
            } else {
                *(f32x4*)&as[32 * i * LDS_STRIDE] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < GB; ++i) {
            if (BDMA) {
                // weights arrive by DMA
            } else if (PREC) {
                pcx4 pc[DEFT_NP];
                deft_split(vb[i], pc);
                deft_piece_t* bp = Bp + (rbase + 32 * i) * LDB + g * 4;
#pragma unroll
                for (int q = 0; q < DEFT_NP; ++q) *(pcx4*)(bp + q * BN * LDB) = pc[q];
            } else {
                *(f32x4*)&bs[32 * i * LDS_STRIDE] = vb[i];
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

    const int frow = lane & 31;        // row of the 32-row MFMA tile this lane feeds
    const int fk = (lane >> 5) * 16 + wk * KK;   // first of this lane's k-values in the chunk (this wave's share)
    float a[TM][KK], b[TN][KK];

    const int fsw = (frow >> 1) & 7;   // DMA image: swizzle term of this lane's fragment rows
    auto read_frags = [&](int stage) {
        const float* as = As + stage * BM * LD + (wm * TM * 32 + frow) * LD + (DMA ? 0 : fk);
        const float* bs = Bs + stage * BN * LD + (wn * TN * 32 + frow) * LD + (DMA ? 0 : fk);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int q = 0; q < KK / 4; ++q) {
                const f32x4 t = *(const f32x4*)&as[i * 32 * LD + (DMA ? (((fk >> 2) + q) ^ fsw) * 4 : 4 * q)];
                a[i][4 * q + 0] = t.x; a[i][4 * q + 1] = t.y; a[i][4 * q + 2] = t.z; a[i][4 * q + 3] = t.w;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < KK / 4; ++q) {
                const f32x4 t = *(const f32x4*)&bs[j * 32 * LD + (DMA ? (((fk >> 2) + q) ^ fsw) * 4 : 4 * q)];
                b[j][4 * q + 0] = t.x; b[j][4 * q + 1] = t.y; b[j][4 * q + 2] = t.z; b[j][4 * q + 3] = t.w;
            }
        }
    };
    auto mfma_chunk = [&]() {
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][kk], b[j][kk], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);       // nothing of finish_store() may be scheduled above the MFMAs
    };

    // PREC = 1: the two K = 16 halves of the chunk; a lane's fragment = 8 consecutive k of its row (k group lane>>5)
    auto split_chunk = [&](int bstage) {
        const int fkg = (lane >> 5) * 8;
        const char* const bdr = Bd + bstage * BN * BROW + (wn * TN * 32 + frow) * BROW;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            pcx8 pa[TM][DEFT_NP], pb[TN][DEFT_NP];
#pragma unroll
            for (int pl = 0; pl < DEFT_NP; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) pa[i][pl] = *(const pcx8*)(Ap + pl * BM * LDB + ((wm * TM + i) * 32 + frow) * LDB + kh * 16 + fkg);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    pb[j][pl] = BDMA ? *(const pcx8*)(bdr + j * 32 * BROW + deft_p3_phys(pl, kh * 2 + (lane >> 5), frow) * 16)
                                     : *(const pcx8*)(Bp + pl * BN * LDB + ((wn * TN + j) * 32 + frow) * LDB + kh * 16 + fkg);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];                                           // smallest terms first
#pragma unroll
                    for (int q = 0; q < DEFT_NPROD; ++q) c = deft_mfma_pc(pa[i][deft_qa(q)], pb[j][deft_qb(q)], c);
                    acc[i][j] = c;
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    if (NSTAGE == 1) {
        // one LDS stage, two barriers per chunk:  store chunk kt | barrier | issue loads kt+1,
        // fragments + MFMAs of chunk kt | barrier
        issue_loads();
        for (int kt = 0; kt < nk; ++kt) {
            finish_store(0);
            if (BDMA) DEFT_WAIT_VM(0);           // this wave's pieces of chunk kt's weight image have landed (the barrier publishes everybody's)
            __syncthreads();
            if (kt + 1 < nk) issue_loads();      // (PREC 2: the weight DMA of chunk kt+1 goes to the stage last read in iteration kt-1)
            if (PREC) {
                split_chunk(kt & 1);
            } else {
                read_frags(0);
                mfma_chunk();
            }
            __syncthreads();  // all waves finished reading this chunk
        }
    } else {
        // two LDS stages, ONE barrier per chunk.  Iteration kt: issue the loads of chunk kt+1,
        // fragments + MFMAs of chunk kt (stage kt&1), then finish chunk kt+1 into the other stage.
        // That stage was last read in iteration kt-1 and is next read in iteration kt+1, each
        // separated from the store by a barrier.
        // (CONV: issue_loads() is the LDS-DMA itself, finish_store() is empty, and the barrier's
        // s_waitcnt vmcnt(0) is what makes chunk kt+1 visible.)
        issue_loads();
        if (!DMA) finish_store(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cs = kt & 1;
            dma_stage = cs ^ 1;
            if (kt + 1 < nk) issue_loads();
            read_frags(cs);
            mfma_chunk();
            if (!DMA && kt + 1 < nk) finish_store(cs ^ 1);
            __syncthreads();
        }
    }

    if (WK > 1) {
        // both loop forms end on a barrier: the staging region is free.  Waves wk > 0 park their
        // partial tiles (lane-contiguous: conflict-free), wave wk == 0 adds them in wk order.
        float* red = smem;
        const int tbase = ((wave % (WM * WN)) * TM * TN) * 1024 + lane;
        constexpr int per_wk = WM * WN * TM * TN * 1024;
        if (wk > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(wk - 1) * per_wk + tbase + (i * TN + j) * 1024 + r * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int w = 1; w < WK; ++w)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(w - 1) * per_wk + tbase + (i * TN + j) * 1024 + r * 64];
    }

    if (SPLIT && S > 1) {               // (S == 1: a group of a grouped launch that is not split)
        // Park this workgroup's partial tile (lane-contiguous: 256 B per store instruction), make it visible
        // device-wide, then take a ticket: the last of the tile's S workgroups adds the partials in split
        // order 0..S-1 (its own included, re-read: the order must not depend on who arrives last).
        const int tile_id = mt * ntiles + nt;
        const size_t tsz = (size_t)BM * BN;
        const int woff = ((wave % (WM * WN)) * TM * TN) * 1024 + lane;
        float* mine = p.ws + ((size_t)tile_id * S + split) * tsz + woff;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) deft_ws_store(&mine[(i * TN + j) * 1024 + r * 64], acc[i][j][r]);
        deft_ws_publish();
        __syncthreads();                 // (waves wk > 0 of a split-K tile have already left)
        int* ticket = (int*)smem;
        if (tid == 0) ticket[0] = deft_ws_ticket(&p.ws_cnt[tile_id]);
        __syncthreads();
        if (ticket[0] != S - 1) return;
        if (tid == 0) deft_ws_reset(&p.ws_cnt[tile_id]);             // ready for the next launch that shares the counters
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

    // P3 output requested: the LDS-transposed epilogue of common.h, which also writes the three bf16 pieces for a following
    // pre-split conv (the launch sizes the dynamic LDS for the BM x (BN + 4) tile when y3 is set: launch_igemm)
    if constexpr ((MODE == MODE_DCN || (MODE == MODE_CONV && PREC >= 1)) && WK == 1) {
        if (p.y3 != nullptr) {
            float* const T = smem;
            __syncthreads();
            deft_epilogue_stage<TM, TN>(T, BN + 4, acc, wm, wn, lane, p, n0, PREC ? DEFT_ASCALE_INV : 1.f);
            __syncthreads();
            deft_epilogue_rows<BM, BN, 256>(T, p, n0, tid, [&](int row) -> long long { return m0 + row < p.M ? (long long)(m0 + row) : -1; });
            return;
        }
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
        const float sc = (p.scale ? p.scale[coc] : 1.f) * (PREC ? DEFT_ASCALE_INV : 1.f);      // (the split path scaled its A operand by DEFT_ASCALE)
        const float sh = p.shift ? p.shift[coc] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int q = 0; q < 4; ++q) {       // 4 rows at a time keeps the epilogue's VGPR footprint small
                float rv0 = 0.f, rv1 = 0.f, rv2 = 0.f, rv3 = 0.f;
                const int mq = mb + 8 * q;
                if (has_res) {
                    const int mc = p.M - 1;
                    rv0 = p.res[(size_t)(mq + 0 < p.M ? mq + 0 : mc) * p.ldr + coc];
                    rv1 = p.res[(size_t)(mq + 1 < p.M ? mq + 1 : mc) * p.ldr + coc];
                    rv2 = p.res[(size_t)(mq + 2 < p.M ? mq + 2 : mc) * p.ldr + coc];
                    rv3 = p.res[(size_t)(mq + 3 < p.M ? mq + 3 : mc) * p.ldr + coc];
                }
                float v0 = acc[i][j][4 * q + 0] * sc + sh + rv0;
                float v1 = acc[i][j][4 * q + 1] * sc + sh + rv1;
                float v2 = acc[i][j][4 * q + 2] * sc + sh + rv2;
                float v3 = acc[i][j][4 * q + 3] * sc + sh + rv3;
                if (p.relu) { v0 = deft_relu(v0); v1 = deft_relu(v1); v2 = deft_relu(v2); v3 = deft_relu(v3); }
                float* yp = p.y + (size_t)mq * p.ldy + co;
                if (cok && mq + 0 < p.M) yp[0] = v0;
                if (cok && mq + 1 < p.M) yp[(size_t)p.ldy] = v1;
                if (cok && mq + 2 < p.M) yp[(size_t)2 * p.ldy] = v2;
                if (cok && mq + 3 < p.M) yp[(size_t)3 * p.ldy] = v3;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int WK, int MODE, int NSTAGE, bool SPLIT, int PREC = 0>
__global__ __launch_bounds__(256) void igemm_kernel(DeftGemmDesc p, int mtiles, int ntiles) {
    igemm_body<BM, BN, WM, WN, WK, MODE, NSTAGE, SPLIT, PREC>(p, mtiles, ntiles, blockIdx.x);
}

// grouped form: blockIdx.y picks one of several independent problems (descriptors in device
// memory, same tile configuration); blocks beyond a problem's tile count exit at once.
template <int BM, int BN, int WM, int WN, int WK, int NSTAGE, bool SPLIT>
__global__ __launch_bounds__(256) void igemm_group_kernel(const DeftGemmDesc* __restrict__ descs) {
    DeftGemmDesc p = descs[blockIdx.y];
    if (SPLIT && p.splitk < 1) p.splitk = 1;
    const int mtiles = (p.M + BM - 1) / BM, ntiles = (p.Cout + BN - 1) / BN;
    if ((int)blockIdx.x >= mtiles * ntiles * (SPLIT ? p.splitk : 1)) return;
    igemm_body<BM, BN, WM, WN, WK, MODE_CONV, NSTAGE, SPLIT>(p, mtiles, ntiles, blockIdx.x);
}

template <auto KERNEL>
static int set_lds_attr(int lds_bytes) {
    static int cur = 64 * 1024;                // > 64 KB of dynamic LDS needs the opt-in, once per instantiation and size
    if (lds_bytes <= cur) return 0;
    hipError_t e = hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    DEFT_CHECK(e == hipSuccess, -101, "igemm: hipFuncSetAttribute(%d B LDS) failed: %s", lds_bytes, hipGetErrorString(e));
    cur = lds_bytes;
    return 0;
}

template <int BM, int BN, int WM, int WN, int WK, int MODE, int NSTAGE>
static int launch_igemm(const DeftGemmDesc& d, const DeftGemmDesc* group_dev, int ngroups, int max_tiles, bool group_split, hipStream_t s) {
    constexpr int lds_bytes = igemm_lds_floats<BM, BN, WK, NSTAGE, MODE>() * 4;
    if (group_dev != nullptr) {
        if constexpr (MODE == MODE_CONV) {
            if (group_split) {
                if (int e = set_lds_attr<igemm_group_kernel<BM, BN, WM, WN, WK, NSTAGE, true>>(lds_bytes)) return e;
                hipLaunchKernelGGL((igemm_group_kernel<BM, BN, WM, WN, WK, NSTAGE, true>), dim3(max_tiles, ngroups), dim3(256), lds_bytes, s, group_dev);
            } else {
                if (int e = set_lds_attr<igemm_group_kernel<BM, BN, WM, WN, WK, NSTAGE, false>>(lds_bytes)) return e;
                hipLaunchKernelGGL((igemm_group_kernel<BM, BN, WM, WN, WK, NSTAGE, false>), dim3(max_tiles, ngroups), dim3(256), lds_bytes, s, group_dev);
            }
        } else {
            DEFT_CHECK(false, -103, "igemm: grouped launches are conv only");
        }
    } else {
        const int mtiles = deft_cdiv(d.M, BM), ntiles = deft_cdiv(d.Cout, BN);
        const int S = d.splitk > 1 ? d.splitk : 1;
        DEFT_CHECK(d.y3 == nullptr || WK == 1, -9, "igemm: y3 (P3 output) is not available on the intra-workgroup split-K tiles (%dx%d)", BM, BN);
        const int lds_y3 = d.y3 != nullptr ? BM * (BN + 4) * 4 : 0;          // the LDS-transposed epilogue's tile
        DEFT_CHECK(S == 1 || (d.ws != nullptr && d.ws_cnt != nullptr && MODE != MODE_PAIR && S <= 32 && (d.Kpad >> 5) >= S), -102,
                   "igemm: splitk=%d needs ws and ws_cnt, conv/dcn, S <= 32 and at least S K chunks (%d)", S, d.Kpad >> 5);
        if constexpr (WK == 1 && NSTAGE == 1 && BN >= 64) {      // BN = 32: the operand split is amortised over too few columns
            if (MODE != MODE_DCN && d.prec == 1 && d.w3 != nullptr && S == 1) {      // ... with the weights pre-split: their chunks arrive by DMA
                constexpr int PB = MODE == MODE_DCN ? 1 : 2;     // (never the DCN: its pre-split-weight form is the patch kernel, dcn.hip)
                constexpr int lds_p = igemm_lds_floats<BM, BN, WK, NSTAGE, MODE, PB>() * 4;
                if (int e = set_lds_attr<igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, false, PB>>(lds_p > lds_y3 ? lds_p : lds_y3)) return e;
                hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, false, PB>), dim3(mtiles * ntiles), dim3(256), lds_p > lds_y3 ? lds_p : lds_y3, s, d, mtiles, ntiles);
                DEFT_CHECK_LAUNCH("igemm");
                return 0;
            }
            if (d.prec == 1) {                   // fp32 through the bf16 matrix cores (DeftGemmDesc.prec)
                constexpr int lds_p = igemm_lds_floats<BM, BN, WK, NSTAGE, MODE, 1>() * 4;
                if (S > 1) {
                    if constexpr (MODE != MODE_PAIR) {
                        if (int e = set_lds_attr<igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, true, 1>>(lds_p > lds_y3 ? lds_p : lds_y3)) return e;
                        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, true, 1>), dim3(mtiles * ntiles * S), dim3(256), lds_p > lds_y3 ? lds_p : lds_y3, s, d,
                                           mtiles, ntiles);
                    }
                } else {
                    if (int e = set_lds_attr<igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, false, 1>>(lds_p > lds_y3 ? lds_p : lds_y3)) return e;
                    hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, false, 1>), dim3(mtiles * ntiles), dim3(256), lds_p > lds_y3 ? lds_p : lds_y3, s, d,
                                       mtiles, ntiles);
                }
                DEFT_CHECK_LAUNCH("igemm");
                return 0;
            }
        }
        // (from here on: the fp32-MFMA instantiations; their plain-conv form has no P3 epilogue)
        DEFT_CHECK(d.y3 == nullptr || MODE == MODE_DCN, -9, "igemm: y3 (P3 output) needs the split-bf16 arithmetic (prec 1, a 1-stage tile with BN >= 64); tile %dx%d prec %d", BM, BN, d.prec);
        if (S > 1) {
            if constexpr (MODE != MODE_PAIR) {
                if (int e = set_lds_attr<igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, true>>(lds_bytes > lds_y3 ? lds_bytes : lds_y3)) return e;
                hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, true>), dim3(mtiles * ntiles * S), dim3(256), lds_bytes > lds_y3 ? lds_bytes : lds_y3, s, d,
                                   mtiles, ntiles);
            }
        } else {
            if (int e = set_lds_attr<igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, false>>(lds_bytes > lds_y3 ? lds_bytes : lds_y3)) return e;
            hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, WK, MODE, NSTAGE, false>), dim3(mtiles * ntiles), dim3(256), lds_bytes > lds_y3 ? lds_bytes : lds_y3, s, d,
                               mtiles, ntiles);
        }
    }
    DEFT_CHECK_LAUNCH("igemm");
    return 0;
}

// `tile` knob: bits 0-15 BN, bits 16-28 BM (0 = automatic), bit 29 = use the 2-stage (1-barrier,
// double LDS) loop instead of the default 1-stage (2-barrier) loop.  64x32 and 32x32 are the
// split-K tiles (WK = 2 / 4).
template <int MODE>
static int dispatch_igemm(const DeftGemmDesc& d, int bm, int bn, bool one_stage, hipStream_t s,
                          const DeftGemmDesc* group_dev = nullptr, int ngroups = 0, int mt = 0, bool gsplit = false) {
#define DEFT_TILE(BM_, BN_, WM_, WN_, WK_)                                                                   \
    if (bm == BM_ && bn == BN_)                                                                              \
        return one_stage ? launch_igemm<BM_, BN_, WM_, WN_, WK_, MODE, 1>(d, group_dev, ngroups, mt, gsplit, s)      \
                         : launch_igemm<BM_, BN_, WM_, WN_, WK_, MODE, 2>(d, group_dev, ngroups, mt, gsplit, s);
    DEFT_TILE(128, 128, 2, 2, 1)
    DEFT_TILE(128, 64, 2, 2, 1)
    DEFT_TILE(128, 32, 4, 1, 1)
    DEFT_TILE(64, 64, 2, 2, 1)
    DEFT_TILE(64, 128, 2, 2, 1)
    DEFT_TILE(64, 32, 2, 1, 2)
    DEFT_TILE(32, 32, 1, 1, 4)
#undef DEFT_TILE
    DEFT_CHECK(false, -15, "igemm: unsupported tile %dx%d", bm, bn);
    return -15;
}

static int check_common(const DeftGemmDesc* d, const char* who) {
    DEFT_CHECK(d != nullptr, -1, "%s: null descriptor", who);
    DEFT_CHECK((d->x || d->x3) && d->w && (d->y || d->y3 || (d->x3 && d->fold_y)), -2, "%s: null x/w/y pointer", who);
    DEFT_CHECK(d->fold_y == nullptr || d->x3 != nullptr, -2, "%s: fold_y is honoured by the pre-split conv kernels only (x3)", who);
    DEFT_CHECK(d->M > 0 && d->Cout > 0, -3, "%s: empty problem M=%d Cout=%d", who, d->M, d->Cout);
    DEFT_CHECK(d->Kpad > 0 && (d->Kpad & 31) == 0 && d->Ktot <= d->Kpad, -4, "%s: Kpad=%d must be a multiple of 32 >= Ktot=%d", who, d->Kpad, d->Ktot);
    DEFT_CHECK((d->ldx & 3) == 0 && (((size_t)d->x) & 15) == 0 && (((size_t)d->w) & 15) == 0, -5, "%s: x/w must be 16-byte aligned, ldx %% 4 == 0", who);
    DEFT_CHECK(d->ldy >= d->Cout, -6, "%s: ldy=%d < Cout=%d", who, d->ldy, d->Cout);
    DEFT_CHECK(!d->res || d->ldr >= d->Cout, -7, "%s: ldr=%d < Cout=%d", who, d->ldr, d->Cout);
    // loaders address with 32-bit element offsets
    DEFT_CHECK((long long)deft_cdiv(d->Cout, 128) * 128 * d->Kpad < (1ll << 31), -8, "%s: weight matrix exceeds 2^31 elements", who);
    return 0;
}

// blocks needed before a bigger tile is worth it: 2 workgroups on each of 256 CUs
#define FILL_BLOCKS 512

static int check_conv(const DeftGemmDesc* d, const char* who) {
    if (int e = check_common(d, who)) return e;
    DEFT_CHECK((d->Cin & 3) == 0, -10, "%s: Cin=%d must be a multiple of 4 (pad channels)", who, d->Cin);
    DEFT_CHECK(d->Ktot == d->KH * d->KW * d->Cin, -11, "%s: Ktot mismatch", who);
    DEFT_CHECK(d->KH * d->KW == 1 || ((d->Cin & (d->Cin - 1)) == 0 && (1 << d->cin_log2) == d->Cin), -12,
               "%s: Cin=%d must be a power of two for KHxKW>1", who, d->Cin);
    DEFT_CHECK(d->rowmap != nullptr || d->M == d->N * d->OH * d->OW, -13, "%s: M != N*OH*OW", who);
    DEFT_CHECK(d->rowmap == nullptr || (d->H < 65536 && d->W < 65536), -19, "%s: rowmap packs y/x in 16 bits", who);
    DEFT_CHECK(d->ldx >= d->Cin, -14, "%s: ldx < Cin", who);
    DEFT_CHECK((long long)d->N * d->H * d->W * d->ldx < (1ll << 29), -16, "%s: input exceeds 2 GiB (split the batch)", who);
    DEFT_CHECK(d->KW >= 1 && d->KW <= 16, -17, "%s: KW=%d out of range", who, d->KW);
    DEFT_CHECK(d->korder == 0 || (d->korder == 1 && (d->Cin & 31) == 0 && d->Kpad == d->Ktot), -20,
               "%s: korder=%d needs Cin %% 32 == 0 (Cin=%d)", who, d->korder, d->Cin);
    DEFT_CHECK(d->Cin >= 32 || d->KH * d->KW == 1 || d->Kpad / d->Cin <= 64, -18,
               "%s: Cin=%d < 32 supports at most 64 taps (incl. K padding)", who, d->Cin);
    return 0;
}

// automatic tile choice for a conv problem of M rows x Cout columns.  The split-K tile changes
// the fp32 summation order, so whether it is used depends on the rows PER IMAGE only: results
// are then bit-identical whatever the batch size (the WK = 1 tiles all sum in the same order) -- as long as no
// cross-workgroup split (DeftGemmDesc.splitk, chosen for launches with few tiles) is in play.
static void pick_conv_tile(int M, int rows_per_image, int Cout, int& bm, int& bn) {
    const long long m128 = deft_cdiv(M, 128);
    if (Cout <= 32) {
        bn = 32;
        bm = deft_cdiv(rows_per_image, 128) >= 256 ? 128 : 32;   // small maps: 32x32 split-K tile, 4 waves per
                                         // output tile (38x68x8 rows 68 -> 48 us, 19x34x8 rows 124 -> 54 us vs 128x32)
    } else if (Cout > 64 && m128 * deft_cdiv(Cout, 128) >= FILL_BLOCKS) { bm = 128; bn = 128; }
    else if (m128 * deft_cdiv(Cout, 64) >= FILL_BLOCKS) { bm = 128; bn = 64; }
    else { bm = 64; bn = 64; }
}

// split factor for a launch of `tiles` output tiles and nk K chunks: double S until the launch has two workgroups
// per compute unit (FILL_BLOCKS), keeping >= 8 chunks per workgroup
static int pick_splitk(long long tiles, int nk, int wgs_per_cu = 2) {
    static const int fill = [] { const char* e = getenv("DEFT_SPLIT_FILL"); return e ? atoi(e) : FILL_BLOCKS; }();   // tuning aid
    static const int minc = [] { const char* e = getenv("DEFT_SPLIT_MINCHUNKS"); return e ? atoi(e) : 8; }();
    int S = 1;
    while (tiles * S < fill * wgs_per_cu / 2 && nk / (2 * S) >= minc && S < 32) S *= 2;
    return S;
}

extern "C" int deft_gemm_plan(const DeftGemmDesc* d, int entry, int* tile, int* splitk, long long* ws_floats, int* ws_tiles) {
    DEFT_CHECK(d && tile && splitk && ws_floats && ws_tiles, -1, "deft_gemm_plan: null pointer");
    DEFT_CHECK(entry == 0 || entry == 1, -2, "deft_gemm_plan: entry %d (0 = conv, 1 = dcn)", entry);
    int bm = (d->tile >> 16) & 0x1fff, bn = d->tile & 0xffff;
    if (bm == 0) {
        if (entry == 0 && d->x3 != nullptr) deft_p3_pick_tile(d, &bm, &bn);
        else if (entry == 0) pick_conv_tile(d->M, d->rowmap ? d->M : d->OH * d->OW, d->Cout, bm, bn);
        else { bm = 64; bn = d->Cout >= 128 ? 128 : 64; }
    }
    const int nk = d->Kpad >> 5;
    long long tiles = (long long)deft_cdiv(d->M, bm) * deft_cdiv(d->Cout, bn);
    // the 8-wave tiles of the pre-split kernel hold a whole CU each: one workgroup per CU fills the chip
    const int per_cu = (entry == 0 && d->x3 != nullptr && bm * bn >= 256 * 128) ? 1 : 2;
    int S = (d->rowmap != nullptr) ? 1 : pick_splitk(tiles, nk, per_cu);
    if (entry == 1 && S > 1 && bn == 128) {          // few tiles: the narrower DCN tile gives twice the workgroups per split
        bn = 64;
        tiles = (long long)deft_cdiv(d->M, bm) * deft_cdiv(d->Cout, bn);
        S = pick_splitk(tiles, nk);
    }
    *tile = (bm << 16) | bn | (d->tile & (3 << 29));
    *splitk = S;
    *ws_floats = S > 1 ? tiles * S * bm * bn : 0;
    *ws_tiles = S > 1 ? (int)tiles : 0;
    return 0;
}

extern "C" int deft_conv2d_nhwc(const DeftGemmDesc* d, void* stream) {
    if (int e = check_conv(d, "deft_conv2d_nhwc")) return e;
    hipStream_t s = (hipStream_t)stream;
    if (d->p3_kernel == 3) return deft_conv3p_dispatch(d, s);       // fp32 patch in LDS, operand split in registers (dcn.hip)
    if (d->x3 != nullptr) {                 // pre-split operands: the LDS-DMA kernel (igemm3.hip)
        if (int e = deft_p3_check(d, "deft_conv2d_nhwc")) return e;
        return d->p3_kernel == 1 ? deft_p3h_dispatch(d, s) : deft_p3_dispatch(d, s);
    }
    DEFT_CHECK(d->y3 == nullptr || (d->rowmap == nullptr && d->stride_w == 0 && (d->Cout & 31) == 0 && (d->ldy3 & 31) == 0 && d->ldy3 >= d->Cout && (d->ldy & 3) == 0 &&
                                    (!d->res || (d->ldr & 3) == 0) && (((size_t)d->y3 | (size_t)d->y | (size_t)d->res) & 15) == 0 && d->y != nullptr),
               -9, "deft_conv2d_nhwc: y3 (P3 output) needs a dense conv with Cout %% 32 == 0, ldy3 %% 32 == 0, ldy/ldr %% 4 == 0, 16-byte aligned y/y3/res");
    int bm = (d->tile >> 16) & 0x1fff, bn = d->tile & 0xffff;
    const bool one_stage = !((d->tile >> 29) & 1);
    if (bm == 0) pick_conv_tile(d->M, d->rowmap ? d->M : d->OH * d->OW, d->Cout, bm, bn);
    return dispatch_igemm<MODE_CONV>(*d, bm, bn, one_stage, s);
}

extern "C" int deft_conv2d_group(const DeftGemmDesc* descs, const DeftGemmDesc* descs_dev, int ngroups, void* stream) {
    DEFT_CHECK(descs && descs_dev && ngroups > 0 && ngroups <= 65535, -40, "deft_conv2d_group: bad arguments");
    int bm = (descs[0].tile >> 16) & 0x1fff, bn = descs[0].tile & 0xffff;
    const bool one_stage = !((descs[0].tile >> 29) & 1);
    if (bm == 0) { bm = 32; bn = 32; }   // one fixed (4 waves per tile) tile: few rows per group
    long long max_tiles = 0;
    bool any_split = false;
    for (int i = 0; i < ngroups; ++i) {
        if (int e = check_conv(descs + i, "deft_conv2d_group")) return e;
        const int S = descs[i].splitk > 1 ? descs[i].splitk : 1;       // per-group cross-workgroup split (own ws / ws_cnt each)
        DEFT_CHECK(S == 1 || (descs[i].ws && descs[i].ws_cnt && S <= 32 && (descs[i].Kpad >> 5) >= S), -102,
                   "deft_conv2d_group: group %d splitk=%d needs ws, ws_cnt, S <= 32 and at least S K chunks", i, S);
        any_split |= S > 1;
        const long long t = (long long)deft_cdiv(descs[i].M, bm) * deft_cdiv(descs[i].Cout, bn) * S;
        max_tiles = t > max_tiles ? t : max_tiles;
    }
    DEFT_CHECK(max_tiles < (1ll << 31), -41, "deft_conv2d_group: too many tiles");
    return dispatch_igemm<MODE_CONV>(descs[0], bm, bn, one_stage, (hipStream_t)stream, descs_dev, ngroups, (int)max_tiles, any_split);
}

extern "C" int deft_dcn_v2_nhwc(const DeftGemmDesc* d, void* stream) {
    if (int e = check_common(d, "deft_dcn_v2_nhwc")) return e;
    DEFT_CHECK(d->x2 != nullptr && d->ldom >= 27, -20, "deft_dcn_v2_nhwc: offset/mask map missing or ldom < 27");
    DEFT_CHECK(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, -21, "deft_dcn_v2_nhwc: only 3x3/s1/p1 (dla.py:652-660)");
    DEFT_CHECK((d->Cin & 31) == 0 && (d->Cin & (d->Cin - 1)) == 0 && (1 << d->cin_log2) == d->Cin, -22,
               "deft_dcn_v2_nhwc: Cin=%d must be a power of two >= 32", d->Cin);
    DEFT_CHECK(d->Ktot == 9 * d->Cin && d->Kpad == d->Ktot, -23, "deft_dcn_v2_nhwc: Ktot/Kpad mismatch");
    DEFT_CHECK(d->OH == d->H && d->OW == d->W && d->M == d->N * d->H * d->W, -24, "deft_dcn_v2_nhwc: geometry mismatch");
    DEFT_CHECK((long long)d->N * d->H * d->W * d->ldx < (1ll << 29), -25, "deft_dcn_v2_nhwc: input exceeds 2 GiB (split the batch)");
    DEFT_CHECK(d->y3 == nullptr || ((d->Cout & 31) == 0 && (d->ldy3 & 31) == 0 && d->ldy3 >= d->Cout && (d->ldy & 3) == 0 && (((size_t)d->y3 | (size_t)d->y) & 15) == 0), -26,
               "deft_dcn_v2_nhwc: y3 needs Cout %% 32 == 0, ldy3 %% 32 == 0, ldy %% 4 == 0, 16-byte aligned outputs");
    hipStream_t s = (hipStream_t)stream;
    if (d->p3_kernel == 2) return deft_dcnp_dispatch(d, s);           // patch form (dcn.hip)
    int bm = (d->tile >> 16) & 0x1fff, bn = d->tile & 0xffff;
    const bool one_stage = !((d->tile >> 29) & 1);
    if (bm == 0) {             // BM = 64: 11.5 KB of sampling records, 39 KB of LDS in all -- four 64x64 (three 64x128) workgroups per CU
        bm = 64; bn = d->Cout >= 128 ? 128 : 64;      // tools/bench_igemm.py dcn (r2, weights by DMA: 64x128 wins from Cout = 128)
    }
    return dispatch_igemm<MODE_DCN>(*d, bm, bn, one_stage, s);
}

extern "C" int deft_pair_layer(const DeftGemmDesc* d, void* stream) {
    if (int e = check_common(d, "deft_pair_layer")) return e;
    DEFT_CHECK(d->x2 != nullptr && d->Q > 0 && d->M % d->Q == 0, -30, "deft_pair_layer: need V' and M %% Q == 0");
    DEFT_CHECK(d->Kpad == d->Ktot && d->ldx >= d->Ktot && (((size_t)d->x2) & 15) == 0, -31, "deft_pair_layer: K must be a multiple of 32 and <= ldx");
    hipStream_t s = (hipStream_t)stream;
    int bm = (d->tile >> 16) & 0x1fff, bn = d->tile & 0xffff;
    const bool one_stage = !((d->tile >> 29) & 1);
    if (bm == 0) {
        bm = 128;
        bn = (d->Cout > 64 && (long long)deft_cdiv(d->M, 128) * deft_cdiv(d->Cout, 128) >= FILL_BLOCKS) ? 128 : 64;
    }
    return dispatch_igemm<MODE_PAIR>(*d, bm, bn, one_stage, s);
}
