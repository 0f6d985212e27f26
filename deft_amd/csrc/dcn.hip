// DCNv2 main contraction for gfx950, "patch" form (DeftGemmDesc.p3_kernel = 2): replaces dcn_v2.DCN.forward's modulated deformable
// im2col + GEMM (third-party CharlesShang/DCNv2: modulated_deformable_im2col_cuda + dmcn_im2col_bilinear; imported dla.py:25-29,
// constructed dla.py:652-660, called dla.py:663) together with DeformConv.actf (dla.py:649-651).
//
// Why another form next to igemm.hip's MODE_DCN.  There the deformed im2col tile (the A operand) is gathered from global memory -- four
// corner loads per sample --, blended, split into its three bf16 pieces and staged through LDS, once per 64-row tile and K chunk; measured
// on MI355X (profiles/r2_*): the matrix pipe 27 % busy, the CU's L1 path carrying 24 B/clk of corner reads, the LDS port as busy with A/B
// fragment reads as the matrix cores with the MFMAs they feed.  Here
//   * a workgroup owns an 8 x 16 pixel output tile (128 GEMM rows, one MFMA row per lane) and all BN output channels of it;
//   * per 16 input channels it stages the fp32 input PATCH that offsets of up to +-R pixels can reach -- (8 + 2 + 2R) x (16 + 2 + 2R)
//     pixels x 64 B -- ONCE by LDS-DMA and takes all nine taps out of it: the corner reads are ds_read_b128 (256 B/clk/CU) instead of
//     L1 traffic, 1/15 of the global bytes.  A sample whose corners leave the patch (|offset| >= R) falls back to global loads, per lane;
//   * the blended A fragment never goes through LDS: lane l of a wave computes exactly the 8 k-values of row l & 31 that
//     v_mfma_f32_32x32x16_bf16 wants from it (k group l >> 5), blends, splits and feeds them from registers -- no ds_write, no A
//     fragment read, no duplicate gather between waves;
//   * the sampling records (four corner weights x sigmoid(mask), patch address) of the lane's row live in 45 VGPRs for the whole K
//     loop (the nine taps are unrolled), not in LDS;
//   * the weights arrive pre-split by LDS-DMA (deft_split_weights_dcn image: one 16-wide K chunk of 64 output channels = 6 KB, lane-
//     linear = conflict-free for the fragment reads), double buffered, one barrier per tap.
// Arithmetic: the prec = 1 arithmetic of igemm.hip (three bf16 pieces per fp32 operand, six v_mfma_f32_32x32x16_bf16 products, fp32
// accumulation); K order (16-channel block, tap, channel) -- another fp32 summation order than MODE_DCN, same error level.
// Sampling rule (upstream dmcn_im2col_bilinear / modulated_deformable_im2col_gpu_kernel, restated in oracle/dcn_scalar.py):
//   h_im = oy - 1 + r + dy, w_im likewise; zero outside (-1, H) x (-1, W); four-corner bilinear with every corner outside the map
//   dropped; the value times sigmoid(mask).
#include <cstdlib>

#include "common.h"

typedef deft_f32x16 f32x16;

#ifndef DCNP_R2
#define DCNP_R2 3
#endif
#ifndef DCNP_R4
#define DCNP_R4 (DEFT_NP == 2 ? 3 : 2)      // 128-column tiles: the margin that still leaves two workgroups per CU (72 KB with two fp16 pieces; 76 KB at R = 2 with three bf16 pieces)
#endif
#ifndef DCNP_GA
#define DCNP_GA 2          // corner reads run this many chunks ahead of the MFMAs (1 or 2): a far sample's global loads have a whole step to land
#endif

// Instruction-order request for the pipelined step: one MFMA, then a few VALU instructions of the next chunk's blend + split, repeated
// (the matrix pipe works 32 cycles per MFMA: the VALU work issues under it).
#if !defined(__HIP_DEVICE_COMPILE__)
#define DCNP_SCHED(TN) do {} while (0)
#else
#define DCNP_SCHED(TN)                                                        \
    do {                                                                      \
        _Pragma("unroll") for (int i_ = 0; i_ < DEFT_NPROD * (TN); ++i_) {    \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                \
            __builtin_amdgcn_sched_group_barrier(0x002, (TN) == 1 ? 7 : (DEFT_NP == 3 ? 12 : 16) / (TN), 0); \
        }                                                                     \
    } while (0)
#endif

#define DP_TH 8
#define DP_TW 16
// patch geometry for a margin of R pixels (offsets of up to +-R stay inside the patch)
#define DP_PH_(R) (DP_TH + 2 + 2 * (R))
#define DP_PW_(R) (DP_TW + 2 + 2 * (R))
#define DP_NPIX_(R) (DP_PH_(R) * DP_PW_(R))
#define DP_PARTS_(R) ((DP_NPIX_(R) + 63) / 64)            // 1 KB DMA pieces per plane (64 pixels x 16 B each)
#define DP_PLANE_(R) (DP_PARTS_(R) * 1024)                // one 4-channel plane of the patch: [pixel][16 B]
#define DP_PBUF_(R) (4 * DP_PLANE_(R))                    // one patch buffer: 16 channels = 4 planes
#define DP_WBLK (DEFT_NP * 2048)                      // one K chunk (16) of 64 output channels: [DEFT_NP pieces][2 k groups][64 rows][8 halves]

// (Round-4 / round-5 experiment arms -- weights global -> registers, far loads outside the compiler's vmcnt scoreboard, accumulators in
// AGPRs, the ablation and in-kernel timing builds -- are kept as profiles/r5_dcn_experiment_switches.patch, not in this file; what they
// measured: profiles/r4_dcn_experiments.md.)

template <int TN, int R>
constexpr int dcnp_lds_bytes() {
    constexpr int loop = 2 * DP_PBUF_(R) + 3 * ((TN + 1) / 2) * DP_WBLK;
    constexpr int tile = 128 * (TN * 32 + 4) * 4;                 // epilogue tile
    return loop > tile ? loop : tile;
}

// Row i (0..31) of a wave's MFMA tile <-> pixel (trow, tx) of the wave's two tile rows.  ds_read_b128 serves a wave in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): this map makes each group one tile row, i.e. 16 consecutive patch pixels = 256
// consecutive bytes of a plane when the offsets are equal -- no bank conflict.
__device__ __forceinline__ void dcnp_row_to_pixel(int i, int& trow, int& tx) {
    const int blk = i >> 2;
    // row 0: lanes 0-3 -> tx 0-3, 12-15 -> 4-7, 20-27 -> 8-15;  row 1: lanes 4-11 -> tx 0-7, 16-19 -> 8-11, 28-31 -> 12-15
    trow = (blk == 1 || blk == 2 || blk == 4 || blk == 7) ? 1 : 0;
    tx = i - (blk == 0 ? 0 : blk <= 2 ? 4 : blk == 3 ? 8 : blk == 4 ? 8 : blk <= 6 ? 12 : 16);
}

// (bf16(lo), bf16(hi)) packed into one register: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned dcnp_cvt_pk(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, p);
}

// Sampling record of (output pixel (oy, ox), tap): the four corner weights x sigmoid(mask) x DEFT_ASCALE and the corner address.  Near (all four
// corners inside the patch of margin R): c = LDS byte address, inside patch buffer 0, of the lane's first 16 bytes (plane 2 g) of the BASE pixel b;
// the corners are b, b + 1, b + PW, b + PW + 1 with the weights w1..w4 (a corner clamped at the map border is folded away: the base moves one
// pixel / line back and the weight of the dropped corner -- zero -- takes its place), so every corner read is `c + immediate`.  Far: c = bit 31 |
// global pixel of the clamped top-left corner << 2 | bit 0: the right-hand corners are one pixel on | bit 1: the lower corners one line on; weights
// in corner order.  (Branch-free: selects instead of nested ifs -- the nine records are 1/7 of the kernel's time at Cin = 64.)
template <int DP_R>
__device__ __forceinline__ void dcnp_record(const DeftGemmDesc& p, const f32x4 (&om)[7], const int tap, const int oy, const int ox, const bool rowok, const int img,
                                    const int py0, const int px0, const int g, float& w1, float& w2, float& w3, float& w4, unsigned& c) {
    constexpr int DP_PH = DP_PH_(DP_R), DP_PW = DP_PW_(DP_R), DP_PLANE = DP_PLANE_(DP_R);
    // (branch-free: selects instead of nested ifs -- the nine records are 1/7 of the kernel's time at Cin = 64)
    const float dy = om[(2 * tap) >> 2][(2 * tap) & 3], dx = om[(2 * tap + 1) >> 2][(2 * tap + 1) & 3];
    const float ml = om[(18 + tap) >> 2][(18 + tap) & 3];
    const int r = tap / 3, s = tap - 3 * r;
    const float h_raw = (float)(oy - 1 + r) + dy, w_raw = (float)(ox - 1 + s) + dx;
    const bool inside = rowok && h_raw > -1.f && w_raw > -1.f && h_raw < (float)p.H && w_raw < (float)p.W;
    const float h_im = inside ? h_raw : 0.f, w_im = inside ? w_raw : 0.f;
    const float hl = floorf(h_im), wl = floorf(w_im);
    const float lh = h_im - hl, lw = w_im - wl;
    const float hh = 1.f - lh, hw_ = 1.f - lw;
    const int h_low = (int)hl, w_low = (int)wl;
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float mask = inside ? DEFT_FAST_RCP(1.f + expf(-ml)) * DEFT_ASCALE : 0.f;     // (the operand scale of the split rides on the corner weights: exact, a power of two)
    const bool t_ok = h_low >= 0, b_ok = h_high <= p.H - 1, l_ok = w_low >= 0, r_ok = w_high <= p.W - 1;
    w1 = (t_ok && l_ok) ? hh * hw_ * mask : 0.f;
    w2 = (t_ok && r_ok) ? hh * lw * mask : 0.f;
    w3 = (b_ok && l_ok) ? lh * hw_ * mask : 0.f;
    w4 = (b_ok && r_ok) ? lh * lw * mask : 0.f;
    // h_low in [-1, H-1], w_low in [-1, W-1]: clamp the four corners into the map (a clamped corner has weight 0)
    const int hl_c = h_low < 0 ? 0 : h_low, wl_c = w_low < 0 ? 0 : w_low;
    const int hh_c = h_high > p.H - 1 ? p.H - 1 : h_high, wh_c = w_high > p.W - 1 ? p.W - 1 : w_high;
    const int fr = wh_c - wl_c, fd = hh_c - hl_c;
    const int bpy = hl_c - (1 - fd) - py0, bpx = wl_c - (1 - fr) - px0;          // base pixel of the near form
    const bool near = !inside || (bpy >= 0 && bpy + 1 < DP_PH && bpx >= 0 && bpx + 1 < DP_PW);
    if (near) {          // (weights only: selects)
        if (!fr) { w2 += w1; w1 = 0.f; w4 += w3; w3 = 0.f; }                   // (one of each pair is zero)
        if (!fd) { w3 += w1; w1 = 0.f; w4 += w2; w2 = 0.f; }
    }
    // (an unused record reads patch pixels 0, 1, PW, PW + 1 with weight 0)
    const unsigned c_near = (unsigned)(g * 2 * DP_PLANE) + (inside ? (unsigned)((bpy * DP_PW + bpx) * 16) : 0u);
    const unsigned c_far = 0x80000000u | (unsigned)((img + hl_c * p.W + wl_c) << 2) | (unsigned)(fr | (fd << 1));
    c = near ? c_near : c_far;
}

// DEFORM = false: the same structure as a PLAIN 3x3 / stride 1 / pad 1 convolution on an fp32 input (the DCN's own conv_offset_mask layer,
// 27 -> 32 output columns: TN = 1): the "records" are the nine fixed taps (one corner, weight 1: no blend, no offset map), the patch has no
// margin (R = 0).  It lets the offset conv read the SAME fp32 map the deformable gather reads -- no bf16-piece copy of the DCN's input has
// to exist (6 more bytes per element written by the producer: the upsample+add pass, the previous DCN).
template <int TN, int DP_R, bool DEFORM>
__global__ __launch_bounds__(256, DEFORM ? 2 : 3) void dcn_patch_kernel(DeftGemmDesc p, int tiles_x, int tiles_y, int ntiles) {
    static_assert(TN == 1 || TN == 2 || TN == 4, "32 (plain conv), 64 or 128 output channels per workgroup");
    static_assert(DEFORM || (TN == 1 && DP_R == 0), "the plain form is the 32-column conv without margin");
    constexpr int DP_PH = DP_PH_(DP_R), DP_PW = DP_PW_(DP_R), DP_NPIX = DP_NPIX_(DP_R), DP_PARTS = DP_PARTS_(DP_R), DP_PLANE = DP_PLANE_(DP_R),
                  DP_PBUF = DP_PBUF_(DP_R);
    constexpr int BN = TN * 32, NBLK = (BN + 63) / 64;
    constexpr int NBP = NBLK * DEFT_NP * 2;           // weight DMA pieces (1 KB) per chunk
    constexpr int BSTAGE = NBLK * DP_WBLK;
    constexpr int NPP = 4 * DP_PARTS / 2;             // patch DMA pieces per wave (waves 2 and 3 issue them)

    DEFT_DYN_LDS(char, smem);
    char* const patch = smem;                          // [2 buffers][4 planes][DP_PARTS * 64 pixels][16 B]
    char* const Bd = smem + 2 * DP_PBUF;               // [3 stages][BSTAGE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (in an SGPR: LDS-DMA destinations and scalar offsets derive from it)
    const int g = lane >> 5;

    // XCD-aware, bijective workgroup remap (see igemm.hip): every XCD gets one contiguous run of tiles, the n-tiles of a
    // pixel tile and neighbouring pixel tiles (shared halo) in the same L2
    int bid = blockIdx.x;
    {
        const int nwg = p.N * tiles_x * tiles_y * ntiles;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nt = bid % ntiles;
    int t = bid / ntiles;
    const int tpi = tiles_x * tiles_y;
    const int n = t / tpi;
    t -= n * tpi;
    const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
    const int ty0 = tyi * DP_TH, tx0 = txi * DP_TW, n0 = nt * BN;
    const int py0 = ty0 - 1 - DP_R, px0 = tx0 - 1 - DP_R;
    const int img = n * p.H * p.W;

    // ---- this lane's GEMM row ----
    int trow, tx;
    dcnp_row_to_pixel(lane & 31, trow, tx);
    const int oy = ty0 + 2 * wave + trow, ox = tx0 + tx;
    const bool rowok = oy < p.H && ox < p.W;
    // offsets + mask logits of the row (27 floats), requested first: they fly while the DMA sources are computed and issued
    f32x4 om[7];
    if (DEFORM) {
        const float* omp = p.x2 + (size_t)(rowok ? img + oy * p.W + ox : 0) * p.ldom;
#pragma unroll
        for (int q = 0; q < 7; ++q) om[q] = *(const f32x4*)(omp + 4 * q);
    }

    // ---- DMA sources.  Patch (waves 2, 3): piece j = (wave - 2) + 2 i = (plane j / PARTS, part j % PARTS): lane l deposits channel group
    // `plane` of patch pixel 64 part + l; pixels outside the map (and beyond the patch) arrive as zeros.  Weights (waves 0, 1). ----
    const deft_rsrc_t rx = deft_make_rsrc(p.x);
    const deft_rsrc_t rw = deft_make_rsrc(p.w3);
    unsigned pv[DP_PARTS];
#pragma unroll
    for (int i = 0; i < DP_PARTS; ++i) {
        const int pp = i * 64 + lane;
        const int py = pp / DP_PW, px = pp - py * DP_PW;
        const int gy = py0 + py, gx = px0 + px;
        const bool ok = pp < DP_NPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        pv[i] = ok ? (unsigned)((img + gy * p.W + gx) * p.ldx * 4) : DEFT_OOB;
    }
    const int ncb = p.Cin >> 4, nchunks = ncb * 9;
    const unsigned vB = (unsigned)((n0 >> 6) * nchunks * DP_WBLK + lane * 16);       // (+ piece and chunk terms in the scalar offset)
    // the first three weight chunks and the first patch are on their way while the records are computed
    // weight pieces per step and wave: waves 0, 1 / waves 2, 3 (which also carry the patch).  3 pieces: 6 * NBLK in all = 2 * (2 + 1) * NBLK;
    // 2 pieces: 4 * NBLK = 2 * (1 + 1) * NBLK
    constexpr int NB_A = DEFT_NP == 3 ? NBP / 3 : NBP / 4, NB_B = DEFT_NP == 3 ? NBP / 6 : NBP / 4;
    auto issue_b3 = [&](int kc, int st) {
        if (wave < 2) {
#pragma unroll
            for (int i = 0; i < NB_A; ++i) {
                const int jp = wave + 2 * i;
                deft_buffer_load_lds_x4s(rw, Bd + st * BSTAGE + jp * 1024, vB, (unsigned)((jp / (DEFT_NP * 2)) * nchunks + kc) * (unsigned)DP_WBLK + (unsigned)(jp % (DEFT_NP * 2)) * 1024u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB_B; ++i) {
                const int jp = 2 * NB_A + (wave - 2) + 2 * i;
                deft_buffer_load_lds_x4s(rw, Bd + st * BSTAGE + jp * 1024, vB, (unsigned)((jp / (DEFT_NP * 2)) * nchunks + kc) * (unsigned)DP_WBLK + (unsigned)(jp % (DEFT_NP * 2)) * 1024u);
            }
        }
    };
    // patch piece `idx` (0 .. 2 PARTS - 1) of this wave (waves 2, 3: planes 0, 1 / 2, 3) of block cb into buffer buf
    auto issue_patch_piece = [&](int cb, int buf, int idx) {
        const int plane = (wave & 1) * 2 + idx / DP_PARTS, part = idx % DP_PARTS;
        deft_buffer_load_lds_x4s(rx, patch + buf * DP_PBUF + plane * DP_PLANE + part * 1024, pv[part], (unsigned)(cb * 64 + plane * 16));
    };
    auto wait_vm = [&](int n) {          // (n is a compile-time constant wherever this is called)
        switch (n) {
            case 0: DEFT_WAIT_VM(0); break;
            case 1: DEFT_WAIT_VM(1); break;
            case 2: DEFT_WAIT_VM(2); break;
            case 3: DEFT_WAIT_VM(3); break;
            case 4: DEFT_WAIT_VM(4); break;
            case 5: DEFT_WAIT_VM(5); break;
            default: DEFT_WAIT_VM(6); break;
        }
    };
    // patch pieces a wave (2, 3) issues at the end of tap t of the block BEFORE the patch's block: its NPP pieces spread over taps 0 .. PT - 1
    // (the corner reads of the next block's tap 0 are issued GA steps ahead, and a piece has two steps to land)
    constexpr int GA = (TN == 2 && DEFORM) ? DCNP_GA : 1;      // how many steps ahead the corner reads run (two: 32 more VGPRs)
    constexpr int PT = 8 - GA;
    auto ps_of = [](int t) { return t >= PT ? NPP : NPP * t / PT; };
    static_assert(NB_B + (NPP + PT - 1) / PT <= 6 && NB_A <= 6, "wait_vm cases");

    issue_b3(0, 0);
    if (nchunks > 1) issue_b3(1, 1);
    if (nchunks > 2) issue_b3(2, 2);
    if (wave >= 2) {
#pragma unroll
        for (int i = 0; i < NPP; ++i) issue_patch_piece(0, 0, i);
    }
    // ---- sampling records of the row, all nine taps, in registers.  Near (all four corners inside the patch): rc = LDS byte address,
    // inside patch buffer 0, of the lane's first 16 bytes (plane 2 g) of the BASE pixel b; the corners are b, b + 1, b + PW, b + PW + 1
    // with the weights rw0..rw3 (a corner clamped at the map border is folded away: the base moves one pixel / line back and the weight
    // of the dropped corner -- zero -- takes its place), so every corner read is `rc + immediate`.  Far: rc = bit 31 | global pixel of the
    // clamped top-left corner << 2 | bit 0: the right-hand corners are one pixel on | bit 1: the lower corners one line on; weights in
    // corner order. ----
    float rw0[9], rw1[9], rw2[9], rw3[9];
    unsigned rc[9];
    {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (!DEFORM) {          // plain conv: tap (r, s) of the lane's pixel, always inside the margin-less patch (zeros outside the map)
                const int r = tap / 3, s = tap - 3 * r;
                rw0[tap] = DEFT_ASCALE; rw1[tap] = rw2[tap] = rw3[tap] = 0.f;
                rc[tap] = (unsigned)(g * 2 * DP_PLANE) + (unsigned)((((oy - ty0) + r) * DP_PW + (ox - tx0) + s) * 16);
                continue;
            }
            float w1, w2, w3, w4;
            unsigned c;
            dcnp_record<DP_R>(p, om, tap, oy, ox, rowok, img, py0, px0, g, w1, w2, w3, w4, c);
            rw0[tap] = w1; rw1[tap] = w2; rw2[tap] = w3; rw3[tap] = w4;
            rc[tap] = c;
        }
    }

    f32x16 acc[1][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    const unsigned far_x = (unsigned)p.ldx * 4u, far_y = (unsigned)(p.W * p.ldx) * 4u;
    const int brow = (lane & 31) * 16 + g * 1024;     // this lane's slot inside a (piece, k group) block of the weight image

    // corner values of one (row, tap, 16-channel block): v[corner][half of the lane's 8 channels].
    // EVERY lane reads the patch (a far lane some valid address: its values are overwritten); the far lanes then load their corners from
    // global memory into the SAME registers.  In this order the only wait between the two is for the LDS reads (the global loads
    // must not be overtaken by them: s_waitcnt lgkmcnt).  Round 3 had the two as the arms of one if / else: the compiler put the global
    // loads first and -- same destination registers -- a `s_waitcnt vmcnt(0)` in front of the LDS reads, i.e. every step of every wave
    // waited out a full global-memory round trip of the loads it had just issued (SQ_WAIT_ANY 0.40 of the wave cycles; rocprof round 4).
    auto gather = [&](unsigned c, int bufoff, int cb, f32x4 (&v)[4][2]) {
        if (!DEFORM) {
            const char* const a = patch + bufoff + c;
            v[0][0] = *(const f32x4*)(a); v[0][1] = *(const f32x4*)(a + DP_PLANE);
            return;
        }
        const bool far = (c & 0x80000000u) != 0u;
        {
            const char* const a = patch + bufoff + (far ? (unsigned)(g * 2 * DP_PLANE) : c);
            v[0][0] = *(const f32x4*)(a); v[0][1] = *(const f32x4*)(a + DP_PLANE);
            v[1][0] = *(const f32x4*)(a + 16); v[1][1] = *(const f32x4*)(a + 16 + DP_PLANE);
            v[2][0] = *(const f32x4*)(a + DP_PW * 16); v[2][1] = *(const f32x4*)(a + DP_PW * 16 + DP_PLANE);
            v[3][0] = *(const f32x4*)(a + DP_PW * 16 + 16); v[3][1] = *(const f32x4*)(a + DP_PW * 16 + 16 + DP_PLANE);
        }
        if (far) {
            const unsigned o1 = ((c & 0x7fffffffu) >> 2) * far_x + (unsigned)(cb * 64 + g * 32);
            const unsigned o2 = o1 + ((c & 1u) ? far_x : 0u);
            const unsigned dyb = (c & 2u) ? far_y : 0u;
            v[0][0] = deft_buffer_load_x4(rx, o1); v[0][1] = deft_buffer_load_x4(rx, o1 + 16u);
            v[1][0] = deft_buffer_load_x4(rx, o2); v[1][1] = deft_buffer_load_x4(rx, o2 + 16u);
            v[2][0] = deft_buffer_load_x4(rx, o1 + dyb); v[2][1] = deft_buffer_load_x4(rx, o1 + dyb + 16u);
            v[3][0] = deft_buffer_load_x4(rx, o2 + dyb); v[3][1] = deft_buffer_load_x4(rx, o2 + dyb + 16u);
        }
    };
    // blend the four corners, split into the operand pieces: the lane's A fragment (8 k of its row)
    auto blend_split = [&](const f32x4 (&v)[4][2], float w0, float w1, float w2, float w3, pcx8 (&pa)[DEFT_NP]) {
        f32x4 b0, b1;
        if (!DEFORM) { b0 = v[0][0] * w0; b1 = v[0][1] * w0; }          // (w0 = DEFT_ASCALE: 1 with three pieces -- exact either way)
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                b0[e] = fmaf(w3, v[3][0][e], fmaf(w2, v[2][0][e], fmaf(w1, v[1][0][e], w0 * v[0][0][e])));
                b1[e] = fmaf(w3, v[3][1][e], fmaf(w2, v[2][1][e], fmaf(w1, v[1][1][e], w0 * v[0][1][e])));
            }
        }
        // the three bf16 pieces (common.h split3: round-to-nearest-even each, exact), two elements at a time so that every conversion
        // is one v_cvt_pk_bf16_f32 of a PAIR and the pieces come back as floats by a shift / a mask: 11 VALU per pair
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        if constexpr (DEFT_NP == 3) {
            u32x4 ph, pm, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x0 = e < 2 ? b0[2 * e] : b1[2 * e - 4], x1 = e < 2 ? b0[2 * e + 1] : b1[2 * e - 3];
                unsigned h = dcnp_cvt_pk(x0, x1);
                DEFT_OPAQUE_NV(h);                   // (otherwise the compiler converts the low element a second time instead of shifting the pair)
                const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
                unsigned m = dcnp_cvt_pk(r0, r1);
                DEFT_OPAQUE_NV(m);
                const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
                ph[e] = h; pm[e] = m; pl[e] = dcnp_cvt_pk(s0, s1);
            }
            pa[0] = __builtin_bit_cast(pcx8, ph);
            pa[1] = __builtin_bit_cast(pcx8, pm);
            pa[DEFT_NP - 1] = __builtin_bit_cast(pcx8, pl);
        } else {
            // two fp16 pieces (the corner weights carry DEFT_ASCALE already): h1 = fp16(x), h2 = fp16(x - h1) -- the residual is exact in
            // fp32 and the matrix instructions' fp16 -> fp32 widening is exact, so this is common.h deft_split element by element
#if DEFT_PIECES == 2
            unsigned h0, h1, h2, h3, m0, m1, m2, m3;
            deft_split2_pair(b0[0], b0[1], h0, m0);
            deft_split2_pair(b0[2], b0[3], h1, m1);
            deft_split2_pair(b1[0], b1[1], h2, m2);
            deft_split2_pair(b1[2], b1[3], h3, m3);
            pa[0] = __builtin_bit_cast(pcx8, u32x4{h0, h1, h2, h3});
            pa[1] = __builtin_bit_cast(pcx8, u32x4{m0, m1, m2, m3});
#endif
        }
    };

    // ---- K loop, software-pipelined.  Chunk kc = (16-channel block cb, tap).  Step kc starts with A(kc) AND the B fragments of kc in
    // registers, so its MFMAs issue right behind the barrier:
    //   wait + barrier   -- the DMA pieces issued two steps ago have landed (vmcnt(N): the N pieces issued ONE step ago may still fly)
    //   corner reads of kc + 1 (LDS, patch buffer of its block)
    //   MFMAs of kc  ||  blend + split of kc + 1 -> A(kc + 1)
    //   B fragments of kc + 1 (LDS; its weights landed two steps ago, visible since this step's barrier)
    //   DMA: weights of kc + 3 into the stage whose fragments were read one step ago; patch pieces of the NEXT block (taps 0..6)
    // Three weight stages, two patch buffers; two blocks per loop iteration, so that buffers and stages are immediates. ----
    pcx8 pa[DEFT_NP], pb[TN][DEFT_NP];
    auto read_b = [&](int st, pcx8 (&o)[TN][DEFT_NP]) {
        const char* const bs = Bd + st * BSTAGE + brow;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const char* const bj = bs + (j >> 1) * DP_WBLK + (j & 1) * 512;
#pragma unroll
            for (int q = 0; q < DEFT_NP; ++q) o[j][q] = *(const pcx8*)(bj + q * 2048);
        }
    };
    f32x4 vq[GA][4][2];                                    // corner values in flight: vq[0] = the next chunk's, vq[1] = the one after (GA = 2)
    {
        DEFT_WAIT_VM(0);
        DEFT_PIPE_BARRIER_ONLY();
        f32x4 v[4][2];
        gather(rc[0], 0, 0, v);
        blend_split(v, rw0[0], rw1[0], rw2[0], rw3[0], pa);
        read_b(0, pb);
        if (GA == 2) gather(rc[1], 0, 0, vq[1]);          // (chunk 1 exists: Cin >= 32 -> at least 18 chunks); slot of an ODD chunk, see the loop
    }
    for (int cb2 = 0; cb2 < ncb; cb2 += 2) {
        const bool lastpair = cb2 + 2 >= ncb;                        // the odd block of this pair is the last block
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cb = cb2 + half, kc = cb * 9 + tap;
                const int st = tap % 3;                                // = kc % 3 (18 steps per iteration)
                const int ntap = (tap + 1) % 9;                        // chunk kc + 1 (blended in this step)
                const int gtap = (tap + GA) % 9, gblk = (tap + GA) / 9;        // chunk kc + GA (its corner reads are issued in this step)
                const bool last_blk = half == 1 && lastpair;           // (runtime only in the odd half)
                const bool more = !(tap == 8 && last_blk);             // is there a chunk kc + 1?
                const bool moreg = !(tap + GA >= 9 && last_blk);       // ... a chunk kc + GA?
                // ---- wait + barrier (step 0 too: everybody has read the B fragments of chunk 0 before its stage is refilled) ----
                {
                    const int pt = tap == 0 ? 8 : tap - 1;             // the previous step's tap; it is in the last block iff this one is and tap > 0
                    if (tap > 0 && last_blk) wait_vm(pt < 6 ? (wave < 2 ? NB_A : NB_B) : 0);
                    else wait_vm(wave < 2 ? NB_A : NB_B + (ps_of(pt + 1) - ps_of(pt)));
                    DEFT_PIPE_BARRIER_ONLY();
                }
                // GA = 2: the corner values live in a two-slot ring with STATIC slots (the 18 steps of an iteration are unrolled, the first chunk
                // of an iteration is even): the reads issued in step kc (chunk kc + 2) go to slot kc & 1 and are blended in step kc + 1,
                // which reads slot (kc + 1) & 1 ^ 1.  (Round 3 copied slot 1 to slot 0 at the end of every step: the compiler hoisted those
                // moves to the top of the step, and with them a wait for the far samples' global loads that had just been issued.)
                constexpr int GS = GA == 2 ? 2 : 1;
                const int gslot = GA == 2 ? ((half * 9 + tap) & 1) : 0;
                if (moreg) gather(rc[gtap], ((half + gblk) & 1) * DP_PBUF, cb + gblk, vq[gslot]);
                // six products per fp32 product, smallest terms first (as igemm.hip); product-major so that consecutive MFMAs go to
                // DIFFERENT accumulators (a dependent MFMA waits for its predecessor's full latency, more with VALU slotted between them)
#pragma unroll
                for (int q = 0; q < DEFT_NPROD; ++q) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[0][j] = deft_mfma_pc(pa[deft_qa(q)], pb[j][deft_qb(q)], acc[0][j]);
                }
                if (more) {
                    pcx8 pn[DEFT_NP];
                    blend_split(vq[GS == 2 ? (gslot ^ 1) : 0], rw0[ntap], rw1[ntap], rw2[ntap], rw3[ntap], pn);
#pragma unroll
                    for (int q = 0; q < DEFT_NP; ++q) pa[q] = pn[q];
                }
                DCNP_SCHED(TN);
#pragma unroll
                for (int q = 0; q < DEFT_NP; ++q) DEFT_OPAQUE(pa[q]);      // A(kc + 1) is finished HERE, under the MFMAs -- not after the next barrier
                if (tap < 6 || !last_blk) issue_b3(kc + 3, st);            // (kc + 3 < nchunks)
                if (wave >= 2 && !last_blk) {
#pragma unroll
                    for (int i = ps_of(tap); i < ps_of(tap + 1); ++i) issue_patch_piece(cb + 1, half ^ 1, i);
                }
                if (more) read_b((tap + 1) % 3, pb);
            }
        }
    }
    __syncthreads();                                   // nobody reads the stages any more, no DMA in flight (the last chunk issued none)

    // ---- epilogue through LDS (common.h): the whole 128-row tile fits the loop's LDS ----
    float* const T = (float*)smem;
    deft_epilogue_stage<1, TN>(T, BN + 4, acc, wave, 0, lane, p, n0);
    DEFT_PIPE_BARRIER_ONLY();
    deft_epilogue_rows<128, BN, 256>(T, p, n0, tid, [&](int R) -> long long {
        int tr, txx;
        dcnp_row_to_pixel(R & 31, tr, txx);
        const int y = ty0 + 2 * (R >> 5) + tr, x = tx0 + txx;
        return (y < p.H && x < p.W) ? (long long)(img + y * p.W + x) : -1;
    });
}

// ---- Producer / consumer form of the 64-column tile (round 6; two fp16 pieces only) --------------------------------------------------------
// What the counters of dcn_patch_kernel<2, 3, true> say (profiles/r6_dcn_producer_consumer.md: matrix pipe 0.20 busy, VALU ~0.4, LDS array 0.31, every wave
// 36 % of its life issuing, two waves per SIMD): nothing is saturated -- one in-order wave does gather -> blend -> split -> MFMA -> weight DMA ->
// fragment reads in sequence, and two such waves per SIMD do not cover each other's latencies.  Here the SAME work of a workgroup (8 x 16 pixels,
// 64 columns, the same patch / weight images, the same arithmetic in the same order: bit-identical results) is split over EIGHT waves:
//   producers (waves 4-7): hold the nine sampling records of their lane's row, gather the corners of the NEXT chunk from the LDS patch (far
//     samples: global loads), blend, split into the two fp16 pieces and WRITE their A fragment -- 16 B per piece and lane -- into an LDS stage,
//     in the lane-linear layout [piece][k group][row] the matrix instruction reads it in (conflict-free for both sides);
//   consumers (waves 0-3): issue every LDS-DMA of the workgroup (weights three chunks ahead, the next block's patch), read the A fragment of
//     their 32 rows (two ds_read_b128) and the B fragments, and run the six MFMAs of the chunk -- nothing else.
// One workgroup barrier per chunk hands stage (kc & 1) over: the producers write A(kc + 1) while the consumers multiply A(kc).  Both roles fit
// 128 VGPRs, so a CU holds two workgroups = FOUR waves per SIMD (one producer and one consumer of each): four instruction streams of different
// kinds (VALU / LDS on one side, MFMA / DMA on the other) instead of two identical ones.  LDS: 2 x 24 KB patch + 3 x 4 KB weights + 2 x 8 KB
// A = 76 KB.
#if DEFT_PIECES == 2
#ifndef DCNPC_R
#define DCNPC_R 3
#endif
#define DPC_ABLK (DEFT_NP * 2 * 128 * 16)              // one A stage: [piece][k group][128 rows][8 halves]
// (The ablation switches behind profiles/r6_dcn_producer_consumer.md -- no far path / corner reads / blend / A write / MFMAs / DMA / A read -- are kept as
// profiles/r6_dcn_pc_ablation_switches.patch, not in this file.)
// Scheduling pins of the producer step (no instruction is emitted).  DCNPC_PIN(b): the four blended values exist HERE and no memory operation moves
// across this point -- the next chunk's corner reads go into the registers the blend has just released (hoisted above it they would need 16 more
// registers per half: spills at 128).  DCNPC_ARRIVED(v): the four corner vectors are consumed no earlier than HERE -- without it the compiler
// software-pipelines the blend of a chunk ABOVE the barrier of the step that issued its reads, with a vmcnt(0) wait for the far lanes' global loads
// (a full L2 round trip) in front of every barrier.
#if defined(__HIP_DEVICE_COMPILE__)
#define DCNPC_PIN(b)                                                     \
    do {                                                                 \
        asm volatile("" : "+v"(b));                                      \
        asm volatile("" ::: "memory");                                   \
    } while (0)
#define DCNPC_ARRIVED(v) asm volatile("" : "+v"((v)[0]), "+v"((v)[1]), "+v"((v)[2]), "+v"((v)[3]))
#else
#define DCNPC_PIN(b) ((void)(b))
#define DCNPC_ARRIVED(v) ((void)(v))
#endif

template <int DP_R>
constexpr int dcnpc_lds_bytes() {
    constexpr int loop = 2 * DP_PBUF_(DP_R) + 3 * DP_WBLK + 2 * DPC_ABLK;
    constexpr int tile = 128 * (64 + 4) * 4;
    return loop > tile ? loop : tile;
}

template <int DP_R>
__global__ __launch_bounds__(512, 4) void dcn_pc_kernel(DeftGemmDesc p, int tiles_x, int tiles_y, int ntiles) {
    constexpr int DP_PW = DP_PW_(DP_R), DP_NPIX = DP_NPIX_(DP_R), DP_PARTS = DP_PARTS_(DP_R), DP_PLANE = DP_PLANE_(DP_R), DP_PBUF = DP_PBUF_(DP_R);
    constexpr int TN = 2, BN = 64;
    constexpr int NBP = DEFT_NP * 2;                   // weight DMA pieces (1 KB) per chunk: one per consumer wave
    constexpr int BSTAGE = DP_WBLK;
    static_assert(NBP == 4, "one weight piece per consumer wave and chunk");
    constexpr int PT = 6;                              // the next block's patch pieces go out during taps 0 .. PT - 1 (a piece has two steps to land;
                                                       // the producers read the block's first chunk during tap PT + 1 of the block before)
    static_assert(DP_PARTS <= PT, "at most one patch piece per consumer wave and step");

    DEFT_DYN_LDS(char, smem);
    char* const patch = smem;                          // [2 buffers][4 planes][DP_PARTS * 64 pixels][16 B]
    char* const Bd = smem + 2 * DP_PBUF;               // [3 stages][BSTAGE]
    char* const Ad = Bd + 3 * BSTAGE;                  // [2 stages][DPC_ABLK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w4 = wave & 3;                           // the 32 rows (two tile rows) this wave produces / consumes
    const bool producer = wave >= 4;
    const int g = lane >> 5;

    int bid = blockIdx.x;                              // XCD-aware, bijective workgroup remap (as dcn_patch_kernel)
    {
        const int nwg = p.N * tiles_x * tiles_y * ntiles;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nt = bid % ntiles;
    int t = bid / ntiles;
    const int tpi = tiles_x * tiles_y;
    const int n = t / tpi;
    t -= n * tpi;
    const int tyi = t / tiles_x, txi = t - tyi * tiles_x;
    const int ty0 = tyi * DP_TH, tx0 = txi * DP_TW, n0 = nt * BN;
    const int py0 = ty0 - 1 - DP_R, px0 = tx0 - 1 - DP_R;
    const int img = n * p.H * p.W;
    const int ncb = p.Cin >> 4, nchunks = ncb * 9;
    const int arow = ((g * 128) + w4 * 32 + (lane & 31)) * 16;      // this lane's 16 bytes inside a piece of an A stage (+ piece * 4096)

    f32x16 acc[1][TN];

    if (producer) {
        // ================================================= producers =================================================
        int trow, tx;
        dcnp_row_to_pixel(lane & 31, trow, tx);
        const int oy = ty0 + 2 * w4 + trow, ox = tx0 + tx;
        const bool rowok = oy < p.H && ox < p.W;
        f32x4 om[7];
        {
            const float* omp = p.x2 + (size_t)(rowok ? img + oy * p.W + ox : 0) * p.ldom;
#pragma unroll
            for (int q = 0; q < 7; ++q) om[q] = *(const f32x4*)(omp + 4 * q);
        }
        const deft_rsrc_t rx = deft_make_rsrc(p.x);
        float rw0[9], rw1[9], rw2[9], rw3[9];
        unsigned rc[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) dcnp_record<DP_R>(p, om, tap, oy, ox, rowok, img, py0, px0, g, rw0[tap], rw1[tap], rw2[tap], rw3[tap], rc[tap]);
        const unsigned far_x = (unsigned)p.ldx * 4u, far_y = (unsigned)(p.W * p.ldx) * 4u;

        // corner values of one (row, tap, 16-channel block), HALF h of the lane's 8 channels (plane 2 g + h): v[corner].  Every lane reads the patch
        // (a far lane some valid address); the far lanes then load their corners from global memory into the same registers (see dcn_patch_kernel).
        auto gather_half = [&](unsigned c, int bufoff, int cb, int h, f32x4 (&v)[4]) {
            const bool far = (c & 0x80000000u) != 0u;
            {
                const char* const a = patch + bufoff + h * DP_PLANE + (far ? (unsigned)(g * 2 * DP_PLANE) : c);
                v[0] = *(const f32x4*)(a);
                v[1] = *(const f32x4*)(a + 16);
                v[2] = *(const f32x4*)(a + DP_PW * 16);
                v[3] = *(const f32x4*)(a + DP_PW * 16 + 16);
            }
            if (far) {
                const unsigned o1 = ((c & 0x7fffffffu) >> 2) * far_x + (unsigned)(cb * 64 + g * 32 + h * 16);
                const unsigned o2 = o1 + ((c & 1u) ? far_x : 0u);
                const unsigned dyb = (c & 2u) ? far_y : 0u;
                v[0] = deft_buffer_load_x4(rx, o1);
                v[1] = deft_buffer_load_x4(rx, o2);
                v[2] = deft_buffer_load_x4(rx, o1 + dyb);
                v[3] = deft_buffer_load_x4(rx, o2 + dyb);
            }
        };
        auto blend_half = [&](const f32x4 (&v)[4], float w0, float w1, float w2, float w3) -> f32x4 {
            f32x4 b;
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = fmaf(w3, v[3][e], fmaf(w2, v[2][e], fmaf(w1, v[1][e], w0 * v[0][e])));
            return b;
        };
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        f32x4 va[4], vb[4];                            // the chunk in flight: corners of half 0 / half 1
        // blend + split the chunk in (va, vb) with the weights of `tap`, write it to A stage `stage`; as soon as a half is blended its registers
        // take the corner reads of the chunk after (tap `gtap` of block `gcb`, patch buffer offset `gbuf`), if there is one
        auto produce = [&](int tap, int stage, bool more, int gtap, int gbuf, int gcb) {
            // (the blend of a half is FINISHED before its registers are handed to the next chunk's reads: the scheduler would otherwise hoist the
            // reads above the blend -- 16 more live registers per half, spills at the 128 the four-waves-per-SIMD occupancy allows)
            DCNPC_ARRIVED(va);
            f32x4 b0 = blend_half(va, rw0[tap], rw1[tap], rw2[tap], rw3[tap]);
            DCNPC_PIN(b0);
            if (more) gather_half(rc[gtap], gbuf, gcb, 0, va);
            DCNPC_ARRIVED(vb);
            f32x4 b1 = blend_half(vb, rw0[tap], rw1[tap], rw2[tap], rw3[tap]);
            DCNPC_PIN(b1);
            if (more) gather_half(rc[gtap], gbuf, gcb, 1, vb);
            unsigned h0, h1, h2, h3, m0, m1, m2, m3;
            deft_split2_pair(b0[0], b0[1], h0, m0);
            deft_split2_pair(b0[2], b0[3], h1, m1);
            deft_split2_pair(b1[0], b1[1], h2, m2);
            deft_split2_pair(b1[2], b1[3], h3, m3);
            char* const ap = Ad + stage * DPC_ABLK + arow;
            *(u32x4*)(ap) = u32x4{h0, h1, h2, h3};
            *(u32x4*)(ap + 4096) = u32x4{m0, m1, m2, m3};
        };

        DEFT_PIPE_BARRIER_ONLY();                      // P0: the first patch has landed (the consumers waited for it)
        gather_half(rc[0], 0, 0, 0, va);
        gather_half(rc[0], 0, 0, 1, vb);
        produce(0, 0, true, 1, 0, 0);                  // A(0) -> stage 0; corner reads of chunk 1 (it exists: Cin >= 32)
        for (int cb2 = 0; cb2 < ncb; cb2 += 2) {
            const bool lastpair = cb2 + 2 >= ncb;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int cb = cb2 + half;
                    const bool last_blk = half == 1 && lastpair;
                    const int ntap = (tap + 1) % 9;                                  // chunk kc + 1: blended and written in this step
                    const int gtap = (tap + 2) % 9, gblk = (tap + 2) / 9;          // chunk kc + 2: its corner reads are issued in this step
                    const bool more = !(tap == 8 && last_blk);                       // is there a chunk kc + 1?
                    const bool moreg = !(tap + 2 >= 9 && last_blk);                  // ... a chunk kc + 2?
                    DEFT_PIPE_BARRIER_ONLY();          // barrier kc: A(kc) is in its stage (this wave's writes have been acknowledged: lgkmcnt(0))
                    if (more) produce(ntap, (half * 9 + tap + 1) & 1, moreg, gtap, ((half + gblk) & 1) * DP_PBUF, cb + gblk);
                }
            }
        }
    } else {
        // ================================================= consumers =================================================
        const deft_rsrc_t rx = deft_make_rsrc(p.x);
        const deft_rsrc_t rw = deft_make_rsrc(p.w3);
        unsigned pv[DP_PARTS];                         // patch DMA sources: lane l deposits channel group `w4` of patch pixel 64 part + l
#pragma unroll
        for (int i = 0; i < DP_PARTS; ++i) {
            const int pp = i * 64 + lane;
            const int py = pp / DP_PW, px = pp - py * DP_PW;
            const int gy = py0 + py, gx = px0 + px;
            const bool ok = pp < DP_NPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            pv[i] = ok ? (unsigned)((img + gy * p.W + gx) * p.ldx * 4) : DEFT_OOB;
        }
        const unsigned vB = (unsigned)((n0 >> 6) * nchunks * DP_WBLK + lane * 16);
        auto issue_b = [&](int kc, int st) {           // weight piece w4 (of NBP = 4) of chunk kc into stage st
            deft_buffer_load_lds_x4s(rw, Bd + st * BSTAGE + w4 * 1024, vB, (unsigned)kc * (unsigned)DP_WBLK + (unsigned)w4 * 1024u);
        };
        auto issue_patch = [&](int cb, int buf, int part) {      // plane w4 of block cb
            deft_buffer_load_lds_x4s(rx, patch + buf * DP_PBUF + w4 * DP_PLANE + part * 1024, pv[part], (unsigned)(cb * 64 + w4 * 16));
        };
        auto wait_vm = [&](int n_) {
            switch (n_) {
                case 0: DEFT_WAIT_VM(0); break;
                case 1: DEFT_WAIT_VM(1); break;
                default: DEFT_WAIT_VM(2); break;
            }
        };
        auto np_of = [](int tap) { return (tap < PT ? (DP_PARTS * (tap + 1) / PT) - (DP_PARTS * tap / PT) : 0); };      // patch pieces issued at this tap
        issue_b(0, 0);
        if (nchunks > 1) issue_b(1, 1);
        if (nchunks > 2) issue_b(2, 2);
#pragma unroll
        for (int i = 0; i < DP_PARTS; ++i) issue_patch(0, 0, i);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
        const int brow = (lane & 31) * 16 + g * 1024;
        pcx8 pb[2][TN][DEFT_NP];
        auto read_b = [&](int st, pcx8 (&o)[TN][DEFT_NP]) {
            const char* const bs = Bd + st * BSTAGE + brow;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < DEFT_NP; ++q) o[j][q] = *(const pcx8*)(bs + (j & 1) * 512 + q * 2048);
        };
        DEFT_WAIT_VM(0);
        DEFT_PIPE_BARRIER_ONLY();                      // P0: patch 0 and the first three weight chunks are in LDS
        read_b(0, pb[0]);
        for (int cb2 = 0; cb2 < ncb; cb2 += 2) {
            const bool lastpair = cb2 + 2 >= ncb;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int cb = cb2 + half, kc = cb * 9 + tap;
                    const int st = tap % 3;                                // = kc % 3
                    const int cur = (half * 9 + tap) & 1;                  // = kc & 1: the A stage and the B register set of this step
                    const bool last_blk = half == 1 && lastpair;
                    const bool more = !(tap == 8 && last_blk);
                    // ---- wait + barrier kc: what this wave issued up to two steps ago has landed (the pieces of ONE step ago may still fly) ----
                    {
                        const int pt = tap == 0 ? 8 : tap - 1;             // the previous step's tap; it is in the last block iff this one is and tap > 0
                        if (tap > 0 && last_blk) wait_vm(pt < 6 ? 1 : 0);
                        else wait_vm(1 + np_of(pt));
                        DEFT_PIPE_BARRIER_ONLY();
                    }
                    pcx8 pa[DEFT_NP];
                    {
                        const char* const ap = Ad + cur * DPC_ABLK + arow;
#pragma unroll
                        for (int q = 0; q < DEFT_NP; ++q) pa[q] = *(const pcx8*)(ap + q * 4096);
                    }
                    // weights three chunks ahead into the stage whose fragments every consumer read one step ago; the next block's patch
                    if (tap < 6 || !last_blk) issue_b(kc + 3, st);
                    if (!last_blk) {
#pragma unroll
                        for (int i = DP_PARTS * tap / PT; i < (tap < PT ? DP_PARTS * (tap + 1) / PT : 0); ++i) issue_patch(cb + 1, half ^ 1, i);
                    }
                    if (more) read_b((tap + 1) % 3, pb[cur ^ 1]);
#pragma unroll
                    for (int q = 0; q < DEFT_NPROD; ++q) {
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[0][j] = deft_mfma_pc(pa[deft_qa(q)], pb[cur][j][deft_qb(q)], acc[0][j]);
                    }
                }
            }
        }
    }
    __syncthreads();                                   // E0: nobody reads the stages any more, no DMA in flight

    // ---- epilogue through LDS (common.h): the consumers park their accumulators, all eight waves store ----
    float* const T = (float*)smem;
    if (!producer) deft_epilogue_stage<1, TN>(T, BN + 4, acc, w4, 0, lane, p, n0);
    DEFT_PIPE_BARRIER_ONLY();
    deft_epilogue_rows<128, BN, 512>(T, p, n0, tid, [&](int R) -> long long {
        int tr, txx;
        dcnp_row_to_pixel(R & 31, tr, txx);
        const int y = ty0 + 2 * (R >> 5) + tr, x = tx0 + txx;
        return (y < p.H && x < p.W) ? (long long)(img + y * p.W + x) : -1;
    });
}

template <int R>
static int launch_dcnpc(const DeftGemmDesc& d, hipStream_t s) {
    constexpr int lds = dcnpc_lds_bytes<R>();
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)dcn_pc_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        DEFT_CHECK(e == hipSuccess, -101, "dcn_pc: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
        done = true;
    }
    const int tiles_x = deft_cdiv(d.W, DP_TW), tiles_y = deft_cdiv(d.H, DP_TH), ntiles = deft_cdiv(d.Cout, 64);
    const long long nwg = (long long)d.N * tiles_x * tiles_y * ntiles;
    DEFT_CHECK(nwg < (1ll << 31), -71, "deft_dcn_v2_nhwc: too many tiles");
    hipLaunchKernelGGL((dcn_pc_kernel<R>), dim3((unsigned)nwg), dim3(512), lds, s, d, tiles_x, tiles_y, ntiles);
    DEFT_CHECK_LAUNCH("dcn_pc");
    return 0;
}

// Which form the 64-column tiles run on when the descriptor leaves it open (tile bits 26 / 27): the one-role kernel unless DEFT_DCN_PC=1 --
// measured on MI355X (profiles/r6_dcn_producer_consumer.md) the producer / consumer form is 6 - 12 % SLOWER on the bench's layer shapes.
static bool dcnpc_enabled() {
    static const bool on = [] { const char* e = getenv("DEFT_DCN_PC"); return e && e[0] == '1'; }();
    return on;
}
#endif

template <int TN, int R, bool DEFORM = true>
static int launch_dcnp(const DeftGemmDesc& d, hipStream_t s) {
    constexpr int lds = dcnp_lds_bytes<TN, R>();
    if (lds > 64 * 1024) {
        static bool done = false;
        if (!done) {
            hipError_t e = hipFuncSetAttribute((const void*)dcn_patch_kernel<TN, R, DEFORM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            DEFT_CHECK(e == hipSuccess, -101, "dcn_patch: hipFuncSetAttribute(%d B LDS) failed: %s", lds, hipGetErrorString(e));
            done = true;
        }
    }
    const int tiles_x = deft_cdiv(d.W, DP_TW), tiles_y = deft_cdiv(d.H, DP_TH), ntiles = deft_cdiv(d.Cout, TN * 32);
    const long long nwg = (long long)d.N * tiles_x * tiles_y * ntiles;
    DEFT_CHECK(nwg < (1ll << 31), -71, "deft_dcn_v2_nhwc: too many tiles");
    hipLaunchKernelGGL((dcn_patch_kernel<TN, R, DEFORM>), dim3((unsigned)nwg), dim3(256), lds, s, d, tiles_x, tiles_y, ntiles);
    DEFT_CHECK_LAUNCH("dcn_patch");
    return 0;
}

// called from deft_dcn_v2_nhwc (igemm.hip) after the common checks, for p3_kernel == 2
int deft_dcnp_dispatch(const DeftGemmDesc* d, hipStream_t s) {
    DEFT_CHECK(d->prec == 1 && d->w3 != nullptr && (((size_t)d->w3 | (size_t)d->x2 | (size_t)d->y | (size_t)d->y3) & 15) == 0, -72,
               "deft_dcn_v2_nhwc: the patch form is the prec = 1 arithmetic and needs w3 (deft_split_weights_dcn); x2 / y / w3 16-byte aligned");
    DEFT_CHECK((d->ldom & 3) == 0 && d->ldom >= 28 && (d->ldy & 3) == 0 && (d->Cout & 7) == 0 && d->y != nullptr, -73,
               "deft_dcn_v2_nhwc: the patch form needs ldom %% 4 == 0, ldom >= 28, ldy %% 4 == 0, Cout %% 8 == 0 (ldom=%d ldy=%d Cout=%d)", d->ldom, d->ldy, d->Cout);
    DEFT_CHECK(d->splitk <= 1 && d->res == nullptr, -74, "deft_dcn_v2_nhwc: the patch form has no split-K and no residual");
    int bn = d->tile & 0xffff;
    if (bn == 0) bn = d->Cout > 64 ? 128 : 64;
    DEFT_CHECK(bn == 64 || bn == 128, -75, "deft_dcn_v2_nhwc: the patch form has 64- and 128-column tiles (tile & 0xffff = %d)", bn);
    // (the weight image has ceil(Cout / 128) * 128 rows: no n-tile reaches past it)
    // margin: as many pixels as still leave two workgroups per CU (DCNP_R2 / DCNP_R4; MI355X A/B of round 5: profiles/r5_dcn_margin_ab.log)
#if DEFT_PIECES == 2
    if (bn == 64 && !(d->tile & (1 << 27)) && ((d->tile & (1 << 26)) || dcnpc_enabled())) return launch_dcnpc<DCNPC_R>(*d, s);      // producer / consumer waves (round 6)
#endif
    return bn == 128 ? launch_dcnp<4, DCNP_R4>(*d, s) : launch_dcnp<2, DCNP_R2>(*d, s);
}

// called from deft_conv2d_nhwc (igemm.hip) for p3_kernel == 3: the plain 3x3 / stride 1 / pad 1 conv with <= 32 output channels on an fp32 input
int deft_conv3p_dispatch(const DeftGemmDesc* d, hipStream_t s) {
    DEFT_CHECK(d->prec == 1 && d->w3 != nullptr && d->x != nullptr && d->y != nullptr && (((size_t)d->w3 | (size_t)d->x | (size_t)d->y) & 15) == 0, -77,
               "deft_conv2d_nhwc: the fp32-patch form (p3_kernel = 3) is the prec = 1 arithmetic and needs x, y and w3 (deft_split_weights_dcn), 16-byte aligned");
    DEFT_CHECK(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->stride_w == 0 && d->rowmap == nullptr && d->OH == d->H && d->OW == d->W, -78,
               "deft_conv2d_nhwc: the fp32-patch form is 3x3 / stride 1 / pad 1, dense");
    DEFT_CHECK((d->Cin & 31) == 0 && d->Cout <= 32 && (d->Cout & 7) == 0 && (d->ldy & 3) == 0 && (d->ldx & 3) == 0 && d->splitk <= 1 && d->res == nullptr &&
               d->y3 == nullptr && d->fold_y == nullptr, -79,
               "deft_conv2d_nhwc: the fp32-patch form needs Cin %% 32 == 0, Cout <= 32 and %% 8 == 0, no residual / split-K / y3 / fold (Cin=%d Cout=%d)", d->Cin, d->Cout);
    DEFT_CHECK((long long)d->N * d->H * d->W * d->ldx < (1ll << 29), -80, "deft_conv2d_nhwc: input exceeds 2 GiB (split the batch)");
    return launch_dcnp<1, 0, false>(*d, s);
}

// ---- weight image of the patch form -----------------------------------------------------------------------------------------------
// w [CoutPad][9 * Cin] fp32 in the DCN K order of deft_dcn_v2_nhwc (k = ((c / 32) * 9 + tap) * 32 + c % 32) ->
// w3 [CoutPad / 64][Cin / 16 * 9 chunks][3 pieces][2 k groups][64 rows][8 bf16]: chunk (cb, tap) = channels 16 cb .. 16 cb + 15 of tap.
__global__ __launch_bounds__(256) void split_weights_dcn_kernel(const float* __restrict__ w, deft_piece_t* __restrict__ w3, int CoutPad, int Cin) {
    constexpr int PER = DEFT_NP * 128;                 // 16-byte slots of one chunk image: [DEFT_NP pieces][2 k groups][64 rows]
    const int nchunks = (Cin >> 4) * 9;
    const long long total = (long long)(CoutPad >> 6) * nchunks * PER;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int row = (int)(idx & 63);
    const int sub = (int)((idx >> 6) % (DEFT_NP * 2));
    const int kc = (int)((idx / PER) % nchunks);
    const int blk = (int)(idx / ((long long)PER * nchunks));
    const int q = sub >> 1, g = sub & 1;
    const int cb = kc / 9, tap = kc - 9 * cb;
    const float* wr = w + (size_t)(blk * 64 + row) * (size_t)(9 * Cin);
    pcx8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cb * 16 + g * 8 + e;
        float r = wr[((c >> 5) * 9 + tap) * 32 + (c & 31)];
        deft_piece_t pc = (deft_piece_t)0.f;
#pragma unroll
        for (int qq = 0; qq < DEFT_NP; ++qq) {
            const deft_piece_t h = (deft_piece_t)r;
            if (qq == q) pc = h;
            r -= (float)h;
        }
        o[e] = pc;
    }
    *(pcx8*)(w3 + idx * 8) = o;
}

extern "C" int deft_split_weights_dcn(const float* w, void* w3, int CoutPad, int Cin, void* stream) {
    DEFT_CHECK(w && w3 && CoutPad > 0 && (CoutPad & 63) == 0 && Cin > 0 && (Cin & 31) == 0 && (((size_t)w3) & 15) == 0, -76,
               "deft_split_weights_dcn: need CoutPad %% 64 == 0, Cin %% 32 == 0, w3 16-byte aligned (%d, %d)", CoutPad, Cin);
    const long long total = (long long)(CoutPad >> 6) * (Cin >> 4) * 9 * (DEFT_NP * 128);
    hipLaunchKernelGGL(split_weights_dcn_kernel, dim3((unsigned)deft_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w, (deft_piece_t*)w3, CoutPad, Cin);
    DEFT_CHECK_LAUNCH("split_weights_dcn");
    return 0;
}
