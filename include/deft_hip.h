/* deft_hip.h -- C ABI of libdeft_hip.so: the MI355X (gfx950) kernels behind DEFT's
 * per-frame hot path.  Plain pointers and sizes only; every pointer is DEVICE memory
 * owned by the caller (PyTorch-ROCm tensors in the Python host), every call is
 * asynchronous on `stream` (a hipStream_t passed as void*), returns 0 on success or
 * a negative code with text in deft_last_error().  The library allocates nothing.
 *
 * Tensors are fp32 NHWC: element (n,y,x,c) of a map lives at
 *     base[((n*H + y)*W + x)*ld + c],   ld >= C, ld % 4 == 0
 * (ld lets a producer write straight into a channel slice of a concat buffer --
 * DLA's Root nodes, dla.py:199-207, never materialise torch.cat).
 *
 * Each entry point names the reference interface it replaces
 * (paths relative to the reference repo MedChaabane/DEFT).
 */
#ifndef DEFT_HIP_H
#define DEFT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DEFT_ABI_VERSION 14

/* ---- implicit-GEMM descriptor shared by the three contraction entry points ---- */
typedef struct DeftGemmDesc {
    const float* x;      /* conv/dcn: input map NHWC.  pair: U' [T][ldx]                    */
    const float* x2;     /* dcn: offset/mask map [M][ldom] (ch 2k=dy_k, 2k+1=dx_k, 18+k=mask
                            logit).  pair: V' [Q][ldx].  conv: unused                        */
    const float* w;      /* packed weights [CoutPad][Kpad], k = (r*KW+s)*Cin + c, zero padded */
    const float* scale;  /* per-Cout epilogue scale (NULL = 1)  -- folded BatchNorm gamma/std */
    const float* shift;  /* per-Cout epilogue shift (NULL = 0)  -- folded bias / BN beta      */
    const float* res;    /* optional residual map added before ReLU (NULL = none)            */
    float* y;            /* output NHWC / [M][ldy]                                            */
    int N, H, W, Cin, ldx;        /* input geometry                                          */
    int OH, OW, Cout, ldy, ldr;   /* output geometry, residual pixel stride                  */
    int KH, KW, stride, pad;
    int Ktot, Kpad;               /* KH*KW*Cin and its multiple-of-32 padding                */
    int cin_log2;                 /* log2(Cin) when KH*KW > 1 (Cin must be a power of two)   */
    int M;                        /* GEMM rows: N*OH*OW (conv/dcn) or T*Q (pair)             */
    int relu;                     /* apply max(.,0) last                                     */
    int Q;                        /* pair: number of current-frame objects                   */
    int ldom;                     /* dcn: pixel stride of x2                                 */
    int tile;                     /* 0 = auto; else (BM<<16)|BN forces a tile; bit 29 selects the
                                     2-stage loop (conv: LDS-DMA form) instead of the 1-stage one.
                                     Pre-split kernels (x3): bit 29 = 3 LDS stages, bit 30 = ONE stage;
                                     halo form: (TH<<16)|BN, bit 28 = tiles are 16 pixels wide (TH x 16),
                                     bit 29 = one tap per weight stage.  deft_dcn_v2_nhwc, patch form (p3_kernel = 2):
                                     64 / 128 = output channels per workgroup; bit 26 puts a 64-column launch on the
                                     producer / consumer form (dcn_pc_kernel), bit 27 on the one-role kernel (dcn_patch_kernel)
                                     -- same bits either way; neither: the library's choice (env DEFT_DCN_PC).
                                     deft_conv_direct, 7x7 image layer: DEFT_TILE_PLANAR = x is the image as [N, 3, H, W] fp32
                                     planes (detector.py:150) instead of 4-channel NHWC -- no deft_nchw_to_nhwc pass in front */
    /* pair, batched form (Tper > 0): rows are (c, t, j) with c = m / (Tper*Q) the current
     * frame, t its history row, j its object; U' row = u0 + c*du + t, V' row = v0 + c*dv + j.
     * Tper == 0: rows are (t, j), U' row = t, V' row = j.                                   */
    int Tper, u0, du, v0, dv;
    /* conv: horizontal stride when > 0 (vertical stays `stride`); 0 = same as `stride`.  Lets the
     * 16-channel full-resolution layers run as "pixel-pair" convs: one GEMM row = two adjacent
     * output pixels (2 x 16 = 32 output columns, KW+1 wide window, stride_w = 2) -- full 32-wide
     * MFMA tiles instead of 16 useful + 16 padding columns.  Zero weights add exact zeros.      */
    int stride_w;
    /* conv K order of the packed weights: 0 = (r, s, c) tap-major; 1 = (c/32, r, s, c%32): all taps of a
     * 32-channel block back to back, so the im2col rows of neighbouring taps (the same input lines shifted
     * by one pixel) are re-read from the CU's L1 instead of L2.  Needs Cin % 32 == 0.                   */
    int korder;
    /* conv, sparse output rows (NULL = dense grid): rowmap[2m] = n*H*W, rowmap[2m+1] = (y << 16) | x
     * of the output pixel GEMM row m stands for, or -1 for an unused row (written as shift/ReLU
     * of a zero accumulator).  Output row m is y + m*ldy.  Used by the embedding head, which
     * needs the selector convs only at the bilinear neighbours of the detection centres.     */
    const int* rowmap;
    /* conv/dcn, cross-workgroup split-K for launches with too few output tiles to fill the chip (one frame per
     * GPU: the 19x34 / 38x68 maps): splitk = S > 1 workgroups share one output tile, each contracting a contiguous
     * 1/S of the K chunks; partial tiles go through `ws` (>= tiles * S * BM*BN floats, see deft_gemm_plan) and the
     * workgroup that arrives last at `ws_cnt[tile]` (ints, zero before the first launch; left zero again) adds the S
     * partials IN SPLIT ORDER and runs the epilogue -- deterministic, but a different fp32 summation order than
     * S = 1.  0 / 1 = off.  Several launches may share ws / ws_cnt as long as they are ordered on one stream. */
    int splitk;
    float* ws;
    int* ws_cnt;
    /* arithmetic of the contraction: 0 = v_mfma_f32_32x32x2_f32, bitwise a k-ordered fp32 fmaf chain; 1 = every fp32
     * operand split into NP = deft_pieces() 16-bit pieces and every fp32 product formed on the 16-bit matrix instructions, fp32
     * accumulation.  NP = 2 (this library): two fp16 pieces (operand to 2^-24 relative inside the fp16 range, csrc/common.h),
     * three v_mfma_f32_32x32x16_f16 products -- 16/3 of the fp32 MFMA rate; the CALLER scales every weight row by a power of two
     * into fp16 range and folds the inverse into `scale` (deft_amd.engine.scale_weight_rows), the kernels scale the activations
     * by 2^4 themselves.  NP = 3 (the `_p3` twin of every entry point, see deft_pieces): three bf16 pieces (exact), six
     * v_mfma_f32_32x32x16_bf16 products (all terms down to 2^-24 relative) -- 16/6 of the fp32 MFMA rate, no scaling.
     * Honoured by the tiles with one wave per output sub-tile, BN >= 64 and the 1-stage loop (the 128x32 / 64x32 /
     * 32x32 tiles and the 2-stage form stay on the fp32 instruction: measured no faster there). */
    int prec;
    /* conv, pre-split operands (the piece or "P3" format; prec = 1 only).  x3 != NULL selects the LDS-DMA kernel (igemm3.hip): the
     * input map and the weights are read as their NP pieces straight into LDS -- no operand split and no register staging in the
     * K loop; results are bit-identical to the prec = 1 path above (same pieces, same products, same k order).  Layouts (16-bit elements):
     *   map   element (pixel p, channel c, piece q < NP) at  p*NP*ld + (c/32)*(32*NP) + q*32 + c%32,
     *         ld = channels per pixel of the underlying (concat) buffer, ld % 32 == 0; x3 / y3 point at the first
     *         channel block of the view (channel offsets are multiples of 32);
     *   w3    [CoutPad/64][Kpad/32][64 rows][4*NP slots of 8 halves]: 64 output channels x one 32-wide K chunk, slot
     *         (piece q, k-slot s) of row r stored at the physical slot csrc/common.h deft_p3_phys gives (NP = 3: q*4 + (s ^ ((r >> 2) & 3));
     *         NP = 2: (q*4 + s) ^ ((r >> 1) & 7)) -- the LDS image of the chunk, copied verbatim (deft_split_weights builds it from the
     *         packed, row-scaled fp32 matrix `w`).
     * Needs Cin % 32 == 0, Kpad == Ktot, KH*KW <= 32, no rowmap, Cout % 8 == 0, ldy % 4 == 0.
     * y3 (nullable): the output is ALSO written in P3 form (pixel stride ldy3 channels) for a following conv;
     * y may then be NULL when no fp32 consumer exists.  y3 is honoured by the pre-split conv kernels and by deft_dcn_v2_nhwc. */
    const void* x3;
    const void* w3;      /* without x3 (conv / pair on igemm.hip, prec = 1): only the weights are pre-split -- their chunk images are
                            copied to LDS by DMA, the activations are still split in the K loop (honoured by the BN >= 64 tiles, S = 1) */
    void* y3;
    int ldx3, ldy3;
    /* which pre-split kernel: 0 = im2col chunks (any conv the x3 rules admit); 1 = halo tiles, 3x3 / stride 1 / pad 1 only:
     * a workgroup stages a (TH+2) x 34 input patch once per 16 channels and takes all nine taps out of it (1/7 of the
     * im2col form's activation traffic through the CU's load path).  w3 must then be the halo-form image
     * (deft_split_weights_halo), korder 1, no split-K; `tile` = (TH << 16) | BN (| 1 << 28 for TH x 16-pixel tiles: 8x16 x
     * {128, 64, 32} are built, next to 4x32 x {128, 64, 32} and 8x32 x {128, 64}), 0 = automatic (4 x 32).  K order (16-channel
     * block, tap): same pieces and products as the other forms, another fp32 summation order.
     * deft_dcn_v2_nhwc: 2 = the "patch" form (dcn.hip): a workgroup owns an 8 x 16 pixel tile and stages, per 16 input channels, the
     * fp32 input patch that offsets of up to +-2 pixels can reach in LDS (corners beyond it are fetched from global memory, per lane);
     * the blended operand goes from registers straight into the matrix cores.  x is read as fp32 (x3 unused); w3 must be the image of
     * deft_split_weights_dcn; `tile` & 0xffff = 64 / 128 output channels per workgroup (0 = automatic); no split-K; K order
     * (16-channel block, tap): another fp32 summation order than the default form.
     * deft_conv2d_nhwc: 3 = the same kernel as a PLAIN 3x3 / stride 1 / pad 1 convolution with Cout <= 32 on the fp32 input x (the
     * conv_offset_mask layer of a DCN, dcn_v2.py): w3 = deft_split_weights_dcn of the weights packed in the DCN K order; no x3, y3,
     * residual or split-K; Cin % 32 == 0. */
    int p3_kernel;
    /* A following 1x1 conv with few outputs folded into the epilogue of the pre-split conv kernels (x3 != NULL) -- the heat-map
     * head's `Conv2d(256, C, 1)` after `Conv2d(64, 256, 3) + ReLU` (base_model.py:37-66): the 256-channel hidden map is neither
     * written nor re-read.  fold_w [fold_n][Cout] fp32 (the 1x1 weights), fold_n <= 16.  Every workgroup multiplies its BN-wide
     * slice of the finished (scaled, shifted, ReLU'd) output rows with the matching slice of fold_w and writes the partial sums to
     * fold_y[(n_tile * M + row) * fold_ld + c]: ceil(Cout / BN) parts (force `tile` to know BN), summed -- together with the 1x1
     * bias -- by deft_fold_finish.  y and y3 may then both be NULL.  NULL = off. */
    const float* fold_w;
    float* fold_y;
    int fold_n, fold_ld;
} DeftGemmDesc;

int deft_version(void);
/* operand pieces of the split arithmetic (DeftGemmDesc.prec = 1) of THIS build: 3 = three bf16 pieces, six v_mfma_f32_32x32x16_bf16 products per
   fp32 product; 2 = two fp16 pieces, three v_mfma_f32_32x32x16_f16 products (csrc/common.h DEFT_PIECES).  Sizes every x3 / y3 / w3 buffer:
   pieces * 2 bytes per element.  With 2 pieces the host scales every weight row by a power of two (deft_amd/engine.py weight_row_shift). */
int deft_pieces(void);
const char* deft_last_error(void);
/* BOTH ARITHMETICS IN ONE LIBRARY (round 6).  Every entry point `deft_X` declared in this header whose source holds device code (everything but
 * the host-side association helpers deft_lapjv / deft_iou3d_matrix / deft_associate_* / deft_kf_*) exists a second time as `deft_X_p3` with the
 * SAME signature: the same source compiled with three bf16 pieces per operand (deft_pieces_p3() == 3; six matrix instructions per fp32 product,
 * no range limit on the activations).  Piece buffers (x3 / y3 / w3), weight images and error strings (deft_last_error_p3) belong to ONE of
 * the two sets -- a caller picks per plan (all launches that exchange piece buffers from the same set).  The Python host builds its plans on
 * the `deft_X` set and moves a detector to the `_p3` set when a frame's activations leave the fp16 range (deft_amd/detector.py
 * Detector._switch_to_safe, deft_amd/hiplib.py HipLib.twin); DEFT_ARITH=bf16x3 starts there.  (Build: deft_amd/build.py links the
 * -DDEFT_PIECES=3 objects with every symbol they define renamed by llvm-objcopy.) */

/* Conv2d (+folded BatchNorm, +residual, +ReLU) as an FP32-MFMA implicit GEMM.
 * Replaces every nn.Conv2d/BatchNorm2d/ReLU triple of DLA-34 and the heads:
 * dla.py:47-87 (BasicBlock), :184-207 (Root), :301-345 (base/levels),
 * base_model.py:37-66 (heads), AFE.py:331-347 (final_net 1x1 stack). */
int deft_conv2d_nhwc(const DeftGemmDesc* d, void* stream);

/* `ngroups` independent conv problems in ONE launch (blockIdx.y = group): `descs` is the host
 * copy (validation, tile choice), `descs_dev` the same array in device memory (read by the
 * kernel).  One tile configuration for all groups (descs[0].tile, 0 = auto).  Used for the 13
 * selector convs of AFE_module.forward_selector_stacker1 (AFE.py:162-188). */
int deft_conv2d_group(const DeftGemmDesc* descs, const DeftGemmDesc* descs_dev, int ngroups, void* stream);

/* DCNv2 main contraction: bilinear offset gather * sigmoid(mask) fused into the
 * A-operand loader (no global im2col buffer), then +bias/BN/ReLU epilogue.
 * Replaces dcn_v2.DCN.forward (third-party CharlesShang/DCNv2, imported at
 * dla.py:25-29, constructed dla.py:652-660, called dla.py:663) together with
 * DeformConv.actf (dla.py:649-651).  x2 is produced by deft_conv2d_nhwc with the
 * conv_offset_mask weights (Cout 27 -> ld 32).  The weight matrix uses the DCN K order
 * k = ((c / 32) * 9 + tap) * 32 + c % 32 (all nine taps of a 32-channel block back to back: the
 * gathered neighbourhood stays in L1 across taps), not the conv order (tap * Cin + c). */
int deft_dcn_v2_nhwc(const DeftGemmDesc* d, void* stream);

/* Pair-MLP layer 2 of the affinity estimator: A[(i,j)][k] = relu(U'[i][k]+V'[j][k])
 * generated on the fly (the separable form of final_net.0+BN+ReLU over the PxQ grid).
 * Replaces AFE_module.forward_stacker2 + the first two blocks of forward_final
 * (AFE.py:190-213) without materialising the [1,832,100,100] tensor. */
int deft_pair_layer(const DeftGemmDesc* d, void* stream);

/* The WHOLE pair MLP of the affinity estimator in one launch (csrc/pairmlp.hip): AFE_module.forward_stacker2 +
 * forward_final (AFE.py:190-213) from the separable first layer's U' / V' rows to the relu'd logit of every pair at its place in the
 * affinity block -- replaces deft_pair_layer + two deft_conv2d_nhwc + phase 1 of deft_affinity_finish; the 448 intermediate floats per pair
 * stay in registers.  Pair row m = (t, j) = (m / Q, m % Q) (Tper > 0: the batched rows of deft_pair_layer); out[(m / Q) * (Q + 1) + m % Q]
 * = relu(w5 . h4 + b5); call deft_affinity_finish(h4 = NULL, ...) on `out` afterwards for the dual softmax.
 * wimg: the weight image of deft_amd.engine.pair_mlp_image (deft_pair_mlp_image_bytes() bytes: 21 chunks of 16 x NP KB, the columns of
 * W3 / W4 permuted to the accumulator layout of the layer before); s / t: per-channel scale / shift of layers 2, 3, 4 (BatchNorm and bias
 * folded; with two fp16 pieces the inverse of the weight rows' power-of-two scale folded into s, like DeftGemmDesc.scale). */
typedef struct DeftPairMlp {
    const float* U;               /* [rows][ldu] history side of layer 1, fp32                */
    const float* V;               /* [rows][ldu] current side (bias included)                 */
    const void* wimg;
    const float *s2, *t2, *s3, *t3, *s4, *t4;      /* [256] [256] [128] [128] [64] [64]       */
    const float* w5;              /* [64]                                                     */
    float* out;
    float b5;
    int ldu, M, Q;
    int Tper, u0, du, v0, dv;     /* as DeftGemmDesc (deft_pair_layer)                        */
} DeftPairMlp;
int deft_pair_mlp(const DeftPairMlp* d, void* stream);
int deft_pair_mlp_image_bytes(void);

/* [N,C,H,W] -> NHWC with channel padding to ld (zeros), and back.  Boundary
 * adapters for detector.py:150 (images) and for handing FeatureMaps out as NCHW. */
int deft_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int ldy, void* stream);
int deft_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int ldx, void* stream);

/* Pre-processing of uint8 frames on the device -- Detector.pre_process (detector.py:377-395) without the host pass:
 * cv2.warpAffine(frame, trans_input, (W, H), INTER_LINEAR) [constant 0 border] + ((v / 255 - mean) / std) + HWC -> NHWC,
 * one kernel, written straight into the network's input buffer (replaces deft_nchw_to_nhwc on that path).
 * src [N][sh][sw][3] uint8 (cv2 channel order); minv [N][6] double: the dst -> src matrix cv2 derives from trans_input
 * (deft_amd.preprocess.invert_affine); lut [3][256] float: the normalisation of every uint8 level per channel, computed in
 * float64 like the reference's numpy expression; y [N][H][W][ldy] (channels >= 3 zeroed).  cv2's fixed-point arithmetic
 * (coordinates cut to 1/32 px, 15-bit weights) restated -- cv2 is not available here: parity unpinned. */
int deft_preprocess_u8(const unsigned char* src, int N, int sh, int sw, const double* minv, const float* lut, float* y,
                       int H, int W, int ldy, void* stream);

/* MaxPool2d(2,2) -- Tree.downsample, dla.py:266-267, 273. */
int deft_maxpool2x2(const float* x, float* y, int N, int H, int W, int C, int ldx, int ldy, void* stream);

/* Depthwise ConvTranspose2d(k=2f, stride f, pad f/2) + skip add:
 * y = up(x) + skip  -- IDAUp.forward dla.py:696-699 (`upsample(project(l[i])) + l[i-1]`).
 * x is [N,H,W,C]; skip and y are [N,f*H,f*W,C]; wup is the module's weight transposed to
 * [2f*2f][C] (tap-major, so 4 channels of one tap are one 16-byte load).  y3 (nullable): the result also as its NP
 * pieces (DeftGemmDesc.x3 layout, pixel stride ldy3 channels) for a following pre-split conv. */
int deft_upsample_add(const float* x, const float* wup, const float* skip, float* y,
                      int N, int H, int W, int C, int f, int ldx, int lds, int ldy, void* y3, int ldy3, void* stream);

/* sigmoid + 3x3 peak NMS + candidate compaction (detector.py:488, utils.py:69-74).
 * hm: NHWC [N,H,W,C] (ld): logits when apply_sigmoid != 0, already-sigmoid'ed scores
 * otherwise (the form model.decode.generic_decode receives).  For every (n) appends (score, c*H*W + y*W + x)
 * of every local maximum to cand_*[n*cap ...]; cand_count[n] must be zeroed by the
 * caller (hipMemsetAsync) before the call. */
int deft_hm_peaks(const float* hm, int N, int H, int W, int C, int ld, int apply_sigmoid,
                  float* cand_score, int* cand_idx, int* cand_count, int cap, void* stream);

/* top-K over the candidates, descending score, ties by ascending index
 * (utils.py:89-104 `_topk`: per-class top-K then top-K of the union == global top-K).
 * Outputs per frame: score[K], ind[K] (= y*W+x), cls[K].  Missing entries (fewer
 * than K peaks) get score 0, ind 0, cls 0. */
int deft_topk(const float* cand_score, const int* cand_idx, const int* cand_count,
              int N, int cap, int K, int HW, float* out_score, int* out_ind, int* out_cls, void* stream);

/* Regression heads evaluated ONLY at the K peak pixels (they are only ever gathered
 * there: decode.py:120-196 via utils.py:32-36): for head h, 3x3(64->256)+bias+ReLU
 * then 1x1(256->c_h)+bias.  w0t [nheads][9*Cf][256] (k=(r*3+s)*Cf+c), b0 [nheads][256],
 * w2 [Ctot][256], b2 [Ctot], head_of [Ctot] (which hidden vector an output uses).
 * out [N][K][Ctot]. */
int deft_heads_at_peaks(const float* feat, int N, int H, int W, int Cf, int ld,
                        const int* inds, int K, const float* w0t, const float* b0,
                        const float* w2, const float* b2, const int* head_of,
                        int nheads, int Ctot, float* out, void* stream);

/* The same heads as three launches with the 3x3 layer on the matrix cores: deft_peak_rows turns the
 * peak indices (ind = y*W+x per frame) into DeftGemmDesc.rowmap rows; deft_conv2d_nhwc with that
 * rowmap computes hid [N*K][nheads*256] = relu(conv3x3(feat) + b0) for all heads at once (weights
 * concatenated along Cout); deft_heads_finish applies each head's 1x1 (256 -> c_h) + bias:
 * out[i][c] = b2[c] + sum_o w2[c][o] * hid[i][head_of[c]*256 + o]. */
int deft_peak_rows(const int* inds, int N, int K, int H, int W, int* rowmap, void* stream);
int deft_heads_finish(const float* hid, int ldh, int NK, const float* w2, const float* b2, const int* head_of,
                      int Ctot, float* out, void* stream);

/* Box assembly of generic_decode (decode.py:118-196): from (ind, head values) to
 * xs, ys, bboxes.  off_* are channel offsets into `heads` rows (-1 = head absent).
 * cts [N][K][2], bboxes [N][K][4].  centers (nullable) [N][K][2]: box centres mapped
 * to [-1,1] by the output-map size (Wm x Hm) -- convert_detection (image.py:391-412)
 * for boxes expressed in output-map pixels. */
int deft_decode_boxes(const int* inds, const float* heads, int N, int K, int Wm, int Hm, int Ctot,
                      int off_reg, int off_wh, int off_ltrb_amodal,
                      float* cts, float* bboxes, float* centers, void* stream);

/* Embedding head for one feature map: ReLU(3x3 selector conv) evaluated only at the
 * 4 bilinear neighbours of each detection centre, then the grid_sample blend
 * (bilinear, padding_mode=border; align_corners: 0 = torch >= 1.3 default, what AFE.py:178 means today and what the
 * oracle is pinned to, 1 = the torch 1.2 behaviour of the authors' environment).  Replaces
 * AFE_module.forward_selector_stacker1 (AFE.py:162-188) for one (map, selector) pair.
 * fmap NHWC [Nf,H,W,C]; wsel_t [9*C][Co] (k=(r*3+s)*C+c); centers [Nf][ndet][2] (x,y in
 * [-1,1]); out[(n*ndet+i)*ldo + col_off + o]. */
int deft_embed_map(const float* fmap, int Nf, int H, int W, int C, int ld,
                   const float* wsel_t, const float* bsel, int Co,
                   const float* centers, int ndet, float* out, int ldo, int col_off, int align_corners, void* stream);

/* Embedding head, fused over all maps (AFE.py:162-188), step 1: for every (map k, frame n, detection
 * i) turn the centre (x,y in [-1,1], grid_sample with the given align_corners, border padding) into the four
 * bilinear corner pixels -> rowmap[k][(n*ndet+i)*4+q] (DeftGemmDesc.rowmap format, q = 2*dy+dx;
 * corners outside the map are unused rows) and the four blend weights bw[k][n*ndet+i][4] (0 for
 * unused corners).  map_hw [nmaps][2] = (H, W) per map, device memory.
 * Step 2 is deft_conv2d_group over the selector convs (ReLU epilogue) into tmp; step 3:
 * out[(n*ndet+i)*ldo + col_off_k + o] = sum_q bw[k][..][q] * tmp_k[((n*ndet+i)*4+q)*ldt_k + o].
 * map_out [nmaps][4] = (float offset of tmp_k inside tmp, ldt_k, Co_k, col_off_k), device memory. */
int deft_embed_rows(const float* centers, int Nf, int ndet, const int* map_hw, int nmaps,
                    int* rowmap, float* bw, int align_corners, void* stream);
int deft_embed_blend(const float* tmp, const float* bw, const int* map_out, int nmaps, int Nf, int ndet,
                     float* out, int ldo, void* stream);

/* Tail of the affinity estimator for F history frames against one current frame:
 * x = relu(h4 . w5 + b5) per pair, then the dual softmax with the analytic padding
 * terms ((max_object - n) * e^0 + e^1) and the max/unmatched-column assembly
 * (AFE.py:119-150).  h4 [T*Q][ldh] (pairs ordered (t,j)), row_start [F+1] prefix of
 * history object counts (device array; T = row_start[F] is also passed by value; every frame's count must be <= max_object <=
 * 112 -- a frame that violates it gets NaN rows, the kernel cannot report an error); out [T][Q+1].  The softmaxes subtract their
 * maximum like F.softmax (finite for any logit).  Two launches:
 * all T*Q pairs in parallel, then one block per history frame for the softmaxes (in place on `out`).  h4 == NULL: the relu'd logits are
 * in `out` already (deft_pair_mlp) -- only the second launch runs. */
int deft_affinity_finish(const float* h4, int ldh, int C4, const float* w5, float b5,
                         const int* row_start, int F, int T, int Q, int max_object,
                         float* out, void* stream);

/* One LSTM step + two Linears for T tracks at once.  Replaces the per-track
 * KalmanFilterLSTM.predict (kalman_filter_lstm.py:65-78).  wih_t [nin][512],
 * whh_t [128][512] (transposed), bias [512] (= b_ih + b_hh), w1_t [128][64], b1 [64],
 * w2_t [64][nout], b2 [nout].  x [T][nin]; h,c [T][128] updated in place; pred [T][nout]. */
int deft_lstm_step(const float* x, float* h, float* c, int T, int nin, int nout,
                   const float* wih_t, const float* whh_t, const float* bias,
                   const float* w1_t, const float* b1, const float* w2_t, const float* b2,
                   float* pred, void* stream);

/* The library's own choice of tile and split factor for a conv (entry 0) or dcn (entry 1) descriptor whose
 * `tile` / `splitk` are 0: *tile as DeftGemmDesc.tile, *splitk (1 = no split), and the workspace the split needs:
 * *ws_floats floats for `ws`, *ws_tiles ints for `ws_cnt`.  The caller allocates, writes tile/splitk/ws/ws_cnt into
 * the descriptor and launches.  Splitting is chosen only when the WHOLE launch has fewer output tiles than the
 * chip has compute units (latency mode); big batches keep S = 1 and their results do not change. */
int deft_gemm_plan(const DeftGemmDesc* d, int entry, int* tile, int* splitk, long long* ws_floats, int* ws_tiles);

/* The whole per-track motion update of one frame in ONE launch, for the T tracks matched/activated in it:
 * feature builder + LSTM step + future boxes.  Replaces, per track, STrack.update_lstm_features (tracker.py:408-480;
 * dim = 4, box = tlwh, nin = 11) or update_lstm_features_ddd (tracker.py:482-580; dim = 7, box = (h,w,l,x,y,z,rot_y),
 * nin = 18) including its KalmanFilterLSTM.predict call and device->host copy (tracker.py:467, 571).
 * slot [T]: row of each track in the persistent state arrays h, c [S][128] (float, zero for a new track) and
 * last [S][DEFT_MOTION_LAST] (double, zero for a new track: [0] = "has a previous observation", [1] = its frame id,
 * [2..] = the previous box quantities the deltas/velocities are taken against).  box [T][dim] double.
 * The features are formed in float64 in the reference's order of operations and rounded to float32 once
 * (bit-identical to the reference's `.float()`); feat [T][nin] returns them.  pred [T][nout/4][dim] double:
 * 2-D (cx, cy, w/h, h) per future step (float32 values widened, tracker.py:471-480), 3-D (h, w, l, x, y, z, rot_y)
 * (tracker.py:573-580). */
#define DEFT_MOTION_LAST 9
int deft_motion_step(const int* slot, const double* box, int T, int dim, int frame_id,
                     float* h, float* c, double* last, int nin, int nout,
                     const float* wih_t, const float* whh_t, const float* bias,
                     const float* w1_t, const float* b1, const float* w2_t, const float* b2,
                     float* feat, double* pred, void* stream);

/* Track x detection similarity of one frame without a host round trip of the affinity blocks
 * (STrack.get_similarity tracker.py:219-252 + Tracker.get_similarity :663-688).  sim [rows][Q+1]: the output of
 * deft_affinity_finish (all stored frames stacked).  For track t, node_row [T][L] lists its selected nodes (oldest
 * first) as absolute rows of `sim`, node_scale [T][L] the decay factor of each node's frame (tracker.py:84-90),
 * node_cnt [T] how many (0..L, L <= 8).  out [T][Q+1] = column-wise numpy.median of the scaled rows (float32; the
 * mean of the two middle values for an even count), a zero row for a track without nodes. */
int deft_track_similarity(const float* sim, int rows, int Q, const int* node_row, const float* node_scale,
                          const int* node_cnt, int T, int L, float* out, void* stream);

/* fp32 -> piece form (DeftGemmDesc.x3 layout, NP = deft_pieces()) for maps produced by kernels without a piece epilogue.
 * x [rows][ldx] fp32 (C channels used, C % 32 == 0), y3 [rows][NP*ldy3] halves.  NP = 3: hi = bf16(x), mid = bf16(x - hi),
 * lo = bf16(x - hi - mid), round-to-nearest-even each (exact).  NP = 2: h1 = fp16(16 x), h2 = fp16(16 x - h1). */
int deft_split_planes(const float* x, void* y3, long long rows, int C, int ldx, int ldy3, void* stream);

/* Packed fp32 weights [CoutPad(128)][Kpad] (DeftGemmDesc.w; row-scaled by the caller when NP = 2) -> the weight image
 * (DeftGemmDesc.w3), CoutPad * Kpad * NP halves.  Done once per layer at load time. */
int deft_split_weights(const float* w, void* w3, int CoutPad, int Kpad, void* stream);

/* The same for the halo form (DeftGemmDesc.p3_kernel = 1): `w` must be packed with korder 1, Kpad = 9 * Cin;
 * image [CoutPad/64][Cin/16][9 taps][64 rows][2*NP slots of 8 halves]. */
int deft_split_weights_halo(const float* w, void* w3, int CoutPad, int Kpad, void* stream);

/* The same for the patch form of deft_dcn_v2_nhwc (DeftGemmDesc.p3_kernel = 2): `w` [CoutPad][9 * Cin] in the DCN K order
 * (k = ((c / 32) * 9 + tap) * 32 + c % 32), CoutPad % 64 == 0, Cin % 32 == 0; image [CoutPad / 64][Cin / 16 * 9 chunks]
 * [NP pieces][2 k groups][64 rows][8 halves] -- chunk (cb, tap) = channels 16 cb .. 16 cb + 15 of tap -- CoutPad * 9 * Cin * NP halves. */
int deft_split_weights_dcn(const float* w, void* w3, int CoutPad, int Cin, void* stream);

/* y[m][c] = bias[c] + sum over the `nparts` partial maps part[(i * M + m) * ldp + c] of a folded 1x1 conv (DeftGemmDesc.fold_y),
 * c < C; parts are added in index order (deterministic). */
int deft_fold_finish(const float* part, int nparts, long long M, int C, int ldp, const float* bias, float* y, int ldy, void* stream);

/* Direct (input patch in LDS) convolution for the layers that read a full-resolution map: DLA-34's base_layer (7x7, 3 -> 16,
 * dla.py:301-306; the image as 4-channel NHWC), level0 (3x3, 16 -> 16, dla.py:307-308) and level1 (3x3, 16 -> 32, stride 2,
 * dla.py:309-310), + folded BatchNorm + ReLU.  pad = KH / 2; stride 1 with Cout <= 16, or stride 2 with Cout <= 32 (Cin = 16);
 * Cin = 16 or Cin = 4 (KW <= 8); built for 3x3x16 (stride 1, 2) and 7x7x4.  A workgroup
 * stages the fp32 patch of an 8 x 32 (stride 2: 4 x 32) output tile once, splits it into the three bf16 pieces on the way into LDS (one split
 * per input element instead of one per output pixel and tap) and feeds v_mfma_f32_16x16x32_bf16 with shifted fragment reads:
 * six bf16 products per fp32 product, fp32 accumulation (the arithmetic of DeftGemmDesc.prec = 1; another summation order
 * than the implicit-GEMM kernels).  Uses x, w3, scale, shift, y, N, H, W, Cin, ldx, OH, OW, Cout, ldy, KH, KW, stride, pad,
 * M, relu of the descriptor; w3 = the fragment image of deft_split_weights_direct.  With DEFT_TILE_PLANAR in `tile` (Cin = 4 only) x is
 * the fp32 image in the reference's own layout, [N, 3, H, W] (ldx unused): the patch loader reads the three planes, bit-identical to
 * deft_nchw_to_nhwc followed by the NHWC form. */
#define DEFT_TILE_PLANAR (1 << 25)
int deft_conv_direct(const DeftGemmDesc* d, void* stream);

/* Packed fp32 weights [>= Cout rows][Kpad], k = (r*KW + s)*Cin + c (DeftGemmDesc.w order), -> the B-fragment image of
 * deft_conv_direct: [ceil(Cout / 16) column blocks][steps][3 pieces][64 lanes][8 bf16], deft_direct_weight_bytes(KH, KW, Cin,
 * Cout) bytes.  Lane l holds output channel 16 * block + (l & 15) and the 8 consecutive k of group g = l >> 4 of an MFMA K step:  Cin = 16: step t = taps 2t, 2t+1
 * (tap 2t + (g >> 1), channels 8 (g & 1) ..+7);  Cin = 4: step t = window row t (pixel 2g + (e >> 2), channel e & 3). */
int deft_split_weights_direct(const float* w, void* w3, int Cout, int Kpad, int KH, int KW, int Cin, void* stream);
long long deft_direct_weight_bytes(int KH, int KW, int Cin, int Cout);

/* ---- host-side association helpers (csrc/assoc.hip): the only entry points that take HOST pointers; synchronous, no stream ---- */

/* matching.linear_assignment's solver (matching.py:40-55: `lap.lapjv(cost_matrix, extend_cost=True, cost_limit=thresh)`; `lap` is a
 * third-party package the reference does not vendor): Jonker-Volgenant on lap's extension of the n_rows x n_cols problem
 * ([[cost, L/2], [L/2, 0]] with L = cost_limit; without a finite limit the zero-padded max(n, m) square).  cost [n_rows][n_cols] double,
 * +inf / NaN = a pair that may not be matched.  x [n_rows]: column of each row or -1; y [n_cols]: row of each column or -1;
 * total (nullable): cost of the matched pairs. */
int deft_lapjv(const double* cost, int n_rows, int n_cols, double cost_limit, int* x, int* y, double* total);

/* matching.iou_ddd_distance (matching.py:107-131): out [T][N] float32 = 1 - iou3d(detection box, track box) (:253-276) for boxes
 * (h, w, l, x, y, z, rot_y) [.][7] double -- convert_3dbox_to_8corner (:207-243), Sutherland-Hodgman clip of the two ground-plane
 * rectangles (:162-204), intersection area x vertical overlap over the union of the volumes. */
int deft_iou3d_matrix(const double* trk, int T, const double* det, int N, float* out);

/* One frame's association cascade of Tracker.update on the 2-D datasets (tracker.py:886-1030) in one call, float64 like the reference:
 *   stage 1: cost = lambda * (1 - sim[t][d]) + w_gate * g[t][d] for rows with gated[t] (g = squared Mahalanobis distance of the detection centre
 *            meas2[d] to mean2[t] under the Cholesky factor chol[t] = (l00, l10, l11) of the 2 x 2 position covariance; pairs with g > gate_thr are
 *            excluded) and lambda * (1 - sim) for the others (matching.fuse_motion, matching.py:311-371), linear_assignment at thr_embed (:40-55);
 *   stage 2: (second_stage != 0: KITTI, tracker.py:954-980) 1 - sim on what is left, same threshold;
 *   stage 3: 1 - IoU (matching.py:71-104, cython_bbox's inclusive-pixel IoU) of the left-over rows with iou_ok[t] (trk_tlbr [T][4]) and the
 *            left-over detections (det_tlbr [N][4]), linear_assignment at thr_iou.
 * sim [T][ld] float32 (columns 0 .. N-1 read; the landing buffer of deft_track_similarity).  match_t / match_d [min(T, N)]: the matched
 * (track row, detection) pairs in the order the reference appends them; lost_t [T]: rows of stage 3 still unmatched; new_d [N]: detections
 * still unmatched.  HOST pointers, synchronous. */
int deft_associate_2d(const float* sim, int ld, int T, int N, const double* mean2, const double* chol, const unsigned char* gated,
                      const double* meas2, double gate_thr, double lambda_, double w_gate, int second_stage,
                      const unsigned char* iou_ok, const double* trk_tlbr, const double* det_tlbr, double thr_embed, double thr_iou,
                      int* match_t, int* match_d, int* n_match, int* lost_t, int* n_lost, int* new_d, int* n_new);

/* The cascade for one class of a nuScenes frame (tracker.py:850-1030): stage 0 (stage0 != 0; every class but pedestrian) 1 - iou3d (float32, the
 * arithmetic of deft_iou3d_matrix) between the rows with recent[t] and all detections, linear_assignment at thr_3d; stage 1 on the unmatched recent
 * rows, then the other rows, x the unmatched detections: lambda * (1 - sim) + w_gate * g, g = the "gaussian" distance of matching.fuse_motion_ddd
 * (matching.py:374-415; metric 0: centre distance, kalman_filter_lstm.py:92-95; metric 1: squared 7-component distance, kalman_filter.py:271-273)
 * of det_ddd [N][7] to trk_ddd [T][7] (h, w, l, x, y, z, rot_y), pairs with g > max(0.2 * depth[t], gate_floor) excluded, thr_embed; stage 2: 1 - sim
 * on what is left; stage 3: 1 - IoU of the 2-D boxes (rows with iou_ok) at thr_iou.  Outputs as deft_associate_2d.  HOST pointers, synchronous. */
int deft_associate_ddd(const float* sim, int ld, int T, int N, int stage0, const unsigned char* recent, const double* trk_ddd,
                       const double* det_ddd, const double* depth, int metric, double gate_floor, double lambda_, double w_gate,
                       const unsigned char* iou_ok, const double* trk_tlbr, const double* det_tlbr, double thr_3d, double thr_embed,
                       double thr_iou, int* match_t, int* match_d, int* n_match, int* lost_t, int* n_lost, int* new_d, int* n_new);

/* The Kalman filter of the 2-D trackers on the pool's arrays, in place (utils/tracking_utils/kalman_filter.py): mean [T][8], cov [T][8][8] double.
 * deft_kf_predict = multi_predict (:165-205) for every row; deft_kf_update = update (:207-240) of the rows rows[0 .. n) with the measurements
 * meas [n][4] (x, y, a, h) -- Cholesky solve of the 4 x 4 innovation covariance like the reference; -94 when it is not positive definite.
 * HOST pointers, synchronous. */
int deft_kf_predict(double* mean, double* cov, int T);
int deft_kf_update(double* mean, double* cov, const int* rows, int n, const double* meas);

/* Greedy NMS of the nuScenes branch (utils/ddd_utils.py:178-245; detector.py:281-288, per tracking class): boxes [n][4] tlbr, scores [n], double.
 * The top_k best-scoring boxes are visited best first; a box is kept and every remaining one whose IoU with it exceeds `overlap` is dropped.
 * keep [n] int64 (zeroed here): the kept indices in its first *count slots -- the reference's return value.  HOST pointers, synchronous. */
int deft_greedy_nms(const double* boxes, const double* scores, int n, double overlap, int top_k, long long* keep, int* count);

/* Which nodes a track's similarity medians over (STrack.get_similarity, tracker.py:221-248) and where their rows live, for every pool row in one
 * host call.  nf, ni [T][L] int64: frame and detection index of the last L = mm + 2 nodes of each track, right-aligned (newest at column L - 1);
 * nn [T]: nodes the track ever had.  Selected: the nodes younger than max_node frames at frame `fid` -- all q of them while q <= mm + 1, else the
 * last mm -- i.e. the last nsel columns.  sel [T][L] (optional): 1 for a selected node.  With rows != NULL also the gather table of
 * deft_track_similarity: the block table (blk_frame / blk_start / blk_len [nblk] int64, blk_delta [nblk] float: the stored frame of each affinity
 * block of the current frame, its first row in the stacked blocks, its row count, its decay -- FeatureRecorder.update, tracker.py:59-90) gives
 * rows [T][L] = blk_start + ni, scale [T][L] = blk_delta, newest node first, zeros behind the cnt [T] = nsel selected ones.  -96: a selected node's
 * frame has no block (the reference's KeyError), -97: its id is past the block (IndexError); *bad = that frame.  HOST pointers, synchronous. */
int deft_track_nodes(const long long* nf, const long long* ni, const long long* nn, int T, int L, long long fid, int mm, int max_node,
                     unsigned char* sel, const long long* blk_frame, const long long* blk_start, const long long* blk_len,
                     const float* blk_delta, int nblk, int* rows, float* scale, int* cnt, long long* bad);

#ifdef __cplusplus
}
#endif
#endif /* DEFT_HIP_H */
