"""Static execution plans that drive libdeft_hip.so for DEFT's per-frame hot path.

Host logic only: weight packing (reference state_dict keys -> MFMA-friendly
layouts), buffer planning (NHWC fp32, concat-by-stride), and the ordered list of
C-ABI calls for one batch of frames.  All arithmetic happens in the HIP library;
torch is used for device memory and streams.  Reference anchors:

  DlaSegPlan      dla.py:758-817 (DLASeg.img2feats), :400-411 (DLA.forward),
                  :271-284 (Tree), :693-735 (IDAUp/DLAUp), base_model.py:111-132
  detect()        detector.py:486-494, decode.py:102-196, utils.py:69-104
  AfePlan         AFE.py:88-213
  LstmPlan        kalman_filter_lstm.py:65-78
"""
import collections
import ctypes as C
import math

import numpy as np
import torch

from . import hiplib
from .hiplib import GemmDesc, ptr

BN_EPS = 1e-5

HEADS = {   # opts.py:500-520 + experiments/*.sh
    "mot": {"hm": 1, "reg": 2, "wh": 2, "tracking": 2, "ltrb_amodal": 4},
    "kitti_tracking": {"hm": 3, "reg": 2, "wh": 2, "tracking": 2},
    "nuscenes": {"hm": 10, "reg": 2, "wh": 2, "tracking": 2, "dep": 1, "rot": 8, "dim": 3, "amodel_offset": 2},
}


def _rup(a, b):
    return (a + b - 1) // b * b


class View:
    """A channel slice [c0, c0+C) of an NHWC buffer [N,H,W,ld]."""

    def __init__(self, buf, N, H, W, C, ld, c0=0):
        self.buf, self.N, self.H, self.W, self.C, self.ld, self.c0 = buf, N, H, W, C, ld, c0

    @property
    def addr(self):
        return self.buf.data_ptr() + 4 * self.c0

    @property
    def shape(self):
        """NCHW shape, as the reference reads it off FeatureMaps (tracker.py:821 `FeatureMaps[0].shape[0]`)."""
        return (self.N, self.C, self.H, self.W)

    def __getitem__(self, n):
        """Frame slice `fm[0]` (tracker.py:821-825 keeps the un-flipped frame of a flip-test pair)."""
        assert isinstance(n, int) and 0 <= n < self.N
        v = View(self.buf, 1, self.H, self.W, self.C, self.ld, self.c0 + n * self.H * self.W * self.ld)
        return v

    def unsqueeze(self, dim):
        assert dim == 0 and self.N == 1
        return self

    def sub(self, c0, C):
        return View(self.buf, self.N, self.H, self.W, C, self.ld, self.c0 + c0)

    def to_nchw(self):
        """Debug/test view (torch indexing, not on the hot path)."""
        return torch.as_strided(self.buf.view(-1), (self.N, self.C, self.H, self.W),
                                (self.H * self.W * self.ld, 1, self.W * self.ld, self.ld), self.c0).contiguous()


class _LazyView(View):
    """Geometry of a map that may never be written (conv(fold=...)): `materialize()` allocates it."""

    def __init__(self, plan, N, H, W, C):
        super().__init__(None, N, H, W, C, _rup(C, 4), 0)
        self._plan = plan

    def materialize(self):
        return self._plan.alloc(self.N, self.H, self.W, self.C)


def _bn_fold(sd, p):
    """Eval BatchNorm as y = x*alpha + beta, computed like ATen's CPU kernel."""
    invstd = 1.0 / torch.sqrt(sd[p + ".running_var"].float() + BN_EPS)
    alpha = invstd * sd[p + ".weight"].float()
    beta = sd[p + ".bias"].float() - sd[p + ".running_mean"].float() * alpha
    return alpha, beta


import os as _os
KORDER_BLOCK = _os.environ.get("DEFT_KORDER", "1") == "1"     # pack k>1 convs with Cin % 32 == 0 in (channel block, tap, channel)
# order (DeftGemmDesc.korder = 1): the KH*KW shifted reads of a 32-channel block follow each other, so only the first comes
# from HBM.  Neutral for the register-staged loop of igemm.hip (instruction-bound); what lets the LDS-DMA loop of igemm3.hip
# run from L2 instead of HBM (tools/probe/dma_rate.hip: 54 vs 10 B/clk/CU).


def conv_korder(w_shape, cin_pad=None):
    Co, Ci, KH, KW = w_shape
    cp = Ci if cin_pad is None else cin_pad
    return 1 if (KORDER_BLOCK and KH * KW > 1 and cp % 32 == 0) else 0


def pack_conv_weight(w, cin_pad=None, korder=None):
    """[Co,Ci,KH,KW] -> [CoPad(128)][Kpad(32)]; korder 0: k = (r*KW+s)*CiPad + c;
    korder 1: k = ((c//32)*KH*KW + r*KW+s)*32 + c%32 (include/deft_hip.h, DeftGemmDesc.korder).
    Default: conv_korder(w.shape) -- callers pass the same value in the descriptor."""
    Co, Ci, KH, KW = w.shape
    cp = Ci if cin_pad is None else cin_pad
    if korder is None:
        korder = conv_korder(w.shape, cin_pad)
    t = torch.zeros(Co, KH, KW, cp, dtype=torch.float32)
    t[..., :Ci] = w.float().permute(0, 2, 3, 1)
    K = KH * KW * cp
    if korder == 1:
        t = t.reshape(Co, KH * KW, cp // 32, 32).permute(0, 2, 1, 3)
    out = torch.zeros(_rup(Co, 128), _rup(K, 32), dtype=torch.float32)
    out[:Co, :K] = t.reshape(Co, K)
    return out, K


def pack_dcn_weight(w):
    """DCN main weight [Co,Ci,3,3] -> [CoPad(128)][9*Ci] in the DCN kernel's K order
    k = ((c // 32) * 9 + tap) * 32 + c % 32  (include/deft_hip.h, deft_dcn_v2_nhwc)."""
    Co, Ci, KH, KW = w.shape
    assert KH == 3 and KW == 3 and Ci % 32 == 0
    t = w.float().permute(0, 2, 3, 1).reshape(Co, 9, Ci // 32, 32).permute(0, 2, 1, 3).reshape(Co, 9 * Ci)
    out = torch.zeros(_rup(Co, 128), 9 * Ci, dtype=torch.float32)
    out[:Co] = t
    return out, 9 * Ci


def pack_pair_conv_weight(w, cin_pad=None):
    """Pixel-pair form of a stride-1 conv with few output channels: output columns
    [p*Co + co] (p = 0,1: left/right pixel of the pair), window KH x (KW+1) with horizontal
    stride 2:  W'[p*Co+co][r][s'][c] = W[co][c][r][s'-p] (0 outside).  The added zeros are exact
    no-ops in the fp32 accumulation chain; only the summation order inside a K chunk differs from
    the plain form (fp32 round-off level).  -> ([CoPad][Kpad], K')."""
    Co, Ci, KH, KW = w.shape
    w2 = torch.zeros(2 * Co, Ci, KH, KW + 1, dtype=torch.float32)
    w2[:Co, :, :, :KW] = w.float()
    w2[Co:, :, :, 1:] = w.float()
    return pack_conv_weight(w2, cin_pad)


# arithmetic of the implicit-GEMM contractions (DeftGemmDesc.prec): 0 = fp32 MFMA (a k-ordered fmaf chain),
# 1 = fp32 through the bf16 matrix cores (three bf16 pieces per operand, six products, fp32 accumulation)
PREC = int(_os.environ.get("DEFT_PREC", "1"))

# pre-split operands (DeftGemmDesc.x3 / w3 / y3, igemm3.hip): convs whose input has Cin % 32 == 0 read the three bf16
# pieces of activations and weights straight into LDS; producers write the pieces next to the fp32 map.  DEFT_P3=0: off
# (the operand split stays in the K loop of igemm.hip).  Same operands and products, but NOT bit-identical: the halo and patch kernels
# sum in (16-channel block, tap) order, and the per-layer choice below depends on the tile count, i.e. on the batch -- the same frame can
# differ in the last bits between batch sizes (tests/parity_checks.check_dcn_patch_batch_invariance pins what IS invariant).
P3 = _os.environ.get("DEFT_P3", "1") != "0"
P3_MIN_COUT = int(_os.environ.get("DEFT_P3_MIN_COUT", "64"))
BDMA = _os.environ.get("DEFT_BDMA", "1") != "0"             # igemm.hip prec-1 launches with >= 128 output columns read pre-split weights by DMA
# (DeftGemmDesc.w3 without x3).  Measured in the pipeline (profiles/r2_*): pair layer 170 -> 189 TFLOP/s, 128-column 1x1 convs +3..6 %;
# 64-column conv tiles lose (the second weight stage costs them a workgroup per CU) and keep splitting weights in the loop.  (Round 2 also had a
# weight-DMA form of the igemm.hip DCN; it produced wrong row groups on the MI355X now and then for a reason that was never found, shipped off, and
# is gone: the DCN's pre-split-weight form is the patch kernel, csrc/dcn.hip, whose batch-invariance test runs several workgroup generations per CU.)
# DCN main contraction on the patch form (csrc/dcn.hip, DeftGemmDesc.p3_kernel = 2): input patch in LDS, blended operand from registers.  Used where
# its 8 x 16 pixel tiles cover the map with at most DCN_PATCH_WASTE padding and the launch has at least DCN_PATCH_MIN_TILES workgroups (fewer: one
# frame per GPU on the small maps, where igemm.hip's cross-workgroup split-K fills the chip).  DEFT_DCN_PATCH=0: off.
DCN_PATCH = _os.environ.get("DEFT_DCN_PATCH", "1") != "0"
DCN_PATCH_MIN_TILES = int(_os.environ.get("DEFT_DCN_PATCH_MIN_TILES", "384"))
DCN_PATCH_WASTE = float(_os.environ.get("DEFT_DCN_PATCH_WASTE", "1.3"))
OFFSET_FP32 = _os.environ.get("DEFT_OFFSET_FP32", "1") != "0"     # the offset / mask conv of a patch-form DCN on the fp32-patch kernel (p3_kernel = 3) ...
OFFSET_FP32_MIN_HW = int(_os.environ.get("DEFT_OFFSET_FP32_MIN_HW", "10000"))   # ... on maps of at least this many pixels (38x68: the halo form on bf16 pieces is
# faster, 0.055 vs 0.066 ms per 16 frames, and the piece copy of so small a map costs nothing; 76x136 and 152x272: equal speed, and the upsample+add pass that
# produces most DCN inputs no longer writes the pieces: 1.80 -> 1.22 ms per step, profiles/r3_layers.md)
DCN_PATCH_MIN_HW = int(_os.environ.get("DEFT_DCN_PATCH_MIN_HW", "0"))        # ... and only on maps of at least this many pixels
P3_HALO = _os.environ.get("DEFT_P3_HALO", "1") != "0"       # 3x3 / stride 1 convs on the halo-tile kernel (DeftGemmDesc.p3_kernel = 1) ...
P3_MIN_TILES = int(_os.environ.get("DEFT_P3_MIN_TILES", "512"))   # ... and give every CU two workgroups (latency mode: 2.39 ms/frame on igemm.hip
# with split-K vs 2.88 on the pre-split kernels, profiles/r2_latency_ab.txt); the im2col form (one 8-wave workgroup per CU) needs half as many
P3_HALO_WASTE_NARROW = float(_os.environ.get("DEFT_P3_HALO_WASTE_NARROW", "1.3"))
P3_HALO16 = _os.environ.get("DEFT_P3_HALO16", "1") != "0"   # ... as 8 x 16 pixel tiles where those pad the map less than 4 x 32
P3_HALO_WASTE = float(_os.environ.get("DEFT_P3_HALO_WASTE", "1.2"))   # ... when its 4 x 32 pixel tiles cover the map with at most this much padding


def halo_waste(H, W, th=4, tw=32):
    return (-(-H // th) * th) * (-(-W // tw) * tw) / float(H * W)


P3H_W16 = 1 << 28          # halo tile flag: TH x 16 pixels instead of TH x 32


def _parse_tile(spec):
    if not spec:
        return 0
    dims, _, st = spec.partition(":")
    bm, bn = (int(v) for v in dims.split("x"))
    return (bm << 16) | bn | {"1": 1 << 30, "2": 0, "3": 1 << 29}[st or "1"]


P3_IM2COL_TILE = _parse_tile(_os.environ.get("DEFT_P3_IM2COL_TILE", ""))
P3_IM2COL_WIDE = _os.environ.get("DEFT_P3_IM2COL_WIDE", "1") != "0"      # 128 x 128 (two-stage) im2col tiles on the deep layers with enough tiles
P3_STRIDE2 = _os.environ.get("DEFT_P3_STRIDE2", "0") == "1"      # tuning aid: stride-2 3x3 convs with Cin >= 64 on the im2col piece kernel
P3H_TPI3 = 1 << 27         # halo tile flag: a filter row (three taps) per weight stage and barrier
P3H_TPI3_64 = _os.environ.get("DEFT_P3H_TPI3_64", "0") == "1"      # tuning aid: the 64-column 8 x 16 tiles on three taps per interval


_T = lambda bm, bn: (bm << 16) | bn
P3_3STAGE = 1 << 29
P3_1STAGE = 1 << 30
P3_TILES = {0, _T(256, 128), _T(128, 256), _T(128, 128), _T(128, 128) | P3_3STAGE, _T(128, 64), _T(128, 64) | P3_3STAGE, _T(256, 64),
            _T(64, 128), _T(64, 128) | P3_3STAGE, _T(64, 64), _T(64, 64) | P3_3STAGE, _T(128, 128) | P3_1STAGE, _T(128, 64) | P3_1STAGE, _T(64, 128) | P3_1STAGE, _T(64, 64) | P3_1STAGE}       # igemm3.hip deft_p3_dispatch; a forced tile outside this set keeps the conv on igemm.hip

# cross-workgroup split-K for launches too small to fill the chip (DeftGemmDesc.splitk); DEFT_SPLITK=0 turns it off
# the two 16-channel full-resolution layers (base_layer 7x7, level0 3x3) on the patch-in-LDS kernel (csrc/direct.hip) instead of the
# pixel-pair implicit GEMM on the fp32 MFMA instruction
DIRECT = _os.environ.get("DEFT_DIRECT", "1") != "0"
PLANAR = 1 << 25                                               # include/deft_hip.h DEFT_TILE_PLANAR: deft_conv_direct reads the [N, 3, H, W] image itself
Y3_INLOOP = _os.environ.get("DEFT_Y3_INLOOP", "1") != "0"    # piece-form output from the in-loop kernel's epilogue (else deft_split_planes)
FOLD = _os.environ.get("DEFT_FOLD", "1") != "0"       # heat-map head: the 1x1 conv folded into the epilogue of the 3x3 conv (DeftGemmDesc.fold_w)
SPLITK = _os.environ.get("DEFT_SPLITK", "1") != "0"
# launch lists of at most DATAFLOW_MAX_N frames CAN run over this many HIP streams along their data dependencies (_Plan.build_schedule): one
# frame per GPU leaves most of the chip idle in most launches, and DLA-34's up path has independent branches -- captured as a hipGraph that is a
# graph with two parallel branches, 1.77 instead of 2.00 ms per frame in the one-frame latency mode.  OFF by default since round 6 (one stream,
# one branch): ROCm 7.2's hipGraphLaunch walks off the end of an executable graph's internal stream list for graphs WITH parallel branches after
# certain process histories (hip::Graph::UpdateStreams, a host segfault; profiles/r6_graph_replay_segfault.md -- moving the launch to another
# stream does not avoid it).  DEFT_DATAFLOW=2 opts back in.
DATAFLOW = int(_os.environ.get("DEFT_DATAFLOW", "1"))
DATAFLOW_MAX_N = int(_os.environ.get("DEFT_DATAFLOW_MAX_N", "1"))
REPLAY_STREAM = _os.environ.get("DEFT_REPLAY_STREAM", "0") != "0"      # 1: a hipGraph replay that would land on the NULL stream runs on a side stream (experiment)


def weight_row_shift(w):
    """Per-row power-of-two exponents k for a packed weight matrix w [rows, K] (any device): the row's largest |entry| times 2^k lies in
    [2^12, 2^13) -- what the two-piece fp16 split (csrc/common.h, DEFT_PIECES = 2) wants: the first piece a normal fp16 number with 3 binades
    of headroom below 65504, the second piece (<= 2^-11 of it) normal for every entry within 2^-10 of the row's largest.  All-zero rows: 0."""
    m = w.abs().amax(1)
    e = torch.frexp(torch.where(m > 0, m, torch.ones_like(m)))[1]           # m = mant * 2^e, mant in [0.5, 1)
    return torch.where(m > 0, 13 - e, torch.zeros_like(e)).to(torch.int32)


def scale_weight_rows(w, scale, cout):
    """(w * 2^k per row, scale * 2^-k): the same GEMM result bit for bit (powers of two commute with every rounding), weights in fp16 range.
    scale None -> ones.  Used by _Plan.prescale for every split-arithmetic launch of a two-piece build; tests that drive the C ABI
    directly with their own descriptors call it themselves."""
    k = weight_row_shift(w)
    w2 = torch.ldexp(w, k.view(-1, 1))
    sc = torch.ones(cout, dtype=torch.float32, device=w.device) if scale is None else scale
    return w2.contiguous(), torch.ldexp(sc, -k[:cout]).contiguous()


PAIR_MLP = _os.environ.get("DEFT_PAIR_MLP", "1") != "0"       # the affinity estimator's pair MLP as ONE launch (csrc/pairmlp.hip) instead of the four-launch chain


def _pieces_of(w, np_):
    """The NP 16-bit pieces of an fp32 matrix, as the kernels form them (common.h deft_split, round-to-nearest each): [NP, ...] int16 views."""
    dt = torch.float16 if np_ == 2 else torch.bfloat16
    out, r = [], w.float()
    for _ in range(np_):
        h = r.to(dt)
        out.append(h.view(torch.int16))
        r = r - h.float()
    return torch.stack(out, 0)


def pair_mlp_image(W2, W3, W4, np_):
    """The weight image of deft_pair_mlp (csrc/pairmlp.hip): 21 chunks of 16 * np_ fragments; a fragment is what ONE matrix instruction's A
    operand reads -- 64 lanes x 8 halves of one piece: lane l = output channel 32 ot + (l & 31) of channel tile ot, k group g = l >> 5,
    element i.  W2 [256, 512], W3 [128, 256], W4 [64, 128] fp32 (row-scaled already when np_ == 2).
      layer 2 (chunks 0-15, k step s of chunk ci): k = 16 (2 ci + s) + 8 g + i;
      layers 3 / 4 (chunks 16-19: two 32-channel k tiles each; chunk 20: four): k step s of input tile kt, k = 32 kt + c(s, g, i) with
      c(s, g, i) = (i & 3) + 8 (2 s + (i >> 2)) + 4 g -- the channel the accumulator of the layer before holds in register 8 s + i of k group g.
    -> int16 tensor [21 * 16 * np_ * 512]."""
    lane = torch.arange(64)
    o, g = lane & 31, lane >> 5
    i = torch.arange(8)
    frags = []

    def add(W, ot, kcols):          # kcols [64 lanes, 8]: the column of W each (lane, element) reads
        rows = (32 * ot + o).view(64, 1).expand(64, 8)
        vals = W[rows, kcols]                                        # [64, 8] fp32
        pc = _pieces_of(vals, np_)                                   # [np_, 64, 8]
        for q in range(np_):
            frags.append(pc[q].reshape(-1))
    for ci in range(16):
        for s_ in range(2):
            for ot in range(8):
                add(W2, ot, (16 * (2 * ci + s_) + 8 * g).view(64, 1) + i.view(1, 8))
    perm = lambda s_: ((i & 3) + 8 * (2 * s_ + (i >> 2))).view(1, 8) + (4 * g).view(64, 1)
    for c3 in range(4):
        for kk in range(2):
            for s_ in range(2):
                for ot in range(4):
                    add(W3, ot, 32 * (2 * c3 + kk) + perm(s_))
    for kt in range(4):
        for s_ in range(2):
            for ot in range(2):
                add(W4, ot, 32 * kt + perm(s_))
    img = torch.cat(frags)
    assert img.numel() == 21 * 16 * np_ * 512
    return img


class _P3Out:
    """A producer's optional P3 (three bf16 pieces) output: written only if some pre-split conv reads it (`used`)."""

    def __init__(self, addr, ld, desc=None):
        self.addr, self.ld, self.desc, self.used = addr, ld, desc, False


def p3_choice(KH, KW, stride, pad, Cin, Cout, H, W, M, korder):
    """Which kernel a conv runs on when its input is available as bf16 pieces: ("halo", tile) = 3x3 / stride 1 on halo tiles
    (igemm3.hip conv3h), ("im2col", tile) = the LDS-DMA chunk loop (igemm3_kernel), None = igemm.hip (operand split in
    the K loop).  Measured per layer shape on MI355X with tools/bench_p3.py (profiles/r2_bench_p3.log): the halo form
    wins where its 4 x 32 pixel tiles cover the map with little padding, the im2col form on the small deep maps; 1x1 convs
    and Cout <= 64 stride-2 convs are HBM- or issue-bound and gain nothing from the 6-byte pieces."""
    if KH * KW == 1 or Cin % 32 or Cout % 8:
        return None
    if (KH, KW, stride, pad) == (3, 3, 1, 1) and korder == 1 and P3_HALO:
        # 4 x 32 or 8 x 16 pixel tiles: whichever pads the map less (widths that are 8 mod 16 -- 136, 272 at config B -- favour 8 x 16:
        # 128->128 @76x136 -9 %, head -5 %); 64-column tiles only as 8 x 16 (64->64 @152x272: 0.30 ms against 0.37 im2col / 0.38 as 4 x 32)
        w32, w16 = halo_waste(H, W, 4, 32), halo_waste(H, W, 8, 16)
        use16 = P3_HALO16 and (w16 < w32 - 0.01 or 32 < Cout <= 64)
        th, tw, waste = (8, 16, w16) if use16 else (4, 32, w32)
        bn = 128 if Cout > 64 else (64 if Cout > 32 else 32)
        # the 32-column offset/mask convs run on the fp32 instruction otherwise (intra-workgroup split-K tiles): the halo form wins with
        # more padding and fewer tiles (256->27 @38x68, 16 frames: 0.097 -> 0.055 ms at 24 % padding and 400 tiles; @19x34 it loses)
        narrow = Cout <= 32
        if waste <= (P3_HALO_WASTE_NARROW if narrow else P3_HALO_WASTE) and (Cout >= 128 or narrow or use16) \
                and (M // (H * W)) * -(-H // th) * -(-W // tw) * -(-Cout // bn) >= (P3_MIN_TILES * 3 // 4 if narrow else P3_MIN_TILES):
            return ("halo", (((th << 16) | bn | P3H_W16) if use16 else 0) | (P3H_TPI3 if (P3H_TPI3_64 and bn == 64 and use16) else 0))
    if Cout < 64 or Cin < 64 or (stride != 1 and not (P3_STRIDE2 and stride == 2)):
        return None                     # stride-2 and 1x1 layers are no faster on the piece form (HBM- or issue-bound, tools/bench_p3.py; round 5 A/B with two pieces: below)
    # the ONE-stage loop with several workgroups per CU (48 / 37 KB of LDS: 3 / 4 of them) beats the 2-stage ring with one 8-wave
    # workgroup on every shape (profiles/r2_bench_p3.log: 256->256 @38x68 179 vs 150 TFLOP/s, 64->64 @152x272 148 vs 120)
    tile = (_T(64, 128) if Cout >= 128 else _T(128, 64)) | P3_1STAGE
    # round 6 (profiles/r6_im2col_tile_ab.log, two repetitions inside one call): the 128 x 128 tile on the TWO-stage ring -- half the weight
    # traffic from L2 per output row -- is +0.7 % of the config-B step on the deep layers (256->256 @38x68, 512->512 @19x34) where it still
    # leaves the launch enough tiles; the one-stage 64 x 128 tile (four workgroups per CU) otherwise
    if Cout >= 128 and P3_IM2COL_WIDE and -(-M // 128) * -(-Cout // 128) >= P3_MIN_TILES:
        tile = _T(128, 128)
    if P3_IM2COL_TILE and Cout >= 128:          # tuning aid: "BMxBN[:stages]" for the deep 128+-column layers
        tile = P3_IM2COL_TILE
    bm, bn = (tile >> 16) & 0x1fff, tile & 0xffff
    if -(-M // bm) * -(-Cout // bn) < P3_MIN_TILES:
        return None                     # few tiles (one frame per GPU): igemm.hip's cross-workgroup split-K fills the chip better
    return ("im2col", tile)


def dcn_patch_choice(N, H, W, Cin, Cout):
    """Does a DCN layer run on the patch form (csrc/dcn.hip)?  8 x 16 pixel tiles: the padding they add to the map and the number of
    workgroups decide (module constants above)."""
    if not (DCN_PATCH and PREC == 1) or Cin % 32 or Cout % 8:
        return False
    return H * W >= DCN_PATCH_MIN_HW and halo_waste(H, W, 8, 16) <= DCN_PATCH_WASTE and N * -(-H // 8) * -(-W // 16) * -(-Cout // (128 if Cout > 64 else 64)) >= DCN_PATCH_MIN_TILES


def _region(v):
    """(storage address, first channel, end channel) of a View -- channel ranges of one NHWC buffer are what concat buffers are written
    through (dla.py:176-181 Root) -- or (storage address, 0, inf) of a whole tensor.  The bf16-piece companion of a buffer (_Plan._p3)
    is written by the producer of the fp32 data and read by its consumers, so it needs no region of its own."""
    if isinstance(v, View):
        ch = v.c0 % v.ld
        return (v.buf.untyped_storage().data_ptr(), ch, ch + v.C)
    return (v.untyped_storage().data_ptr(), 0, 1 << 30)


class _Plan:
    """Common machinery: device buffers + an ordered list of bound C-ABI calls."""

    def __init__(self, device, lib=None):
        self.device = torch.device(device)
        self.lib = lib if lib is not None else hiplib.get_lib()
        if (self.device.type == "cuda") == bool(getattr(self.lib, "host_pointers", False)):
            raise hiplib.DeftHipError("deft_amd runs on an MI355X only: device %s with %s (there is no CPU path)"
                                      % (self.device, self.lib.path))
        self.np = int(getattr(self.lib, "pieces", 3))      # operand pieces of the library's split arithmetic: 3 (bf16) or 2 (fp16), csrc/common.h
        self._by_ptr = {}      # data_ptr -> device tensor uploaded through dev() (prescale looks weights / scales up by descriptor pointer)
        self._wscaled = {}     # packed weight data_ptr -> (row-scaled copy, exponents) of a two-piece build
        self._wscaled_ptrs = set()
        self._sscaled = {}     # (weight data_ptr, scale data_ptr or None, Cout) -> compensated epilogue scale
        self._p3 = {}          # id(fp32 buffer) -> 16-bit tensor holding its piece (P3) form
        self._p3_cover = {}    # id(fp32 buffer) -> [(ch_lo, ch_hi, producing descriptor or None)] channel ranges with valid P3 data
        self._p3_outs = []     # _P3Out records of every producer that can write a P3 copy of its output
        self._w3 = {}          # packed fp32 weight data_ptr -> P3 weight image
        self.ops = []          # (kind, name, callable, flops)
        self.io = []           # per op: None (= ordered against everything) or (regions read, regions written), see dependencies()
        self.sched = None      # set by build_schedule(): the op list spread over several HIP streams along its data dependencies
        self._gemms = []       # (entry, name, descriptor) of every implicit-GEMM launch, for autotune()
        self._op_desc = {}     # op index -> its descriptor (build_schedule gives split-K launches on a side stream their own workspace)
        self._keep = []        # tensors / descriptors kept alive
        self.profile = None    # when set to a list, run() appends (name, kind, flops, ms)

    def dev(self, t):
        t = t.contiguous().to(self.device)
        self._keep.append(t)
        self._by_ptr[t.data_ptr()] = t
        return t

    def _scaled_weight(self, w):
        """Two-piece builds: the row-scaled copy of a packed weight matrix (cached) and its exponents; else (w, None)."""
        if self.np != 2 or PREC != 1:
            return w, None
        hit = self._wscaled.get(w.data_ptr())
        if hit is None:
            k = weight_row_shift(w)
            hit = self._wscaled[w.data_ptr()] = (torch.ldexp(w, k.view(-1, 1)).contiguous(), k)
            self._wscaled_ptrs.add(hit[0].data_ptr())
            self._keep.append(w)
            self._by_ptr[hit[0].data_ptr()] = hit[0]
        return hit

    def prescale(self, d):
        """Two-piece (fp16) builds: point a split-arithmetic descriptor at the row-scaled weights and the compensated epilogue scale
        (weight_row_shift).  Idempotent; a no-op for three-piece builds and fp32-MFMA descriptors."""
        if self.np != 2 or d.prec != 1 or not d.w:
            return
        if d.w in self._wscaled_ptrs:
            return                                                       # already scaled
        w = self._by_ptr.get(d.w)
        if w is None:
            raise hiplib.DeftHipError("prescale: weight matrix %#x was not uploaded through _Plan.dev()" % d.w)
        w2, k = self._scaled_weight(w)
        key = (w.data_ptr(), d.scale or 0, d.Cout)
        sc2 = self._sscaled.get(key)
        if sc2 is None:
            sc = self._by_ptr.get(d.scale) if d.scale else None
            if d.scale and sc is None:
                raise hiplib.DeftHipError("prescale: scale vector %#x was not uploaded through _Plan.dev()" % d.scale)
            base = torch.ones(d.Cout, dtype=torch.float32, device=self.device) if sc is None else sc[:d.Cout]
            sc2 = self._sscaled[key] = torch.ldexp(base, -k[:d.Cout]).contiguous()
        d.w, d.scale = w2.data_ptr(), sc2.data_ptr()

    def alloc(self, N, H, W, C, ld=None):
        ld = _rup(C, 4) if ld is None else ld
        buf = torch.zeros(N * H * W * ld, dtype=torch.float32, device=self.device)
        self._keep.append(buf)
        return View(buf, N, H, W, C, ld)

    _stream_cache = None       # set for the duration of run(): one torch.cuda.current_stream() query per launch list

    def _stream(self):
        s = self._stream_cache
        return s if s is not None else hiplib.stream_ptr(self.device)

    def add(self, kind, name, fn, flops=0.0, reads=None, writes=None):
        """reads / writes: the Views (or whole tensors) the launch touches.  Ops that do not say are ordered against every other op."""
        self.ops.append((kind, name, fn, flops))
        self.io.append(None if reads is None or writes is None else
                       ([_region(v) for v in reads if v is not None], [_region(v) for v in writes if v is not None]))

    # ---- data-flow schedule: the launch list over several HIP streams --------------------------------------------------------------
    def dependencies(self):
        """deps[i] = the earlier ops launch i has to wait for: the last writers of what it reads, and the readers and writers since
        of what it writes (regions = channel ranges of NHWC buffers, _region()).  Ops without I/O information wait for everything
        before them and everything after them waits for them."""
        n = len(self.ops)
        deps = [set() for _ in range(n)]
        fence = -1
        writes, reads = {}, {}                         # storage -> [(lo, hi, op)]
        for i, io in enumerate(self.io):
            if io is None:
                deps[i] = set(range(max(fence, 0), i))
                fence, writes, reads = i, {}, {}
                continue
            if fence >= 0:
                deps[i].add(fence)
            R, W = io
            for st, lo, hi in R:
                deps[i].update(j for a, b, j in writes.get(st, ()) if a < hi and lo < b)
            for st, lo, hi in W:
                deps[i].update(j for a, b, j in writes.get(st, ()) if a < hi and lo < b)
                deps[i].update(j for a, b, j in reads.get(st, ()) if a < hi and lo < b)
            deps[i].discard(i)
            for st, lo, hi in R:
                reads.setdefault(st, []).append((lo, hi, i))
            for st, lo, hi in W:
                # entries this write covers completely are ordered before it now: later ops only need to order against this op
                writes[st] = [e for e in writes.get(st, ()) if not (lo <= e[0] and e[1] <= hi)] + [(lo, hi, i)]
                reads[st] = [e for e in reads.get(st, ()) if not (lo <= e[0] and e[1] <= hi) or e[2] == i]
        return deps

    def build_schedule(self, nstreams, dur_ms=None, sync_ms=0.003):
        """Spread the launch list over `nstreams` HIP streams along its data dependencies (one frame per GPU: most launches fill a
        fraction of the chip, and the up-path of DLA-34 has independent branches -- dla.py:693-699 projects every level before it
        merges them, and three of DLAUp's projections read backbone outputs only).  List scheduling in program order: every op goes
        to the stream where it can start first, given measured (or estimated) durations; a stream runs its ops in program order, an
        event orders an op after a dependency on another stream.  Results cannot change: the same launches on the same buffers, every
        read after its write.  Returns the modelled (serial, scheduled) time in ms."""
        n = len(self.ops)
        deps = self.dependencies()
        if dur_ms is None:
            dur_ms = [0.004 + f / 1.0e11 for _, _, _, f in self.ops]
        assert len(dur_ms) == n and nstreams >= 1
        succ = [[] for _ in range(n)]
        for i in range(n):
            for j in deps[i]:
                succ[j].append(i)
        level = [0.0] * n                                  # longest path from the start of op i to the end of the list
        for i in range(n - 1, -1, -1):
            level[i] = dur_ms[i] + max((level[k] for k in succ[i]), default=0.0)
        where, finish, free, order = [0] * n, [0.0] * n, [0.0] * nstreams, []
        missing = [len(deps[i]) for i in range(n)]
        ready = [i for i in range(n) if not missing[i]]
        while ready:
            i = max(ready, key=lambda k: (level[k], -k))   # the op with the longest tail first; program order among equals
            ready.remove(i)
            late = max(deps[i], key=lambda j: finish[j], default=None)
            s_, start = None, None
            for c in range(1 if self.io[i] is None else nstreams):          # ops without I/O information stay on the first stream
                t = max([free[c]] + [finish[j] + (sync_ms if where[j] != c else 0.0) for j in deps[i]])
                better = start is None or t < start - 1e-9
                if not better and abs(t - start) <= 1e-9 and late is not None and where[late] == c and where[late] != s_:
                    better = True                           # tie: continue the chain of the dependency that finishes last
                if better:
                    s_, start = c, t
            where[i], finish[i] = s_, start + dur_ms[i]
            free[s_] = finish[i]
            order.append(i)
            for k in succ[i]:
                missing[k] -= 1
                if not missing[k]:
                    ready.append(k)
        assert len(order) == n
        # events: op i waits for its latest dependency on every OTHER stream, unless an earlier op of i's stream already waited for it
        pos = {i: k for k, i in enumerate(order)}
        waits, seen = [[] for _ in range(n)], [[-1] * nstreams for _ in range(nstreams)]
        for i in order:
            need = {}
            for j in deps[i]:
                if where[j] != where[i] and pos[j] > need.get(where[j], (-1, None))[0]:
                    need[where[j]] = (pos[j], j)
            for c, (pj, j) in sorted(need.items()):
                if pj > seen[where[i]][c]:
                    waits[i].append(j)
                    seen[where[i]][c] = pj
        signals = sorted({j for w in waits for j in w})
        self.sched = {"n": nstreams, "where": where, "waits": waits, "signals": signals, "order": order, "streams": None, "events": None,
                      "model_ms": (sum(dur_ms), max(finish) if n else 0.0)}
        # cross-workgroup split-K launches share one workspace per plan: one per stream now.  The extra workspaces are keyed by STREAM
        # INDEX and kept across rebuilds; every split-K descriptor is re-assigned from `where` on every build (a second tune_schedule /
        # tune_dataflow(nstreams=..) must not leave a descriptor on the workspace of the stream it sat on before), and reset_splitk
        # zeroes all of them.
        sp = getattr(self, "_split", None)
        if sp is not None and sp["ws"] is not None:
            extra = sp.setdefault("extra", {})
            for i, d in self._op_desc.items():
                if d.splitk <= 1:
                    continue
                c = where[i] if nstreams > 1 else 0
                if c == 0:
                    d.ws, d.ws_cnt = sp["ws"].data_ptr(), sp["cnt"].data_ptr()
                    continue
                if c not in extra or extra[c][0].numel() < sp["ws"].numel() or extra[c][1].numel() < sp["cnt"].numel():
                    extra[c] = (torch.empty_like(sp["ws"]), torch.zeros_like(sp["cnt"]))
                    self._keep += list(extra[c])
                d.ws, d.ws_cnt = extra[c][0].data_ptr(), extra[c][1].data_ptr()
        return self.sched["model_ms"]

    def tune_schedule(self, nstreams=None, reps=3):
        """Measure every launch of the list alone (HIP events, best of `reps`) and build the data-flow schedule from those durations.
        Returns the modelled (serial, scheduled) ms, or None when the list stays on one stream."""
        if nstreams is None:
            nstreams = DATAFLOW if getattr(self, "N", 1) <= DATAFLOW_MAX_N else 1
        # Two streams is what is shipped and what was exercised (~200 captures of the real plans and 30 of a small one without an incident).
        # THREE streams: hipStreamEndCapture segfaulted in 3 of ~25 runs of the capture test on a small plan (2 x 96 x 160; inside the runtime,
        # with or without empty branches) -- and three were no faster (2.12 vs 2.07 ms).  So captures are held to two.
        nstreams = min(nstreams, 2)
        self.sched = None
        if self.device.type != "cuda" or nstreams <= 1:
            return None
        dur = None
        for _ in range(reps + 1):
            self.profile = []
            try:
                self.run()
                torch.cuda.synchronize(self.device)
                ms = [e0.elapsed_time(e1) for (_, _, _, e0, e1) in self.profile]
            finally:
                self.profile = None
            dur = ms if dur is None else [min(a, b) for a, b in zip(dur, ms)]
        model = self.build_schedule(nstreams, [max(0.002, t - 0.005) for t in dur])       # an event pair around a launch costs ~5 us
        self.run()                                         # creates the side streams and events (not inside a graph capture)
        torch.cuda.synchronize(self.device)
        return model

    def capture_graph(self, then=None):
        """The launch list (plus whatever `then()` launches behind it) as a hipGraph; small-batch lists are spread over several streams
        first (tune_schedule), which become parallel branches of the graph.  Buffers are plan-owned and static, so replays are valid."""
        assert self.device.type == "cuda"
        if self.sched is None and DATAFLOW > 1:
            self.tune_schedule()
        import gc
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        # no cyclic garbage collection while the stream is capturing: a collected object that owns a HIP resource (an event, a stream, another graph of
        # a Detector that went out of scope) would call into the runtime in the middle of the capture
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            cap = torch.cuda.Stream(device=self.device)
            if _os.environ.get("DEFT_DEBUG_CAPTURE") == "1":
                import sys as _sys
                side = [] if not self.sched or self.sched.get("streams") is None else [s_.cuda_stream for s_ in self.sched["streams"][1:]]
                _sys.stderr.write("capture_graph: capture stream %#x, side streams %s, sched n=%s\n" % (cap.cuda_stream, [hex(v) for v in side], None if not self.sched else self.sched.get("n")))
                if self.sched:
                    import json as _json
                    _Plan._dbg_n = getattr(_Plan, "_dbg_n", 0) + 1
                    _os.makedirs("gpurun_out/r6x", exist_ok=True)
                    _json.dump({"shape": [getattr(self, "N", None), getattr(self, "H", None), getattr(self, "W", None)], "kinds": [o[0] for o in self.ops],
                                "names": [o[1] for o in self.ops], **{k: self.sched[k] for k in ("n", "where", "waits", "signals", "order")}},
                               open("gpurun_out/r6x/sched_%02d.json" % _Plan._dbg_n, "w"))
            with torch.cuda.graph(g, stream=cap):
                self.run()
                if then is not None:
                    then()
        finally:
            if was_enabled:
                gc.enable()
        return g

    @staticmethod
    def replay_graph(g, device):
        """g.replay(): the one place the package launches a captured hipGraph from.  (DEFT_REPLAY_STREAM=1: a replay that would land on the NULL
        stream runs on a process-wide side stream instead -- the first workaround tried for the runtime fault described at DATAFLOW above; it
        survives the first reproducer and dies in another, so it is off and graphs are captured without parallel branches instead.)"""
        cur = torch.cuda.current_stream(device) if REPLAY_STREAM else None
        if cur is None or cur.cuda_stream != 0:
            g.replay()
            return
        rs = _Plan._replay_streams.get(device.index)
        if rs is None:
            rs = _Plan._replay_streams[device.index] = torch.cuda.Stream(device=device)
        rs.wait_stream(cur)
        with torch.cuda.stream(rs):
            g.replay()
        cur.wait_stream(rs)

    _replay_streams = {}

    def _run_dataflow(self):
        sc = self.sched
        main = torch.cuda.current_stream(self.device)
        if sc["streams"] is None:
            sc["streams"] = [None] + [torch.cuda.Stream(device=self.device) for _ in range(sc["n"] - 1)]
            sc["events"] = {j: torch.cuda.Event() for j in sc["signals"]}
            sc["fork"], sc["join"] = torch.cuda.Event(), [torch.cuda.Event() for _ in range(sc["n"])]
        streams = [main] + sc["streams"][1:]
        used = sorted(set(sc["where"]) - {0})            # only side streams that carry launches fork and join (no empty graph branches)
        sc["fork"].record(main)
        for c in used:
            streams[c].wait_event(sc["fork"])            # side streams start behind whatever precedes the plan (and join a graph capture)
        where, waits, events = sc["where"], sc["waits"], sc["events"]
        for i in sc["order"]:                            # a topological order: every event is recorded before it is waited for
            fn = self.ops[i][2]
            st = streams[where[i]]
            for j in waits[i]:
                st.wait_event(events[j])
            self._stream_cache = C.c_void_p(st.cuda_stream)
            with torch.cuda.stream(st):
                fn()
            if i in events:
                events[i].record(st)
        for c in used:
            sc["join"][c].record(streams[c])
            main.wait_event(sc["join"][c])

    def run(self):
        self._stream_cache = hiplib.stream_ptr(self.device)
        try:
            if self.sched is not None and self.sched["n"] > 1 and self.profile is None and self.device.type == "cuda":
                self._run_dataflow()
            else:
                self._run_ops()
        except hiplib.DeftHipError:
            self.reset_splitk()            # a launch list that stopped half way may leave split-K tickets taken: start clean next time
            raise
        finally:
            self._stream_cache = None

    def reset_splitk(self):
        """Zero the cross-workgroup split-K ticket counters (DeftGemmDesc.ws_cnt).  They are zero after every COMPLETED launch (the
        last arriver resets its tile's counter); only an aborted launch can leave them non-zero."""
        sp = getattr(self, "_split", None)
        if sp is not None and sp["cnt"] is not None:
            sp["cnt"].zero_()
            for _, cnt in sp.get("extra", {}).values():
                cnt.zero_()

    def _run_ops(self):
        if self.profile is None:
            for _, _, fn, _ in self.ops:
                fn()
            return
        for kind, name, fn, flops in self.ops:
            if self.device.type == "cuda":
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                self.profile.append((name, kind, flops, e0, e1))
            else:
                fn()

    # ---- pre-split (P3) bookkeeping ----------------------------------------
    def p3_capable(self, v):
        return v.ld % 32 == 0 and (v.c0 % v.ld) % 32 == 0 and v.C % 32 == 0

    def p3_addr(self, v):
        """Address of view v inside the P3 companion of its buffer (allocated on first use)."""
        assert self.p3_capable(v)
        t = self._p3.get(id(v.buf))
        if t is None:
            t = torch.zeros(v.buf.numel() * self.np, dtype=torch.int16, device=self.device)      # (bf16 or fp16 pieces: the kernels' business)
            self._p3[id(v.buf)] = t
        pix, ch = divmod(v.c0, v.ld)
        return t.data_ptr() + 2 * (pix * self.np * v.ld + (ch // 32) * 32 * self.np)

    def _p3_input(self, name, v):
        """P3 address of an input view; channel ranges no producer has written in P3 form yet are converted now
        (deft_split_planes, one pass over the view) -- the op list is in program order, so the producer has run."""
        ch = v.c0 % v.ld
        need = [(ch, ch + v.C)]
        for lo, hi, d in self._p3_cover.get(id(v.buf), []):
            nxt = []
            for a, b in need:
                if hi <= a or lo >= b:
                    nxt.append((a, b)); continue
                if d is not None:
                    d.used = True
                if a < lo: nxt.append((a, lo))
                if hi < b: nxt.append((hi, b))
            need = nxt
        addr = self.p3_addr(v)
        if need:
            lib = self.lib
            a = (C.c_void_p(v.addr), C.c_void_p(addr), C.c_longlong(v.N * v.H * v.W), v.C, v.ld, v.ld)
            self.add("deft_split_planes", name + ".p3", lambda: lib.call("deft_split_planes", *a, self._stream()), reads=[v], writes=[v])
            self._p3_cover.setdefault(id(v.buf), []).append((ch, ch + v.C, None))
        return addr

    def weights_p3(self, w_packed, halo=False):
        """P3 image of a packed fp32 weight matrix (deft_split_weights / deft_split_weights_halo / deft_split_weights_dcn, once per matrix).
        halo = "dcn": the patch form's image of a DCN weight matrix (pack_dcn_weight)."""
        key = (w_packed.data_ptr(), halo)
        if key not in self._w3:
            w3 = torch.empty(w_packed.numel() * self.np, dtype=torch.int16, device=self.device)
            w_packed = self._scaled_weight(w_packed)[0]              # (two-piece builds: the row-scaled matrix; prescale() compensates in the epilogue scale)
            if halo == "dcn":
                self.lib.call("deft_split_weights_dcn", ptr(w_packed), C.c_void_p(w3.data_ptr()), w_packed.shape[0], w_packed.shape[1] // 9,
                              hiplib.stream_ptr(self.device))
            else:
                self.lib.call("deft_split_weights_halo" if halo else "deft_split_weights", ptr(w_packed), C.c_void_p(w3.data_ptr()),
                              w_packed.shape[0], w_packed.shape[1], hiplib.stream_ptr(self.device))
            self._w3[key] = w3
            self._keep.append(w_packed)
        return self._w3[key]

    def p3_output(self, out, desc=None):
        """Register `out` as producible in P3 form -> _P3Out (its `used` flag is set when a pre-split conv reads it) or None."""
        if not (P3 and PREC == 1 and self.p3_capable(out)):
            return None
        h = _P3Out(self.p3_addr(out), out.ld, desc)
        ch = out.c0 % out.ld
        self._p3_cover.setdefault(id(out.buf), []).append((ch, ch + out.C, h))
        self._p3_outs.append(h)
        return h

    def finalize_p3(self):
        """Drop the P3 outputs nobody reads (call once the whole launch list is built)."""
        for h in self._p3_outs:
            if h.desc is not None:
                h.desc.y3, h.desc.ldy3 = (h.addr, h.ld) if h.used else (None, 0)

    # ---- op builders -------------------------------------------------------
    def gemm(self, entry, name, desc, flops, reads=None, writes=None):
        desc.prec = PREC
        self.prescale(desc)
        self._keep.append(desc)
        lib, ref = self.lib, C.byref(desc)
        self._op_desc[len(self.ops)] = desc
        self.add(entry, name, lambda: lib.call(entry, ref, self._stream()), flops, reads, writes)
        self._gemms.append((entry, name, desc))
        if SPLITK and entry in ("deft_conv2d_nhwc", "deft_dcn_v2_nhwc") and not desc.splitk and not desc.p3_kernel and not desc.fold_y:
            self._plan_splitk(entry, desc)

    def _plan_splitk(self, entry, desc):
        """Launches with fewer output tiles than the chip has compute units (one frame per GPU: the 19x34 and
        38x68 maps) split K across workgroups (DeftGemmDesc.splitk).  The library picks tile and split factor
        (`deft_gemm_plan`); the plan owns ONE workspace + ticket array shared by all its launches (they are
        ordered on the plan's stream), sized for the largest."""
        tile, S, wsf, wst = C.c_int(), C.c_int(), C.c_longlong(), C.c_int()
        rc = self.lib._fn["deft_gemm_plan"](C.byref(desc), 0 if entry == "deft_conv2d_nhwc" else 1, C.byref(tile), C.byref(S),
                                          C.byref(wsf), C.byref(wst))
        if rc != 0:
            raise hiplib.DeftHipError("deft_gemm_plan failed (%d): %s" % (rc, self.lib.last_error()))
        if S.value <= 1:
            return
        desc.tile, desc.splitk = tile.value, S.value
        if not hasattr(self, "_split"):
            self._split = {"descs": [], "floats": 0, "tiles": 0, "ws": None, "cnt": None}
        sp = self._split
        sp["descs"].append(desc)
        if wsf.value > sp["floats"] or wst.value > sp["tiles"] or sp["ws"] is None:
            sp["floats"], sp["tiles"] = max(sp["floats"], wsf.value), max(sp["tiles"], wst.value)
            sp["ws"] = torch.empty(sp["floats"], dtype=torch.float32, device=self.device)
            sp["cnt"] = torch.zeros(sp["tiles"], dtype=torch.int32, device=self.device)
            for d in sp["descs"]:
                d.ws, d.ws_cnt = sp["ws"].data_ptr(), sp["cnt"].data_ptr()
        desc.ws, desc.ws_cnt = sp["ws"].data_ptr(), sp["cnt"].data_ptr()

    def autotune(self, reps=4, verbose=False):
        """Per-layer tile search on the GPU this plan will run on (the cuDNN-benchmark / MIOpen-find
        step of the reference stack, done once per plan): every implicit-GEMM launch is timed alone
        with each candidate (tile, loop form) and the fastest is written into its descriptor.
        Only result-identical candidates are tried, so tuning never changes an output bit
        (tests/test_gpu_parity.py::test_autotune_keeps_every_bit): with prec 0 every WK = 1 tile and both loop forms are the
        same k-ordered fp32 chain; with prec 1 only the 1-stage BN >= 64 tiles implement the split-bf16 arithmetic
        (launch_igemm falls back to the fp32 MFMA for the 2-stage form and BN = 32), so only those are candidates.
        Launches on the pre-split kernels (x3) and cross-workgroup split-K launches keep their configuration."""
        if self.device.type != "cuda":
            return
        T = lambda bm, bn: (bm << 16) | bn
        two = 1 << 29
        cands = {"deft_conv2d_nhwc": [T(128, 128), T(128, 64), T(64, 64), T(64, 128)],
                 "deft_dcn_v2_nhwc": [T(64, 64), T(64, 128), T(128, 64)],
                 "deft_pair_layer": [T(128, 128), T(128, 64), T(64, 64)]}
        lib, s = self.lib, self._stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        memo = {}
        for entry, name, d in self._gemms:
            if d.splitk > 1 or d.x3 or d.p3_kernel:       # cross-workgroup split-K launches keep their (tile, split, workspace); so do the x3 kernels
                continue
            key = (entry, d.M, d.Cout, d.Ktot, d.Cin, d.KH, d.KW, d.stride, d.H, d.W, d.ldx, d.ldy, bool(d.res), bool(d.rowmap))
            if key not in memo:
                if d.Cout <= 32 or d.rowmap:
                    opts = [0, two] if not d.rowmap else [d.tile]          # keep the (split-K) tile, try both loop forms
                    if d.Cout <= 32 and -(-(d.OH * d.OW) // 128) >= 256:
                        opts = [T(128, 32), T(128, 32) | two]
                else:
                    opts = [t | st for t in cands[entry] for st in ((0,) if d.prec == 1 else (0, two))]
                best, best_t = d.tile, None
                for t in opts:
                    d.tile = t
                    try:
                        lib.call(entry, C.byref(d), s)                      # warm-up + validity
                    except hiplib.DeftHipError:
                        continue
                    e0.record()
                    for _ in range(reps):
                        lib.call(entry, C.byref(d), s)
                    e1.record(); e1.synchronize()
                    ms = e0.elapsed_time(e1) / reps
                    if best_t is None or ms < best_t:
                        best, best_t = t, ms
                memo[key] = best
                if verbose:
                    print("autotune %-28s M=%d N=%d K=%d -> %dx%d %s (%.3f ms)" % (name, d.M, d.Cout, d.Ktot, (best >> 16) & 0x1fff, best & 0xffff,
                                                                            "2st" if best & two else "1st", best_t or 0.0))
            d.tile = memo[key]

    def conv(self, name, x, w_packed, K, KH, KW, stride, pad, Cout, scale, shift, relu, out=None, res=None, tile=0,
             true_cin=None, korder=None, p3=None, true_cout=None, fold=None):
        """fold = (w [n][Cout] device fp32, bias [n] or None, n): a following 1x1 conv with n <= 16 outputs.  When this conv runs on
        a pre-split kernel it is folded into the epilogue (DeftGemmDesc.fold_w) and the return value is the [N, OH, OW, n] map of
        the 1x1 conv -- the Cout-channel map is never written; otherwise `fold` is ignored and the caller adds the 1x1 conv."""
        OH = (x.H + 2 * pad - KH) // stride + 1
        OW = (x.W + 2 * pad - KW) // stride + 1
        if out is None:
            out = self.alloc(x.N, OH, OW, Cout) if fold is None else _LazyView(self, x.N, OH, OW, Cout)
        assert (out.N, out.H, out.W, out.C) == (x.N, OH, OW, Cout), (name, out.H, out.W, out.C, OH, OW, Cout)
        if res is not None:
            assert (res.H, res.W, res.C) == (OH, OW, Cout), name
        d = GemmDesc()
        for t_ in (w_packed, scale):                     # (callers outside the plans hand over tensors that did not go through dev())
            if t_ is not None:
                self._by_ptr.setdefault(t_.data_ptr(), t_)
        d.x = x.addr; d.x2 = None; d.w = w_packed.data_ptr()
        d.scale = scale.data_ptr() if scale is not None else None
        d.shift = shift.data_ptr() if shift is not None else None
        d.res = res.addr if res is not None else None
        d.y = out.addr if out.buf is not None else None
        d.N, d.H, d.W, d.Cin, d.ldx = x.N, x.H, x.W, x.C, x.ld
        d.OH, d.OW, d.Cout, d.ldy, d.ldr = OH, OW, Cout, out.ld, (res.ld if res is not None else 0)
        d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
        d.Ktot, d.Kpad = K, w_packed.shape[1]
        d.cin_log2 = int(math.log2(x.C)) if KH * KW > 1 else 0
        d.M = x.N * OH * OW
        d.relu = int(relu); d.Q = 0; d.ldom = 0; d.tile = tile
        d.korder = conv_korder((Cout, x.C, KH, KW)) if korder is None else korder
        # ---- pre-split operands: which kernel (p3: None = by shape, False = never, "halo" / "im2col" = forced by a test) ----
        choice = None
        if P3 and PREC == 1 and p3 is not False and self.p3_capable(x) and K == w_packed.shape[1] and out.ld % 4 == 0 and out.c0 % 4 == 0 \
                and Cout % 8 == 0 and (KH * KW == 1 or x.C & (x.C - 1) == 0) and KH * KW <= 32:
            if p3 == "halo":
                assert (KH, KW, stride, pad) == (3, 3, 1, 1) and d.korder == 1, "halo form needs 3x3 / stride 1 / pad 1 and korder 1"
                choice = ("halo", tile)
            elif p3 in ("im2col", True):
                choice = ("im2col", tile) if tile in P3_TILES else None
            elif tile == 0:
                choice = p3_choice(KH, KW, stride, pad, x.C, Cout, x.H, x.W, d.M, d.korder)
            elif tile in P3_TILES and Cout >= P3_MIN_COUT:
                choice = ("im2col", tile)          # a forced igemm3 tile
        folded = None
        if fold is not None:
            if choice is not None and FOLD and res is None:
                fw, fb, fn = fold
                halo = choice[0] == "halo"
                t = choice[1] if choice[1] & 0xffff else ((4 << 16) | (128 if Cout > 64 else (64 if Cout > 32 else 32)) if halo else 0)
                if t & 0xffff:
                    choice = (choice[0], t)
                    nparts = -(-Cout // (t & 0xffff))
                    part = torch.empty(nparts * d.M * fn, dtype=torch.float32, device=self.device)
                    folded = (part, nparts, fw, fb, fn)
            if folded is None and out.buf is None:
                out = out.materialize()
                d.y = out.addr; d.ldy = out.ld
        if choice is not None:
            halo = choice[0] == "halo"
            d.tile = choice[1]
            d.x3, d.ldx3 = self._p3_input(name, x), x.ld
            d.p3_kernel = 1 if halo else 0
            d.w3 = self.weights_p3(w_packed, halo).data_ptr()
            if Cout % 32 == 0 and folded is None:
                h = self.p3_output(out, d)
                if h is not None:
                    d.y3, d.ldy3 = h.addr, h.ld
        if choice is None and BDMA and PREC == 1 and Cout >= 128:
            d.w3 = self.weights_p3(w_packed).data_ptr()         # igemm.hip: weight chunks by DMA, activations split in the loop
        if choice is None and Y3_INLOOP and P3 and PREC == 1 and tile == 0 and Cout >= 64 and Cout % 32 == 0 and out.buf is not None \
                and out.ld % 4 == 0 and out.c0 % 4 == 0 and (res is None or res.ld % 4 == 0):
            # the in-loop kernel's split-bf16 tiles (BN >= 64: every automatic tile from Cout = 64) can write the piece form too:
            # no deft_split_planes pass for a pre-split conv that reads this output
            h = self.p3_output(out, d)
            if h is not None:
                d.y3, d.ldy3 = h.addr, h.ld
        cin = x.C if true_cin is None else true_cin
        d.flop_k = KH * KW * cin
        d.flop_n = 0 if true_cout is None else true_cout
        if folded is not None:
            part, nparts, fw, fb, fn = folded
            d.y = None; d.ldy = Cout
            d.fold_w, d.fold_y, d.fold_n, d.fold_ld = fw.data_ptr(), part.data_ptr(), fn, fn
            self._keep += [part, fw, fb]
        self.gemm("deft_conv2d_nhwc", name, d, 2.0 * d.M * (Cout if true_cout is None else true_cout) * KH * KW * cin,
                  reads=[x, res], writes=[folded[0] if folded is not None else out])
        if folded is not None:
            y2 = self.alloc(x.N, OH, OW, fn)
            lib = self.lib
            a = (ptr(part), nparts, C.c_longlong(d.M), fn, fn, ptr(fb), C.c_void_p(y2.addr), y2.ld)
            self.add("deft_fold_finish", name + ".fold", lambda: lib.call("deft_fold_finish", *a, self._stream()), 2.0 * d.M * fn * Cout,
                     reads=[part], writes=[y2])
            return y2
        return out

    def conv_pair(self, name, x, w_packed, K, KH, KW, pad, Cout, scale2, shift2, relu, true_k):
        """Stride-1 conv with Cout <= 16 computed on pixel pairs (see pack_pair_conv_weight):
        GEMM rows = (n, oy, ox/2), 2*Cout columns, window KH x (KW+1), stride_w 2."""
        OH, OW = x.H + 2 * pad - KH + 1, x.W + 2 * pad - KW + 1
        assert OW % 2 == 0 and Cout % 4 == 0
        out = self.alloc(x.N, OH, OW, Cout)
        assert out.ld == Cout
        d = GemmDesc()
        d.x = x.addr; d.x2 = None; d.w = w_packed.data_ptr()
        d.scale = scale2.data_ptr() if scale2 is not None else None
        d.shift = shift2.data_ptr() if shift2 is not None else None
        d.res = None; d.y = out.addr
        d.N, d.H, d.W, d.Cin, d.ldx = x.N, x.H, x.W, x.C, x.ld
        d.OH, d.OW, d.Cout, d.ldy, d.ldr = OH, OW // 2, 2 * Cout, 2 * Cout, 0
        d.KH, d.KW, d.stride, d.pad, d.stride_w = KH, KW + 1, 1, pad, 2
        d.Ktot, d.Kpad = K, w_packed.shape[1]
        d.cin_log2 = int(math.log2(x.C))
        d.M = x.N * OH * (OW // 2)
        d.relu = int(relu); d.Q = 0; d.ldom = 0; d.tile = 0
        d.flop_k = true_k
        self.gemm("deft_conv2d_nhwc", name, d, 2.0 * x.N * OH * OW * Cout * true_k, reads=[x], writes=[out])
        return out

    def conv_direct(self, name, x, w_packed, K, KH, pad, Cout, scale, shift, relu, true_cin, stride=1):
        """Conv with <= 16 (stride 2: <= 32) output channels on the patch-in-LDS kernel (deft_conv_direct): x is fp32 NHWC with 16
        (or, for the image, 4) channels; `w_packed` the ordinary packed fp32 matrix, re-laid once into B fragments."""
        assert pad == KH // 2 and x.C in (4, 16) and Cout <= 16 * stride and stride in (1, 2)
        OH, OW = (x.H + 2 * pad - KH) // stride + 1, (x.W + 2 * pad - KH) // stride + 1
        out = self.alloc(x.N, OH, OW, Cout)
        key = (w_packed.data_ptr(), "direct")
        if key not in self._w3:
            nb = self.lib._fn["deft_direct_weight_bytes"](KH, KH, x.C, Cout)
            assert nb > 0
            w3 = torch.empty(nb, dtype=torch.uint8, device=self.device)
            self._by_ptr.setdefault(w_packed.data_ptr(), w_packed)
            self.lib.call("deft_split_weights_direct", ptr(self._scaled_weight(w_packed)[0]), ptr(w3), Cout, w_packed.shape[1], KH, KH, x.C, hiplib.stream_ptr(self.device))
            self._w3[key] = w3
            self._keep.append(w_packed)
        d = GemmDesc()
        d.x = x.addr; d.w = w_packed.data_ptr(); d.w3 = self._w3[key].data_ptr()
        d.scale = scale.data_ptr() if scale is not None else None
        d.shift = shift.data_ptr() if shift is not None else None
        d.y = out.addr
        d.N, d.H, d.W, d.Cin, d.ldx = x.N, x.H, x.W, x.C, x.ld
        d.OH, d.OW, d.Cout, d.ldy, d.ldr = OH, OW, Cout, out.ld, 0
        d.KH, d.KW, d.stride, d.pad = KH, KH, stride, pad
        d.Ktot, d.Kpad = K, w_packed.shape[1]
        d.M = x.N * OH * OW
        d.relu = int(relu)
        d.flop_k = KH * KH * true_cin
        d.prec = 1
        if scale is not None:
            self._by_ptr.setdefault(scale.data_ptr(), scale)
        self.prescale(d)                     # (two-piece builds: the image above was built from the row-scaled matrix; the scale compensates)
        self._keep.append(d)
        if name == "base_layer":
            self._base_desc = d
        lib, ref = self.lib, C.byref(d)
        self.add("deft_conv_direct", name, lambda: lib.call("deft_conv_direct", ref, self._stream()), 2.0 * d.M * Cout * KH * KH * true_cin,
                 reads=[x], writes=[out])
        return out

    def maxpool(self, name, x, out=None):
        if out is None:
            out = self.alloc(x.N, x.H // 2, x.W // 2, x.C)
        lib = self.lib
        a = (C.c_void_p(x.addr), C.c_void_p(out.addr), x.N, x.H, x.W, x.C, x.ld, out.ld)
        self.add("deft_maxpool2x2", name, lambda: lib.call("deft_maxpool2x2", *a, self._stream()), reads=[x], writes=[out])
        return out

    def upsample_add(self, name, x, wup, skip, f):
        out = self.alloc(x.N, x.H * f, x.W * f, x.C)
        assert (skip.H, skip.W, skip.C) == (out.H, out.W, out.C), name
        lib = self.lib
        a = (C.c_void_p(x.addr), ptr(wup), C.c_void_p(skip.addr), C.c_void_p(out.addr),
             x.N, x.H, x.W, x.C, f, x.ld, skip.ld, out.ld)
        h = self.p3_output(out)                       # written in P3 form too if a pre-split conv reads it (decided by run time)
        self.add("deft_upsample_add", name, lambda: lib.call("deft_upsample_add", *a, C.c_void_p(h.addr) if h is not None and h.used else None,
                                                             h.ld if h is not None else 0, self._stream()), reads=[x, skip], writes=[out])
        return out


class DlaSegPlan(_Plan):
    """DLA-34 + DLAUp/IDAUp (DCNv2) + heads + decode for a fixed (N, H, W)."""

    def __init__(self, sd, N, H, W, dataset="mot", K=100, device="cuda", lib=None, dense_heads=False):
        super().__init__(device, lib)
        assert H % 32 == 0 and W % 32 == 0, "DLA-34 needs H, W divisible by 32"
        self.sd, self.N, self.H, self.W, self.K = sd, N, H, W, K
        self.heads = HEADS[dataset]
        self.dataset = dataset
        self._wcache = {}
        self.image = torch.zeros(N, 3, H, W, dtype=torch.float32, device=self.device)
        x4 = self.alloc(N, H, W, 4)
        self._x4 = x4
        lib = self.lib
        a = (ptr(self.image), C.c_void_p(x4.addr), N, 3, H, W, 4)
        self.add("deft_nchw_to_nhwc", "image", lambda: lib.call("deft_nchw_to_nhwc", *a, self._stream()), reads=[self.image], writes=[x4])
        self._image_op = self.ops[0], self.io[0]
        self._build_base(x4)
        self._read_image_planes(True)
        self._build_neck()
        self._build_heads(dense_heads)
        self.finalize_p3()

    # ---- weights -----------------------------------------------------------
    def _conv_bn(self, name, x, wkey, bnkey, KH, stride, pad, relu, out=None, res=None, cin_pad=None):
        w = self.sd[wkey + ".weight"]
        if (DIRECT and PREC == 1 and w.shape[0] <= 16 * stride and out is None and res is None and pad == KH // 2
                and ((KH, x.C, stride) in ((3, 16, 1), (7, 4, 1), (3, 16, 2))) and x.N * ((x.H + 7) // 8) * ((x.W + 31) // 32) >= P3_MIN_TILES * stride * stride):
            if ("d", wkey) not in self._wcache:
                wp, K = pack_conv_weight(w, cin_pad)
                alpha, beta = _bn_fold(self.sd, bnkey)
                self._wcache[("d", wkey)] = (self.dev(wp), K, self.dev(alpha), self.dev(beta))
            wp, K, alpha, beta = self._wcache[("d", wkey)]
            return self.conv_direct(name, x, wp, K, KH, pad, w.shape[0], alpha, beta, relu, w.shape[1], stride=stride)
        if w.shape[0] <= 16 and stride == 1 and out is None and res is None and x.W % 2 == 0 and KH > 1:
            # 16-channel full-resolution layers (base_layer, level0): pixel-pair GEMM, 32 useful columns
            if wkey not in self._wcache:
                wp, K = pack_pair_conv_weight(w, cin_pad)
                alpha, beta = _bn_fold(self.sd, bnkey)
                self._wcache[wkey] = (self.dev(wp), K, self.dev(torch.cat([alpha, alpha])), self.dev(torch.cat([beta, beta])))
            wp, K, alpha2, beta2 = self._wcache[wkey]
            return self.conv_pair(name, x, wp, K, KH, KH, pad, w.shape[0], alpha2, beta2, relu, KH * KH * w.shape[1])
        if wkey not in self._wcache:
            wp, K = pack_conv_weight(self.sd[wkey + ".weight"], cin_pad)
            alpha, beta = _bn_fold(self.sd, bnkey)
            self._wcache[wkey] = (self.dev(wp), K, self.dev(alpha), self.dev(beta))
        wp, K, alpha, beta = self._wcache[wkey]
        Cout = self.sd[wkey + ".weight"].shape[0]
        return self.conv(name, x, wp, K, KH, KH, stride, pad, Cout, alpha, beta, relu, out=out, res=res,
                         true_cin=self.sd[wkey + ".weight"].shape[1])

    def _block(self, p, x, stride, residual, out):
        """BasicBlock dla.py:73-87."""
        cout = self.sd[p + ".conv1.weight"].shape[0]
        t = self._conv_bn(p + ".conv1", x, p + ".conv1", p + ".bn1", 3, stride, 1, True)
        return self._conv_bn(p + ".conv2", t, p + ".conv2", p + ".bn2", 3, 1, 1, True, out=out, res=residual)

    def _tree1(self, p, x, cin, cout, stride, level_root, cat=None, bottom=None, out=None):
        """Tree(levels=1) dla.py:271-284; `cat` = the Root's concat buffer
        [x2 | x1 | children...] (children already written when passed in)."""
        N = x.N
        oh, ow = x.H // stride, x.W // stride
        if cat is None:
            cat = self.alloc(N, oh, ow, 2 * cout + (cin if level_root else 0))
        if bottom is None:
            if stride > 1:
                bottom = self.maxpool(p + ".downsample", x, out=cat.sub(2 * cout, cin) if level_root else None)
            else:
                bottom = x
        if (p + ".project.0.weight") in self.sd:
            residual = self._conv_bn(p + ".project", bottom, p + ".project.0", p + ".project.1", 1, 1, 0, False)
        else:
            residual = bottom
        x1 = self._block(p + ".tree1", x, stride, residual, cat.sub(cout, cout))
        self._block(p + ".tree2", x1, 1, x1, cat.sub(0, cout))
        root_in = View(cat.buf, cat.N, cat.H, cat.W, cat.C, cat.ld, cat.c0)
        return self._conv_bn(p + ".root", root_in, p + ".root.conv", p + ".root.bn", 1, 1, 0, True, out=out)

    def _tree2(self, p, x, cin, cout):
        """Tree(levels=2, stride 2, level_root=True) dla.py:271-284.  The outer
        project(bottom) is dead compute in the reference and skipped."""
        oh, ow = x.H // 2, x.W // 2
        cat = self.alloc(x.N, oh, ow, 2 * cout + cin + cout)       # [b2 | b1 | bottom | x1]
        bottom = self.maxpool(p + ".downsample", x, out=cat.sub(2 * cout, cin))
        x1 = self._tree1(p + ".tree1", x, cin, cout, 2, False, bottom=bottom, out=cat.sub(2 * cout + cin, cout))
        return self._tree1(p + ".tree2", x1, cout, cout, 1, False, cat=cat, bottom=x1)

    def _build_base(self, x4):
        b = self._conv_bn("base_layer", x4, "base.base_layer.0", "base.base_layer.1", 7, 1, 3, True, cin_pad=4)
        y0 = self._conv_bn("level0", b, "base.level0.0", "base.level0.1", 3, 1, 1, True)
        y1 = self._conv_bn("level1", y0, "base.level1.0", "base.level1.1", 3, 2, 1, True)
        y2 = self._tree1("base.level2", y1, 32, 64, 2, False)
        y3 = self._tree2("base.level3", y2, 64, 128)
        y4 = self._tree2("base.level4", y3, 128, 256)
        y5 = self._tree1("base.level5", y4, 256, 512, 2, True)
        self.base = [y0, y1, y2, y3, y4, y5]

    def _deform(self, p, x, om=None):
        """DeformConv dla.py:646-665: DCN (offset conv + modulated gather GEMM) -> BN -> ReLU.
        om: an already computed offset / mask-logit map [N, H, W, >= 28] (channel 2k = dy_k, 2k + 1 = dx_k, 18 + k = logit_k) -- the
        conv_offset_mask launch is then skipped (tests feed crafted offsets this way)."""
        sd = self.sd
        cin = x.C
        cout = sd[p + ".conv.weight"].shape[0]
        key = p + ".conv"
        if key not in self._wcache:
            wo, Ko = pack_conv_weight(sd[p + ".conv.conv_offset_mask.weight"])
            wm, Km = pack_dcn_weight(sd[p + ".conv.weight"])
            alpha, beta = _bn_fold(sd, p + ".actf.0")
            shift = sd[p + ".conv.bias"].float() * alpha + beta
            bo = torch.zeros(32); bo[:27] = sd[p + ".conv.conv_offset_mask.bias"].float()
            self._wcache[key] = (self.dev(wo), Ko, self.dev(bo), self.dev(wm), Km, self.dev(alpha), self.dev(shift))
        wo, Ko, bo, wm, Km, alpha, shift = self._wcache[key]
        # offset/mask conv as a 32-column problem (27 channels + 5 zero columns: zero weight rows, zero bias) so that it can run
        # on the pre-split halo kernel, which writes 8 channels per thread; the DCN reads channels 0..26 (ldom = 32)
        patch = dcn_patch_choice(x.N, x.H, x.W, cin, cout)
        if om is None and patch and OFFSET_FP32 and x.H * x.W >= OFFSET_FP32_MIN_HW and x.ld % 4 == 0:
            # the offset / mask conv on the fp32-patch form (csrc/dcn.hip, DeftGemmDesc.p3_kernel = 3): it reads the SAME fp32 map the
            # deformable gather reads, so no producer has to write a bf16-piece copy of a DCN's input
            om = self.alloc(x.N, x.H, x.W, 32, ld=32)
            okey = p + ".conv.offset_patch"
            if okey not in self._wcache:
                wod, _ = pack_dcn_weight(sd[p + ".conv.conv_offset_mask.weight"])
                self._wcache[okey] = self.dev(wod)
            wod = self._wcache[okey]
            do = GemmDesc()
            do.x = x.addr; do.w = wod.data_ptr(); do.w3 = self.weights_p3(wod, "dcn").data_ptr()
            do.scale = None; do.shift = bo.data_ptr(); do.res = None; do.y = om.addr
            do.N, do.H, do.W, do.Cin, do.ldx = x.N, x.H, x.W, cin, x.ld
            do.OH, do.OW, do.Cout, do.ldy, do.ldr = x.H, x.W, 32, om.ld, 0
            do.KH, do.KW, do.stride, do.pad = 3, 3, 1, 1
            do.Ktot, do.Kpad = 9 * cin, wod.shape[1]
            do.cin_log2 = int(math.log2(cin)); do.korder = 1
            do.M = x.N * x.H * x.W
            do.relu = 0; do.tile = 0; do.p3_kernel = 3
            do.flop_k = 9 * cin; do.flop_n = 27
            self.gemm("deft_conv2d_nhwc", p + ".offset", do, 2.0 * do.M * 27 * 9 * cin, reads=[x], writes=[om])
        elif om is None:
            om = self.alloc(x.N, x.H, x.W, 32, ld=32)
            self.conv(p + ".offset", x, wo, Ko, 3, 3, 1, 1, 32, None, bo, False, out=om, true_cout=27)
        out = self.alloc(x.N, x.H, x.W, cout)
        d = GemmDesc()
        d.x = x.addr; d.x2 = om.addr; d.w = wm.data_ptr()
        d.scale = alpha.data_ptr(); d.shift = shift.data_ptr(); d.res = None; d.y = out.addr
        d.N, d.H, d.W, d.Cin, d.ldx = x.N, x.H, x.W, cin, x.ld
        d.OH, d.OW, d.Cout, d.ldy, d.ldr = x.H, x.W, cout, out.ld, 0
        d.KH, d.KW, d.stride, d.pad = 3, 3, 1, 1
        d.Ktot, d.Kpad = Km, wm.shape[1]
        d.cin_log2 = int(math.log2(cin))
        d.M = x.N * x.H * x.W
        d.relu = 1; d.Q = 0; d.ldom = om.ld; d.tile = 0
        if patch and out.ld % 4 == 0:
            d.p3_kernel = 2
            d.w3 = self.weights_p3(wm, "dcn").data_ptr()
        if cout % 32 == 0 and out.ld % 4 == 0 and _os.environ.get("DEFT_DCN_Y3", "1") != "0":
            h = self.p3_output(out, d)               # pruned by finalize_p3() when no pre-split conv reads it
            if h is not None:
                d.y3, d.ldy3 = h.addr, h.ld
        self.gemm("deft_dcn_v2_nhwc", p + ".dcn", d, 2.0 * d.M * cout * 9 * cin, reads=[x, om], writes=[out])
        return out

    def _ida_up(self, layers, p, startp, endp):
        """IDAUp.forward dla.py:693-699."""
        for i in range(startp + 1, endp):
            k = i - startp
            wkey = p + ".up_%d.weight" % k
            if wkey not in self._wcache:
                w = self.sd[wkey].float()
                self._wcache[wkey] = (self.dev(w.reshape(w.shape[0], -1).t()), w.shape[2] // 2)   # [taps][C]
            wup, f = self._wcache[wkey]
            t = self._deform(p + ".proj_%d" % k, layers[i])
            u = self.upsample_add(p + ".up_%d" % k, t, wup, layers[i - 1], f)
            layers[i] = self._deform(p + ".node_%d" % k, u)

    def _build_neck(self):
        layers = list(self.base)
        out = [layers[-1]]                                   # DLAUp.forward dla.py:728-735
        for i in range(len(layers) - 2 - 1):
            self._ida_up(layers, "dla_up.ida_%d" % i, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        y = [out[0], out[1], out[2]]                         # img2feats dla.py:795-799 (clones are views here)
        self._ida_up(y, "ida_up", 0, 3)
        self.fmaps = list(self.base) + out + y
        self.feat = y[-1]

    # ---- heads + decode ------------------------------------------------------
    def _build_heads(self, dense_heads):
        sd, N, K = self.sd, self.N, self.K
        h, w = self.feat.H, self.feat.W
        self.out_h, self.out_w = h, w
        self.dense = {}
        names = ["hm"] + ([k for k in self.heads if k != "hm"] if dense_heads else [])
        for hd in names:
            c = self.heads[hd]
            w0, K0 = pack_conv_weight(sd[hd + ".0.weight"])
            w1, K1 = pack_conv_weight(sd[hd + ".2.weight"])
            b1 = self.dev(sd[hd + ".2.bias"].float())
            fold = (self.dev(sd[hd + ".2.weight"].float().reshape(c, 256)), b1, c) if c <= 16 else None
            hid = self.conv(hd + ".0", self.feat, self.dev(w0), K0, 3, 3, 1, 1, 256, None, self.dev(sd[hd + ".0.bias"].float()), True, fold=fold)
            # (on a pre-split kernel the 1x1 conv was folded into the 3x3 conv's epilogue: `hid` is then already its output)
            self.dense[hd] = hid if hid.C == c and fold is not None and hid.C != 256 else \
                self.conv(hd + ".2", hid, self.dev(w1), K1, 1, 1, 1, 0, c, None, b1, False)
        hm = self.dense["hm"]
        chm = self.heads["hm"]
        lib = self.lib
        cap = h * w * chm
        self.cand_s = torch.zeros(N * cap, dtype=torch.float32, device=self.device)
        self.cand_i = torch.zeros(N * cap, dtype=torch.int32, device=self.device)
        self.cand_n = torch.zeros(N, dtype=torch.int32, device=self.device)
        self.scores = torch.zeros(N, K, dtype=torch.float32, device=self.device)
        self.inds = torch.zeros(N, K, dtype=torch.int32, device=self.device)
        self.clses = torch.zeros(N, K, dtype=torch.int32, device=self.device)
        self.add("zero", "cand_count", lambda: self.cand_n.zero_())
        a = (C.c_void_p(hm.addr), N, h, w, chm, hm.ld, 1, ptr(self.cand_s), ptr(self.cand_i), ptr(self.cand_n), cap)
        self.add("deft_hm_peaks", "hm_peaks", lambda: lib.call("deft_hm_peaks", *a, self._stream()))
        b = (ptr(self.cand_s), ptr(self.cand_i), ptr(self.cand_n), N, cap, K, h * w, ptr(self.scores), ptr(self.inds), ptr(self.clses))
        self.add("deft_topk", "topk", lambda: lib.call("deft_topk", *b, self._stream()))
        # regression heads only at the K peaks
        reg = [k for k in self.heads if k != "hm"]
        self.reg_heads = reg
        self.reg_off = {}
        off = 0
        for k in reg:
            self.reg_off[k] = off
            off += self.heads[k]
        Ctot = off
        self.Ctot = Ctot
        Cf = self.feat.C
        nh = len(reg)
        w0 = torch.cat([sd[k + ".0.weight"].float() for k in reg])                       # [nh*256, Cf, 3, 3]
        w0p, K0 = pack_conv_weight(w0)
        b0 = torch.cat([sd[k + ".0.bias"].float() for k in reg])
        w2 = torch.cat([sd[k + ".2.weight"].float().reshape(self.heads[k], 256) for k in reg])
        b2 = torch.cat([sd[k + ".2.bias"].float() for k in reg])
        head_of = torch.tensor([i for i, k in enumerate(reg) for _ in range(self.heads[k])], dtype=torch.int32)
        w0p, b0, w2, b2, head_of = map(self.dev, (w0p, b0, w2, b2, head_of))
        self.head_vals = torch.zeros(N, K, Ctot, dtype=torch.float32, device=self.device)
        self.cts = torch.zeros(N, K, 2, dtype=torch.float32, device=self.device)
        self.bboxes = torch.zeros(N, K, 4, dtype=torch.float32, device=self.device)
        self.centers = torch.zeros(N, K, 2, dtype=torch.float32, device=self.device)
        # regression heads at the K peaks: peak rows -> sparse-row conv GEMM (all heads' 3x3 layers as one
        # [N*K] x [nh*256] x [9*Cf] problem on the matrix cores) -> per-head 1x1
        self.peak_rows = torch.zeros(N * K * 2, dtype=torch.int32, device=self.device)
        self.peak_hid = torch.zeros(N * K, nh * 256, dtype=torch.float32, device=self.device)
        r_ = (ptr(self.inds), N, K, h, w, ptr(self.peak_rows))
        self.add("deft_peak_rows", "peak_rows", lambda: lib.call("deft_peak_rows", *r_, self._stream()))
        d = GemmDesc()
        d.x = self.feat.addr; d.x2 = None; d.w = w0p.data_ptr(); d.scale = None; d.shift = b0.data_ptr(); d.res = None
        d.y = self.peak_hid.data_ptr()
        d.N, d.H, d.W, d.Cin, d.ldx = N, h, w, Cf, self.feat.ld
        d.OH, d.OW, d.Cout, d.ldy, d.ldr = 1, 1, nh * 256, nh * 256, 0
        d.KH, d.KW, d.stride, d.pad = 3, 3, 1, 1
        d.Ktot, d.Kpad, d.cin_log2, d.M = K0, w0p.shape[1], int(math.log2(Cf)), N * K
        d.relu = 1; d.Q = 0; d.ldom = 0; d.tile = (64 << 16) | 64
        d.korder = conv_korder(w0.shape)
        d.rowmap = self.peak_rows.data_ptr()
        self.gemm("deft_conv2d_nhwc", "heads_at_peaks.0", d, 2.0 * N * K * nh * 256 * 9 * Cf)
        f_ = (ptr(self.peak_hid), nh * 256, N * K, ptr(w2), ptr(b2), ptr(head_of), Ctot, ptr(self.head_vals))
        self.add("deft_heads_finish", "heads_at_peaks.2", lambda: lib.call("deft_heads_finish", *f_, self._stream()),
                 2.0 * N * K * Ctot * 256)
        d_ = (ptr(self.inds), ptr(self.head_vals), N, K, w, h, Ctot, self.reg_off.get("reg", -1), self.reg_off.get("wh", -1),
              self.reg_off.get("ltrb_amodal", -1), ptr(self.cts), ptr(self.bboxes), ptr(self.centers))
        self.add("deft_decode_boxes", "decode_boxes", lambda: lib.call("deft_decode_boxes", *d_, self._stream()))

    def _read_image_planes(self, on):
        """When the 7x7 image layer runs on deft_conv_direct, its patch loader reads the [N, 3, H, W] planes itself (DEFT_TILE_PLANAR) and the
        plan's first launch -- the layout pass to 4-channel NHWC -- is dropped (slot 0 stays, as a no-launch placeholder, so that the op
        numbering is the same for every input form); off = the NHWC buffer again (the uint8 path writes it)."""
        if len(self.ops) < 2 or self.ops[1][:2] != ("deft_conv_direct", "base_layer"):
            return
        d = self._base_desc
        if on:
            d.x, d.tile = self.image.data_ptr(), d.tile | PLANAR
            self.ops[0] = ("image_in_place", "image", lambda: 0, 0.0)
            self.io[0] = ([], [])
            self.io[1] = ([_region(self.image)], self.io[1][1])
        else:
            d.x, d.tile = self._x4.addr, d.tile & ~PLANAR
            self.ops[0], self.io[0] = self._image_op
            self.io[1] = ([_region(self._x4)], self.io[1][1])

    @property
    def input_kind(self):
        """'fp32' (frames as [N, 3, H, W], detector.py:150) or 'u8' (camera frames, after use_u8_input)."""
        return "u8" if self.ops[0][0] == "deft_preprocess_u8" else "fp32"

    # ---- public --------------------------------------------------------------
    def use_u8_input(self, sh, sw, minv=None):
        """Switch the plan's first launch from `deft_nchw_to_nhwc` (fp32 NCHW frames, detector.py:150) to `deft_preprocess_u8`:
        uint8 HWC frames [N, sh, sw, 3] are warped (Detector.pre_process' affine, fix_res mode), normalised and written straight
        into the network's input buffer.  minv: dst -> src matrices [N,6] float64 (default: the reference's for sh x sw frames)."""
        from . import preprocess as PR
        assert self.ops[0][0] in ("deft_nchw_to_nhwc", "image_in_place", "deft_preprocess_u8")
        self._read_image_planes(False)
        if minv is None:
            M, _, _ = PR.input_affine(sh, sw, self.H, self.W)
            minv = np.tile(PR.invert_affine(M)[None], (self.N, 1))
        self.image_u8 = torch.zeros(self.N, sh, sw, 3, dtype=torch.uint8, device=self.device)
        self._minv = self.dev(torch.from_numpy(np.ascontiguousarray(minv, np.float64)))
        self._lut = self.dev(torch.from_numpy(PR.normalisation_table()))
        lib, x4 = self.lib, self._x4
        a = (ptr(self.image_u8), self.N, sh, sw, ptr(self._minv), ptr(self._lut), C.c_void_p(x4.addr), self.H, self.W, x4.ld)
        self.ops[0] = ("deft_preprocess_u8", "image", lambda: lib.call("deft_preprocess_u8", *a, self._stream()), 0.0)
        self.io[0] = ([_region(self.image_u8)], [_region(x4)])

    def forward_u8(self, frames_u8):
        """frames_u8 [N, sh, sw, 3] uint8 on the device (after use_u8_input)."""
        self.image_u8.copy_(frames_u8, non_blocking=True)
        self.run()
        return self

    def forward(self, images):
        """images [N,3,H,W] fp32 (detector.py:150).  Runs backbone + neck + hm head + decode.
        Results stay on the device: self.scores/inds/clses/cts/bboxes/head_vals, self.fmaps."""
        self.image.copy_(images, non_blocking=True)
        self.run()
        return self

    def dets(self):
        """generic_decode's dict (decode.py:102-196) for the heads present; device tensors."""
        r = {"scores": self.scores, "clses": self.clses.float(), "inds": self.inds.long(),
             "xs": self.cts[..., 0], "ys": self.cts[..., 1], "cts": self.cts, "bboxes": self.bboxes}
        if "ltrb_amodal" in self.reg_off:
            r["bboxes_amodal"] = self.bboxes
        for k in ("tracking", "dep", "rot", "dim", "amodel_offset"):
            if k in self.reg_off:
                o = self.reg_off[k]
                r[k] = self.head_vals[..., o:o + self.heads[k]]
        return r

    def flops(self):
        return sum(f for _, _, _, f in self.ops)


class AfePlan(_Plan):
    """Embedding extraction at detection centres + pairwise affinity (AFE.py:88-213)."""
    EGROUP_CACHE = 8

    def __init__(self, sd, max_object=100, device="cuda", lib=None, align_corners=False):
        """align_corners: how grid_sample (AFE.py:178, no flag passed) maps [-1,1] onto pixels -- False = torch >= 1.3
        (the reference as it runs today; the oracle), True = the torch 1.2 behaviour of the authors' environment."""
        super().__init__(device, lib)
        self.max_object = max_object
        self.align_corners = bool(align_corners)
        nsel = 13
        self.sel, self.sel_t = [], []
        off = 0
        for k in range(nsel):
            w = sd["AFE.selector.%d.weight" % k].float()          # [Co,C,3,3]
            Co, Cc = w.shape[0], w.shape[1]
            wp, K = pack_conv_weight(w)                            # implicit-GEMM layout [CoPad][Kpad]
            self.sel.append((self.dev(wp), K, self.dev(sd["AFE.selector.%d.bias" % k].float()), Co, Cc, off))
            self.sel_t.append(self.dev(w.permute(2, 3, 1, 0).reshape(9 * Cc, Co)))   # [(r,s,c)][Co] for deft_embed_map
            off += Co
        self.D = D = off
        # ---- pair MLP, separable first layer with both BatchNorms folded in fp64 ----
        a0, b0 = [t.double() for t in _bn_fold(sd, "AFE.stacker2_bn")]
        W1 = sd["AFE.final_net.0.weight"].double().reshape(512, 2 * D)
        bias1 = sd["AFE.final_net.0.bias"].double()
        a1, b1 = [t.double() for t in _bn_fold(sd, "AFE.final_net.1")]
        W1a, W1b = W1[:, :D], W1[:, D:]
        Ua = (W1a * a0.view(1, D)) * a1.view(512, 1)                              # U' = x_hist @ Ua^T
        Vb = (W1b * a0.view(1, D)) * a1.view(512, 1)                              # V' = x_cur @ Vb^T + cb
        cb = a1 * (bias1 + W1a @ b0 + W1b @ b0) + b1
        Kd = _rup(D, 32)

        def pack(Wm):
            out = torch.zeros(_rup(Wm.shape[0], 128), Kd, dtype=torch.float32)
            out[:Wm.shape[0], :D] = Wm.float()
            return out
        self.Ua, self.Vb, self.cb, self.Kd = self.dev(pack(Ua)), self.dev(pack(Vb)), self.dev(cb.float()), Kd
        self.layers = []
        for i, bn in ((3, True), (6, True), (9, False)):
            w = sd["AFE.final_net.%d.weight" % i].float()
            wp, K = pack_conv_weight(w)
            bias = sd["AFE.final_net.%d.bias" % i].float()
            if bn:
                al, be = _bn_fold(sd, "AFE.final_net.%d" % (i + 1))
                scale, shift = al, bias * al + be
            else:
                scale, shift = None, bias
            self.layers.append((self.dev(wp), K, w.shape[0], self.dev(scale) if scale is not None else None, self.dev(shift)))
        self.w5 = self.dev(sd["AFE.final_net.11.weight"].float().reshape(-1))
        self.b5 = float(sd["AFE.final_net.11.bias"].float().item())
        self._pair_mlp = None
        if PAIR_MLP and PREC == 1 and "deft_pair_mlp" in self.lib._fn:
            # the three matrix layers of the pair MLP as ONE weight image (pair_mlp_image) + per-channel scale / shift (BatchNorm and bias
            # folded; two-piece builds: rows scaled into fp16 range, the inverse folded into the scale -- weight_row_shift, as _Plan.prescale)
            Ws, scs, shs = [], [], []
            for i, bn in ((3, True), (6, True), (9, False)):
                w = sd["AFE.final_net.%d.weight" % i].float().reshape(sd["AFE.final_net.%d.weight" % i].shape[0], -1)
                bias = sd["AFE.final_net.%d.bias" % i].float()
                if bn:
                    al, be = _bn_fold(sd, "AFE.final_net.%d" % (i + 1))
                    sc, sh = al.float(), (bias * al + be).float()
                else:
                    sc, sh = torch.ones_like(bias), bias
                if self.np == 2:
                    k = weight_row_shift(w)
                    w, sc = torch.ldexp(w, k.view(-1, 1)), torch.ldexp(sc, -k)
                Ws.append(w); scs.append(sc.contiguous()); shs.append(sh.contiguous())
            assert [tuple(w.shape) for w in Ws] == [(256, 512), (128, 256), (64, 128)], "deft_pair_mlp is the 512-256-128-64-1 net of AFE.py:331-347"
            img = pair_mlp_image(Ws[0], Ws[1], Ws[2], self.np)
            assert img.numel() * 2 == self.lib._fn["deft_pair_mlp_image_bytes"]()
            self._pair_mlp = {"img": self.dev(img), "s": [self.dev(x) for x in scs], "t": [self.dev(x) for x in shs]}

    def _pair_mlp_launch(self, U, V, M, Q, out, batched=None):
        """deft_pair_mlp: U' / V' [rows, 512] -> relu'd logits at out[(m / Q) * (Q + 1) + m % Q]."""
        pm = self._pair_mlp
        d = hiplib.PairMlpDesc()
        d.U, d.V, d.wimg = U.data_ptr(), V.data_ptr(), pm["img"].data_ptr()
        d.s2, d.t2, d.s3, d.t3, d.s4, d.t4 = (pm["s"][0].data_ptr(), pm["t"][0].data_ptr(), pm["s"][1].data_ptr(), pm["t"][1].data_ptr(),
                                              pm["s"][2].data_ptr(), pm["t"][2].data_ptr())
        d.w5, d.out, d.b5 = self.w5.data_ptr(), out if isinstance(out, int) else out.data_ptr(), self.b5
        d.ldu, d.M, d.Q = 512, M, Q
        if batched is not None:
            d.Tper, d.u0, d.du, d.v0, d.dv = batched
        self.lib.call("deft_pair_mlp", C.byref(d), self._stream())

    def _embed_group(self, fmaps, Nf, ndet):
        """Per (feature-map buffers, Nf, ndet): the 13 sparse-row conv descriptors (host + device
        copies) and the scratch of the fused embedding head."""
        key = (tuple(fm.addr for fm in fmaps), Nf, ndet)
        if not hasattr(self, "_egroups"):
            self._egroups = collections.OrderedDict()
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if key in self._egroups:
            self._egroups.move_to_end(key)
            if capturing:
                self._egroups[key]["pinned"] = True          # its buffers are baked into a hipGraph now: never evicted
            return self._egroups[key]
        assert not capturing, "AfePlan.extract: warm the (maps, frames, ndet) shape up before capturing it (allocates)"
        # drop-in tracker path: ndet changes from frame to frame -- keep the few most recent shapes (each entry owns rowmaps, scratch and
        # split-K workspaces: MBs); entries a captured graph replays into stay
        victims = [k for k, g_ in self._egroups.items() if not g_.get("pinned")]
        while len(victims) >= self.EGROUP_CACHE:
            del self._egroups[victims.pop(0)]
        dev = self.device
        nm = len(self.sel)
        M = Nf * ndet * 4
        ldts = [_rup(Co, 4) for (_, _, _, Co, _, _) in self.sel]
        toffs = [0]
        for ld in ldts:
            toffs.append(toffs[-1] + M * ld)
        g = {"rowmap": torch.zeros(nm * M * 2, dtype=torch.int32, device=dev),
             "bw": torch.zeros(nm * Nf * ndet * 4, dtype=torch.float32, device=dev),
             "tmp": torch.zeros(toffs[-1], dtype=torch.float32, device=dev),
             "map_hw": torch.tensor([[fm.H, fm.W] for fm in fmaps], dtype=torch.int32).to(dev),
             "map_out": torch.tensor([[toffs[k], ldts[k], self.sel[k][3], self.sel[k][5]] for k in range(nm)], dtype=torch.int32).to(dev)}
        descs = (GemmDesc * nm)()
        for k, (fm, (wp, K, bias, Co, Cc, off)) in enumerate(zip(fmaps, self.sel)):
            assert fm.C == Cc and fm.N == Nf
            d = descs[k]
            d.x = fm.addr; d.x2 = None; d.w = wp.data_ptr(); d.scale = None; d.shift = bias.data_ptr(); d.res = None
            d.y = g["tmp"].data_ptr() + 4 * toffs[k]
            d.N, d.H, d.W, d.Cin, d.ldx = fm.N, fm.H, fm.W, fm.C, fm.ld
            d.OH, d.OW, d.Cout, d.ldy, d.ldr = 1, 1, Co, ldts[k], 0
            d.KH, d.KW, d.stride, d.pad = 3, 3, 1, 1
            d.Ktot, d.Kpad, d.cin_log2, d.M = K, wp.shape[1], int(math.log2(fm.C)), M
            d.relu = 1; d.Q = 0; d.ldom = 0; d.tile = 0
            d.korder = conv_korder((Co, fm.C, 3, 3))
            d.rowmap = g["rowmap"].data_ptr() + 4 * (k * M * 2)
        # few rows in total (one frame): the long-K groups (512-channel maps: 144 chunks) would be the tail of the
        # launch -> split their K over workgroups so that every workgroup contracts ~9-16 chunks
        tiles = [-(-M // 32) * -(-descs[k].Cout // 32) for k in range(nm)]
        if SPLITK and sum(tiles) < 512:
            S = [min(16, 1 << max(0, (descs[k].Kpad // 32 // 9).bit_length() - 1)) for k in range(nm)]
            woff, coff = [0], [0]
            for k in range(nm):
                woff.append(woff[-1] + (tiles[k] * S[k] * 1024 if S[k] > 1 else 0))
                coff.append(coff[-1] + (tiles[k] if S[k] > 1 else 0))
            g["ws"] = torch.empty(max(1, woff[-1]), dtype=torch.float32, device=dev)
            g["ws_cnt"] = torch.zeros(max(1, coff[-1]), dtype=torch.int32, device=dev)
            for k in range(nm):
                if S[k] > 1:
                    descs[k].tile, descs[k].splitk = (32 << 16) | 32, S[k]
                    descs[k].ws, descs[k].ws_cnt = g["ws"].data_ptr() + 4 * woff[k], g["ws_cnt"].data_ptr() + 4 * coff[k]
        g["descs"] = descs
        g["descs_dev"] = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
        self._egroups[key] = g
        return g

    def extract(self, fmaps, centers, out=None):
        """fmaps: 13 Views (DlaSegPlan.fmaps); centers [Nf, ndet, 2] (x,y in [-1,1], as
        convert_detection image.py:391-412 produces) -> embeddings [Nf, ndet, D] (written
        into `out` when given: a contiguous [Nf, ndet, D] device tensor).
        Three launches: corner rows + blend weights, ONE grouped sparse-row conv GEMM over the 13
        selector convs (ReLU epilogue), bilinear blend into the embedding columns."""
        Nf, ndet = centers.shape[0], centers.shape[1]
        centers = centers.to(self.device, torch.float32).contiguous()
        if out is None:
            out = torch.empty(Nf, ndet, self.D, dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and tuple(out.shape) == (Nf, ndet, self.D)
        if Nf * ndet == 0:
            return out
        g = self._embed_group(fmaps, Nf, ndet)
        s = self._stream()
        nm = len(self.sel)
        self.lib.call("deft_embed_rows", ptr(centers), Nf, ndet, ptr(g["map_hw"]), nm, ptr(g["rowmap"]), ptr(g["bw"]), int(self.align_corners), s)
        self.lib.call("deft_conv2d_group", g["descs"], C.c_void_p(g["descs_dev"].data_ptr()), nm, s)
        self.lib.call("deft_embed_blend", ptr(g["tmp"]), ptr(g["bw"]), ptr(g["map_out"]), nm, Nf, ndet, ptr(out), self.D, s)
        return out

    def extract_per_map(self, fmaps, centers, out=None):
        """Same result through deft_embed_map (one launch per map); kept for the single-map ABI entry."""
        Nf, ndet = centers.shape[0], centers.shape[1]
        centers = centers.to(self.device, torch.float32).contiguous()
        if out is None:
            out = torch.empty(Nf, ndet, self.D, dtype=torch.float32, device=self.device)
        s = self._stream()
        for fm, (wp, K, b, Co, Cc, off), wt in zip(fmaps, self.sel, self.sel_t):
            assert fm.C == Cc and fm.N == Nf
            self.lib.call("deft_embed_map", C.c_void_p(fm.addr), Nf, fm.H, fm.W, fm.C, fm.ld, ptr(wt), ptr(b), Co,
                          ptr(centers), ndet, ptr(out), self.D, off, int(self.align_corners), s)
        return out

    def _lin(self, x, M, Cin, ldx, wp, Kpad, Cout, scale, shift, relu, y, ldy):
        d = GemmDesc()
        d.x = x.data_ptr(); d.x2 = None; d.w = wp.data_ptr()
        d.scale = scale.data_ptr() if scale is not None else None
        d.shift = shift.data_ptr() if shift is not None else None
        d.res = None; d.y = y.data_ptr()
        d.N, d.H, d.W, d.Cin, d.ldx = M, 1, 1, Cin, ldx
        d.OH, d.OW, d.Cout, d.ldy, d.ldr = 1, 1, Cout, ldy, 0
        d.KH, d.KW, d.stride, d.pad = 1, 1, 1, 0
        d.Ktot, d.Kpad, d.cin_log2, d.M = Cin, Kpad, 0, M
        d.relu = int(relu); d.Q = 0; d.ldom = 0; d.tile = 0; d.prec = PREC
        if BDMA and PREC == 1 and Cout >= 128:
            d.w3 = self.weights_p3(wp).data_ptr()
        self.prescale(d)
        self.lib.call("deft_conv2d_nhwc", C.byref(d), self._stream())

    def _staged_ints(self, values):
        """A small int32 host list as a device tensor without a blocking copy: written into one of a few pinned staging rows, copied
        non_blocking on the plan's stream.  Eight rows in rotation, each guarded by an event recorded behind its copy."""
        n = len(values)
        if self.device.type != "cuda":
            return torch.tensor(values, dtype=torch.int32, device=self.device)
        st = getattr(self, "_stage", None)
        if st is None or st[0].shape[1] < n:
            st = self._stage = (torch.empty(8, max(64, 2 * n), dtype=torch.int32).pin_memory(), [0], [None] * 8)
        buf, turn, evs = st
        k = turn[0] % 8
        turn[0] += 1
        if evs[k] is not None:
            evs[k].synchronize()                   # the copy that read this row eight calls ago (warm-up loops call this back to back)
        row = buf[k]
        row[:n] = torch.as_tensor(values, dtype=torch.int32)
        out = row[:n].to(self.device, non_blocking=True)
        evs[k] = torch.cuda.Event()
        evs[k].record(torch.cuda.current_stream(self.device))
        return out

    def _work(self, name, numel):
        """Grow-only fp32 device workspace `name` of at least `numel` elements (1.5x headroom on growth)."""
        ws = self.__dict__.setdefault("_workspaces", {})
        t = ws.get(name)
        if t is None or t.numel() < numel:
            t = ws[name] = torch.empty(max(numel + numel // 2, 1 << 16), dtype=torch.float32, device=self.device)
        return t[:numel]

    def affinity(self, hist, cur):
        """hist: list of [P_f, D] embeddings of F stored frames; cur [Q, D].
        Returns the F matrices [P_f, Q+1] of forward_stacker_features (AFE.py:110-160,
        fill_up_column=False) as one device tensor [sum P_f, Q+1] plus the row offsets."""
        dev, D, Kd = self.device, self.D, self.Kd
        Q = cur.shape[0]
        starts = [0]
        for hx in hist:
            assert hx.shape[0] <= self.max_object
            starts.append(starts[-1] + hx.shape[0])
        T = starts[-1]
        assert 0 < Q <= self.max_object and T > 0
        # grow-only workspaces (h2 alone is T*Q rows: up to 0.5 GB for 50 stored frames of 100 objects).  Fresh torch.empty calls of a size
        # that changes every frame (the lazily-computed block set does) keep missing the caching allocator: each miss is a hipMalloc of
        # hundreds of MB, and each garbage-collection of the cache a hipFree -- which waits for EVERYTHING the device is running, e.g. a
        # lookahead pass on another stream (profiles/r4_hw_queue_stall.md).  Safe to reuse: launches of consecutive calls are
        # ordered on the caller's stream(s), and only `out` outlives the call.
        M = T * Q
        (w2, K2, c2, s2, t2), (w3, K3, c3, s3, t3), (w4, K4, c4, s4, t4) = self.layers
        xh = self._work("xh", T * Kd).view(T, Kd)
        xc = self._work("xc", Q * Kd).view(Q, Kd)
        hd = [hx.to(dev, torch.float32) for hx in hist]
        if Kd == D:
            torch.cat(hd, 0, out=xh)                       # one launch, straight into the workspace
        else:
            xh[:, D:].zero_(); xc[:, D:].zero_()
            xh[:, :D] = torch.cat(hd, 0)
        xc[:, :D].copy_(cur, non_blocking=True)
        U = self._work("U", T * 512).view(T, 512)
        V = self._work("V", Q * 512).view(Q, 512)
        self._lin(xh, T, Kd, Kd, self.Ua, Kd, 512, None, None, False, U, 512)
        self._lin(xc, Q, Kd, Kd, self.Vb, Kd, 512, None, self.cb, False, V, 512)
        out = torch.empty(T, Q + 1, dtype=torch.float32, device=dev)
        rs = self._staged_ints(starts)                # (pinned staging: a pageable torch.tensor(..., device=) is a blocking copy behind the whole chain)
        if self._pair_mlp is not None:
            self._pair_mlp_launch(U, V, M, Q, out)    # layers 2-5 in one launch: the logits land in `out`
            self.lib.call("deft_affinity_finish", None, c4, c4, None, C.c_float(0.0), ptr(rs), len(hist), T, Q, self.max_object, ptr(out), self._stream())
            return out, starts
        h2 = self._work("h2", M * c2).view(M, c2)
        d = GemmDesc()
        d.x = U.data_ptr(); d.x2 = V.data_ptr(); d.w = w2.data_ptr()
        d.scale = s2.data_ptr(); d.shift = t2.data_ptr(); d.res = None; d.y = h2.data_ptr()
        d.N, d.H, d.W, d.Cin, d.ldx = M, 1, 1, 512, 512
        d.OH, d.OW, d.Cout, d.ldy, d.ldr = 1, 1, c2, c2, 0
        d.KH, d.KW, d.stride, d.pad = 1, 1, 1, 0
        d.Ktot, d.Kpad, d.cin_log2, d.M = 512, 512, 0, M
        d.relu = 1; d.Q = Q; d.ldom = 0; d.tile = 0; d.prec = PREC
        if BDMA and PREC == 1:
            d.w3 = self.weights_p3(w2).data_ptr()
        self.prescale(d)
        self.lib.call("deft_pair_layer", C.byref(d), self._stream())
        h3 = self._work("h3", M * c3).view(M, c3)
        self._lin(h2, M, c2, c2, w3, w3.shape[1], c3, s3, t3, True, h3, c3)
        h4 = self._work("h4", M * c4).view(M, c4)
        self._lin(h3, M, c3, c3, w4, w4.shape[1], c4, s4, t4, True, h4, c4)
        self.lib.call("deft_affinity_finish", ptr(h4), c4, c4, ptr(self.w5), C.c_float(self.b5), ptr(rs), len(hist), T, Q,
                      self.max_object, ptr(out), self._stream())
        return out, starts

    def affinity_ring(self, ring, g0, Bc, hist):
        """Batched steady-state form used by the frame pipeline.  ring [R, K, D]: embeddings of
        consecutive frames of one stream (every frame K objects); current frames are ring[g0+c],
        c < Bc, each scored against ring[g0+c-hist : g0+c] (tracker.py:76-90 for `hist` stored
        frames).  U'/V' are computed once per ring frame; one pair-GEMM chain covers up to
        `CH` current frames (the kernels address activations with 32-bit byte offsets: <= 2 GiB
        per tensor), so large batches run as a few chains.  Returns [Bc, hist*K, K+1]."""
        R, K, D = ring.shape
        assert ring.is_contiguous() and D == self.D and D == self.Kd and g0 - hist >= 0 and g0 + Bc <= R
        assert K <= self.max_object
        (w2, K2, c2, s2, t2), (w3, K3, c3, s3, t3), (w4, K4, c4, s4, t4) = self.layers
        fused = self._pair_mlp is not None
        # (the fused launch keeps the per-pair intermediates in registers: no h2 / h3 / h4 buffers, no 2 GiB-per-tensor limit on a chain)
        CH = max(1, min(Bc, ((1 << 31) - 1) // (hist * K * (K + 1)))) if fused else max(1, min(Bc, ((1 << 29) - 1) // (hist * K * K * max(c2, 512))))
        key = (R, K, Bc, hist)
        if not hasattr(self, "_ring"):
            self._ring = {}
        if key not in self._ring:
            dev = self.device
            Mc = CH * hist * K * K
            buf = {"U": torch.empty(R * K, 512, dtype=torch.float32, device=dev),
                   "V": torch.empty(R * K, 512, dtype=torch.float32, device=dev),
                   "out": torch.empty(Bc, hist * K, K + 1, dtype=torch.float32, device=dev),
                   "rs": torch.arange(0, CH * hist + 1, dtype=torch.int32, device=dev) * K}
            if not fused:
                buf.update({"h2": torch.empty(Mc, c2, dtype=torch.float32, device=dev), "h3": torch.empty(Mc, c3, dtype=torch.float32, device=dev),
                            "h4": torch.empty(Mc, c4, dtype=torch.float32, device=dev)})
            self._ring[key] = buf
        b = self._ring[key]
        self._lin(ring, R * K, D, D, self.Ua, self.Kd, 512, None, None, False, b["U"], 512)
        self._lin(ring, R * K, D, D, self.Vb, self.Kd, 512, None, self.cb, False, b["V"], 512)
        for c0 in range(0, Bc, CH):
            nc = min(CH, Bc - c0)
            M = nc * hist * K * K
            if self._pair_mlp is not None:
                o_ptr = b["out"].data_ptr() + 4 * c0 * hist * K * (K + 1)
                self._pair_mlp_launch(b["U"], b["V"], M, K, o_ptr, batched=(hist * K, (g0 + c0 - hist) * K, K, (g0 + c0) * K, K))
                self.lib.call("deft_affinity_finish", None, c4, c4, None, C.c_float(0.0), ptr(b["rs"]), nc * hist, nc * hist * K, K, self.max_object,
                              C.c_void_p(o_ptr), self._stream())
                continue
            d = GemmDesc()
            d.x = b["U"].data_ptr(); d.x2 = b["V"].data_ptr(); d.w = w2.data_ptr()
            d.scale = s2.data_ptr(); d.shift = t2.data_ptr(); d.res = None; d.y = b["h2"].data_ptr()
            d.N, d.H, d.W, d.Cin, d.ldx = M, 1, 1, 512, 512
            d.OH, d.OW, d.Cout, d.ldy, d.ldr = 1, 1, c2, c2, 0
            d.KH, d.KW, d.stride, d.pad = 1, 1, 1, 0
            d.Ktot, d.Kpad, d.cin_log2, d.M = 512, 512, 0, M
            d.relu = 1; d.Q = K; d.ldom = 0; d.tile = 0; d.prec = PREC
            if BDMA and PREC == 1:
                d.w3 = self.weights_p3(w2).data_ptr()
            self.prescale(d)
            d.Tper, d.u0, d.du, d.v0, d.dv = hist * K, (g0 + c0 - hist) * K, K, (g0 + c0) * K, K
            self.lib.call("deft_pair_layer", C.byref(d), self._stream())
            self._lin(b["h2"], M, c2, c2, w3, w3.shape[1], c3, s3, t3, True, b["h3"], c3)
            self._lin(b["h3"], M, c3, c3, w4, w4.shape[1], c4, s4, t4, True, b["h4"], c4)
            self.lib.call("deft_affinity_finish", ptr(b["h4"]), c4, c4, ptr(self.w5), C.c_float(self.b5), ptr(b["rs"]),
                          nc * hist, nc * hist * K, K, self.max_object, C.c_void_p(b["out"].data_ptr() + 4 * c0 * hist * K * (K + 1)), self._stream())
        return b["out"]

    @staticmethod
    def affinity_flops(T, Q, D):
        return 2.0 * (T + Q) * D * 512 + 2.0 * T * Q * (512 * 256 + 256 * 128 + 128 * 64 + 64)


class LstmPlan(_Plan):
    """Batched KalmanFilterLSTM.predict (kalman_filter_lstm.py:65-78)."""

    def __init__(self, lsd, device="cuda", lib=None):
        super().__init__(device, lib)
        self.nin = lsd["lstm.weight_ih_l0"].shape[1]
        self.nout = lsd["out2.weight"].shape[0]
        self.wih_t = self.dev(lsd["lstm.weight_ih_l0"].float().t())
        self.whh_t = self.dev(lsd["lstm.weight_hh_l0"].float().t())
        self.bias = self.dev(lsd["lstm.bias_ih_l0"].float() + lsd["lstm.bias_hh_l0"].float())
        self.w1_t = self.dev(lsd["out1.weight"].float().t())
        self.b1 = self.dev(lsd["out1.bias"].float())
        self.w2_t = self.dev(lsd["out2.weight"].float().t())
        self.b2 = self.dev(lsd["out2.bias"].float())

    def step(self, x, h, c):
        """x [T,nin]; h,c [T,128] updated IN PLACE; returns pred [T, nout//4, 4]."""
        T = x.shape[0]
        x = x.to(self.device, torch.float32).contiguous()
        assert h.is_contiguous() and c.is_contiguous() and h.device.type == self.device.type
        pred = torch.empty(T, self.nout, dtype=torch.float32, device=self.device)
        self.lib.call("deft_lstm_step", ptr(x), ptr(h), ptr(c), T, self.nin, self.nout, ptr(self.wih_t), ptr(self.whh_t),
                      ptr(self.bias), ptr(self.w1_t), ptr(self.b1), ptr(self.w2_t), ptr(self.b2), ptr(pred), self._stream())
        return pred.view(T, -1, 4)

    def motion_step(self, slot, box, frame_id, h, c, last):
        """Feature builder + LSTM step + future boxes for the tracks updated in one frame, ONE launch
        (tracker.py:408-480 / 482-580).  slot int32 [T]; box float64 [T, 4|7]; h, c [S,128] float32 and
        last [S,9] float64 are the persistent per-track rows (updated in place).  Returns
        (feat float32 [T,nin], pred float64 [T, nout//4, 4|7])."""
        T, dim = box.shape
        assert slot.dtype == torch.int32 and box.dtype == torch.float64 and last.dtype == torch.float64
        assert slot.is_contiguous() and box.is_contiguous() and h.is_contiguous() and c.is_contiguous() and last.is_contiguous()
        feat = torch.empty(T, self.nin, dtype=torch.float32, device=self.device)
        pred = torch.empty(T, self.nout // 4, dim, dtype=torch.float64, device=self.device)
        self.lib.call("deft_motion_step", ptr(slot), ptr(box), T, dim, int(frame_id), ptr(h), ptr(c), ptr(last), self.nin, self.nout,
                      ptr(self.wih_t), ptr(self.whh_t), ptr(self.bias), ptr(self.w1_t), ptr(self.b1), ptr(self.w2_t), ptr(self.b2),
                      ptr(feat), ptr(pred), self._stream())
        return feat, pred
