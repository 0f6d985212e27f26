"""TEST INFRASTRUCTURE ONLY -- an independent, scalar, tap-by-tap restatement of the DCNv2 forward pass that
`from dcn_v2 import DCN` resolves to in the reference (dla.py:25-29; constructed dla.py:652-660, called dla.py:663).

PARITY UNPINNED: the arithmetic lives in the third-party extension CharlesShang/DCNv2, which is neither vendored in
/root/reference nor pinned by it (README.md:72-78 only says to build it).  This file restates upstream's PUBLISHED
algorithm loop by loop, in the shape upstream wrote it:

  * `DCN.forward` (dcn_v2.py):  out = conv_offset_mask(x);  o1, o2, mask = chunk(out, 3, dim=1);
    offset = cat(o1, o2);  mask = sigmoid(mask);  dcn_v2_conv(x, offset, mask, weight, bias, stride 1, pad 1, dil 1, groups 1)
  * `modulated_deformable_im2col_gpu_kernel` (dcn_v2_im2col_cuda.cu):  for kernel tap (i, j) of output pixel (h_col, w_col):
    offset_h = offset[2 (i kw + j)], offset_w = offset[2 (i kw + j) + 1], m = mask[i kw + j];
    h_im = h_col * stride - pad + i * dil + offset_h, w_im likewise;
    val = 0;  if (h_im > -1 && w_im > -1 && h_im < height && w_im < width) val = dmcn_im2col_bilinear(...);  col = val * m
  * `dmcn_im2col_bilinear`:  h_low = floor(h), h_high = h_low + 1 (w likewise); lh = h - h_low, hh = 1 - lh; the four corner values,
    each 0 unless its row AND column are inside [0, height-1] x [0, width-1]; val = hh hw v1 + hh lw v2 + lh hw v3 + lh lw v4
  * the contraction: y[n, co, h, w] = bias[co] + sum over (c, i, j) of weight[co, c, i, j] * col[n, c, i, j, h, w]

It shares no code with oracle/deft_oracle.py::dcn_v2_forward (a vectorised torch formulation) nor with the HIP kernels
(csrc/igemm.hip MODE_DCN, csrc/dcn.hip); tests/test_oracle.py checks the three against each other and against the
vectors this file wrote to tests/golden/dcn_v2_*.npz (`python oracle/dcn_scalar.py` regenerates them).

Arithmetic: the sampling in float32 (numpy scalars, the order of operations of the upstream kernel), the contraction in float64.
"""
import math
import os

import numpy as np

F32 = np.float32


def sigmoid32(v):
    return F32(1.0) / (F32(1.0) + np.exp(-F32(v), dtype=F32))


def bilinear(plane, height, width, h, w):
    """dmcn_im2col_bilinear on one [height, width] float32 plane at (h, w) (float32)."""
    h_low = int(math.floor(h)); w_low = int(math.floor(w))
    h_high = h_low + 1; w_high = w_low + 1
    lh = F32(h - F32(h_low)); lw = F32(w - F32(w_low))
    hh = F32(F32(1) - lh); hw = F32(F32(1) - lw)
    v1 = plane[h_low, w_low] if (h_low >= 0 and w_low >= 0) else F32(0)
    v2 = plane[h_low, w_high] if (h_low >= 0 and w_high <= width - 1) else F32(0)
    v3 = plane[h_high, w_low] if (h_high <= height - 1 and w_low >= 0) else F32(0)
    v4 = plane[h_high, w_high] if (h_high <= height - 1 and w_high <= width - 1) else F32(0)
    w1 = F32(hh * hw); w2 = F32(hh * lw); w3 = F32(lh * hw); w4 = F32(lh * lw)
    return F32(F32(F32(F32(w1 * v1) + F32(w2 * v2)) + F32(w3 * v3)) + F32(w4 * v4))


def deform_im2col(x, offset, mask):
    """modulated_deformable_im2col for 3x3 / stride 1 / pad 1 / dilation 1 / one deformable group.
    x [N, C, H, W], offset [N, 18, H, W] (channel 2k = dy of tap k, 2k + 1 = dx), mask [N, 9, H, W] (already sigmoid'ed)
    -> col [N, C, 9, H, W] float32."""
    N, C, H, W = x.shape
    col = np.zeros((N, C, 9, H, W), dtype=F32)
    for n in range(N):
        for h_col in range(H):
            for w_col in range(W):
                for i in range(3):
                    for j in range(3):
                        k = i * 3 + j
                        h_im = F32(F32(h_col - 1 + i) + offset[n, 2 * k, h_col, w_col])
                        w_im = F32(F32(w_col - 1 + j) + offset[n, 2 * k + 1, h_col, w_col])
                        if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
                            continue
                        m = mask[n, k, h_col, w_col]
                        for c in range(C):
                            col[n, c, k, h_col, w_col] = F32(bilinear(x[n, c], H, W, h_im, w_im) * m)
    return col


def conv3x3_scalar(x, w, b):
    """Plain 3x3 / stride 1 / pad 1 convolution, float64 accumulation (the conv_offset_mask layer)."""
    N, C, H, W = x.shape
    Co = w.shape[0]
    xp = np.zeros((N, C, H + 2, W + 2), dtype=np.float64)
    xp[:, :, 1:-1, 1:-1] = x
    y = np.zeros((N, Co, H, W), dtype=np.float64)
    for i in range(3):
        for j in range(3):
            y += np.einsum("oc,nchw->nohw", w[:, :, i, j].astype(np.float64), xp[:, :, i:i + H, j:j + W])
    return (y + b.astype(np.float64).reshape(1, -1, 1, 1)).astype(F32)


def dcn_v2_from_offsets(x, offset, mask_logit, w, b):
    """dcn_v2_conv with explicit offsets and mask LOGITS (sigmoid applied here, as DCN.forward does)."""
    mask = np.vectorize(sigmoid32, otypes=[F32])(mask_logit)
    col = deform_im2col(x.astype(F32), offset.astype(F32), mask)
    N, C, _, H, W = col.shape
    y = np.einsum("ok,nkhw->nohw", w.reshape(w.shape[0], C * 9).astype(np.float64), col.reshape(N, C * 9, H, W).astype(np.float64))
    return y + b.astype(np.float64).reshape(1, -1, 1, 1)


def dcn_v2_module(x, w_off, b_off, w, b):
    """DCN.forward: conv_offset_mask -> chunk -> (offset, sigmoid(mask)) -> dcn_v2_conv.  Returns (y float64, om float32 [N,27,H,W])."""
    om = conv3x3_scalar(x.astype(F32), w_off, b_off)
    o1, o2, ml = om[:, 0:9], om[:, 9:18], om[:, 18:27]
    offset = np.concatenate([o1, o2], axis=1)
    return dcn_v2_from_offsets(x, offset, ml, w, b), om


# ---------------------------------------------------------------------------------------------------------------------
def border_case_offsets(H, W, rng):
    """Offsets [1, 18, H, W] and mask logits [1, 9, H, W] that drive samples through every branch of the sampling rule:
    exactly on integer coordinates, just inside / outside -1 and H (W), whole-pixel jumps far outside the map, corners with one,
    two, three and four taps outside, and ordinary fractional positions."""
    off = rng.uniform(-1.5, 1.5, size=(1, 18, H, W)).astype(F32)
    special = [0.0, 1.0, -1.0, -0.999, -1.001, 0.5, -0.5, 2.0, -2.0, 3.25, -3.25, float(H), float(-H), float(W) + 0.5, 1e-4, -1e-4,
               float(H) - 1.0, float(W) - 1.0, 7.75, -7.75]
    k = 0
    for y in range(H):
        for x in range(W):
            for t in range(18):
                if (y * W + x + t) % 3 == 0:
                    off[0, t, y, x] = F32(special[k % len(special)]); k += 1
    ml = rng.normal(0, 2.0, size=(1, 9, H, W)).astype(F32)
    return off, ml


def make_vectors(out_dir):
    rng = np.random.default_rng(20260926)
    cases = {
        # name: (N, C, Co, H, W, offset scale of the module path)
        "small": (1, 32, 8, 5, 7, 0.5),
        "borders": (1, 32, 16, 9, 12, None),          # crafted offsets
        "wide": (2, 64, 24, 11, 19, 3.0),             # two images, partial 8 x 16 tiles, offsets of several pixels
    }
    for name, (N, C, Co, H, W, sc) in cases.items():
        x = rng.normal(0, 1, size=(N, C, H, W)).astype(F32)
        w = (rng.normal(0, 1, size=(Co, C, 3, 3)) / math.sqrt(9 * C)).astype(F32)
        b = rng.normal(0, 0.1, size=(Co,)).astype(F32)
        if sc is None:
            off, ml = border_case_offsets(H, W, rng)
            y = dcn_v2_from_offsets(x, off, ml, w, b)
            om = np.concatenate([off[:, 0:18], ml], axis=1)          # the 27-channel map the kernels read (2k = dy, 2k+1 = dx, 18+k = logit)
            np.savez_compressed(os.path.join(out_dir, "dcn_v2_%s.npz" % name), x=x, om=om, w=w, b=b, y=y.astype(np.float64))
        else:
            w_off = (rng.normal(0, 1, size=(27, C, 3, 3)) * (0.3 / math.sqrt(9 * C))).astype(F32)
            b_off = (rng.normal(0, 1, size=(27,)) * sc).astype(F32)
            y, om = dcn_v2_module(x, w_off, b_off, w, b)
            # the module's channel order is (o1 | o2 | mask) = exactly the 27-channel map: offset = cat(o1, o2) keeps channels 0..17
            np.savez_compressed(os.path.join(out_dir, "dcn_v2_%s.npz" % name), x=x, w_off=w_off, b_off=b_off, om=om, w=w, b=b,
                                y=y.astype(np.float64))
        print("wrote dcn_v2_%s.npz  y range [%.3f, %.3f]" % (name, float(y.min()), float(y.max())))


if __name__ == "__main__":
    make_vectors(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
