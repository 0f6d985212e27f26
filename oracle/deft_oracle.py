"""CPU ORACLE for the DEFT per-frame hot path.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, in functional PyTorch-CPU fp32, of the reference
algorithm on the hot path named by BASELINE.json (DLA-34/CenterNet forward incl.
DCNv2 -> embedding head -> pairwise affinity -> LSTM motion step -> decode).  It
is the *checker* for the HIP path: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product package (deft_amd/)
never imports it and has no CPU fallback.

Every function cites the reference file:line (relative to /root/reference) it
follows.  Pinning status:

* Everything except DCNv2 is pinned against the reference's own modules
  (imported from /root/reference in the build container) by
  oracle/make_golden.py, which also writes the fixtures in tests/golden/.
* DCNv2 (`dcn_v2_forward`): PARITY UNPINNED.  The reference imports the
  arithmetic from the un-vendored, un-pinned third-party extension
  CharlesShang/DCNv2 (README.md:72-78, src/lib/model/networks/dla.py:25-29);
  its source is not under /root/reference and the reference holds no test or
  golden vector at that boundary.  The function below restates upstream's
  published algorithm (dcn_v2.py `DCN.forward`, `modulated_deformable_im2col`
  and `dmcn_im2col_bilinear` in src/cuda/dcn_v2_im2col_cuda.cu), anchored on the
  reference's call sites dla.py:652-660 (ctor: 3x3, stride 1, pad 1, dil 1,
  deformable_groups 1) and dla.py:663 (forward).  It is cross-checked against an
  independent formulation (F.grid_sample, zeros padding, align_corners=True) in
  tests/test_oracle.py.

Weights are addressed by the reference's state_dict key names (SURVEY.md §5
"Checkpoint / resume"), so a real DEFT checkpoint drives the oracle unchanged.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5

from deft_amd.synth import (HEADS, SELECTOR_IN, FEATURE_STRIDES, selector_out,   # noqa: F401  (weight table only,
                            synth_state_dict, synth_lstm_state_dict)              # no hot-path algorithm lives there)

# --------------------------------------------------------------------------
# DCNv2 (PARITY UNPINNED -- see module docstring)
# --------------------------------------------------------------------------
def dcn_v2_forward(x, w_off, b_off, w, b):
    """Modulated deformable 3x3 conv, stride 1, pad 1, dil 1, 1 deformable group.
    Call sites: dla.py:652-660, 663.  Upstream: DCN.forward -> dcn_v2_conv."""
    return dcn_v2_from_om(x, F.conv2d(x, w_off, b_off, stride=1, padding=1), w, b)          # 27 channels


def dcn_v2_from_om(x, out, w, b):
    """The sampling + contraction of dcn_v2_forward on a given conv_offset_mask output `out` [N, 27, H, W]."""
    N, C, H, W = x.shape
    o1, o2, mask = torch.chunk(out, 3, dim=1)
    offset = torch.cat((o1, o2), dim=1)                           # [N,18,H,W]: 2k=dy, 2k+1=dx
    mask = torch.sigmoid(mask)                                    # [N,9,H,W]
    hh = torch.arange(H, dtype=x.dtype).view(1, 1, H, 1)
    ww = torch.arange(W, dtype=x.dtype).view(1, 1, 1, W)
    ki = torch.arange(3, dtype=x.dtype).repeat_interleave(3).view(1, 9, 1, 1)
    kj = torch.arange(3, dtype=x.dtype).repeat(3).view(1, 9, 1, 1)
    h_im = (hh - 1 + ki) + offset[:, 0::2]                        # [N,9,H,W]
    w_im = (ww - 1 + kj) + offset[:, 1::2]
    inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
    h_low = torch.floor(h_im); w_low = torch.floor(w_im)
    lh = h_im - h_low; lw = w_im - w_low
    hh_ = 1 - lh; hw_ = 1 - lw
    h_low = h_low.long(); w_low = w_low.long()
    h_high = h_low + 1; w_high = w_low + 1
    xf = x.reshape(N, C, H * W)

    def corner(hi, wi, ok):
        ok = ok & inside
        idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).view(N, 1, 9 * H * W).expand(N, C, -1)
        v = torch.gather(xf, 2, idx).view(N, C, 9, H, W)
        return v * ok.view(N, 1, 9, H, W).to(x.dtype)

    v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
    v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
    v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
    v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))
    w1 = (hh_ * hw_).unsqueeze(1); w2 = (hh_ * lw).unsqueeze(1)
    w3 = (lh * hw_).unsqueeze(1); w4 = (lh * lw).unsqueeze(1)
    col = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * mask.unsqueeze(1)   # [N,C,9,H,W]
    col = col.reshape(N, C * 9, H * W)
    y = torch.matmul(w.reshape(w.shape[0], C * 9), col) + b.view(1, -1, 1)
    return y.view(N, -1, H, W)


# --------------------------------------------------------------------------
# DLA-34 + DLAUp/IDAUp + heads
# --------------------------------------------------------------------------
def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)


def _basic_block(x, sd, p, stride, residual=None):
    """dla.py:73-87."""
    if residual is None:
        residual = x
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    return F.relu(out + residual)


def _root(xs, sd, p):
    """dla.py:199-207 (root_residual False for dla34)."""
    return F.relu(_bn(F.conv2d(torch.cat(xs, 1), sd[p + ".conv.weight"]), sd, p + ".bn"))


def _tree(x, sd, p, levels, stride, level_root, children=None):
    """dla.py:271-284.  (The `residual` argument handed to an inner Tree is
    overwritten there -- dla.py:274 -- so it is not threaded through.)"""
    children = [] if children is None else children
    bottom = F.max_pool2d(x, stride, stride) if stride > 1 else x
    if (p + ".project.0.weight") in sd:
        residual = _bn(F.conv2d(bottom, sd[p + ".project.0.weight"]), sd, p + ".project.1")
    else:
        residual = bottom
    if level_root:
        children.append(bottom)
    if levels == 1:
        x1 = _basic_block(x, sd, p + ".tree1", stride, residual)
        x2 = _basic_block(x1, sd, p + ".tree2", 1)
        return _root([x2, x1] + children, sd, p + ".root")
    x1 = _tree(x, sd, p + ".tree1", levels - 1, stride, False)
    children.append(x1)
    return _tree(x1, sd, p + ".tree2", levels - 1, 1, False, children)


def dla34_base(x, sd):
    """dla.py:400-411 with dla34 = levels [1,1,1,2,2,1] (dla.py:433-436)."""
    y = []
    x = F.relu(_bn(F.conv2d(x, sd["base.base_layer.0.weight"], None, 1, 3), sd, "base.base_layer.1"))
    x = F.relu(_bn(F.conv2d(x, sd["base.level0.0.weight"], None, 1, 1), sd, "base.level0.1")); y.append(x)
    x = F.relu(_bn(F.conv2d(x, sd["base.level1.0.weight"], None, 2, 1), sd, "base.level1.1")); y.append(x)
    lv = [1, 1, 1, 2, 2, 1]
    for L in range(2, 6):
        x = _tree(x, sd, "base.level%d" % L, lv[L], 2, L != 2)
        y.append(x)
    return y


def deform_conv(x, sd, p):
    """DeformConv.forward dla.py:662-665: DCN -> BN -> ReLU."""
    y = dcn_v2_forward(x, sd[p + ".conv.conv_offset_mask.weight"], sd[p + ".conv.conv_offset_mask.bias"],
                       sd[p + ".conv.weight"], sd[p + ".conv.bias"])
    return F.relu(_bn(y, sd, p + ".actf.0"))


def _ida_up(layers, sd, p, startp, endp):
    """IDAUp.forward dla.py:693-699."""
    for i in range(startp + 1, endp):
        k = i - startp
        wup = sd[p + ".up_%d.weight" % k]
        f = wup.shape[2] // 2
        t = deform_conv(layers[i], sd, p + ".proj_%d" % k)
        t = F.conv_transpose2d(t, wup, None, stride=f, padding=f // 2, groups=wup.shape[0])
        layers[i] = deform_conv(t + layers[i - 1], sd, p + ".node_%d" % k)


def dlaseg_features(x, sd):
    """DLASeg.img2feats dla.py:789-802 -> (feat 64@/4, FeatureMaps[13])."""
    base = dla34_base(x, sd)
    fmaps = list(base)
    layers = list(base)
    out = [layers[-1]]                                  # DLAUp.forward dla.py:728-735
    for i in range(len(layers) - 2 - 1):
        _ida_up(layers, sd, "dla_up.ida_%d" % i, len(layers) - i - 2, len(layers))
        out.insert(0, layers[-1])
    fmaps += out
    y = [out[i].clone() for i in range(3)]
    _ida_up(y, sd, "ida_up", 0, 3)
    fmaps += y
    return y[-1], fmaps


def head_forward(feat, sd, h):
    """base_model.py:37-66: 3x3(64->256)+bias, ReLU, 1x1(256->c)+bias."""
    t = F.relu(F.conv2d(feat, sd[h + ".0.weight"], sd[h + ".0.bias"], 1, 1))
    return F.conv2d(t, sd[h + ".2.weight"], sd[h + ".2.bias"])


def dlaseg_forward(x, sd, dataset="mot"):
    """BaseModel.forward base_model.py:111-132 -> (dict head->map, FeatureMaps)."""
    feat, fmaps = dlaseg_features(x, sd)
    return {h: head_forward(feat, sd, h) for h in HEADS[dataset]}, fmaps


# --------------------------------------------------------------------------
# decode (detector.py:486-494, decode.py:102-196, utils.py:69-104)
# --------------------------------------------------------------------------
def sigmoid_output(output, depth_scale=1.0):
    out = dict(output)
    out["hm"] = torch.sigmoid(out["hm"])
    if "dep" in out:
        out["dep"] = (1.0 / (torch.sigmoid(out["dep"]) + 1e-6) - 1.0) * depth_scale
    return out


def _gather_at(feat, ind):
    """utils.py:25-36."""
    B, C = feat.shape[:2]
    f = feat.permute(0, 2, 3, 1).reshape(B, -1, C)
    return f.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))


def generic_decode(output, K=100):
    """decode.py:102-196 for the heads DEFT uses (hm reg wh ltrb_amodal + regression heads)."""
    heat = output["hm"]
    B, cat, H, W = heat.shape
    hmax = F.max_pool2d(heat, 3, 1, 1)                                  # utils.py:69-74
    heat = heat * (hmax == heat).float()
    sc, inds = torch.topk(heat.view(B, cat, -1), K)                     # utils.py:89-104
    inds = inds % (H * W)
    ys = (inds / W).int().float(); xs = (inds % W).int().float()
    score, ind = torch.topk(sc.view(B, -1), K)
    clses = (ind / K).int()
    inds = inds.view(B, -1).gather(1, ind)
    ys0 = ys.view(B, -1).gather(1, ind); xs0 = xs.view(B, -1).gather(1, ind)
    ret = {"scores": score, "clses": clses.float(), "xs": xs0, "ys": ys0,
           "cts": torch.stack([xs0, ys0], 2), "inds": inds}
    if "reg" in output:
        reg = _gather_at(output["reg"], inds)
        xs = xs0.view(B, K, 1) + reg[:, :, 0:1]; ys = ys0.view(B, K, 1) + reg[:, :, 1:2]
    else:
        xs = xs0.view(B, K, 1) + 0.5; ys = ys0.view(B, K, 1) + 0.5
    if "wh" in output:
        wh = _gather_at(output["wh"], inds).clamp(min=0)                # decode.py:135
        ret["bboxes"] = torch.cat([xs - wh[..., 0:1] / 2, ys - wh[..., 1:2] / 2,
                                   xs + wh[..., 0:1] / 2, ys + wh[..., 1:2] / 2], 2)
    for h in ["tracking", "dep", "rot", "dim", "amodel_offset"]:
        if h in output:
            ret[h] = _gather_at(output[h], inds)
    if "ltrb_amodal" in output:                                         # decode.py:178-196
        l = _gather_at(output["ltrb_amodal"], inds)
        x0 = xs0.view(B, K, 1); y0 = ys0.view(B, K, 1)
        ret["bboxes_amodal"] = torch.cat([x0 + l[..., 0:1], y0 + l[..., 1:2],
                                          x0 + l[..., 2:3], y0 + l[..., 3:4]], 2)
        ret["bboxes"] = ret["bboxes_amodal"]
    return ret


# --------------------------------------------------------------------------
# AFE: embedding extraction and pairwise affinity
# --------------------------------------------------------------------------
def convert_detection(boxes, h, w):
    """image.py:391-412 (tlbr boxes in image px -> centres in [-1,1]); no .cuda()."""
    d = np.array(boxes, dtype=np.float64).copy()
    d[:, 2] -= d[:, 0]; d[:, 3] -= d[:, 1]
    d[:, 0] /= w; d[:, 2] /= w; d[:, 1] /= h; d[:, 3] /= h
    c = (2 * d[:, 0:2] + d[:, 2:4]) - 1.0
    return torch.from_numpy(c.astype(float)).float().view(1, -1, 1, 1, 2)


def afe_extract(fmaps, centers, sd, align_corners=False):
    """AFE.forward_feature_extracter AFE.py:88-92 -> forward_selector_stacker1
    AFE.py:162-188: ReLU(3x3 selector conv) on all 13 maps, bilinear grid_sample
    (border padding) at each centre.  AFE.py:178 passes no align_corners: False on torch >= 1.3
    (what the reference computes today and what the fixtures pin); True reproduces torch 1.2."""
    srcs = [F.relu(F.conv2d(x, sd["AFE.selector.%d.weight" % k], sd["AFE.selector.%d.bias" % k], 1, 1))
            for k, x in enumerate(fmaps)]
    N = centers.shape[1]
    grid = centers.view(1, N, 1, 2)
    res = [F.grid_sample(s, grid, mode="bilinear", padding_mode="border", align_corners=align_corners)
           .squeeze(3).permute(0, 2, 1) for s in srcs]                  # each [1,N,C]
    return torch.cat(res, 2)


def afe_affinity(xp, xn, sd, max_object=100):
    """AFE.forward_stacker_features AFE.py:110-160 (fill_up_column=False), with
    forward_stacker2 AFE.py:190-207 and forward_final AFE.py:209-213.
    xp [1,P,D], xn [1,Q,D] -> float32 numpy [P, Q+1]."""
    P, Q, D = xp.shape[1], xn.shape[1], xp.shape[2]
    M = max_object
    a = torch.cat([xp, torch.zeros(1, M - P, D)], 1)
    b = torch.cat([xn, torch.zeros(1, M - Q, D)], 1)
    a = a.unsqueeze(2).repeat(1, 1, M, 1).permute(0, 3, 1, 2).contiguous()
    b = b.unsqueeze(1).repeat(1, M, 1, 1).permute(0, 3, 1, 2).contiguous()
    x = torch.cat([_bn(a, sd, "AFE.stacker2_bn"), _bn(b, sd, "AFE.stacker2_bn")], 1)
    for i in (0, 3, 6):
        x = F.conv2d(x, sd["AFE.final_net.%d.weight" % i], sd["AFE.final_net.%d.bias" % i])
        x = F.relu(_bn(x, sd, "AFE.final_net.%d" % (i + 1)))
    for i in (9, 11):
        x = F.relu(F.conv2d(x, sd["AFE.final_net.%d.weight" % i], sd["AFE.final_net.%d.bias" % i]))
    x = x[0, 0].clone()
    x[:, Q:] = 0
    x[P:, :] = 0
    x = torch.cat([x, torch.ones(1, M)], 0)
    x = torch.cat([x, torch.ones(M + 1, 1)], 1)
    x_f = F.softmax(x, dim=1); x_t = F.softmax(x, dim=0)
    rows = list(range(P)) + [M]; cols = list(range(Q)) + [M]
    x_f = x_f[rows][:, cols]; x_t = x_t[rows][:, cols]
    out = torch.zeros(P, Q + 1)
    out[:, :Q] = torch.max(x_f[:P, :Q], x_t[:P, :Q])
    out[:, Q] = x_f[:P, Q]
    return out.numpy()


# --------------------------------------------------------------------------
# LSTM motion model
# --------------------------------------------------------------------------
def lstm_predict(h0, c0, x, lsd):
    """KalmanFilterLSTM.predict kalman_filter_lstm.py:65-78 for T tracks at once.
    h0,c0 [T,128]; x [T,nin] -> hn, cn [T,128], pred [T, nfut, 4] (one LSTM step,
    Linear 128->64, Linear 64->4*nfut, no activation between the Linears)."""
    gates = x @ lsd["lstm.weight_ih_l0"].t() + lsd["lstm.bias_ih_l0"] \
        + h0 @ lsd["lstm.weight_hh_l0"].t() + lsd["lstm.bias_hh_l0"]
    i, f, g, o = gates.chunk(4, 1)                     # torch.nn.LSTM gate order
    c = torch.sigmoid(f) * c0 + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    y = (h @ lsd["out1.weight"].t() + lsd["out1.bias"]) @ lsd["out2.weight"].t() + lsd["out2.bias"]
    return h, c, y.view(x.shape[0], -1, 4)


class MotionTrack:
    """One track's motion-model state: STrack.update_lstm_features (tracker.py:408-480, 2-D, box = tlwh) and
    STrack.update_lstm_features_ddd (tracker.py:482-580, 3-D, box = (h, w, l, x, y, z, rot_y)).  Python-float
    (float64) feature arithmetic, `.float()` once, lstm_predict, then the in-place numpy post-processing of the
    predictions (float32 arrays `+=` float64 scalars: summed in float64, stored as float32)."""

    def __init__(self, lsd, ddd=False):
        self.lsd, self.ddd = lsd, ddd
        self.h = torch.zeros(1, 128); self.c = torch.zeros(1, 128)
        self.prev = None                 # (frame_id, quantities of the previous observation)
        self.features = None

    def update(self, box, frame_id):
        b = [float(v) for v in np.asarray(box, dtype=np.float64)]
        if not self.ddd:
            cx, cy, w, h = b[0] + b[2] / 2, b[1] + b[3] / 2, b[2], b[3]
            if self.prev is None:
                dcx = dcy = dh = dw = 0.0
            else:
                f0, pcx, pcy, pw, ph = self.prev
                dcx, dcy = (cx - pcx) / (frame_id - f0), (cy - pcy) / (frame_id - f0)
                dh, dw = h - ph, w - pw
            self.prev = (frame_id, cx, cy, w, h)
            feats = [cx, cy, dcx, dcy, h, w, w / h, dh, dw, dcx, dcy]                    # tracker.py:448-462
        else:
            h, w, l, cx, cy, cz, rot = b
            if self.prev is None:
                d = [0.0] * 7; v = [0.0] * 4
            else:
                f0, ph, pw, pl, px, py, pz, prot = self.prev
                d = [cx - px, cy - py, cz - pz, h - ph, w - pw, l - pl, rot - prot]
                v = [(cx - px) / (frame_id - f0), (cy - py) / (frame_id - f0), (cz - pz) / (frame_id - f0),
                     (rot - prot) / (frame_id - f0)]
            self.prev = (frame_id, h, w, l, cx, cy, cz, rot)
            feats = [cx, cy, cz, d[0], d[1], d[2], h, w, l, d[3], d[4], d[5], v[0], v[1], v[2], rot, d[6], v[3]]   # :544-566
        x = torch.from_numpy(np.array([feats])).float()
        self.features = x[0].numpy().copy()
        self.h, self.c, pred = lstm_predict(self.h, self.c, x, self.lsd)
        pred = pred[0].numpy().copy()                   # float32 [nfut, 4]
        out = {}
        for i in range(pred.shape[0]):
            p = pred[i]
            if not self.ddd:                            # tracker.py:471-480 -> (cx, cy, w/h, h)
                p[:2] += np.array([cx, cy]); p[2] += np.float64(h); p[3] += np.float64(w)      # float64 sums, float32 stores
                ph_, pw_ = p[2], p[3]
                p[3] = ph_; p[2] = pw_
                p[2] /= p[3]
                out[1 + i] = p
            else:                                       # tracker.py:573-580 -> (h, w, l, x, y, z, rot_y) float64
                p[:3] += np.array([cx, cy, cz]); p[3] += np.float64(rot)
                out[1 + i] = np.array([h, w, l] + p.tolist())
        return out


def track_similarity(sim_frame, tracks_nodes, frame_index, num_detections, dataset, max_track_node=50):
    """Tracker.get_similarity (tracker.py:663-688) with STrack.get_similarity (:219-252) for every track.
    sim_frame: {previous frame -> float32 [P, Q+1]} (= recorder.all_similarity[frame_index], decay applied);
    tracks_nodes: per track its nodes [(frame_index, id), ...] oldest first.  -> float64 [T, Q+1]."""
    mm = 2 if dataset == "nuscenes" else 4
    rows_out = []
    for nodes in tracks_nodes:
        rows = [sim_frame[f][i, :] for f, i in nodes if frame_index - f < max_track_node]
        if not rows:
            rows_out.append([0.0] * (num_detections + 1))
            continue
        a = np.array(rows)
        if a.shape[0] > mm + 1:
            a = a[a.shape[0] - mm:]
        rows_out.append(np.median(a, axis=0).tolist())
    return np.array(rows_out)


# --------------------------------------------------------------------------
# pre-processing (detector.py:377-395): cv2.warpAffine(INTER_LINEAR, constant 0 border) + normalisation, numpy restatement
# of cv2's fixed-point arithmetic (cv2 itself is absent here: PARITY UNPINNED; checked against a float bilinear resampling)
# --------------------------------------------------------------------------
def warp_affine_u8(img, M, out_w, out_h):
    """img [h,w,3] uint8, M 2x3 forward (src -> dst) -> [out_h, out_w, 3] uint8."""
    M = np.asarray(M, np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    i00, i01, i10, i11 = M[1, 1] * D, -M[0, 1] * D, -M[1, 0] * D, M[0, 0] * D
    b0, b1 = -i00 * M[0, 2] - i01 * M[1, 2], -i10 * M[0, 2] - i11 * M[1, 2]
    xs, ys = np.arange(out_w, dtype=np.float64), np.arange(out_h, dtype=np.float64)
    X = (np.rint(i00 * xs * 1024.0).astype(np.int64)[None, :] + np.rint((i01 * ys + b0) * 1024.0).astype(np.int64)[:, None] + 16) >> 5
    Y = (np.rint(i10 * xs * 1024.0).astype(np.int64)[None, :] + np.rint((i11 * ys + b1) * 1024.0).astype(np.int64)[:, None] + 16) >> 5
    sx, sy, fx, fy = X >> 5, Y >> 5, X & 31, Y & 31
    h, w = img.shape[:2]
    pad = np.zeros((h + 2, w + 2, 3), np.int64)
    pad[1:-1, 1:-1] = img

    def tap(yy, xx):
        ok = (yy >= -1) & (yy <= h) & (xx >= -1) & (xx <= w)
        return np.where(ok[..., None], pad[np.clip(yy + 1, 0, h + 1), np.clip(xx + 1, 0, w + 1)], 0)
    w00 = ((32 - fy) * (32 - fx) * 32)[..., None]; w01 = ((32 - fy) * fx * 32)[..., None]
    w10 = (fy * (32 - fx) * 32)[..., None]; w11 = (fy * fx * 32)[..., None]
    v = (w00 * tap(sy, sx) + w01 * tap(sy, sx + 1) + w10 * tap(sy + 1, sx) + w11 * tap(sy + 1, sx + 1) + (1 << 14)) >> 15
    return v.astype(np.uint8)


def preprocess_u8(img, M, out_w, out_h, mean, std):
    """detector.py:389-394: warp, ((x / 255. - mean) / std).astype(float32), HWC -> [1,3,H,W]."""
    inp = warp_affine_u8(img, M, out_w, out_h)
    x = ((inp / 255.0 - mean.reshape(1, 1, 3)) / std.reshape(1, 1, 3)).astype(np.float32)
    return torch.from_numpy(x.transpose(2, 0, 1)[None].copy())
