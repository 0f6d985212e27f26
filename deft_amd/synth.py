"""Seeded synthetic DEFT weights with the reference's state_dict key names and shapes
(no pretrained checkpoints exist offline: README.md:79 is a Google-Drive link).

Pure torch-CPU host utility shared by bench.py, the tests and the oracle: it
contains NO algorithm of the hot path, only the parameter table of
DLASeg / BaseModel / AFE_module (dla.py:758-787, base_model.py:24-103,
AFE.py:20-72, 331-366) and DecoderRNN (kalman_filter_lstm.py:9-21).  The table is
validated by oracle/make_golden.py, which loads the result into the reference's own
DLASeg with strict=True.
"""
import math

import numpy as np
import torch

# --------------------------------------------------------------------------
# configuration tables (reference: opts.py:500-520 heads; AFE.py:15-55 channels)
# --------------------------------------------------------------------------
HEADS = {
    "mot": {"hm": 1, "reg": 2, "wh": 2, "tracking": 2, "ltrb_amodal": 4},
    "kitti_tracking": {"hm": 3, "reg": 2, "wh": 2, "tracking": 2},
    "nuscenes": {"hm": 10, "reg": 2, "wh": 2, "tracking": 2, "dep": 1, "rot": 8,
                 "dim": 3, "amodel_offset": 2},
}
SELECTOR_IN = [16, 32, 64, 128, 256, 512, 64, 128, 256, 512, 64, 64, 64]  # AFE.py:15
SELECTOR_OUT = {
    "nuscenes": [48, 48, 64, 64, 64, 64, 64, 64, 64, 64, 32, 32, 32],     # AFE.py:23-38
    "default": [32] * 13,                                                 # AFE.py:40-55
}
FEATURE_STRIDES = [1, 2, 4, 8, 16, 32, 4, 8, 16, 32, 4, 4, 4]


def selector_out(dataset):
    return SELECTOR_OUT["nuscenes" if dataset == "nuscenes" else "default"]


# --------------------------------------------------------------------------
# deterministic synthetic weights (no reference needed; same on the GPU box)
# --------------------------------------------------------------------------
def _param_table(dataset, heads=None):
    """Ordered (name, shape, kind) list with the reference's state_dict names
    (dla.py DLASeg / base_model.py BaseModel / AFE.py AFE_module).  heads: {name: channels} when it differs from the
    dataset's default table (opts.py:500-520)."""
    T = []

    def conv(name, co, ci, k, bias=False, kind="conv"):
        T.append((name + ".weight", (co, ci, k, k), kind))
        if bias:
            T.append((name + ".bias", (co,), "bias"))

    def bn(name, c):
        T.append((name + ".weight", (c,), "bn_w"))
        T.append((name + ".bias", (c,), "bn_b"))
        T.append((name + ".running_mean", (c,), "bn_m"))
        T.append((name + ".running_var", (c,), "bn_v"))
        T.append((name + ".num_batches_tracked", (), "bn_n"))

    def block(p, ci, co):
        conv(p + ".conv1", co, ci, 3); bn(p + ".bn1", co)
        conv(p + ".conv2", co, co, 3); bn(p + ".bn2", co)

    def tree(p, levels, ci, co, level_root, root_dim=0):
        if root_dim == 0:
            root_dim = 2 * co
        if level_root:
            root_dim += ci
        if levels == 1:
            block(p + ".tree1", ci, co)
            block(p + ".tree2", co, co)
            conv(p + ".root.conv", co, root_dim, 1); bn(p + ".root.bn", co)
        else:
            tree(p + ".tree1", levels - 1, ci, co, False, 0)
            tree(p + ".tree2", levels - 1, co, co, False, root_dim + co)
        if ci != co:
            conv(p + ".project.0", co, ci, 1); bn(p + ".project.1", co)

    def dcn(p, ci, co):
        bn(p + ".actf.0", co)
        T.append((p + ".conv.weight", (co, ci, 3, 3), "conv"))
        T.append((p + ".conv.bias", (co,), "bias"))
        T.append((p + ".conv.conv_offset_mask.weight", (27, ci, 3, 3), "off_w"))
        T.append((p + ".conv.conv_offset_mask.bias", (27,), "off_b"))

    def ida(p, o, channels, up_f):
        for i in range(1, len(channels)):
            dcn(p + ".proj_%d" % i, channels[i], o)
            f = int(up_f[i])
            T.append((p + ".up_%d.weight" % i, (o, 1, 2 * f, 2 * f), "up"))
            dcn(p + ".node_%d" % i, o, o)

    heads = HEADS[dataset] if heads is None else heads
    for h, c in heads.items():
        conv(h + ".0", 256, 64, 3, bias=True)
        conv(h + ".2", c, 256, 1, bias=True, kind="hm_out" if h == "hm" else "head_out")
    D = sum(selector_out(dataset))
    bn("AFE.stacker2_bn", D)
    fin = [2 * D, 512, 256, 128, 64, 1]
    idx = 0
    cin = fin[0]
    for v in fin[1:-2]:
        conv("AFE.final_net.%d" % idx, v, cin, 1, bias=True); bn("AFE.final_net.%d" % (idx + 1), v)
        idx += 3; cin = v
    for v in fin[-2:]:
        conv("AFE.final_net.%d" % idx, v, cin, 1, bias=True, kind="conv" if v != 1 else "aff_out")
        idx += 2; cin = v
    for k, (ci, co) in enumerate(zip(SELECTOR_IN, selector_out(dataset))):
        conv("AFE.selector.%d" % k, co, ci, 3, bias=True)
    conv("base.base_layer.0", 16, 3, 7); bn("base.base_layer.1", 16)
    conv("base.level0.0", 16, 16, 3); bn("base.level0.1", 16)
    conv("base.level1.0", 32, 16, 3); bn("base.level1.1", 32)
    ch = [16, 32, 64, 128, 256, 512]
    lv = [1, 1, 1, 2, 2, 1]
    for L in range(2, 6):
        tree("base.level%d" % L, lv[L], ch[L - 1], ch[L], L != 2)
    # DLAUp(2, [64,128,256,512], [1,2,4,8]) -- dla.py:702-726
    channels = [64, 128, 256, 512]
    in_ch = list(channels)
    scales = np.array([1, 2, 4, 8])
    for i in range(3):
        j = -i - 2
        ida("dla_up.ida_%d" % i, channels[j], in_ch[j:], scales[j:] // scales[j])
        scales[j + 1:] = scales[j]
        in_ch[j + 1:] = [channels[j] for _ in channels[j + 1:]]
    ida("ida_up", 64, [64, 128, 256], [1, 2, 4])
    return T


def _up_weight(shape):
    """dla.py:565-573 fill_up_weights."""
    w = torch.zeros(shape)
    k = shape[2]
    f = math.ceil(k / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(k):
        for j in range(k):
            w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    w[1:] = w[0:1]
    return w


def _is_trunk(name):
    return name.startswith(("base.", "dla_up.", "ida_up."))


def synth_state_dict(dataset="mot", seed=317):
    """Seeded synthetic weights with the reference's key names and shapes.
    BN statistics are randomised and the DCN offset conv is NON-zero (upstream
    zero-inits it, which would degenerate DCN into a plain conv).
    Every dataset gets the SAME backbone + neck (the MOT table's draw): one generator walks the table heads-first, so a different head
    table used to shift the whole random stream and the KITTI / nuScenes trunks came out an order of magnitude worse conditioned than
    MOT's (heat-map logits in [-13, 4.6], 7e-4 .. 1e-3 between any two fp32 summation orders) -- a property of the draw, not of the
    architecture.  Heads and AFE of the other datasets come from their own generator."""
    trunk = None
    if dataset != "mot":
        trunk = {k: v for k, v in synth_state_dict("mot", seed).items() if _is_trunk(k)}
        seed = seed + 1000 + sum(ord(c) for c in dataset)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape, kind in _param_table(dataset):
        if trunk is not None and _is_trunk(name):
            assert tuple(trunk[name].shape) == tuple(shape), name
            sd[name] = trunk[name]
            continue
        if kind in ("conv", "head_out", "hm_out", "aff_out"):
            fan_in = shape[1] * shape[2] * shape[3]
            gain = {"conv": 2.0, "aff_out": 6.0}.get(kind, 1.0)
            t = torch.randn(shape, generator=g) * math.sqrt(gain / fan_in)
        elif kind == "bias":
            t = torch.randn(shape, generator=g) * 0.1
        elif kind == "bn_w":
            t = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif kind == "bn_b":
            t = torch.randn(shape, generator=g) * 0.2
        elif kind == "bn_m":
            t = torch.randn(shape, generator=g) * 0.3
        elif kind == "bn_v":
            t = torch.rand(shape, generator=g) * 1.5 + 0.5
        elif kind == "bn_n":
            t = torch.tensor(1, dtype=torch.long)
        elif kind == "off_w":
            t = torch.randn(shape, generator=g) * (0.5 / math.sqrt(shape[1] * 9))
        elif kind == "off_b":
            t = torch.randn(shape, generator=g) * 0.5
        elif kind == "up":
            t = _up_weight(shape)
        else:
            raise KeyError(kind)
        sd[name] = t
    sd["hm.2.bias"] = torch.full_like(sd["hm.2.bias"], -4.6)  # base_model.py:91-92, opts.py:151
    sd["AFE.final_net.11.bias"] = torch.full_like(sd["AFE.final_net.11.bias"], 1.0)  # spread the affinities
    return sd


def synth_lstm_state_dict(dataset="mot", seed=318):
    """DecoderRNN parameters (kalman_filter_lstm.py:9-21)."""
    g = torch.Generator().manual_seed(seed)
    nin, nout = (18, 16) if dataset == "nuscenes" else (11, 20)
    k = 1.0 / math.sqrt(128)

    def u(*s, a=k):
        return (torch.rand(*s, generator=g) * 2 - 1) * a

    return {
        "lstm.weight_ih_l0": u(512, nin), "lstm.weight_hh_l0": u(512, 128),
        "lstm.bias_ih_l0": u(512), "lstm.bias_hh_l0": u(512),
        "out1.weight": u(64, 128), "out1.bias": u(64),
        "out2.weight": u(nout, 64, a=0.125), "out2.bias": u(nout, a=0.125),
    }




# --------------------------------------------------------------------------
# a heat map shaped like a TRAINED detector's (tests + bench.py's peaked gate stream)
# --------------------------------------------------------------------------
def peaked_head(sd, H, W, nblobs, seed=11, classes=1):
    """A final feature map and hm-head weights that give a heat map like a trained CenterNet's: `nblobs` Gaussian bumps
    with distinct amplitudes (peak logits spread over [-2, 6]) on the prior_bias = -4.6 background (base_model.py:91-92,
    opts.py:151), plus low-level feature noise.  `classes` > 1: blob b belongs to class b % classes; class c reads its own
    group of the 64 feature channels and of the 256 hidden channels, so every class map is a blob map of its own.
    -> (feat [1,64,h,w], state_dict with the hm head replaced)."""
    g = torch.Generator().manual_seed(seed)
    h, w = H // 4, W // 4
    rows = max(1, int((nblobs * h / w) ** 0.5))
    cols = -(-nblobs // rows)
    ch, cw = h // rows, w // cols
    assert ch >= 8 and cw >= 8, "blobs too dense for this map"
    amp = torch.linspace(2.6, 10.6, nblobs)[torch.randperm(nblobs, generator=g)]       # peak logit = -4.6 + amp
    amp = amp + (torch.rand(nblobs, generator=g) - 0.5) * 0.02
    yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
    bump = torch.zeros(classes, h, w)
    for b in range(nblobs):
        r, c = divmod(b, cols)
        cy = r * ch + 3 + int(torch.randint(0, ch - 6, (1,), generator=g))
        cx = c * cw + 3 + int(torch.randint(0, cw - 6, (1,), generator=g))
        sig = 1.2 + 0.6 * float(torch.rand(1, generator=g))
        bump[b % classes] += amp[b] * torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
    sd2 = dict(sd)
    if classes == 1:
        v = torch.rand(64, generator=g) + 0.2
        v = v / v.norm()
        feat = (bump[0][None] * v[:, None, None] + 0.02 * torch.rand(64, h, w, generator=g)).unsqueeze(0)
        u = torch.rand(256, generator=g) + 0.5
        w0 = torch.randn(256, 64, 3, 3, generator=g) * 0.01
        w0[:, :, 1, 1] += u[:, None] * v[None, :]                     # the centre tap reads the blob direction
        w2 = (torch.rand(1, 256, 1, 1, generator=g) + 0.5)
        w2 = w2 / float((w2.view(-1) * u).sum())                      # so that logit ~= -4.6 + bump
    else:
        fc, hc = 64 // classes, 256 // classes                       # feature / hidden channels per class
        v = torch.zeros(classes, 64)
        u = torch.zeros(classes, 256)
        for c in range(classes):
            vc = torch.rand(fc, generator=g) + 0.2
            v[c, c * fc:(c + 1) * fc] = vc / vc.norm()
            u[c, c * hc:(c + 1) * hc] = torch.rand(hc, generator=g) + 0.5
        feat = (torch.einsum("chw,cf->fhw", bump, v) + 0.02 * torch.rand(64, h, w, generator=g)).unsqueeze(0)
        w0 = torch.randn(256, 64, 3, 3, generator=g) * 0.01
        w0[:, :, 1, 1] += torch.einsum("cj,cf->jf", u, v)
        w2 = torch.zeros(classes, 256, 1, 1)
        for c in range(classes):
            wc = torch.rand(hc, generator=g) + 0.5
            w2[c, c * hc:(c + 1) * hc, 0, 0] = wc / float((wc * u[c, c * hc:(c + 1) * hc]).sum())
    sd2["hm.0.weight"], sd2["hm.0.bias"] = w0, torch.zeros(256)
    sd2["hm.2.weight"], sd2["hm.2.bias"] = w2, torch.full((classes,), -4.6)
    return feat, sd2
