"""Drop-in seams for the reference's inference path.

Each object here has the name, call signature and return types the reference's host
code expects at one of its plugin seams, and routes the work to libdeft_hip.so:

  seam 1  `from dcn_v2 import DCN`            dla.py:25-29, 652-663   -> DCN (nn.Module)
  seam 2  model.AFE.forward_feature_extracter  tracker.py:776, 826    -> AfeSeam
  seam 3  model.AFE.forward_stacker_features   tracker.py:87          -> AfeSeam
  seam 4  KalmanFilterLSTM.predict             tracker.py:467, 571    -> KalmanFilterLSTM
  seam 5  generic_decode(output, K, opt)       detector.py:544        -> generic_decode
  seam 6  model(images, pre_img, pre_hm)       detector.py:535        -> DeftModel

`create_model(opt, state_dict)` is the one-line replacement for
`create_model(...)`/`load_model(...)` in detector.py:80-83.  There is no CPU fallback:
everything raises DeftHipError if the HIP library is missing.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import engine, hiplib
from .engine import View
from .hiplib import GemmDesc, ptr


def _lib(lib):
    return lib if lib is not None else hiplib.get_lib()


# ---------------------------------------------------------------------------------------------
# seam 1: dcn_v2.DCN
# ---------------------------------------------------------------------------------------------
class DCN(nn.Module):
    """Upstream-compatible modulated deformable conv module (CharlesShang/DCNv2 `DCN`):
    parameters `weight [Co,Ci,3,3]`, `bias [Co]`, submodule `conv_offset_mask` =
    Conv2d(Ci -> deformable_groups*27, 3x3, same stride/pad, bias) -- the state_dict keys
    pretrained DEFT checkpoints carry.  forward(x [N,Ci,H,W]) -> [N,Co,H,W]."""

    lib = None     # tests may inject the emulator build here

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        ks = tuple(kernel_size) if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        if ks != (3, 3) or stride != 1 or padding != 1 or dilation != 1 or deformable_groups != 1:
            raise NotImplementedError("deft_amd DCN implements the configuration DEFT uses (dla.py:652-660): 3x3, s1, p1, d1, 1 group")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.conv_offset_mask = nn.Conv2d(in_channels, 27, 3, 1, 1, bias=True)
        n = in_channels * 9
        self.weight.data.uniform_(-1.0 / n ** 0.5, 1.0 / n ** 0.5)     # upstream reset_parameters
        self.conv_offset_mask.weight.data.zero_()                       # upstream init_offset
        self.conv_offset_mask.bias.data.zero_()
        self._packed = None

    def _pack(self, dev):
        ver = (self.weight._version, self.bias._version, self.conv_offset_mask.weight._version,
               self.conv_offset_mask.bias._version, str(dev))
        if self._packed is None or self._packed[0] != ver:
            wo, Ko = engine.pack_conv_weight(self.conv_offset_mask.weight.detach().cpu())
            wm, Km = engine.pack_dcn_weight(self.weight.detach().cpu())
            self._packed = (ver, wo.to(dev), Ko, self.conv_offset_mask.bias.detach().float().to(dev).contiguous(),
                            wm.to(dev), Km, self.bias.detach().float().to(dev).contiguous())
        return self._packed[1:]

    def forward(self, x):
        lib = _lib(DCN.lib)
        dev = x.device
        wo, Ko, bo, wm, Km, bm = self._pack(dev)
        N, Ci, H, W = x.shape
        plan = engine._Plan(dev, lib)
        xv = plan.alloc(N, H, W, Ci)
        s = hiplib.stream_ptr(dev)
        xc = x.detach().float().contiguous()
        lib.call("deft_nchw_to_nhwc", ptr(xc), C.c_void_p(xv.addr), N, Ci, H, W, xv.ld, s)
        om = plan.alloc(N, H, W, 27, ld=32)
        plan.conv("offset", xv, wo, Ko, 3, 3, 1, 1, 27, None, bo, False, out=om)
        out = plan.alloc(N, H, W, self.out_channels)
        d = GemmDesc()
        d.x = xv.addr; d.x2 = om.addr; d.w = wm.data_ptr(); d.scale = None; d.shift = bm.data_ptr(); d.res = None; d.y = out.addr
        d.N, d.H, d.W, d.Cin, d.ldx = N, H, W, Ci, xv.ld
        d.OH, d.OW, d.Cout, d.ldy, d.ldr = H, W, self.out_channels, out.ld, 0
        d.KH, d.KW, d.stride, d.pad = 3, 3, 1, 1
        d.Ktot, d.Kpad, d.cin_log2, d.M = Km, wm.shape[1], int(np.log2(Ci)), N * H * W
        d.relu = 0; d.Q = 0; d.ldom = om.ld; d.tile = 0
        plan.run()
        lib.call("deft_dcn_v2_nhwc", C.byref(d), s)
        y = torch.empty(N, self.out_channels, H, W, dtype=torch.float32, device=dev)
        lib.call("deft_nhwc_to_nchw", C.c_void_p(out.addr), ptr(y), N, self.out_channels, H, W, out.ld, s)
        return y


# ---------------------------------------------------------------------------------------------
# seams 2+3: AFE methods
# ---------------------------------------------------------------------------------------------
class AfeSeam:
    """Stands in for `model.AFE` (AFE.py:18): the two methods the tracker calls."""

    def __init__(self, state_dict, max_object=100, device="cuda", lib=None, align_corners=False):
        self.plan = engine.AfePlan(state_dict, max_object, device, _lib(lib), align_corners=align_corners)
        self.device = self.plan.device
        self.max_object = max_object
        self.host_copy = True       # affinity_many also returns the blocks as numpy (the reference's recorder layout)
        self.last_device = None

    def _views(self, FeatureMaps):
        """Accept the 13 maps either as NHWC Views (DeftModel) or as NCHW tensors (reference model)."""
        out = []
        for fm in FeatureMaps:
            if isinstance(fm, View):
                out.append(fm)
                continue
            N, Cc, H, W = fm.shape
            v = self.plan.alloc(N, H, W, Cc)
            src = fm.detach().float().contiguous().to(self.device)
            self.plan._keep.append(src)
            self.plan.lib.call("deft_nchw_to_nhwc", ptr(src), C.c_void_p(v.addr), N, Cc, H, W, v.ld, hiplib.stream_ptr(self.device))
            out.append(v)
        return out

    def _to_device_f32(self, t):
        """Host -> device through PINNED staging (a ring of 4 buffers, each guarded by an event recorded behind the copy that reads it).  A pageable source makes the copy synchronous, and on this runtime it then also waits for whatever else the device is
        running -- e.g. a whole lookahead pass on another stream (profiles/r4_hw_queue_stall.md)."""
        if self.device.type != "cuda" or t.device.type == "cuda":
            return t.to(self.device, torch.float32)
        ring = getattr(self, "_pin_ring", None)
        if ring is None or ring[0][0].numel() < t.numel():
            ring = self._pin_ring = [[torch.empty(max(1024, 2 * t.numel()), dtype=torch.float32).pin_memory(), None] for _ in range(4)]
            self._pin_turn = 0
        self._pin_turn = (self._pin_turn + 1) % len(ring)
        slot = ring[self._pin_turn]
        if slot[1] is not None:                    # the copy that last read this slot (an event recorded behind it): normally long done -- but seven
            slot[1].synchronize()                  # per-class trackers on one seam call this 7 times a frame, with nothing in between on frame 1
        stage = slot[0][:t.numel()].view(t.shape)
        stage.copy_(t)
        out = stage.to(self.device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(self.device))
        return out

    def forward_feature_extracter(self, s, l):
        """AFE.py:88-92.  s: FeatureMaps (13), l: centres [1,N,1,1,2] in [-1,1] -> [1,N,D]."""
        keep = len(self.plan._keep)
        views = self._views(s)
        Nf = views[0].N
        n = l.shape[1]
        centers = self._to_device_f32(l.reshape(1, n, 2)).expand(Nf, n, 2).contiguous()
        emb = self.plan.extract(views, centers)
        if len(self.plan._keep) > keep:               # NCHW adapter made temporaries: wait for the launches that read them, then drop them
            if self.device.type == "cuda":            # and the descriptors built on their addresses (NHWC Views -- the plan's own maps -- need
                torch.cuda.current_stream(self.device).synchronize()          # neither: no host wait in the tracker's per-frame path)
            del self.plan._keep[keep:]
            self.plan._egroups.clear()
        return emb[0:1]

    def forward_stacker_features(self, xp, xn, fill_up_column=True):
        """AFE.py:110-160.  xp [1,P,D], xn [1,Q,D] -> numpy float32 [P,Q+1] (or [P,Q+P] with
        fill_up_column, AFE.py:147-150)."""
        out, _ = self.plan.affinity([xp[0]], xn[0])
        y = out.detach().cpu().numpy()
        P, Q = xp.shape[1], xn.shape[1]
        if fill_up_column and P > 1:
            y = np.concatenate([y, np.repeat(y[:, Q:Q + 1], P - 1, axis=1)], axis=1)
        return y

    def affinity_many(self, hist, cur):
        """All stored frames against the current one in ONE launch chain -- what
        FeatureRecorder.update's loop (tracker.py:76-90) asks for, without the per-pair D2H."""
        out, starts = self.plan.affinity(hist, cur)
        self.last_device = (out, starts)          # kept for deft_amd.tracker.get_similarity (no host round trip)
        if not self.host_copy:
            return [None] * len(hist)
        y = out.detach().cpu().numpy()
        return [y[starts[f]:starts[f + 1]] for f in range(len(hist))]


# ---------------------------------------------------------------------------------------------
# seam 4: LSTM motion model
# ---------------------------------------------------------------------------------------------
class KalmanFilterLSTM(object):
    """kalman_filter_lstm.py:32-102 `KalmanFilterLSTM`: same constructor (`KalmanFilterLSTM(opt)` loads
    `opt.load_model_traj`; tracker.py:144, 301, 661 build one per Tracker AND one per activated track), same
    `predict` and `gating_distance`.  Packed weights are cached per (checkpoint, device): the per-track
    constructions of tracker.py:301 cost nothing."""

    _plans = {}

    def __init__(self, opt, lstm_state_dict=None, device=None, lib=None):
        self.opt = opt
        self.MAX_dis_fut = 4 if opt.dataset == "nuscenes" else 5
        if device is None:
            gpus = getattr(opt, "gpus", [0])
            device = "cuda" if gpus and gpus[0] >= 0 else "cpu"          # kalman_filter_lstm.py:55
        if lstm_state_dict is not None:
            self.plan = engine.LstmPlan(lstm_state_dict, device, _lib(lib))
            return
        path = getattr(opt, "load_model_traj", "")
        key = (path, opt.dataset, str(device), id(lib))
        if key not in KalmanFilterLSTM._plans:
            KalmanFilterLSTM._plans[key] = engine.LstmPlan(self._load(path, opt.dataset), device, _lib(lib))
        self.plan = KalmanFilterLSTM._plans[key]

    @staticmethod
    def _load(path, dataset):
        """model.py:40-53 for the trajectory checkpoint; with no path the reference keeps a freshly
        initialised DecoderRNN (kalman_filter_lstm.py:51-52) -- same module construction here."""
        if path:
            ck = torch.load(path, map_location="cpu")
            sd = ck["state_dict"] if "state_dict" in ck else ck
            return {(k[7:] if k.startswith("module.") and not k.startswith("module_list") else k): v for k, v in sd.items()}
        nin, nout = (18, 16) if dataset == "nuscenes" else (11, 20)
        m = nn.ModuleDict({"lstm": nn.LSTM(nin, 128), "out1": nn.Linear(128, 64), "out2": nn.Linear(64, nout)})
        return {k: v.detach() for k, v in m.state_dict().items()}

    def predict(self, h0, c0, new_features):
        """h0, c0 [1,1,128]; new_features [1,1,nin] -> (hn, cn, {1..MAX_dis_fut: float32[4]})."""
        dev = self.plan.device
        h = h0.reshape(1, 128).to(dev, torch.float32).clone()
        c = c0.reshape(1, 128).to(dev, torch.float32).clone()
        pred = self.plan.step(new_features.reshape(1, -1), h, c)
        x = pred.view(self.MAX_dis_fut, -1).cpu().numpy()
        return h.view(1, 1, 128), c.view(1, 1, 128), {1 + i: x[i] for i in range(self.MAX_dis_fut)}

    def predict_batch(self, h, c, feats):
        """All tracks updated this frame at once: h, c [T,128] in place, feats [T,nin] -> [T,fut,4]."""
        return self.plan.step(feats, h, c)

    def gating_distance(self, mean, covariance, measurements, only_position=False, metric="maha"):
        """kalman_filter_lstm.py:80-102 (host numpy, float64; called by matching.fuse_motion / fuse_motion_ddd).
        Kept as written: the "gaussian" metric compares components 3:-1 of the vectors AFTER the
        only_position slice, so on the 2-D path (matching.py:353-366) the slice is empty and the
        distance is 0 for every detection; on the 3-D path it is the centre distance."""
        mean = np.asarray(mean); measurements = np.asarray(measurements)
        if only_position:
            mean, covariance = mean[:2], covariance[:2, :2]
            measurements = measurements[:, :2]
        if metric == "gaussian":
            d = measurements[:, 3:-1] - mean[3:-1]
            return np.sqrt(np.sum(d * d, axis=1))
        if metric == "maha":
            import scipy.linalg
            d = measurements - mean
            chol = np.linalg.cholesky(covariance)
            z = scipy.linalg.solve_triangular(chol, d.T, lower=True, check_finite=False, overwrite_b=True)
            return np.sum(z * z, axis=0)
        raise ValueError("invalid distance metric")


# ---------------------------------------------------------------------------------------------
# seam 5: generic_decode on dense head maps
# ---------------------------------------------------------------------------------------------
def generic_decode(output, K=100, opt=None, lib=None):
    """decode.py:102-196 for dense NCHW head maps (`output["hm"]` already sigmoid'ed, as
    detector.py:537 leaves it).  Peak NMS + top-K run in the HIP library; the K-row gathers
    are torch indexing."""
    lib = _lib(lib)
    heat = output["hm"]
    dev = heat.device
    N, Cc, H, W = heat.shape
    s = hiplib.stream_ptr(dev)
    ld = (Cc + 3) // 4 * 4
    hv = torch.zeros(N, H, W, ld, dtype=torch.float32, device=dev)
    hc = heat.detach().float().contiguous()
    lib.call("deft_nchw_to_nhwc", ptr(hc), ptr(hv), N, Cc, H, W, ld, s)
    cap = H * W * Cc
    cs = torch.zeros(N * cap, dtype=torch.float32, device=dev)
    ci = torch.zeros(N * cap, dtype=torch.int32, device=dev)
    cn = torch.zeros(N, dtype=torch.int32, device=dev)
    sc = torch.zeros(N, K, dtype=torch.float32, device=dev)
    ind = torch.zeros(N, K, dtype=torch.int32, device=dev)
    cl = torch.zeros(N, K, dtype=torch.int32, device=dev)
    lib.call("deft_hm_peaks", ptr(hv), N, H, W, Cc, ld, 0, ptr(cs), ptr(ci), ptr(cn), cap, s)
    lib.call("deft_topk", ptr(cs), ptr(ci), ptr(cn), N, cap, K, H * W, ptr(sc), ptr(ind), ptr(cl), s)
    inds = ind.long()
    ys0 = (inds // W).float(); xs0 = (inds % W).float()
    ret = {"scores": sc, "clses": cl.float(), "xs": xs0, "ys": ys0, "cts": torch.stack([xs0, ys0], 2)}

    def gat(name):
        f = output[name]
        return f.permute(0, 2, 3, 1).reshape(N, H * W, f.shape[1]).gather(1, inds.unsqueeze(2).expand(N, K, f.shape[1]))
    if "reg" in output:
        reg = gat("reg")
        xs = xs0.view(N, K, 1) + reg[..., 0:1]; ys = ys0.view(N, K, 1) + reg[..., 1:2]
    else:
        xs = xs0.view(N, K, 1) + 0.5; ys = ys0.view(N, K, 1) + 0.5
    if "wh" in output:
        wh = gat("wh").clamp(min=0)
        ret["bboxes"] = torch.cat([xs - wh[..., 0:1] / 2, ys - wh[..., 1:2] / 2, xs + wh[..., 0:1] / 2, ys + wh[..., 1:2] / 2], 2)
    for h in ["tracking", "dep", "rot", "dim", "amodel_offset", "nuscenes_att", "velocity"]:
        if h in output:
            ret[h] = gat(h)
    if "ltrb_amodal" in output:
        l = gat("ltrb_amodal")
        x0 = xs0.view(N, K, 1); y0 = ys0.view(N, K, 1)
        ret["bboxes_amodal"] = torch.cat([x0 + l[..., 0:1], y0 + l[..., 1:2], x0 + l[..., 2:3], y0 + l[..., 3:4]], 2)
        ret["bboxes"] = ret["bboxes_amodal"]
    return ret


# ---------------------------------------------------------------------------------------------
# seam 6: the model callable
# ---------------------------------------------------------------------------------------------
class DeftModel(object):
    """Callable with DLASeg's inference contract (base_model.py:111-132):
    `model(images, pre_img=None, pre_hm=None) -> ([{head: NCHW tensor}], FeatureMaps[13])`,
    attribute `.AFE` with the two tracker-facing methods.  Plans are cached per input shape."""

    def __init__(self, state_dict, dataset="mot", K=100, max_object=100, device="cuda", lib=None):
        self.sd, self.dataset, self.K = state_dict, dataset, K
        self.device = torch.device(device)
        self.lib = _lib(lib)
        self.AFE = AfeSeam(state_dict, max_object, device, self.lib)
        self._plans = {}
        self._graphs = {}          # (N,H,W) -> None after the first (eager) call, then the captured hipGraph
        self.hip_graphs = self.device.type == "cuda"

    def eval(self):
        return self

    def to(self, device):
        return self

    def plan_for(self, N, H, W):
        key = (N, H, W)
        if key not in self._plans:
            self._plans[key] = engine.DlaSegPlan(self.sd, N, H, W, self.dataset, K=self.K, device=self.device,
                                                 lib=self.lib, dense_heads=True)
        return self._plans[key]

    def __call__(self, images, pre_img=None, pre_hm=None):
        assert pre_img is None and pre_hm is None, "DEFT inference never passes pre_img/pre_hm (detector.py:153,162)"
        N, _, H, W = images.shape
        plan = self.plan_for(N, H, W)
        images = images.to(self.device)
        key = (N, H, W)
        if not self.hip_graphs or key not in self._graphs:
            plan.forward(images)                    # first call of a shape: eager
            self._graphs.setdefault(key, None)
        else:                                       # then replay the launch list (one-frame calls are launch-bound on the host)
            if self._graphs[key] is None:
                self._graphs[key] = plan.capture_graph()
            plan.image.copy_(images, non_blocking=True)
            engine._Plan.replay_graph(self._graphs[key], self.device)
        s = hiplib.stream_ptr(self.device)
        out = {}
        for h, v in plan.dense.items():
            y = torch.empty(N, v.C, v.H, v.W, dtype=torch.float32, device=self.device)
            self.lib.call("deft_nhwc_to_nchw", C.c_void_p(v.addr), ptr(y), N, v.C, v.H, v.W, v.ld, s)
            out[h] = y
        return [out], plan.fmaps


def create_model(opt, state_dict, device="cuda", lib=None):
    """Replacement for `create_model(opt.arch, opt.heads, opt.head_conv, opt)` + `load_model`
    (detector.py:80-83).  Only `--arch dla_34` is a DEFT inference architecture."""
    if getattr(opt, "arch", "dla_34") != "dla_34":
        raise NotImplementedError("DEFT inference is defined for arch dla_34 only (dla.py:765, detector.py:535)")
    return DeftModel(state_dict, opt.dataset, getattr(opt, "K", 100), getattr(opt, "max_object", 100), device, lib)
