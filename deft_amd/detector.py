"""Host-side mirror of the reference `Detector` for the kernel path (detector.py:72-344).

Same names / argument meaning as the reference for the seam this package replaces:
`Detector(opt)`, `.process(images, ...) -> (output, dets, FeatureMaps)`,
`.img_height/.img_width`, `.reset_tracking(opt)`.  The forward, sigmoid, decode and the
regression heads run as ONE plan of HIP launches (deft_amd.engine.DlaSegPlan) with a
single D2H copy of the K detection records, instead of the reference's four
`torch.cuda.synchronize()` points (detector.py:188, 534, 541, 545).
"""
import numpy as np
import torch

from . import engine, hiplib


import os as _os
PRIORITY_TRACKER = _os.environ.get("DEFT_TRACKER_PRIORITY", "1") != "0"      # run(prefetch=): the tracker's launches on a high-priority stream
# run(prefetch=): when the next frame's pass is queued -- "hook" = when the tracker announces the end of its device work (the pass then runs
# beside the host-only rest of the association), "early" = before the tracker starts (the pass runs beside the tracker's own launches,
# which go to the high-priority stream)
# "auto": hook for multi-frame passes (measured 2.12 vs 2.19 ms per frame at 4 frames per pass), early for one-frame passes (2.91 vs 3.18:
# a one-frame pass queued at the hook is not finished when the next call wants it) -- tools/probe/lookahead_probe.py
LOOKAHEAD_AT = _os.environ.get("DEFT_LOOKAHEAD_AT", "auto")
# run() on a recorded stream: when the host already holds the NEXT frame's detections (same lookahead pass, or the other slot's finished pass), the
# tracker's device half for that frame (ArrayTracker.begin) is queued right behind this frame's update().  "0": every frame's device half inside its own update()
BEGIN_AHEAD = _os.environ.get("DEFT_BEGIN_AHEAD", "1") != "0"
PREPARE_AHEAD = _os.environ.get("DEFT_PREPARE_AHEAD", "1") != "0"     # ... and the embeddings + affinity blocks of the frame after that (ArrayTracker.prepare)


def _fetch(d):
    """{name: device tensor} -> {name: numpy array} with ONE device->host copy (the fields of a decoded frame are ten small tensors: ten
    copies are ten stream synchronisations).  float64 holds every field exactly (float32 values, int64 indices < 2^53)."""
    if not d:
        return {}
    ref = next(iter(d.values()))
    if ref.device.type == "cpu":
        return {k: v.detach().numpy() for k, v in d.items()}
    flat = torch.cat([v.detach().reshape(-1).double() for v in d.values()]).cpu().numpy()
    out, o = {}, 0
    for k, v in d.items():
        n = v.numel()
        out[k] = flat[o:o + n].astype(_NP[v.dtype]).reshape(tuple(v.shape))
        o += n
    return out


_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.int64: np.int64, torch.int32: np.int32}


def _flag_finite(plan, d, lib):
    """Two-fp16-piece builds carry activations of |x| < 4094 (csrc/common.h); beyond that an operand is +-inf, the frame's heat map NaN -- and a
    NaN map has no peaks: the frame would come back EMPTY, silently.  One more field in the frame's record (is every heat-map logit finite?)
    turns that into an error in _check_finite.  Three-bf16-piece builds have no range limit: no flag."""
    if getattr(lib, "pieces", 3) == 2:
        d["_finite"] = torch.isfinite(plan.dense["hm"].buf).all().to(torch.float32).reshape(1)
    return d


def _check_finite(dets):
    """Raises FloatingPointError when the frame's heat map, or any decoded float field (scores, boxes, the regression heads at the peaks: they run the
    same split arithmetic), is not finite.  Detector.process / run catch it ONCE per detector: the frame is re-run on the three-bf16-piece entry
    points of the same library (no range limit) and the detector stays there (Detector._switch_to_safe)."""
    f = dets.pop("_finite", None)
    bad = f is not None and not bool(np.all(f))
    if not bad:
        bad = any(v.dtype.kind == "f" and not np.isfinite(v).all() for v in dets.values())
    if bad:
        raise FloatingPointError("non-finite heat map or detection record: an activation left the range of the two-fp16-piece arithmetic (|x| >= 4094, "
                                 "csrc/common.h); the three-bf16-piece entry points of the same library have no range limit (DEFT_ARITH=bf16x3)")
    return dets


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class Detector(object):
    def __init__(self, opt, state_dict=None):
        """opt: the reference's argparse namespace (opts.py); fields used: dataset, K,
        max_object, gpus (gpus[0] >= 0 -> cuda).  state_dict: DLASeg weights with the
        reference's key names (what model.load_model returns, model.py:40-121)."""
        if state_dict is None:
            ck = torch.load(opt.load_model, map_location="cpu")
            state_dict = {k[7:] if k.startswith("module.") else k: v for k, v in ck["state_dict"].items()}  # model.py:49-53
        self.opt = opt
        self.device = torch.device("cuda" if getattr(opt, "gpus", [0])[0] >= 0 else "cpu")
        self.lib = hiplib.get_lib()            # the plans below refuse a CPU device with the HIP library (no CPU path)
        # opt.deft_arith: "auto" (default) = two fp16 pieces, moving to the range-free three-bf16-piece entry points of the same library by itself when a
        # frame overflows (_switch_to_safe); "bf16x3" = start there (a model whose activations are known to be large: tools/validate_checkpoint.py
        # reports a checkpoint's largest activation against the fp16-piece range); "fp16x2" = never move (an overflow raises FloatingPointError)
        want = getattr(opt, "deft_arith", "auto")
        assert want in ("auto", "fp16x2", "bf16x3"), want
        if want == "bf16x3" and getattr(self.lib, "pieces", 3) == 2:
            twin = self.lib.twin()
            if twin is None:
                raise hiplib.DeftHipError("opt.deft_arith = 'bf16x3': this library carries no three-bf16-piece entry points")
            self.lib = twin
        self._arith_fixed = want == "fp16x2"
        self.arith = "fp16x2" if getattr(self.lib, "pieces", 3) == 2 else "bf16x3"
        self.sd = state_dict
        self.dataset = opt.dataset
        self.K = getattr(opt, "K", 100)
        self._plans = {}
        self._graphs = {}          # (N,H,W) -> None after the first (eager) frame, then the captured hipGraph
        self.hip_graphs = bool(getattr(opt, "hip_graphs", True)) and self.device.type == "cuda"
        self.afe = engine.AfePlan(state_dict, getattr(opt, "max_object", 100), self.device, self.lib)
        self.img_height = 100          # detector.py:108-109
        self.img_width = 100
        self.pre_images = None
        self.tracker = None            # set_tracker(): the reference's Tracker (or {class: Tracker} for nuScenes), detector.py:102-107
        self.times = {}                # stage seconds of the last run(), under the reference's names (detector.py:113-114, 340-344)
        self._ahead = {}               # (inp_h, inp_w, sh, sw) -> the two _Slot objects of run(..., prefetch=): frame k+1's network beside frame k's tracker
        self._ahead_turn = 0

    def set_tracker(self, tracker, factory=None):
        """The object `run()` hands the frame's detections to: `utils.tracker.Tracker(opt, model, h, w)` of the reference (bind
        deft_amd.tracker.accelerate first for the device forms), or {class name: Tracker} for nuScenes (detector.py:102-107).
        factory(opt, h, w) -> a fresh tracker (or dict): what `reset_tracking` calls per video, like the reference rebuilds its
        Tracker(s) there (detector.py:677-686).  Without one, reset_tracking re-constructs trackers of the same class with the same
        model (`type(t)(opt, t.model, h=, w=)`, the reference's constructor signature, tracker.py:632)."""
        self.tracker = tracker
        self._tracker_factory = factory

    def _plan(self, N, H, W):
        """The plan for fp32 NCHW input (process())."""
        key = (N, H, W)
        if key not in self._plans:
            self._plans[key] = engine.DlaSegPlan(self.sd, N, H, W, self.dataset, K=self.K, device=self.device, lib=self.lib)
            self._plans[key]._gkey = key
        return self._plans[key]

    def _minv(self, sh, sw, n=1):
        """dst -> src matrices [n, 6] of the input warp for sh x sw frames in this detector's input mode (preprocess.input_geometry)."""
        from . import preprocess as PR
        return np.tile(PR.invert_affine(PR.input_geometry(self.opt, sh, sw)[0])[None], (n, 1))

    def _plan_u8(self, N, H, W, sh, sw):
        """The plan for uint8 HWC frames of sh x sw (run() on a camera frame): its OWN plan and graph key -- use_u8_input rewrites the
        plan's first launch for good, so sharing the fp32 plan of process() would make a later process() call on the same input size
        run the network on the last uint8 frame."""
        key = ("u8", N, H, W, sh, sw)
        if key not in self._plans:
            p = engine.DlaSegPlan(self.sd, N, H, W, self.dataset, K=self.K, device=self.device, lib=self.lib)
            p.use_u8_input(sh, sw, minv=self._minv(sh, sw, N))
            p._gkey = key
            self._plans[key] = p
        return self._plans[key]

    arith = None      # "fp16x2" / "bf16x3": which entry points of the library this detector's plans run on (set in __init__)

    def _switch_to_safe(self, why):
        """An activation left the range of the two-fp16-piece arithmetic: move this detector -- plans, graphs, lookahead slots, the embedding /
        affinity plan the tracker shares -- to the three-bf16-piece entry points of the SAME library (hiplib.HipLib.twin(): six matrix instructions
        per fp32 product instead of three, no range limit, ~25 % slower) for the rest of its life.  Returns False when there is nothing to switch
        to (the library carries one arithmetic, or this detector is on the range-free one already)."""
        twin = self.lib.twin() if hasattr(self.lib, "twin") and not getattr(self, "_arith_fixed", False) else None
        if twin is None:
            return False
        import warnings
        warnings.warn("deft_amd: %s -- this detector continues on the three-bf16-piece arithmetic (no range limit)" % why, RuntimeWarning)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._drop_ahead()
        self._ahead, self._plans, self._graphs = {}, {}, {}
        self._launch_next = self._peeked = self._fm_ready = None
        self.lib = twin
        def move(plan):                                # in place: whoever holds the plan object keeps it
            if getattr(plan, "lib", None) is not None and plan.lib is not twin and getattr(plan.lib, "pieces", 3) == 2:
                new = engine.AfePlan(self.sd, plan.max_object, self.device, twin, align_corners=plan.align_corners)
                plan.__dict__.clear()
                plan.__dict__.update(new.__dict__)
        move(self.afe)
        # the trackers' own embedding / affinity plans (model.AFE.plan: integrate.AfeSeam builds one per model) run the same split arithmetic
        trks = list(self.tracker.values()) if isinstance(self.tracker, dict) else ([self.tracker] if self.tracker is not None else [])
        for t in trks:
            plan = getattr(getattr(getattr(t, "model", None), "AFE", None), "plan", None)
            if isinstance(plan, engine.AfePlan):
                move(plan)
        self.arith = "bf16x3"
        return True

    def process(self, images, pre_images=None, pre_hms=None, pre_inds=None, return_time=False):
        """detector.py:530-551 (see _process_once).  A frame whose activations leave the two-fp16-piece range is re-run on the range-free
        arithmetic (_switch_to_safe) instead of raising."""
        try:
            return self._process_once(images, pre_images, pre_hms, pre_inds, return_time)
        except FloatingPointError as e:
            if not self._switch_to_safe(str(e)):
                raise
            return self._process_once(images, pre_images, pre_hms, pre_inds, return_time)

    def _process_once(self, images, pre_images=None, pre_hms=None, pre_inds=None, return_time=False):
        """detector.py:530-551.  images [N,3,H,W] fp32.  Returns (output, dets, FeatureMaps):
        output = {"hm": dense sigmoid'ed map as an NHWC View}, dets = generic_decode's dict as
        numpy (one D2H), FeatureMaps = the 13 NHWC Views consumed by AFE (tracker.py:826)."""
        assert pre_images is None and pre_hms is None, "DEFT inference never passes pre_img/pre_hm (detector.py:153,162)"
        if getattr(self.opt, "flip_test", False):
            return self._process_flip(images, pre_inds, return_time)
        N, _, H, W = images.shape
        plan = self._plan(N, H, W)
        assert plan.input_kind == "fp32", "process() needs the fp32-input plan"
        images = images.to(self.device, non_blocking=True)
        key = plan._gkey
        if not self.hip_graphs or key not in self._graphs:
            plan.forward(images)                    # first frame of a shape: eager (sets kernel attributes, fills caches)
            self._graphs.setdefault(key, None)
        else:
            # one frame per call is launch-bound on the host (~100 launches of 5-50 us): replay the plan's launch
            # list as a hipGraph (buffers are plan-owned and static, so the capture stays valid)
            if self._graphs[key] is None:
                self._graphs[key] = plan.capture_graph()
            plan.image.copy_(images, non_blocking=True)
            engine._Plan.replay_graph(self._graphs[key], self.device)
        d = plan.dets()
        if "dep" in d:      # _sigmoid_output, detector.py:491-493, applied at the K peaks
            d["dep"] = (1.0 / (torch.sigmoid(d["dep"]) + 1e-6) - 1.0) * getattr(self.opt, "depth_scale", 1.0)
        dets = _check_finite(_fetch(_flag_finite(plan, d, self.lib)))
        output = {"hm": plan.dense["hm"], "pre_inds": pre_inds}
        if return_time:
            import time
            return output, dets, time.time(), plan.fmaps
        return output, dets, plan.fmaps

    # heads `_flip_output` (detector.py:496-528) averages over the frame and its mirror image / averages with the odd channels negated; every
    # other head of DEFT's configurations is taken from the un-flipped frame ("single_flips")
    FLIP_AVERAGE = ("hm", "wh", "dep", "dim")
    FLIP_NEG_AVERAGE = ("amodel_offset",)

    def _process_flip(self, images, pre_inds=None, return_time=False):
        """`--flip_test` on the device (detector.py:396-399, 536-540, 496-528): the frame and its mirror image are the two frames of ONE
        plan (images [2,3,H,W] as Detector.pre_process stacks them, or [1,3,H,W]: mirrored here on the device), every head evaluated densely,
        `_sigmoid_output` + `_flip_output` as elementwise device operations on the head maps, then peak NMS / top-K / the K-row gathers of
        `generic_decode` on the device (integrate.generic_decode).  FeatureMaps are the 13 maps of BOTH frames as the reference returns
        them -- its tracker keeps the un-flipped frame (tracker.py:821-825), and so does ArrayTracker."""
        from . import integrate
        images = images.to(self.device, torch.float32)
        if images.shape[0] == 1:
            images = torch.cat([images, images.flip(3)], 0)
        assert images.shape[0] == 2, "flip_test runs one frame (and its mirror image) per call (detector.py:396-399)"
        _, _, H, W = images.shape
        key = ("flip", 2, H, W)
        if key not in self._plans:
            self._plans[key] = engine.DlaSegPlan(self.sd, 2, H, W, self.dataset, K=self.K, device=self.device, lib=self.lib, dense_heads=True)
            self._plans[key]._gkey = key
        plan = self._plans[key]
        if not self.hip_graphs or key not in self._graphs:
            plan.forward(images)
            self._graphs.setdefault(key, None)
        else:
            if self._graphs[key] is None:
                self._graphs[key] = plan.capture_graph()
            plan.image.copy_(images, non_blocking=True)
            engine._Plan.replay_graph(self._graphs[key], self.device)
        out = {}
        for h, v in plan.dense.items():
            y = v.to_nchw()                                              # [2, C, h, w] (a device tensor)
            if h == "hm":
                y = torch.sigmoid(y)                                     # _sigmoid_output, detector.py:488-493
            elif h == "dep":
                y = (1.0 / (torch.sigmoid(y) + 1e-6) - 1.0) * getattr(self.opt, "depth_scale", 1.0)
            if h in self.FLIP_AVERAGE:                                   # detector.py:510-512 (flip_tensor = mirror along x)
                y = (y[0:1] + y[1:2].flip(3)) / 2
            elif h in self.FLIP_NEG_AVERAGE:                             # detector.py:513-516
                f = y[1:2].flip(3).clone()
                f[:, 0::2] *= -1
                y = (y[0:1] + f) / 2
            else:                                                        # detector.py:517-518
                y = y[0:1]
            out[h] = y.contiguous()
        d = integrate.generic_decode(out, K=self.K, opt=self.opt, lib=self.lib)
        dets = _check_finite(_fetch(_flag_finite(plan, d, self.lib)))
        out["pre_inds"] = pre_inds
        if return_time:
            import time
            return out, dets, time.time(), plan.fmaps
        return out, dets, plan.fmaps

    # ---- between process() and Tracker.update(): the reference's post-processing, vectorised (deft_amd/postprocess.py) ----
    def post_process(self, dets, meta, scale=1):
        """detector.py:553-575: network output grid -> original image coordinates (+ depth un-projection for the 3-D heads).
        Returns the reference's list of per-detection dicts."""
        from . import postprocess as PP
        post = PP.generic_post_process(dets, meta["c"], meta["s"], meta["out_height"], meta["out_width"],
                                       getattr(self.opt, "out_thresh", 0.0), calib=meta.get("calib"))
        self.this_calib = meta.get("calib")
        if scale != 1 and "bbox" in post:
            post["bbox"] = post["bbox"] / np.float32(scale)
        self._post = post
        return PP.as_result_list(post)

    def merge_outputs(self, detections):
        """detector.py:577-583 (single test scale)."""
        thr = getattr(self.opt, "out_thresh", 0.0)
        det = detections[0]
        post = det.arrays() if hasattr(det, "arrays") else None
        if post is None:
            return [d for d in det if d["score"] > thr]
        from . import postprocess as PP
        keep = post["score"] > thr                                         # the same rows, decided on the array the dicts were built from
        out = PP.ResultList(det if keep.all() else [d for d, k in zip(det, keep.tolist()) if k])
        out.post = post if keep.all() else {k: v[keep] for k, v in post.items()}
        return out

    def nuscenes_targets(self, results, image_info, nms=True):
        """The nuScenes branch of Detector.run up to the tracker calls (detector.py:200-338): per tracking class the arguments of
        `self.tracker[class_name].update(results, FeatureMaps, ddd_boxes=, depths_by_class=, ddd_org_boxes=, submission=, classe=)`."""
        from . import postprocess as PP
        keys = ("score", "class", "bbox", "dim", "loc", "rot_y")
        arr = results.arrays() if hasattr(results, "arrays") else None
        if arr is not None and all(k in arr for k in keys):                # post_process' own arrays (postprocess.ResultList): no parsing back
            post = {k: arr[k] for k in keys}
        else:
            post = {k: np.stack([np.asarray(r[k]) for r in results]) if results else np.zeros((0,)) for k in keys}
        if not results:
            return {n: {"results": [], "ddd_boxes": [], "depths": [], "ddd_org_boxes": [], "submission": []} for n in PP.NUSCENES_TRACKING_NAMES}
        return PP.nuscenes_frame(post, image_info, nms=nms, lib=getattr(self, "lib", None))

    # ---- frame in -> tracks out: Detector.run (detector.py:112-344) on the fused path ----------------------------------------
    def _meta_for(self, height, width, inp_h, inp_w, input_meta):
        """The meta dict of Detector.pre_process (detector.py:346-415: fix_short / fix_res / keep_res, preprocess.input_geometry), without
        touching the pixels."""
        from . import preprocess as PR
        M, c, s, gh, gw = PR.input_geometry(self.opt, height, width)
        assert (gh, gw) == (inp_h, inp_w)
        calib = np.array(input_meta["calib"], dtype=np.float32) if "calib" in input_meta else \
            np.array([[1200.0, 0, width / 2, 0], [0, 1200.0, height / 2, 0], [0, 0, 1, 0]], np.float32)      # _get_default_calib, detector.py:424-428
        meta = {"calib": calib, "c": c, "s": s, "height": height, "width": width, "out_height": inp_h // 4, "out_width": inp_w // 4,
                "inp_height": inp_h, "inp_width": inp_w, "trans_input": M}
        for k in ("pre_dets", "cur_dets"):
            if k in input_meta:
                meta[k] = input_meta[k]
        return meta

    def run(self, image_or_path_or_tensor, meta={}, image_info=None, nms=True, prefetch=None):
        """detector.py:112-344 (see _run_once).  A frame whose network pass leaves the two-fp16-piece range raises inside the detection stage,
        before the tracker has seen it: the detector moves to the range-free arithmetic (_switch_to_safe: queued lookahead passes are dropped)
        and the frame is run again -- the video goes on.  An overflow found in the TRACKER's stage (embedding / affinity chain: array_tracker
        checks the similarity matrix) also moves the detector over, but that frame's association has begun: the error is passed on."""
        st = {"tracker": False}
        try:
            return self._run_once(image_or_path_or_tensor, meta, image_info, nms, prefetch, st)
        except FloatingPointError as e:
            if not self._switch_to_safe(str(e)) or st["tracker"]:
                raise
            return self._run_once(image_or_path_or_tensor, meta, image_info, nms, prefetch, st)

    def _run_once(self, image_or_path_or_tensor, meta={}, image_info=None, nms=True, prefetch=None, _stage=None):
        """detector.py:112-344 -- same argument forms, same return value (`Tracker.update`'s online targets; the post-processed
        `results` when no tracker is set), same stage timers (self.times: load / pre / net / dec / post / merge / track / tot):
          * numpy uint8 HWC frame (cv2 channel order): warp + normalise + layout ON THE DEVICE (deft_preprocess_u8, fix_res mode),
            straight into the plan's input -- the host never touches the pixels;
          * the prefetch-loader dict of src/test.py:106-112, 213 ({"image", "images": {scale: [tensor]}, "meta": {scale: {...}}});
          * a path: read with cv2 when that is importable (it is not in this image).
        Then process() -> post_process() -> merge_outputs() -> nuScenes branch / `self.tracker.update(results, FeatureMaps)`.
        One test scale (asserted like detector.py:578); `opt.flip_test` runs the frame and its mirror image as one two-frame plan
        (_process_flip) without lookahead.
        prefetch: the NEXT uint8 frame of the stream (same size), if the caller has it already: its network pass is queued on a second
        set of plan buffers BEFORE this frame's post-processing and tracker run, so the GPU works on frame k+1 while the host associates
        frame k (the reference's loop is strictly serial: detector.py:112-344).  The next run() call must pass that same array object,
        unmodified (its pixels are copied when the pass is queued; a different array simply drops the queued pass); results
        are identical to the serial order (tests/parity_checks.check_fused_run_prefetch)."""
        import time
        opt = self.opt
        scales = list(getattr(opt, "test_scales", [1.0]))
        assert len(scales) == 1, "the fused run() is the single-scale configuration (detector.py:578)"
        scale = scales[0]
        t_start = time.time()
        pre_processed, frame = False, None
        self._peek_src, meta_given = None, bool(meta)
        known, self._peeked = getattr(self, "_peeked", None) or [], None       # [(frame, results)]: post-processed during earlier calls (_begin_next)
        if isinstance(image_or_path_or_tensor, np.ndarray):
            frame = image_or_path_or_tensor
        elif isinstance(image_or_path_or_tensor, str):
            try:
                import cv2
            except ImportError as e:
                raise RuntimeError("reading %r needs cv2, which this environment does not have; pass the decoded uint8 frame" % image_or_path_or_tensor) from e
            frame = cv2.imread(image_or_path_or_tensor)
        else:
            pre_processed = True
        t_loaded = time.time()
        if not pre_processed:
            assert frame.dtype == np.uint8 and frame.ndim == 3 and frame.shape[2] == 3
            sh, sw = frame.shape[:2]
            from . import preprocess as PR
            _, _, _, inp_h, inp_w = PR.input_geometry(opt, sh, sw)        # fix_short / fix_res / keep_res (detector.py:346-376)
            meta = self._meta_for(sh, sw, inp_h, inp_w, meta)
            akey = (inp_h, inp_w, sh, sw)
            if getattr(opt, "flip_test", False):
                # detector.py:396-399 mirrors the PRE-PROCESSED network input (not the camera frame: cv2's fixed-point warp is not mirror
                # symmetric): warp + normalise on the device with the one-frame plan's first launch, mirror that tensor, two-frame plan
                if self._ahead_busy():
                    self._drop_ahead()
                plan = self._plan_u8(1, inp_h, inp_w, sh, sw)
                t_pre = time.time()
                plan.image_u8.copy_(torch.from_numpy(np.ascontiguousarray(frame)).unsqueeze(0).to(self.device, non_blocking=True), non_blocking=True)
                plan.ops[0][2]()
                x4 = plan._x4
                img = x4.buf.view(1, inp_h, inp_w, x4.ld)[..., :3].permute(0, 3, 1, 2).contiguous()
                output, dets, t_fwd, fmaps = self._process_flip(img, None, return_time=True)
            elif prefetch is not None or any(sl.frame is frame for slots in self._ahead.values() for sl in slots):
                t_pre = time.time()
                output, dets, t_fwd, fmaps = self._process_ahead(akey, frame, prefetch)
            else:
                if self._ahead_busy():
                    self._drop_ahead()                 # a frame nobody announced, and no announcement with it: the stream left the lookahead
                plan = self._plan_u8(1, inp_h, inp_w, sh, sw)
                src = torch.from_numpy(np.ascontiguousarray(frame)).unsqueeze(0)
                t_pre = time.time()
                output, dets, t_fwd, fmaps = self._process_u8(plan, src)
        else:
            d = image_or_path_or_tensor
            images = d["images"][scale][0]
            meta = {k: v.numpy()[0] for k, v in d["meta"][scale].items()}
            for k in ("pre_dets", "cur_dets"):
                if k in d["meta"]:
                    meta[k] = d["meta"][k]
            t_pre = time.time()
            output, dets, t_fwd, fmaps = self.process(images, None, None, None, return_time=True)
        t_dec = time.time()
        mine = next((r for f, r in known if f is frame), None) if not meta_given else None
        if mine is not None:
            results = mine                                             # post-processed one or two calls ago (_begin_next): the tracker holds THIS list
            t_post = time.time()
        else:
            result = self.post_process(dets, meta, scale)
            t_post = time.time()
            results = self.merge_outputs([result])
        t_merge = time.time()
        if _stage is not None:
            _stage["tracker"] = True                    # from here on the tracker's state moves with the frame
        if getattr(opt, "public_det", False) and pre_processed:
            results = image_or_path_or_tensor["meta"]["cur_dets"]                     # detector.py:190-196
        # the lookahead pass of the NEXT frame (prefetch=): a tracker that says when its device work is over (array_tracker.Tracker2D:
        # after the similarity medians, ~40 % into update()) gets it queued at that point, so the pass overlaps the host-only rest of the
        # association instead of competing with the tracker's own launches; any other tracker gets it queued up front
        nxt, self._launch_next = getattr(self, "_launch_next", None), None
        per_class = self.nuscenes_targets(results, image_info, nms=nms) if self.tracker is not None and self.dataset == "nuscenes" else None
        hook_on = None                                  # the tracker object whose update() will fire the queued pass
        if nxt is not None and self.tracker is not None:
            last = self.tracker[list(per_class)[-1]] if per_class is not None else self.tracker      # nuScenes: the last class's tracker
            at = LOOKAHEAD_AT if LOOKAHEAD_AT != "auto" else ("hook" if self.lookahead_frames > 1 else "early")
            if hasattr(last, "after_device_work") and at == "hook":
                hook_on = last
        if nxt is not None and hook_on is None:
            nxt()
        # a lookahead pass running (or about to be queued) beside us: the tracker's many small launches go to a HIGH-priority stream, so
        # that they are dispatched ahead of the pass's workgroups instead of queueing behind them
        prio = getattr(self, "_trk_stream", None) if (self._ahead_busy() or nxt is not None) else None
        ready, self._fm_ready = getattr(self, "_fm_ready", None), None
        if prio is not None:
            main = torch.cuda.current_stream(self.device)
            if ready is not None:
                # NOT prio.wait_stream(main): the caller's stream is normally the null stream, and HIP multiplexes streams onto a few
                # hardware queues -- when the null stream shares its queue with the pass's stream (or a branch stream of its graph), an
                # event recorded on it completes only when the pass does, and the tracker would wait ~a whole pass for nothing
                # (tools/probe/stall_probe.py, profiles/r4_hw_queue_stall.md).  The feature maps' own event is all the tracker needs.
                prio.wait_event(ready)
            else:
                prio.wait_stream(main)
        with (torch.cuda.stream(prio) if prio is not None else _null()):
            if self.tracker is None:
                targets = results
            elif per_class is not None:                                                    # detector.py:198-338
                targets = []
                trks = [self.tracker[name] for name in per_class]
                if BEGIN_AHEAD and all(hasattr(t, "begin") and hasattr(t, "extract_together") for t in trks):
                    # every class's device half first (ArrayTracker.begin), on ONE embedding extraction for the detections of all classes ...
                    pres = [t.detections_as_arrays(a["results"], a["ddd_boxes"], a["depths"]) for t, a in zip(trks, per_class.values())]
                    feats = type(trks[0]).extract_together(trks, pres, fmaps)
                    for t, a, p, f in zip(trks, per_class.values(), pres, feats):
                        t.begin(a["results"], fmaps, ddd_boxes=a["ddd_boxes"], depths_by_class=a["depths"], pre=p, feats=f)
                for name, a in per_class.items():                                          # ... then the associations, class by class
                    trk = self.tracker[name]
                    if trk is hook_on:
                        trk.after_device_work = nxt
                    targets += trk.update(a["results"], fmaps, ddd_boxes=a["ddd_boxes"], depths_by_class=a["depths"],
                                          ddd_org_boxes=a["ddd_org_boxes"], submission=a["submission"], classe=name)
                if hook_on is not None and hook_on.after_device_work is not None:
                    hook_on.after_device_work = None
                    nxt()
            elif hook_on is not None:
                self.tracker.after_device_work = nxt
                targets = self.tracker.update(results, fmaps)
                if self.tracker.after_device_work is not None:             # (update() returned early)
                    self.tracker.after_device_work = None
                    nxt()
            else:
                targets = self.tracker.update(results, fmaps)                              # detector.py:340-342
            if (BEGIN_AHEAD and self._peek_src and per_class is None and self.tracker is not None and hasattr(self.tracker, "begin")
                    and not meta_given and not getattr(opt, "public_det", False)):
                self._begin_next(meta, scale, prio, known)
        if prio is not None:
            main.wait_stream(prio)
        t_end = time.time()
        self.times = {"load": t_loaded - t_start, "pre": t_pre - t_loaded, "net": t_fwd - t_pre, "dec": t_dec - t_fwd, "post": t_post - t_dec,
                      "merge": t_merge - t_post, "track": t_end - t_merge, "tot": t_end - t_start}
        self.last_results = results
        return targets

    def _process_u8(self, plan, frames_u8):
        """process() for a uint8 frame batch [N, sh, sw, 3] (host or device): the plan's first launch is deft_preprocess_u8."""
        import time
        key = plan._gkey
        assert plan.input_kind == "u8"
        frames_u8 = frames_u8.to(self.device, non_blocking=True)
        if not self.hip_graphs or key not in self._graphs:
            plan.forward_u8(frames_u8)
            self._graphs.setdefault(key, None)
        else:
            if self._graphs[key] is None:
                self._graphs[key] = plan.capture_graph()
            plan.image_u8.copy_(frames_u8, non_blocking=True)
            engine._Plan.replay_graph(self._graphs[key], self.device)
        d = plan.dets()
        if "dep" in d:
            d["dep"] = (1.0 / (torch.sigmoid(d["dep"]) + 1e-6) - 1.0) * getattr(self.opt, "depth_scale", 1.0)
        dets = _check_finite(_fetch(_flag_finite(plan, d, self.lib)))
        return {"hm": plan.dense["hm"], "pre_inds": None}, dets, time.time(), plan.fmaps

    # ---- one frame of lookahead: two sets of plan buffers, frame k+1's network pass beside frame k's host work ---------------------------
    class _Slot:
        """One set of plan buffers of the lookahead: `n` frames per pass (Detector.lookahead_frames)."""

        def __init__(self, det, inp_h, inp_w, sh, sw, n=1):
            self.n = n
            self.plan = engine.DlaSegPlan(det.sd, n, inp_h, inp_w, det.dataset, K=det.K, device=det.device, lib=det.lib)
            self.plan.use_u8_input(sh, sw, minv=det._minv(sh, sw, n))
            cuda = det.device.type == "cuda"
            self.stage = torch.empty(n, sh, sw, 3, dtype=torch.uint8, pin_memory=cuda)
            self.stage_np = self.stage.numpy()
            self.fields = None
            self.graph, self.warm, self.host = None, False, None
            self.frames, self.pos = None, 0              # the frame arrays of the pass in flight / being handed out, and how many were consumed
            self.done = torch.cuda.Event() if cuda else None

        @property
        def frame(self):                                 # the frame the next run() call may claim (None: the slot is free)
            return self.frames[self.pos] if self.frames is not None else None

        def __del__(self):                       # a queued pass (announced, never consumed) must not outlive its graph and buffers
            try:
                if self.done is not None and self.frames is not None:
                    self.done.synchronize()
            except Exception:
                pass

    def _launch_ahead(self, sl, frames):
        """Queue frames -> detections on slot `sl` (its own stream): staging copy, H2D, the plan (hipGraph from the second use), one D2H
        of every decoded field into pinned memory, an event.  Returns without waiting.  frames: 1 .. sl.n uint8 arrays (a pass with
        fewer frames than the plan holds repeats the last one: its results are never handed out)."""
        p = sl.plan
        cuda = self.device.type == "cuda"
        frames = list(frames)
        srcs = []
        for i in range(sl.n):
            f = frames[min(i, len(frames) - 1)]
            src = torch.from_numpy(f) if f.flags["C_CONTIGUOUS"] else None
            if cuda and src is not None and src.is_pinned():
                srcs.append(src)                           # the frame already lives in pinned host memory (a decoder's output buffer): no staging copy
            else:
                np.copyto(sl.stage_np[i], f)               # (a plain memcpy: torch's multi-threaded CPU copy_ stalls for milliseconds next to the tracker's BLAS threads)
                srcs.append(sl.stage[i])
        if cuda:
            if not hasattr(self, "_net_stream"):
                self._net_stream = torch.cuda.Stream(device=self.device)
                self._trk_done = torch.cuda.Event()
                # the tracker's stream: never the null stream (see run()); high priority = its own hardware-queue pool, and its many
                # small launches are dispatched ahead of the pass's workgroups
                self._trk_stream = torch.cuda.Stream(device=self.device, priority=-1 if PRIORITY_TRACKER else 0)
            main = torch.cuda.current_stream(self.device)
            self._trk_done.record(main)
            self._net_stream.wait_event(self._trk_done)            # whatever read this slot's feature maps (two passes ago) is finished
            if self.hip_graphs and sl.warm and sl.graph is None:
                sl.graph = p.capture_graph()
        ctx = torch.cuda.stream(self._net_stream) if cuda else _null()
        with ctx:
            for i, src in enumerate(srcs):
                p.image_u8[i].copy_(src, non_blocking=True)
            if sl.graph is not None:
                engine._Plan.replay_graph(sl.graph, self.device)
            else:
                p.run()
                sl.warm = True
            d = _flag_finite(p, p.dets(), self.lib)
            if "dep" in d:
                d["dep"] = (1.0 / (torch.sigmoid(d["dep"]) + 1e-6) - 1.0) * getattr(self.opt, "depth_scale", 1.0)
            # every decoded field in ONE contiguous record (float64 holds the int64 indices exactly), ONE D2H into pinned memory
            flat = torch.cat([v.detach().reshape(-1).double() for v in d.values()])
            if sl.host is None:
                sl.host = torch.empty(flat.shape, dtype=torch.float64, pin_memory=cuda)
                sl.fields = [(k, tuple(v.shape), v.numel(), _NP[v.dtype]) for k, v in d.items()]
            sl.host.copy_(flat, non_blocking=True)
            if cuda:
                sl.done.record(self._net_stream)
        sl.frames, sl.pos = frames, 0

    def _drop_ahead(self):
        """Forget every lookahead pass that was announced and never claimed (the caller moved on to another frame, another video or the
        serial form of run()): let it finish -- its graph writes into the slot's buffers -- then release the slot and its frames."""
        for slots in self._ahead.values():
            for sl in slots:
                if sl.frames is not None and sl.done is not None:
                    sl.done.synchronize()
                sl.frames = None
        self._launch_next = None

    def _ahead_busy(self):
        return any(sl.frames is not None for slots in self._ahead.values() for sl in slots)

    lookahead_frames = 1           # n > 1: run(prefetch=[the next 2n-1 frames]) runs n frames per lookahead pass (an n-frame plan: the batch-1
    #                                launch list is latency-bound, n frames cost less than n passes), queued up to 2n-1 frames ahead; the
    #                                results are handed out one per call.  For recorded streams (test.py reads files); a live camera pays
    #                                n-1 frame periods of latency for it

    @staticmethod
    def _slot_dets(sl, j):
        """The decoded fields of frame j of slot sl's finished pass, from its pinned host record (rewritten two passes on: copies)."""
        rec, dets, o = sl.host.numpy(), {}, 0
        for k, shape, cnt, dt in sl.fields:
            a = rec[o:o + cnt].astype(dt).reshape(shape)
            dets[k] = a[j:j + 1] if shape and shape[0] == sl.n and k != "_finite" else a
            o += cnt
        return dets

    def _begin_next(self, meta, scale, prio, known=()):
        """The detections of the NEXT frame(s) are already on the host (later frames of the pass this frame came from, or the first frames of the
        other slot's pass when that has finished): post-process them now and let the tracker queue device work for them behind this frame's
        update() -- the whole device half of the next frame (ArrayTracker.begin), and the embeddings + affinity blocks of the frame after it
        (ArrayTracker.prepare: they do not need the track table, and the pair MLP is the longest launch of a tracked frame).  The next run()
        call finds its embedding / affinity / similarity round trip finished instead of waiting for it.  known: [(frame, results)] post-processed
        by an earlier call -- the tracker was handed THOSE lists."""
        done = []
        for i, (sl, j) in enumerate(self._peek_src):
            frame = sl.frames[j]
            results = next((r for f, r in known if f is frame), None)
            if results is None:
                try:
                    dets = _check_finite(self._slot_dets(sl, j))
                except FloatingPointError:
                    break                                              # the run() call of that frame raises it
                results = self.merge_outputs([self.post_process(dets, meta, scale)])
            fmaps = sl.plan.fmaps if sl.n == 1 else [fm[j] for fm in sl.plan.fmaps]
            if sl.done is not None:
                (prio if prio is not None else torch.cuda.current_stream(self.device)).wait_event(sl.done)
            if i == 0:
                self.tracker.begin(results, fmaps)
            else:
                self.tracker.prepare(results, fmaps)
            done.append((frame, results))
        self._peeked = done

    def _process_ahead(self, akey, frame, prefetch):
        import time
        n = int(self.lookahead_frames)
        upcoming = list(prefetch) if isinstance(prefetch, (list, tuple)) else ([] if prefetch is None else [prefetch])
        # the pass that already holds this frame -- in whichever slot pair (the caller may have changed the frame size or
        # lookahead_frames since it was announced) -- else a fresh one
        slots = next((sl_ for sl_ in self._ahead.values() if any(sl.frame is frame for sl in sl_)), None)
        cur = None if slots is None else next(sl for sl in slots if sl.frame is frame)
        if cur is None:                                                # not announced by an earlier call: launch it now (with what is known to follow)
            self._drop_ahead()                                         # announced passes that were never claimed are dropped, everywhere
            key = akey + (n,)
            if key not in self._ahead:
                self._ahead[key] = [Detector._Slot(self, *akey, n=n), Detector._Slot(self, *akey, n=n)]
            slots = self._ahead[key]
            cur = slots[self._ahead_turn]
            self._launch_ahead(cur, [frame] + upcoming[:n - 1])
        n = cur.n
        other = slots[1 - slots.index(cur)]
        self._ahead_turn = slots.index(other)
        if cur.done is not None:
            cur.done.synchronize()
        j = cur.pos
        dets = _check_finite(self._slot_dets(cur, j))
        fmaps = cur.plan.fmaps if cur.n == 1 else [fm[j] for fm in cur.plan.fmaps]
        hm = cur.plan.dense["hm"] if cur.n == 1 else cur.plan.dense["hm"][j]
        cur.pos += 1
        left = len(cur.frames) - cur.pos                               # frames of this pass still to be handed out
        if left == 0:
            cur.frames = None
        if cur.done is not None:
            torch.cuda.current_stream(self.device).wait_event(cur.done)    # the tracker's launches read this slot's feature maps
        self._fm_ready = cur.done
        # whose detections the host holds next: the pass of this frame, or the other slot's when it has finished
        nxt2 = [(cur, q) for q in range(cur.pos, len(cur.frames))] if cur.frames is not None else []
        if len(nxt2) < 2 and other.frames is not None and (other.done is None or other.done.query()):
            nxt2 += [(other, q) for q in range(other.pos, len(other.frames))]
        self._peek_src = nxt2[:2 if PREPARE_AHEAD and hasattr(self.tracker, "prepare") else 1]
        t_fwd = time.time()
        self._launch_next = None
        if other.frames is None and len(upcoming) > left:              # the free slot takes the frames behind the ones this slot still holds
            nxt = upcoming[left:left + n]
            for f in nxt:
                assert f.dtype == np.uint8 and f.shape == frame.shape, "prefetch: the next frames of the same stream"
            if len(nxt) == n or left == 0:                             # (a short pass only when nothing else is in flight: the stream is ending)
                self._launch_next = lambda: self._launch_ahead(other, nxt)      # run() decides when: see there
        return {"hm": hm, "pre_inds": None}, dets, t_fwd, fmaps

    def track_stream(self, frames, image_infos=None, frames_per_pass=None, nms=True):
        """The per-video loop of src/test.py:100-150 (`for ind, (img_id, pre_processed_images) in enumerate(data_loader): ret =
        detector.run(...)`) over a recorded stream: yields run()'s return value (the tracks / detections of one frame) per frame, in
        order.  `frames`: any iterable of uint8 HWC arrays (a decoder, a list); it is read at most 2n-1 frames ahead, n =
        frames_per_pass (default: self.lookahead_frames), and those frames ride in the lookahead passes (n per pass) while the host
        associates -- the caller does not build prefetch lists.  Every frame must be its own array and stay untouched until its tracks
        have been yielded (a decoder that recycles ONE buffer needs a copy per frame); pinned host memory skips the staging copy.
        image_infos: per-frame `image_info` records (nuScenes), same order."""
        import collections
        import itertools
        if frames_per_pass is not None:
            self.lookahead_frames = int(frames_per_pass)
        n = int(self.lookahead_frames)
        it = iter(frames)
        infos = iter(image_infos) if image_infos is not None else itertools.repeat(None)
        window = collections.deque(itertools.islice(it, 2 * n))        # the frame of this call + the 2n-1 behind it
        while window:
            frame = window.popleft()
            ahead = list(window)
            yield self.run(frame, image_info=next(infos), prefetch=(ahead if n > 1 else ahead[0]) if ahead else None, nms=nms)
            nxt = next(it, None)
            if nxt is not None:
                window.append(nxt)

    def reset_tracking(self, opt):
        """detector.py:677-686: a new video -- fresh Tracker(s) built with the current img_height / img_width (tracks, recorder and
        frame counter must not survive into the next sequence), no previous image, and no lookahead pass left over from the last video."""
        self._drop_ahead()
        if self.tracker is not None:
            for t in (self.tracker.values() if isinstance(self.tracker, dict) else (self.tracker,)):
                if hasattr(t, "close"):
                    t.close()                       # the old trackers' MotionBank slots go back to the (shared) bank
            fac = getattr(self, "_tracker_factory", None)
            if fac is not None:
                self.tracker = fac(opt, self.img_height, self.img_width)
            elif isinstance(self.tracker, dict):
                self.tracker = {k: type(t)(opt, t.model, h=self.img_height, w=self.img_width) for k, t in self.tracker.items()}
            else:
                self.tracker = type(self.tracker)(opt, self.tracker.model, h=self.img_height, w=self.img_width)
        self.pre_images = None
