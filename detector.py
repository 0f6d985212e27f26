"""`from detector import Detector` (the reference's src/test.py:19) with this repository AHEAD of the reference's `src/lib`
on PYTHONPATH: the reference's own `Detector` -- `pre_process`, `run`, `post_process`, `merge_outputs`,
`reset_tracking`, the `img_height` / `img_width` attributes, all unchanged and inherited -- with the hot path behind it
replaced:

  * the model is `deft_amd.integrate.create_model` (HIP kernels) instead of `create_model` + `load_model`
    (detector.py:78-83); `model.AFE` carries the two tracker-facing methods (tracker.py:776, 826, 87);
  * `process()` is the fused launch list of `deft_amd.detector.Detector` (sigmoid, peak NMS, top-K and the regression heads
    at the K peaks on the device, ONE device->host copy); with `--flip_test` the frame and its mirror image run as one two-frame
    plan and `_flip_output` (detector.py:496-528) is applied on the device (deft_amd.detector.Detector._process_flip);
  * the tracker's per-frame forms are bound (`deft_amd.tracker.accelerate`: one affinity chain per frame, device-side
    similarity medians, vectorised gating / assignment, and with `--lstm` one motion-update launch per frame).

Nothing in src/test.py changes.  The reference module is the next `detector.py` on sys.path; it is loaded under the name
`deft_reference_detector`."""
import importlib.util
import os
import sys

import torch


def _load_reference_detector():
    here = os.path.realpath(__file__)
    for p in sys.path:
        cand = os.path.join(p or ".", "detector.py")
        if os.path.isfile(cand) and os.path.realpath(cand) != here:
            spec = importlib.util.spec_from_file_location("deft_reference_detector", cand)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[spec.name] = mod
            spec.loader.exec_module(mod)
            return mod
    raise ImportError("detector.py of the reference (its src/lib) must be on sys.path behind this repository")


_ref = _load_reference_detector()
globals().update({k: v for k, v in vars(_ref).items() if not k.startswith("__")})      # the module's other public names


class Detector(_ref.Detector):
    def __init__(self, opt):
        from deft_amd import integrate, tracker as DT
        from deft_amd.detector import Detector as FusedDetector
        import utils.tracker as RT                                   # the reference's tracker module
        from deft_amd import checkpoint
        sd = checkpoint.load_model_state(opt.load_model, opt)         # model.py:40-90: module. prefixes, shape mismatches, reset_hm / reuse_hm, missing keys
        dev = "cuda" if opt.gpus[0] >= 0 else "cpu"
        kf = integrate.KalmanFilterLSTM(opt) if getattr(opt, "lstm", False) else None
        self._undo_tracker = DT.accelerate(RT, kf)                    # before the Trackers of __init__ / reset_tracking are built
        saved = (_ref.create_model, _ref.load_model)
        _ref.create_model = lambda arch, heads, head_conv, opt=None: integrate.create_model(opt, sd, device=dev)
        _ref.load_model = lambda model, path, opt=None: model
        try:
            super().__init__(opt)                                     # the reference's own constructor, our model
        finally:
            _ref.create_model, _ref.load_model = saved
        self._fused = FusedDetector(opt, sd)

    def process(self, images, pre_images=None, pre_hms=None, pre_inds=None, return_time=False):
        if pre_images is not None or pre_hms is not None:              # (never at inference: detector.py:153, 162)
            return super().process(images, pre_images, pre_hms, pre_inds, return_time)
        return self._fused.process(images, None, None, pre_inds, return_time)
