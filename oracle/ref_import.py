"""TEST INFRASTRUCTURE ONLY (oracle side).

Imports the reference's own numerical modules from /root/reference/src/lib so the
oracle restatement in deft_oracle.py can be pinned against them.  Works only in
the build container (the GPU box has no /root/reference); nothing in `-m gpu`
tests, smoke() or bench.py may import this file.

Stubs follow SURVEY.md §8(c): AFE.py:10 imports cv2 (viz only), dla.py:25-29
imports `dcn_v2.DCN` (third-party, un-vendored), kalman_filter_lstm.py:4 imports
model.model which pulls torchvision.
"""
import sys
import types
from types import SimpleNamespace

REF_LIB = "/root/reference/src/lib"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs(dcn_cls):
    """dcn_cls: the nn.Module class to expose as `dcn_v2.DCN`."""
    if REF_LIB not in sys.path:
        sys.path.insert(0, REF_LIB)
    if "cv2" not in sys.modules:
        _stub("cv2")
    _stub("dcn_v2", DCN=dcn_cls)
    for missing in ("lap", "cython_bbox"):       # utils/matching.py:1,4 (host association, not on the kernel path)
        if missing not in sys.modules:
            _stub(missing, bbox_overlaps=None, lapjv=None)
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tvm = _stub("torchvision.models")
        tvu = _stub("torchvision.models.utils", load_state_dict_from_url=lambda *a, **k: {})
        tv.models = tvm
        tvm.utils = tvu


def make_opt(dataset="mot", max_object=100):
    return SimpleNamespace(
        dataset=dataset, max_object=max_object, dla_node="dcn", load_model="x",
        pre_img=False, pre_hm=False, head_kernel=3, prior_bias=-4.6,
        model_output_list=False, zero_tracking=False, load_model_traj="",
    )


HEADS = {
    "mot": {"hm": 1, "reg": 2, "wh": 2, "tracking": 2, "ltrb_amodal": 4},
    "kitti_tracking": {"hm": 3, "reg": 2, "wh": 2, "tracking": 2},
    "nuscenes": {"hm": 10, "reg": 2, "wh": 2, "tracking": 2, "dep": 1, "rot": 8,
                 "dim": 3, "amodel_offset": 2},
}


def build_reference_model(dataset, dcn_cls, max_object=100):
    install_stubs(dcn_cls)
    from model.networks.dla import DLASeg  # noqa
    opt = make_opt(dataset, max_object)
    heads = HEADS[dataset]
    head_convs = {h: [256] for h in heads}
    model = DLASeg(34, heads, head_convs, opt)
    model.eval()
    return model, opt
