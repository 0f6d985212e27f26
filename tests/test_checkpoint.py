"""-m "not gpu": deft_amd.checkpoint.load_model_state against the rules of the reference's load_model (model/model.py:40-90)
-- and, in the build container, against that function itself run on the reference's own DLASeg."""
import os
from types import SimpleNamespace

import pytest
import torch

from deft_amd import checkpoint, synth

HAVE_REF = os.path.isdir("/root/reference/src/lib")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _opt(**kw):
    d = dict(dataset="mot", reset_hm=False, reuse_hm=False, prior_bias=-4.6, heads=None, resume=False)
    d.update(kw)
    return SimpleNamespace(**d)


def _ckpt(dataset="mot", **changes):
    sd = synth.synth_state_dict(dataset)
    sd.update(changes)
    return sd


def test_module_prefix_and_bare_state_dict(tmp_path):
    sd = _ckpt()
    wrapped = {"epoch": 7, "state_dict": {("module." + k): v for k, v in sd.items()}}
    wrapped["state_dict"]["module_list.0.weight"] = torch.zeros(3)           # model.py:51: NOT stripped -> unknown -> dropped
    path = str(tmp_path / "ck.pth")
    torch.save(wrapped, path)
    logs = []
    got = checkpoint.load_model_state(path, _opt(), log=logs.append)
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert "Drop parameter module_list.0.weight." in logs and any(l.startswith("loaded") and "epoch 7" in l for l in logs)
    got2 = checkpoint.load_model_state(sd, _opt(), log=logs.append)          # a bare state dict
    assert all(torch.equal(got2[k], sd[k]) for k in sd)


def test_hm_head_with_80_classes_skip_and_reuse():
    """A COCO-pretrained checkpoint (80-class heat map) loaded into the 1-class MOT model."""
    g = torch.Generator().manual_seed(0)
    w80, b80 = torch.randn(80, 256, 1, 1, generator=g), torch.randn(80, generator=g)
    sd = _ckpt(**{"hm.2.weight": w80, "hm.2.bias": b80})
    logs = []
    got = checkpoint.load_model_state(sd, _opt(), log=logs.append)                       # model.py:77-85: skipped, the model's init stays
    assert got["hm.2.weight"].shape == (1, 256, 1, 1) and float(got["hm.2.weight"].abs().max()) == 0.0
    assert torch.equal(got["hm.2.bias"], torch.full((1,), -4.6))
    assert any(l.startswith("Skip loading parameter hm.2.weight") for l in logs)
    got = checkpoint.load_model_state(sd, _opt(reuse_hm=True), log=logs.append)          # :63-76: the first rows are re-used
    assert torch.equal(got["hm.2.weight"], w80[:1]) and torch.equal(got["hm.2.bias"], b80[:1])
    # reset_hm: an hm parameter with 80 (or 1) rows is treated as mismatched even when the shapes agree (:58-62)
    sd1 = _ckpt()
    got = checkpoint.load_model_state(sd1, _opt(reset_hm=True), log=logs.append)
    assert torch.equal(got["hm.2.bias"], torch.full((1,), -4.6)) and float(got["hm.2.weight"].abs().max()) == 0.0
    assert torch.equal(got["hm.0.weight"], sd1["hm.0.weight"])                            # [256, 64, 3, 3]: 256 rows, shapes agree -> loaded
    # fewer rows than the model wants: the reference's load_state_dict raises -- so does this
    sd_small = _ckpt("kitti_tracking", **{"hm.2.weight": torch.zeros(1, 256, 1, 1), "hm.2.bias": torch.zeros(1)})
    with pytest.raises(RuntimeError):
        checkpoint.load_model_state(sd_small, _opt(dataset="kitti_tracking", reuse_hm=True), log=logs.append)


def test_missing_keys_keep_the_initial_value():
    sd = _ckpt()
    for k in ("ltrb_amodal.2.weight", "ltrb_amodal.2.bias", "base.level2.tree1.bn1.running_var", "ida_up.up_1.weight"):
        del sd[k]
    logs = []
    got = checkpoint.load_model_state(sd, _opt(), log=logs.append)
    assert float(got["ltrb_amodal.2.bias"].abs().max()) == 0.0                            # fill_fc_weights (base_model.py:16-20)
    assert torch.equal(got["base.level2.tree1.bn1.running_var"], torch.ones(64))
    assert torch.equal(got["ida_up.up_1.weight"], synth._up_weight((64, 1, 4, 4)))        # dla.py:565-573
    assert sum(l.startswith("No param") for l in logs) == 4


def test_custom_heads_change_the_template():
    heads = {"hm": 2, "reg": 2, "wh": 2}
    t = checkpoint.model_template(_opt(heads=heads))
    assert t["hm.2.weight"].shape == (2, 256, 1, 1) and "tracking.0.weight" not in t and torch.equal(t["hm.2.bias"], torch.full((2,), -4.6))


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference only exists in the build container")
def test_against_the_reference_load_model(tmp_path):
    """The reference's load_model on its own DLASeg: which parameters are loaded, skipped, re-used, dropped or kept is the same, and
    every value that does not come from the reference's RANDOM initialisation is equal."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import make_golden as MG
    import ref_import
    model, ropt = ref_import.build_reference_model("mot", MG.OracleDCN)
    from model.model import load_model
    g = torch.Generator().manual_seed(1)
    sd = _ckpt(**{"hm.2.weight": torch.randn(80, 256, 1, 1, generator=g), "hm.2.bias": torch.randn(80, generator=g),
                  "wh.2.weight": torch.randn(4, 256, 1, 1, generator=g)})                 # a second mismatch outside hm
    del sd["reg.2.bias"]
    sd["not_a_parameter.weight"] = torch.zeros(2)
    path = str(tmp_path / "ck.pth")
    torch.save({"epoch": 1, "state_dict": {"module." + k: v for k, v in sd.items()}}, path)
    for reuse in (False, True):
        init = {k: v.clone() for k, v in model.state_dict().items()}
        ropt.reset_hm, ropt.reuse_hm, ropt.resume = False, reuse, False
        ref = load_model(model, path, ropt).state_dict()
        got = checkpoint.load_model_state(path, _opt(reuse_hm=reuse), log=lambda *_: None)
        assert set(got) == set(ref)
        for k in ref:
            kept_random = torch.equal(ref[k], init[k]) and k in ("hm.2.weight", "wh.2.weight", "wh.2.bias") and not torch.equal(got[k].float(), ref[k].float())
            if kept_random:          # the reference kept its random initial weights; this loader's deterministic stand-in is zero
                assert float(got[k].abs().max()) == 0.0, k
                continue
            assert torch.equal(got[k].to(ref[k].dtype), ref[k]), (k, reuse)
        model.load_state_dict(init)


def test_wrong_architecture_fails_loudly():
    """a checkpoint / opt of another architecture must not load as the all-initial dla_34 template with a log line per key."""
    import warnings
    from types import SimpleNamespace
    from deft_amd import checkpoint as CK
    with pytest.raises(ValueError):
        CK.model_template(SimpleNamespace(dataset="mot", arch="res_18"))
    with pytest.raises(ValueError):
        CK.model_template(SimpleNamespace(dataset="mot", arch="dla_34", head_conv=64))
    with pytest.raises(ValueError):
        CK.model_template(SimpleNamespace(dataset="mot", arch="dla_34", head_conv={"hm": [256, 256]}))
    CK.model_template(SimpleNamespace(dataset="mot", arch="dla_34", head_conv={"hm": [256], "reg": [256]}))
    foreign = {"layer%d.conv.weight" % i: torch.zeros(4, 4, 3, 3) for i in range(30)}
    with pytest.raises(ValueError, match="wrong architecture"):
        CK.load_model_state({"state_dict": foreign}, SimpleNamespace(dataset="mot"), log=lambda *_: None)
    # a checkpoint that lacks a conv layer: loads (model.py:88-91), but says so as a warning, not only as a log line
    tpl = CK.model_template(SimpleNamespace(dataset="mot"))
    part = {k: v for k, v in tpl.items() if k != "base.level2.tree1.conv1.weight"}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        CK.load_model_state(part, SimpleNamespace(dataset="mot"), log=lambda *_: None)
    assert any("left at ZERO" in str(x.message) for x in w)


def test_validate_checkpoint_tool_reports_key_coverage(tmp_path):
    """tools/validate_checkpoint.py: the one-command check for the day a real DEFT checkpoint exists -- here on a
    fabricated checkpoint in the reference's format (DataParallel prefixes, an epoch, one key missing, one foreign key)."""
    import subprocess
    import sys
    from deft_amd import synth
    sd = synth.synth_state_dict("mot")
    ck = {"epoch": 70, "state_dict": {"module." + k: v for k, v in sd.items() if k != "wh.2.bias"}}
    ck["state_dict"]["module.some_other.weight"] = torch.zeros(3)
    p = str(tmp_path / "ck.pth")
    torch.save(ck, p)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "validate_checkpoint.py"), p], capture_output=True, text=True,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert "epoch 70" in r.stdout and "missing: 1" in r.stdout and "No param wh.2.bias." in r.stdout and "Drop parameter some_other.weight." in r.stdout
    assert "taken from the checkpoint: 461" in r.stdout and r.returncode == 1          # a missing parameter is not OK
    torch.save({"epoch": 1, "state_dict": dict(sd)}, p)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "validate_checkpoint.py"), p], capture_output=True, text=True,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert "missing: 0" in r.stdout and r.returncode == 0, r.stdout + r.stderr
