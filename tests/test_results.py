"""-m "not gpu": deft_amd.results -- the result writers of the reference's evaluation loop (src/test.py:213-310, 322-342; SURVEY.md 8(f)
rank 4) -- against text written by the REFERENCE's own `write_results` (tests/golden/result_writers.npz, oracle/make_golden.py
run_result_writers) and, for the nuScenes submission records, against the loop body of test.py restated field by field."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

from deft_amd import results as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _frames():
    fx = np.load(os.path.join(GOLD, "result_writers.npz"))
    out = []
    for k in range(int(fx["nframes"])):
        tl = [row.astype(np.float32) if f32 else row for row, f32 in zip(fx["f%d_tlwh" % k], fx["f%d_f32" % k])]
        out.append((int(fx["f%d_frame" % k]), tl, [int(v) for v in fx["f%d_ids" % k]]))
    return fx, out


@pytest.mark.parametrize("data_type", ["mot", "kitti_tracking"])
def test_write_results_matches_reference_text(tmp_path, data_type):
    fx, frames = _frames()
    p = tmp_path / "out.txt"
    R.write_results(str(p), frames, data_type)
    assert p.read_text() == str(fx["text_" + data_type])
    assert len(p.read_text().splitlines()) == sum(1 for _, _, ids in frames for i in ids if i >= 0) > 5
    with pytest.raises(ValueError):
        R.write_results(str(p), frames, "nuscenes")


def test_frame_record_applies_min_box_area():
    targets = [SimpleNamespace(tlwh=np.array([1.0, 2.0, 4.0, 5.0]), track_id=3),            # area 20: NOT > min_box_area
               SimpleNamespace(tlwh=np.array([1.0, 2.0, 4.0, 5.1]), track_id=4)]
    frame, tlwhs, ids = R.frame_record(7, targets)
    assert frame == 7 and ids == [4] and np.array_equal(tlwhs[0], targets[1].tlwh)


def test_nuscenes_records_and_cap(tmp_path):
    g = np.random.RandomState(0)

    def target(k, name):
        return SimpleNamespace(tlwh=np.array([10.0, 10.0, 30.0, 40.0]) if k != 2 else np.array([0.0, 0.0, 2.0, 3.0]), track_id=100 + k, classe=name,
                               score=np.float64(0.3 + 0.001 * k), ddd_submission=g.rand(10), ddd_bbox=g.rand(7), org_ddd_box=g.rand(7))
    res = R.NuScenesResults()
    t_cam0 = [target(k, n) for k, n in enumerate(["car", "pedestrian", "bus", "bicycle"])]
    rec = res.add(t_cam0, "tok", sensor_id=1)
    assert [r["tracking_id"] for r in rec] == [100, 101, 103]                                 # the tiny box is dropped (test.py:220)
    assert [r["attribute_name"] for r in rec] == ["vehicle.moving", "pedestrian.moving", "cycle.with_rider"]        # test.py:228-233, nuscenes_att == 0
    r0 = rec[0]
    sub = t_cam0[0].ddd_submission.tolist()
    assert r0["translation"] == sub[:3] and r0["size"] == sub[3:6] and r0["rotation"] == sub[6:] and r0["velocity"] == [0, 0]
    assert r0["detection_name"] == r0["tracking_name"] == "car" and r0["detection_score"] == r0["tracking_score"] == t_cam0[0].score
    assert r0["sensor_id"] == 1 and r0["det_id"] == -1 and r0["sample_token"] == "tok"
    many = [target(10 + k, "car") for k in range(600)]
    for k, t in enumerate(many):
        t.score = np.float64(k / 1000.0)
    res.add(many, "tok", sensor_id=2)                                                         # a second camera of the same sample
    out = res.finalize()
    kept = out["results"]["tok"]
    assert len(kept) == 500 and kept[0]["detection_score"] == max(r["detection_score"] for r in kept)     # test.py:297-308
    assert all(a["detection_score"] >= b["detection_score"] for a, b in zip(kept, kept[1:]))
    res.dump(str(tmp_path / "results.json"))
    back = json.load(open(tmp_path / "results.json"))
    assert back["meta"]["use_camera"] is True and len(back["results"]["tok"]) == 500
