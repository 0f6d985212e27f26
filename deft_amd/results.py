"""Result writers of the reference's evaluation loop (src/test.py:213-310, 322-342) for the tracks this package returns -- what turns
`Detector.run`'s targets into the files the MOT17 / KITTI / nuScenes evaluators read.  Same selection rule (`tlwh[2] * tlwh[3] >
min_box_area`), same text formats character for character (`write_results`), same nuScenes submission records, same per-sample cap
(the 500 best by detection score) -- checked against the reference's own `write_results` and its loop body by
tests/test_results.py (fixture written from /root/reference by oracle/make_golden.py)."""
import json

import numpy as np

min_box_area = 20                                   # test.py:24

_vehicles = ["car", "truck", "bus", "trailer", "construction_vehicle"]        # test.py:26-40
_cycles = ["motorcycle", "bicycle"]
_pedestrians = ["pedestrian"]
attribute_to_id = {"": 0, "cycle.with_rider": 1, "cycle.without_rider": 2, "pedestrian.moving": 3, "pedestrian.standing": 4,
                   "pedestrian.sitting_lying_down": 5, "vehicle.moving": 6, "vehicle.parked": 7, "vehicle.stopped": 8}
id_to_attribute = {v: k for k, v in attribute_to_id.items()}
nuscenes_att = np.zeros(8, np.float32)              # test.py:41 (never updated by the reference: the arg-max below is always index 0)


def frame_record(frame_id, online_targets):
    """test.py:214-222, 270-272: (frame_id, tlwhs, ids) of the targets whose box area exceeds min_box_area."""
    tlwhs, ids = [], []
    for t in online_targets:
        tlwh = t.tlwh
        if tlwh[2] * tlwh[3] > min_box_area:
            tlwhs.append(tlwh)
            ids.append(t.track_id)
    return (frame_id, tlwhs, ids)


def write_results(filename, results, data_type):
    """test.py:322-342: MOT challenge / KITTI tracking text files from [(frame_id, tlwhs, track_ids), ...]."""
    if data_type == "mot":
        save_format = "{frame},{id},{x1},{y1},{w},{h},1,-1,-1,-1\n"
    elif data_type == "kitti_tracking":
        save_format = "{frame} {id} Car 0 0 -10 {x1} {y1} {x2} {y2} -10 -10 -10 -1000 -1000 -1000 -10\n"
    else:
        raise ValueError(data_type)
    with open(filename, "w") as f:
        for frame_id, tlwhs, track_ids in results:
            if data_type == "kitti_tracking":
                frame_id -= 1
            for tlwh, track_id in zip(tlwhs, track_ids):
                if track_id < 0:
                    continue
                x1, y1, w, h = tlwh
                x2, y2 = x1 + w, y1 + h
                f.write(save_format.format(frame=frame_id, id=track_id, x1=x1, y1=y1, x2=x2, y2=y2, w=w, h=h))


def nuscenes_sample_results(online_targets, sample_token, sensor_id):
    """test.py:224-262: the submission records of one camera frame's targets (nuScenes tracking task format)."""
    out = []
    for t in online_targets:
        tlwh = t.tlwh
        if not tlwh[2] * tlwh[3] > min_box_area:
            continue
        name = t.classe
        if name in _cycles:
            att = id_to_attribute[int(np.argmax(nuscenes_att[0:2])) + 1]
        elif name in _pedestrians:
            att = id_to_attribute[int(np.argmax(nuscenes_att[2:5])) + 3]
        elif name in _vehicles:
            att = id_to_attribute[int(np.argmax(nuscenes_att[5:8])) + 6]
        else:
            raise KeyError(name)                    # (the reference would reuse the previous target's attribute, or fail on the first)
        sub = np.asarray(t.ddd_submission).tolist()
        out.append({"sample_token": sample_token, "translation": sub[:3], "size": sub[3:6], "rotation": sub[6:], "velocity": [0, 0],
                    "detection_name": name, "attribute_name": att, "detection_score": t.score, "tracking_name": name,
                    "tracking_score": t.score, "tracking_id": t.track_id, "sensor_id": sensor_id, "det_id": -1})
    return out


class NuScenesResults(object):
    """The `ret` dictionary of test.py:123-133, 264-268, 297-310: per-sample record lists merged over the six cameras, capped at the 500
    best by detection score, dumped as results.json."""

    def __init__(self):
        self.ret = {"meta": {"use_camera": True, "use_lidar": False, "use_radar": False, "use_map": False, "use_external": False}, "results": {}}

    def add(self, online_targets, sample_token, sensor_id):
        rec = nuscenes_sample_results(online_targets, sample_token, sensor_id)
        res = self.ret["results"]
        res[sample_token] = res[sample_token] + rec if sample_token in res else rec
        return rec

    def finalize(self):
        res = self.ret["results"]
        for token in res.keys():
            confs = sorted([(-d["detection_score"], ind) for ind, d in enumerate(res[token])])
            res[token] = [res[token][ind] for _, ind in confs[:min(500, len(confs))]]
        return self.ret

    def dump(self, path):
        def plain(v):
            if isinstance(v, (np.floating, np.integer)):
                return v.item()
            if isinstance(v, np.ndarray):
                return v.tolist()
            raise TypeError(type(v))
        with open(path, "w") as f:
            json.dump(self.finalize(), f, default=plain)
