"""Per-frame latency of the two tracker-side device forms against the call pattern they replace (GPU box):

  motion update    T per-track `KalmanFilterLSTM.predict` calls (tracker.py:467: H2D features, launch, D2H, per track)
                   vs ONE `MotionBank.step` (deft_motion_step) for the T tracks;
  track similarity D2H of the frame's affinity blocks + per-track numpy medians (tracker.py:219-252, 663-688)
                   vs `deft_amd.tracker.get_similarity` (deft_track_similarity; only [T, N+1] comes back).

  tracker update   deft_amd.array_tracker.Tracker2D.update (the 2-D association loop on the device forms) over a synthetic scene of
                   `tracks` objects with the recorder full (`stored` frames): total per frame, and inside it the embedding extraction,
                   FeatureRecorder.update (the affinity chain), get_similarity, fuse_motion, lapjv, bbox_overlaps, the batched Kalman
                   steps, and the remaining per-track Python.

    python tools/bench_tracker_ops.py [--tracks 100] [--dets 100] [--stored 49]  -> gpurun_out/tracker_ops.json
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deft_amd import engine, integrate, synth, tracker as DT  # noqa: E402


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def tracker_update_split(T, R, dev, H=608, W=1088, nframes=40):
    """Tracker2D.update on T drifting objects (every object detected every frame -> T tracks x T detections), real embedding /
    affinity kernels on random FeatureMaps of the config-B shapes, recorder filled with R frames before timing."""
    from deft_amd import association as A, array_tracker as MT, hiplib
    from deft_amd.engine import View
    sd = synth.synth_state_dict("mot")
    lib = hiplib.get_lib()
    afe = integrate.AfeSeam(sd, 100, dev, lib)
    afe.host_copy = False
    opt = SimpleNamespace(dataset="mot", track_buffer=30, max_object=100, lstm=False)
    trk = MT.Tracker2D(opt, SimpleNamespace(AFE=afe), h=H, w=W)
    # the 13 FeatureMaps of a 1088x608 frame (random values: the tracker only samples them)
    fm = []
    for c, s_ in zip(synth.SELECTOR_IN, synth.FEATURE_STRIDES):
        v = afe.plan.alloc(1, H // s_, W // s_, c)
        v.buf.normal_()
        fm.append(v)
    g = np.random.RandomState(1)
    x0 = g.uniform(40, W - 120, T); y0 = g.uniform(40, H - 200, T)
    vx = g.uniform(-2, 2, T); vy = g.uniform(-1, 1, T)
    bw = g.uniform(30, 60, T); bh = g.uniform(80, 160, T)

    def frame(t):
        return [{"score": float(0.9 - 0.004 * i), "class": 1,
                 "bbox": np.array([x0[i] + vx[i] * t, y0[i] + vy[i] * t, x0[i] + vx[i] * t + bw[i], y0[i] + vy[i] * t + bh[i]], np.float32)} for i in range(T)]

    acc = {}

    def wrap(obj, name, key):
        fn = getattr(obj, name)

        def timed(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize(); acc[key] = acc.get(key, 0.0) + time.perf_counter() - t0
            return out
        setattr(obj, name, timed)
        return lambda: setattr(obj, name, fn)

    for t in range(R + 2):                    # fill the recorder (R stored frames), un-timed
        trk.update(frame(t), fm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(R + 2, R + 2 + nframes):
        trk.update(frame(t), fm)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / nframes * 1e3
    undo = [wrap(afe, "forward_feature_extracter", "embed_extract"), wrap(trk.recorder, "update", "recorder_update (affinity chain)"),
            wrap(trk, "get_similarity", "get_similarity"), wrap(A, "fuse_motion", "fuse_motion"), wrap(A, "lapjv", "lapjv"),
            wrap(A, "bbox_overlaps", "bbox_overlaps"), wrap(MT, "kf_multi_predict", "kf_multi_predict"), wrap(MT, "kf_multi_update", "kf_multi_update")]
    t0 = time.perf_counter()
    for t in range(R + 2 + nframes, R + 2 + 2 * nframes):
        trk.update(frame(t), fm)
    torch.cuda.synchronize()
    total_instr = (time.perf_counter() - t0) / nframes * 1e3
    for u in undo:
        u()
    split = {k: round(v / nframes * 1e3, 4) for k, v in acc.items()}
    split["other per-track Python"] = round(total_instr - sum(split.values()), 4)
    return {"ms_per_frame": round(total, 4), "ms_per_frame_with_sync_per_call": round(total_instr, 4), "split_ms": split,
            "tracks_alive": len(trk.tracked_stracks), "stored_frames": len(trk.recorder.all_frame_index), "detections": T}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=100)
    ap.add_argument("--dets", type=int, default=100)
    ap.add_argument("--stored", type=int, default=49)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda")
    T, N, R = a.tracks, a.dets, a.stored
    g = np.random.RandomState(0)
    res = {"tracks": T, "dets": N, "stored_frames": R}

    # ---- motion update ----
    opt = SimpleNamespace(dataset="mot", load_model_traj="", gpus=[0])
    kf = integrate.KalmanFilterLSTM(opt, synth.synth_lstm_state_dict("mot"))
    bank = DT.MotionBank(kf, capacity=T)
    slots = [bank.alloc() for _ in range(T)]
    boxes = np.abs(g.randn(T, 4)) * 50 + 20
    frame = [0]

    def batched():
        frame[0] += 1
        bank.step(slots, boxes, frame[0])
    hs = [torch.zeros(1, 1, 128, device=dev) for _ in range(T)]
    cs = [torch.zeros(1, 1, 128, device=dev) for _ in range(T)]
    feats = np.abs(g.randn(T, 11))

    def per_track():
        for t in range(T):
            x = torch.from_numpy(feats[t:t + 1]).unsqueeze(0).to(dev).float()        # tracker.py:463-465
            hs[t], cs[t], _ = kf.predict(hs[t], cs[t], x)
    res["motion_per_track_ms"] = timeit(per_track, max(2, a.reps // 4))
    res["motion_batched_ms"] = timeit(batched, a.reps)

    # ---- track similarity ----
    sim = torch.rand(R * N, N + 1, device=dev)
    starts = [k * N for k in range(R + 1)]
    fr = R + 1
    index = {p: (p - 1, np.float32(1.0)) for p in range(1, R + 1)}
    nodes = [[SimpleNamespace(frame_index=p, id=int(g.randint(N))) for p in range(max(1, fr - 1 - int(g.randint(1, 12))), fr)] for _ in range(T)]
    pool = [SimpleNamespace(nodes=n) for n in nodes]
    me = SimpleNamespace(recorder=SimpleNamespace(_dev=(fr, sim, starts, index)), dataset="mot",
                         model=SimpleNamespace(AFE=SimpleNamespace(plan=engine._Plan("cuda", None))))

    def device_form():
        return DT.get_similarity(me, fr, pool, N)

    def host_form():
        y = sim.cpu().numpy()                                                        # what the recorder copies per frame
        blocks = {p: y[starts[p - 1]:starts[p]] for p in range(1, R + 1)}
        out = []
        for trk in pool:
            rows = [blocks[n.frame_index][n.id, :] for n in trk.nodes if fr - n.frame_index < 50]
            arr = np.array(rows)
            if arr.shape[0] > 5:
                arr = arr[arr.shape[0] - 4:]
            out.append(np.median(arr, axis=0).tolist())
        return np.array(out)
    assert np.array_equal(device_form(), host_form())
    res["similarity_host_ms"] = timeit(host_form, a.reps)
    res["similarity_device_ms"] = timeit(device_form, a.reps)
    res["tracker_update"] = tracker_update_split(T, R, dev)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/tracker_ops.json", "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
