#!/usr/bin/env python
"""Drive ONE video stream over the GPUs of a node with deft_amd.stream.ShardedStream (BASELINE configs[2]: consecutive frames
sharded one per GPU; two small all-gathers per step over RCCL/xGMI; association on rank 0).

    python run_stream.py --frames 16                          # one GPU (no collective)
    python run_stream.py --frames 16 --force-dist --check     # one GPU, the collectives run through a 1-rank RCCL group and are
                                                              # checked against the direct path (used by the -m gpu test)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_stream.py --frames 64

Synthetic weights / frames (no dataset here).  With the reference's src/lib on PYTHONPATH (and its third-party imports
available) rank 0 runs the reference's own `Tracker` from the gathered records; otherwise the run stops after the exchange
(records + affinity blocks on every rank), which is everything that involves the GPUs.  Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", default="608x1088")
    ap.add_argument("--dets", type=int, default=30, help="detections kept per frame (random weights: a score threshold is meaningless)")
    ap.add_argument("--force-dist", action="store_true", help="run the collectives in a 1-rank process group")
    ap.add_argument("--check", action="store_true", help="compare records / blocks with a second, collective-free stream")
    ap.add_argument("--bench", action="store_true", help="timed run: --warmup frames untimed, then --frames frames; a second pass splits the step into its phases")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-tracker", action="store_true", help="exchange only (records + affinity blocks on every rank)")
    ap.add_argument("--host-detect", action="store_true", help="the round-2 detect path (NCHW adapter, host decode): for A/B")
    ap.add_argument("--out", default=None, help="also write the JSON line to this file")
    ap.add_argument("--tracks-out", default=None, help="rank 0: write every frame's tracks [(frame, [(id, tlwh)])] to this JSON file (tools/scale_check.sh compares N ranks with one)")
    args = ap.parse_args()
    H, W = [int(v) for v in args.size.split("x")]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
    from deft_amd import integrate, synth
    from deft_amd.stream import DeviceDetect, ShardedStream
    torch.set_grad_enabled(False)
    sd = synth.synth_state_dict("mot")
    model = integrate.DeftModel(sd, "mot", K=100, max_object=100, device=dev)

    def detect_host(x):
        out, fmaps = model(x.to(dev), None, None)
        o = dict(out[-1]); o["hm"] = o["hm"].sigmoid()
        dets = {k: v.detach().cpu() for k, v in integrate.generic_decode(o, K=100).items()}
        res = [{"score": float(dets["scores"][0, i]), "class": int(dets["clses"][0, i]) + 1,
                "bbox": dets["bboxes"][0, i].numpy().astype(np.float32) * 4.0} for i in range(args.dets)]     # down_ratio 4 (opts.py:138-143)
        return res, fmaps

    # the fused device-side front half: hipGraph replay + on-device post-process / centres / embeddings (deft_amd.stream.DeviceDetect)
    detect = detect_host if args.host_detect else DeviceDetect(sd, H, W, "mot", K=100, device=dev, img_h=H, img_w=W, out_thresh=-1.0,
                                                               first_n=args.dets, afe_plan=model.AFE.plan)
    tracker, why = None, None
    if rank == 0 and not args.no_tracker:
        try:                                              # the reference's Tracker, when its tree is importable
            argv, sys.argv = sys.argv, ["test.py", "tracking"]
            from opts import opts
            from utils import tracker as RT
            sys.argv = argv
            tracker = RT.Tracker(opts().parse(["tracking", "--dataset", "mot"]), model, h=H, w=W)
        except Exception as e:                            # not importable here (the GPU box has no reference tree): this repository's own
            why = "%s: %s" % (type(e).__name__, e)        # 2-D tracker (identical tracks to the reference's, tests/test_mot_tracker.py)
            sys.argv = sys.argv if sys.argv[0] != "test.py" else [__file__]
            from types import SimpleNamespace
            from deft_amd.array_tracker import Tracker2D
            tracker = Tracker2D(SimpleNamespace(dataset="mot", track_buffer=30, max_object=100, lstm=False), SimpleNamespace(AFE=model.AFE), h=H, w=W)
    mk = lambda trk, coll: ShardedStream(detect, model.AFE, model.AFE.plan.D, tracker=trk, dataset="mot", kmax=100, img_h=H, img_w=W, batch=1, device=dev,
                                         force_collective=coll, snapshot=lambda tg: [(int(t.track_id), [float(v) for v in t.tlwh]) for t in tg])
    st = mk(tracker, args.force_dist)
    ref = mk(None, False) if args.check else None
    if ref is not None:
        ref.collective, ref.world, ref.rank = False, 1, 0
    ok, ntracks, track_log = True, 0, []
    nsteps = args.frames // world
    nwarm = (args.warmup // world) if args.bench else 0
    frames = [torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(500 + t)).to(dev) for t in range(min(nsteps + nwarm, 24) * world)]
    fr = lambda s: frames[(s * world + rank) % len(frames)]
    for s in range(nwarm):
        st.step([fr(s)])
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for s in range(nwarm, nwarm + nsteps):
        out = st.step([fr(s)])
        ntracks += sum(len(tg) for _, tg in out)
        if args.tracks_out and rank == 0:
            track_log += [[int(f), [[int(i), [round(float(v), 4) for v in b]] for i, b in tg]] for f, tg in out]
        if ref is not None and world == 1:
            ref.step([fr(s)])
            ok &= torch.equal(st.all_rec, ref.all_rec) and torch.equal(st.all_blk, ref.all_blk)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    dt = time.perf_counter() - t0
    phases = None
    if args.bench:                                        # second pass with a device synchronisation after every phase: where the step goes
        st.timers = {}
        for s in range(nwarm + nsteps, nwarm + 2 * nsteps):
            st.step([fr(s)])
        phases = {k: round(v / nsteps * 1e3, 4) for k, v in st.timers.items()}          # ms per STEP (= per `world` frames)
        st.timers = None
    try:                                   # RCCL prints its banner through C stdio; flush it BEFORE the JSON line (last line of stdout)
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if rank == 0:
        check = None
        if args.check:
            check = ("ok" if ok else "MISMATCH") if world == 1 else "skipped (compared only in a one-process run)"
        line = {"frames": nsteps * world, "world": world, "size": "%dx%d" % (W, H), "collectives": bool(st.collective),
                "backend": dist.get_backend() if dist.is_initialized() else None, "ms_per_frame": round(dt / (nsteps * world) * 1e3, 4),
                "frames_per_s": round(nsteps * world / dt, 2), "detect": "host" if args.host_detect else "device",
                "bytes_gathered_per_step": st.bytes_gathered // max(nsteps + nwarm + (nsteps if args.bench else 0), 1),
                "reference_tracker": tracker is not None and why is None, "reference_tracker_unavailable": why,
                "tracker": None if tracker is None else (type(tracker).__module__ + "." + type(tracker).__name__),
                "track_outputs": ntracks, "check": check, "phase_ms_per_step": phases}
        print(json.dumps(line), flush=True)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            json.dump(line, open(args.out, "w"), indent=1)
        if args.tracks_out:
            os.makedirs(os.path.dirname(os.path.abspath(args.tracks_out)), exist_ok=True)
            json.dump(track_log, open(args.tracks_out, "w"))
    if dist.is_initialized():
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
