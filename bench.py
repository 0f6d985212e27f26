#!/usr/bin/env python
"""Benchmark of DEFT's per-frame hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--batch B]

A "step" = one batch of B synthetic 1088x608 frames per GPU through
DLA-34 + DCNv2 neck + hm head + decode (K=100) + sparse regression heads +
embedding head (100 detections) + 100x500 affinity (5 history frames x 100).
Inputs are resident in HBM when the timed region starts.  With N>1 (launched by
torch.distributed.run, one rank per GPU) consecutive frames are sharded over the
ranks and one RCCL all-gather of the embedding records per step provides the
cross-rank history (deft_amd/pipeline.py); `value` is whole-job frames/s.

Extra objects in the JSON line:
  roofline      the implicit-GEMM kernel family (conv + DCNv2 launches of the step),
                algorithmic FLOPs (2*M*Cout*K of each conv, no padding) / measured
                launch time (HIP events on the launch stream) vs the FP32-MFMA peak.
  cpu_baseline  the oracle (PyTorch-CPU restatement pinned against the reference
                modules) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: bf16 dense (no sparsity); the split path spends 6 bf16 products per fp32 product
H, W, KDET, HIST = 608, 1088, 100, 5


def cpu_baseline(frames=3, budget_s=25.0):
    """kind=port: oracle/deft_oracle.py on the host cores (GPU not used).  Bounded: stops
    after `budget_s` seconds of CPU work (at least one timed frame)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import deft_oracle as O
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(ncpu, 32)))     # ATen's CPU convs stop scaling (and thrash) far below 256 threads
    sd = O.synth_state_dict("mot")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 3, H, W, generator=g)
    hist = [torch.rand(1, KDET, 416, generator=g) * 3 for _ in range(HIST)]
    t_all = []
    with torch.no_grad():
        O.dlaseg_forward(torch.randn(1, 3, 128, 160, generator=g), sd, "mot")       # small warm-up (thread pool, allocator)
        t_begin = time.time()
        for it in range(frames):
            t0 = time.time()
            out, maps = O.dlaseg_forward(x, sd, "mot")
            dets = O.generic_decode(O.sigmoid_output(out), K=KDET)
            b = dets["bboxes"][0]
            c = torch.stack([(b[:, 0] + b[:, 2]) / (W / 4) - 1, (b[:, 1] + b[:, 3]) / (H / 4) - 1], 1).view(1, KDET, 1, 1, 2)
            emb = O.afe_extract(maps, c, sd)
            for hx in hist:
                O.afe_affinity(hx, emb, sd, 100)
            t_all.append(time.time() - t0)
            if time.time() - t_begin > budget_s:
                break
    t = sorted(t_all)[len(t_all) // 2]
    return {"value": round(1.0 / t, 4), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d frame(s) of 1088x608, median (DLA-34+DCNv2+decode+embed(100)+5x(100x100) affinity), small warm-up, %d threads" % (len(t_all), torch.get_num_threads())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="frames per step per GPU")
    ap.add_argument("--streams", type=int, default=2, help="independent sub-batches on separate HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--autotune", action="store_true", help="per-layer tile search at plan-build time (engine._Plan.autotune)")
    ap.add_argument("--graphs", action="store_true", help="replay each sub-batch's launch list as a captured hipGraph")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--serialize", action="store_true",
                    help="profiling aid: same sub-batch plans, launched back to back on one stream (per-kernel durations "
                         "comparable with the roofline's per-launch HIP events)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or os.environ.get("DEFT_FORCE_DIST") == "1":      # DEFT_FORCE_DIST: 1-rank RCCL group (path check on a 1-GPU box)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    from deft_amd import engine, hiplib, synth
    from deft_amd.pipeline import HipCompute, FramePipeline
    engine_prec = engine.PREC
    lib = hiplib.get_lib()                      # no fallback: raises if the HIP extension is missing
    sd = synth.synth_state_dict("mot")
    B = args.batch
    comp = HipCompute(sd, B, H, W, "mot", K=KDET, device=dev, lib=lib, streams=args.streams)
    comp.serialize = args.serialize
    pipe = FramePipeline(comp, B, KDET, comp.D, history=HIST, device=dev)
    g = torch.Generator().manual_seed(1000 + rank)
    images = torch.randn(B, 3, H, W, generator=g).to(dev)      # resident in HBM before timing

    if args.autotune:                                    # plan-build time, outside the timed region
        comp.autotune(images, verbose=args.verbose and rank == 0)

    if args.graphs:
        comp.capture(images)

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.step(images)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.step(images)
    host_dt = time.perf_counter() - t0        # time the host needed to ENQUEUE the steps (it runs ahead of the GPU)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    frames = args.steps * B * world
    fps = frames / dt

    # ---- roofline of the dominant kernel family: one profiled step, HIP events per launch
    #      (torch events on the stream every kernel is launched on) ----
    roof = None
    prof = []
    gr, comp.graphs = comp.graphs, None          # per-launch events need eager launches
    ser, comp.serialize = comp.serialize, True   # same sub-batch plans, back to back on one stream: per-launch events
    pipe.step(images)                            # un-profiled step queued first: the host then runs AHEAD of the GPU, so the
    if rank == 0:                                # event intervals below hold kernel time, not Python launch latency
        lib.profile = prof
    pipe.step(images)                            # EVERY rank runs the step (it contains the all-gather); rank 0 records
    comp.serialize = ser
    comp.graphs = gr
    torch.cuda.synchronize()
    lib.profile = None
    if rank == 0:
        GEMM = ("deft_conv2d_nhwc", "deft_conv2d_group", "deft_dcn_v2_nhwc", "deft_pair_layer")
        gemm_ms = sum(e0.elapsed_time(e1) for (k, _, e0, e1, _i, _b) in prof if k in GEMM)
        gemm_fl = sum(fl for (k, fl, _, _, _i, _b) in prof if k in GEMM)
        n_launch = sum(1 for p in prof if p[0] in GEMM)
        all_ms = sum(e0.elapsed_time(e1) for (_, _, e0, e1, _i, _b) in prof)
        # fixed cost of one (event, launch, event) bracket: the smallest kernels of the step (a few us of real work)
        tiny = sorted(e0.elapsed_time(e1) for (k, _, e0, e1, _i, _b) in prof if k in ("deft_peak_rows", "deft_embed_rows", "deft_decode_boxes"))
        ev_over_ms = tiny[len(tiny) // 2] if tiny else 0.0
        ach = gemm_fl / (gemm_ms * 1e-3) / 1e12
        ach_corr = gemm_fl / (max(gemm_ms - n_launch * ev_over_ms, 1e-6) * 1e-3) / 1e12
        traffic, tsrc = None, None                    # HBM bytes per launch from the committed PMC passes of this build, if any
        tfile = os.path.join(ROOT, "profiles", "r1_traffic.json")
        if os.path.exists(tfile):
            tj = json.load(open(tfile))
            traffic, tsrc = round(tj["traffic_bytes_per_launch"]), "profiles/r1_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH x2)"
        roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": round(ach / FP32_MFMA_PEAK_TF, 4), "traffic": traffic, "traffic_unit": "B/launch", "traffic_source": tsrc,
                "algorithmic_bytes_per_launch": round(sum(b for (k, _, _, _, _i, b) in prof if k in GEMM) / max(1, n_launch)),
                "kernel": "igemm_kernel<*> (implicit GEMM with fp32 results: conv / DCNv2 / pair loaders)",
                # what the contraction runs on.  prec 1: each fp32 operand = 3 bf16 pieces, each fp32 product = 6 bf16 MFMA
                # products, fp32 accumulation (error of an fp32 chain); BN < 64 tiles and prec 0: the fp32 MFMA instruction.
                # `peak` stays the fp32 MFMA peak (what an fp32 result is priced against); the split path's own ceiling
                # is the bf16 dense peak / 6.
                "arithmetic": ("fp32 via 3 x bf16 operand split, 6 x v_mfma_f32_32x32x16_bf16 per fp32 product (tiles with BN >= 64); "
                               "v_mfma_f32_32x32x2_f32 elsewhere") if engine_prec == 1 else "v_mfma_f32_32x32x2_f32",
                "peak_split_bf16": round(BF16_MFMA_PEAK_TF / 6, 1) if engine_prec == 1 else None,
                "frac_of_split_peak": round(ach / (BF16_MFMA_PEAK_TF / 6), 4) if engine_prec == 1 else None,
                "launches_per_step": n_launch, "gflop_per_step": round(gemm_fl / 1e9, 2),
                "avg_launch_us": round(gemm_ms * 1e3 / max(1, n_launch), 2),
                "event_bracket_overhead_us": round(ev_over_ms * 1e3, 2),     # median bracket of the step's ~2 us kernels
                "achieved_minus_bracket_overhead": round(ach_corr, 3),
                "ms_per_step_in_kernel": round(gemm_ms, 3), "ms_per_step_all_kernels": round(all_ms, 3),
                # the same FLOPs over the TIMED steps (sub-batches overlapped on their streams, every other kernel included)
                "pipeline_achieved": round(gemm_fl / (dt / args.steps) / 1e12, 3),
                "pipeline_frac": round(gemm_fl / (dt / args.steps) / 1e12 / FP32_MFMA_PEAK_TF, 4)}
        by = {}
        for (k, fl, e0, e1, _i, _b) in prof:
            by.setdefault(k, [0.0, 0, 0.0]); by[k][0] += e0.elapsed_time(e1); by[k][1] += 1; by[k][2] += fl
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_ops.json"), "w") as f:
            json.dump({"by_entry_ms_launches_flops": by, "calls": [(k, fl, e0.elapsed_time(e1), info) for (k, fl, e0, e1, info, _b) in prof]}, f)

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline()

    if rank == 0:
        out = {"metric": "frames/sec (detect+embed+affinity) at 1088x608", "value": round(fps, 3), "unit": "frames/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "host_enqueue_ms_per_step": round(host_dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "MOT17 1088x608 DLA-34 + DCNv2 + 100x500 affinity (BASELINE configs[1])",
                          "contraction": "split-bf16 x6, fp32 accumulate" if engine_prec == 1 else "fp32 MFMA", "frames_per_step_per_gpu": B, "hip_streams": args.streams, "hip_graphs": bool(args.graphs), "detections": KDET, "history_frames": HIST,
                          "parallelism": "frames sharded dp%d, 1 all-gather/step" % world},
               "roofline": roof, "cpu_baseline": cpu}
    # RCCL prints its version banner through C stdio on every rank; it would otherwise be flushed at process exit,
    # AFTER the JSON line.  Flush it everywhere first, then rank 0 prints the JSON as the last stdout line.
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
