#!/bin/bash
# Readiness check of the multi-GPU paths for the day an N-GPU node is available.  No scaling curve is computed here (the
# driver does that from bench.py's per-N lines): this asserts that every path that has only ever run as ONE RCCL rank works as N.
#
#   tools/scale_check.sh [N ...]          default: 2 4 8, each skipped when the node has fewer GPUs
#
# For each N:  (i) bench.py --gpus N (config B: frames sharded, one all-gather per step): the line comes from N connected ranks over RCCL and
#              its N-rank parity gate passes;  (ii) bench.py --gpus N --config E (one camera stream per GPU: replicas, no collective);
#              (iii) run_stream.py over N ranks (config C: ONE video stream, frame k on GPU k % N, association on rank 0): the tracks are IDENTICAL to
#              the one-GPU run's, frame by frame (ids and boxes).
# Exit code 0 = every assertion held.  Logs: gpurun_out/scale_check/.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/scale_check; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())")
NS=${@:-2 4 8}
FRAMES=64
fail=0
say() { echo "[scale_check] $*"; }
check() { python - "$@" <<'PY'
import json, sys
kind, path, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
line = [l for l in open(path) if l.startswith("{")]
assert line, "no JSON line in %s" % path
j = json.loads(line[-1])
if kind == "bench":
    d = j["distributed"]
    assert j["n_gpus"] == n and d["ranks"] == n and d["launcher_world_size"] == n and d["backend"] == ("nccl" if n > 1 else None), d
    if j["config"]["config"] == "E" or n == 1:
        assert d["collectives_per_step"] == 0 and (n == 1 or "replicas" in j["config"]["parallelism"]), (d, j["config"]["parallelism"])
    else:
        assert d["collectives_per_step"] == 1.0 and d["bytes_gathered_per_step_per_rank"] > 0, d
    p = j["config"]["parity"]
    assert p["pass"] is True and p["error"] is None, p
    print("   %s: %.1f frames/s on %d ranks, gate pass (raw strict: %s)" % (j["config"]["config"], j["value"], n, p["raw"]["pass"]))
else:
    assert j["world"] == n and (n == 1 or (j["collectives"] and j["backend"] == "nccl")), j
    print("   stream: %.1f frames/s on %d ranks, %d track outputs" % (j["frames_per_s"], n, j["track_outputs"]))
PY
}
say "node has $NG GPU(s); one-GPU reference run of the stream"
python run_stream.py --frames $FRAMES --tracks-out $OUT/tracks_n1.json > $OUT/stream_n1.log 2>&1 && check stream $OUT/stream_n1.log 1 || { say "FAILED: one-GPU stream"; fail=1; }
for N in $NS; do
    if [ "$N" -gt "$NG" ]; then say "N=$N skipped ($NG GPUs)"; continue; fi
    say "N=$N"
    python bench.py --gpus $N --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench_B_n$N.log 2>&1 && check bench $OUT/bench_B_n$N.log $N || { say "FAILED: bench B N=$N"; fail=1; }
    python bench.py --gpus $N --config E --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench_E_n$N.log 2>&1 && check bench $OUT/bench_E_n$N.log $N || { say "FAILED: bench E N=$N"; fail=1; }
    [ "$N" = 1 ] && continue
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) run_stream.py --frames $FRAMES \
        --tracks-out $OUT/tracks_n$N.json > $OUT/stream_n$N.log 2>&1 && check stream $OUT/stream_n$N.log $N || { say "FAILED: stream N=$N"; fail=1; }
    python - $OUT/tracks_n1.json $OUT/tracks_n$N.json <<'PY' || { echo "[scale_check] FAILED: tracks of N ranks differ from one GPU's"; fail=1; }
import json, sys
a, b = (json.load(open(p)) for p in sys.argv[1:3])
assert len(a) == len(b) and len(a) > 0, (len(a), len(b))
for (fa, ta), (fb, tb) in zip(a, b):
    assert fa == fb and ta == tb, ("frame", fa, fb, ta[:2], tb[:2])
print("   tracks identical to the one-GPU run over %d frames (%d track outputs)" % (len(a), sum(len(t) for _, t in a)))
PY
done
[ $fail = 0 ] && say "all checks held" || say "SOME CHECKS FAILED (logs in $OUT)"
exit $fail
