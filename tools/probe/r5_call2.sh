#!/bin/bash
# round 5, GPU call 2: first hardware run of the two-fp16-piece build (libdeft_hip.so) next to the three-bf16-piece build of the same sources
mkdir -p gpurun_out/r5b
O=gpurun_out/r5b
./tools/probe/f16_denorm.bin > $O/f16_denorm.log 2>&1
timeout 120 ./tools/probe/pkf32_min.bin 20 > $O/pkf32_min.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q --maxfail=12 -k "not seed_sweep" --durations=25 > $O/pytest_np2.log 2>&1
for lib in hip bf16x3; do
  DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$lib.so timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_$lib.json 2> $O/bench_$lib.err
done
tail -4 $O/pytest_np2.log; cat $O/f16_denorm.log; tail -3 $O/pkf32_min.log
python - <<'PY'
import json
for lib in ("hip", "bf16x3"):
    try:
        d = json.loads([l for l in open("gpurun_out/r5b/bench_%s.json" % lib) if l.startswith("{")][-1])
        print(lib, d["value"], "frames/s", d["ms_per_step"], "ms", d["config"]["contraction"], "parity", json.dumps(d["config"].get("parity")))
    except Exception as e:
        print(lib, "no line:", e)
PY
