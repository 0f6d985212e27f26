#!/bin/bash
# round 5, final GPU call: the full -m gpu suite on the product library, the kernel / model parity tests on the three-bf16-piece build, the default
# bench line as the driver runs it, and the rocprofv3 passes (stats, SQ, FETCH, WRITE) of the same build
mkdir -p gpurun_out/r5z
O=gpurun_out/r5z
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_bf16x3.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "not seed_sweep and not full_size and not bench_parity and not rccl" > $O/pytest_gpu_bf16x3.log 2>&1
tail -2 $O/pytest_gpu_bf16x3.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
cp gpurun_out/bench_ops.json $O/bench_ops.json
cp gpurun_out/topk_sweep_*.json $O/ 2>/dev/null
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5z/bench_full.json") if l.startswith("{")][-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms", json.dumps(d["config"]["parity"]), json.dumps(d["config"]["side"]))
PY
bash tools/prof.sh r5 > $O/prof.log 2>&1
tail -3 $O/prof.log
