# round 6, call 11: the whole GPU suite on the final tree, the kernel / model tests on the three-bf16-piece entry points, the closing bench line
mkdir -p gpurun_out/r6y; O=gpurun_out/r6y
timeout 3000 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.log 2>&1; tail -16 $O/pytest_gpu.log
DEFT_ARITH=bf16x3 timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "not seed_sweep and not full_size and not bench_parity and not rccl and not out_of_range and not twin and not float_errors and not fp16_split and not pair_mlp" > $O/pytest_gpu_bf16x3.log 2>&1; tail -3 $O/pytest_gpu_bf16x3.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6y/bench_full.json") if l.startswith("{")][-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms", d["roofline"]["frac"], json.dumps(d["config"]["parity"]["pass"]), json.dumps(d["config"]["side"])[:900]); print(json.dumps(d["cpu_baseline"])[:500])
PY
