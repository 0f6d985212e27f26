#!/bin/bash
# round 5, GPU call 3: hand-written fp16 split sequences (v_cvt_pk_f16_f32 + v_fma_mix) + three workgroups per CU on the 128-column halo tiles
mkdir -p gpurun_out/r5c
O=gpurun_out/r5c
./tools/probe/f16_split_asm.bin > $O/f16_split_asm.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q --maxfail=12 -k "not seed_sweep and not full_size_other and not fused_run_with_lookahead" > $O/pytest.log 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_hip.json 2> $O/bench_hip.err
cp gpurun_out/bench_ops.json $O/bench_ops.json
tail -3 $O/pytest.log; cat $O/f16_split_asm.log
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5c/bench_hip.json") if l.startswith("{")][-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms", d["config"]["contraction"], "parity", json.dumps(d["config"].get("parity")))
PY
python tools/layers_table.py r5c | head -40
