mkdir -p gpurun_out/df
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k dataflow 2>&1 | tail -5
for cfg in "DATAFLOW=1" "DATAFLOW=2" "DATAFLOW=3" "DATAFLOW=4"; do timeout 120 python tools/probe/latency_ab.py $cfg 2>&1 | tail -1; done
