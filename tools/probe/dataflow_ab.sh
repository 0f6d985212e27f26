mkdir -p gpurun_out/df
for cfg in "DATAFLOW=1 OVERLAP=0" "DATAFLOW=2 OVERLAP=0" "DATAFLOW=1 OVERLAP=1" "DATAFLOW=2 OVERLAP=1" "DATAFLOW=3 OVERLAP=1"; do timeout 120 python tools/probe/latency_ab.py $cfg 2>&1 | tail -1 | cut -c1-60; done
for fill in 768 1024 2048; do echo "DEFT_SPLIT_FILL=$fill"; DEFT_SPLIT_FILL=$fill timeout 120 python tools/probe/latency_ab.py DATAFLOW=2 OVERLAP=1 2>&1 | tail -1 | cut -c1-200; done
echo "DEFT_SPLIT_MINCHUNKS=4 FILL=1024";  DEFT_SPLIT_MINCHUNKS=4 DEFT_SPLIT_FILL=1024 timeout 120 python tools/probe/latency_ab.py DATAFLOW=2 OVERLAP=1 2>&1 | tail -1 | cut -c1-200
