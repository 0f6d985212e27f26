#!/bin/bash
# round 6: the GPU suite in REVERSE order (the graph-replay fault hid behind the file order for three rounds: what else does?)
mkdir -p gpurun_out/r6x
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" > gpurun_out/r6x/ids.txt
tac gpurun_out/r6x/ids.txt > gpurun_out/r6x/ids_rev.txt
wc -l gpurun_out/r6x/ids_rev.txt
timeout 2400 python -X faulthandler -m pytest -q -p no:cacheprovider $(cat gpurun_out/r6x/ids_rev.txt | tr '\n' ' ') > gpurun_out/r6x/reverse.log 2>&1
echo "rc=$?"; tail -5 gpurun_out/r6x/reverse.log | cut -c1-200
