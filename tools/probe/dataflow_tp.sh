# throughput mode (config B) with the data-flow schedule inside every 16-frame sub-batch plan
for cfg in "1 1" "2 16" "3 16" "1 1" "2 16"; do set -- $cfg; echo "DATAFLOW=$1 MAX_N=$2"; DEFT_DATAFLOW=$1 DEFT_DATAFLOW_MAX_N=$2 timeout 200 python bench.py --no-extras --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"; done
