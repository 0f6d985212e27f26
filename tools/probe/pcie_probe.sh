timeout 200 python -m pytest tests/test_gpu_parity.py -q -x -k "preprocess or fused or device_detect or frame_feeder" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline > /tmp/pp.log 2>/dev/null
grep '"metric"' /tmp/pp.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d.get('value_incl_pcie'), d.get('latency_mode',{}).get('ms_per_step'), d.get('config_C',{}).get('ms_per_step'), d.get('end_to_end',{}).get('ms_per_frame'))"
f=$(find /tmp/pp -name '*kernel_stats.csv' | head -1)
grep -i "preprocess\|nchw_to_nhwc" $f | cut -c1-160
