#!/bin/bash
# round 5, GPU call 7: DCN 64-column tile at three workgroups per CU (margin 2, lookahead 1, 168 VGPRs with spills) against the product build
mkdir -p gpurun_out/r5g
for rep in 1 2; do
for v in hip occ3; do
  DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$v.so timeout 200 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')
"
  python - <<PY
import json, collections
d=json.load(open('gpurun_out/bench_ops.json'))
r=collections.OrderedDict()
for c in d['calls']:
    if c[0]=='deft_dcn_v2_nhwc':
        q=r.setdefault(c[3],[0,0.0]); q[0]+=1; q[1]+=c[2]
print('     dcn total %.3f ms: ' % sum(v[1] for v in r.values()) + '; '.join('%s %.3f' % (k.split(' 3x3')[0], v[1]) for k,v in r.items()))
PY
done
done > gpurun_out/r5g/dcn_occ3_ab.log 2>&1
DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_occ3.so timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "dcn" 2>&1 | tail -2 >> gpurun_out/r5g/dcn_occ3_ab.log
cat gpurun_out/r5g/dcn_occ3_ab.log
