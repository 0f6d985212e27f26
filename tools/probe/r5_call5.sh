#!/bin/bash
# round 5, GPU call 5: DCN patch margin (R) and gather lookahead (GA) A/B with two fp16 pieces (the weight stages shrank: R = 4 fits two workgroups per CU now)
mkdir -p gpurun_out/r5e
for rep in 1 2; do
for v in hip r24 r43 r24r43 ga1; do
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
done > gpurun_out/r5e/dcn_margin_ab.log 2>&1
cat gpurun_out/r5e/dcn_margin_ab.log
