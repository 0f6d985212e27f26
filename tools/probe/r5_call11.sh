#!/bin/bash
# round 5, GPU call 11: stride-2 3x3 convs on the im2col piece kernel (flag) A/B; end-to-end config B with the deferred similarity read
mkdir -p gpurun_out/r5k
for rep in 1 2; do
for v in 0 1; do
  DEFT_P3_STRIDE2=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('stride2_on_pieces=$v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')
"
  python - <<PY
import json, collections
d=json.load(open('gpurun_out/bench_ops.json'))
r=collections.OrderedDict()
for c in d['calls']:
    if ' s2 ' in c[3] and c[0]=='deft_conv2d_nhwc':
        q=r.setdefault(c[3],[0,0.0]); q[0]+=1; q[1]+=c[2]
print('     stride-2 total %.3f ms: ' % sum(v[1] for v in r.values()) + '; '.join('%s %.3f' % (k.split(' 3x3')[0]+k.split('split')[-1], v[1]) for k,v in r.items()))
PY
done
done > gpurun_out/r5k/stride2_ab.log 2>&1
cat gpurun_out/r5k/stride2_ab.log
timeout 300 python tools/probe/r5_e2e_profile.py B 2>&1 | grep -E "ms_per_frame|profiled" > gpurun_out/r5k/e2e_B.log
cat gpurun_out/r5k/e2e_B.log
