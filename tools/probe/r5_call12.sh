#!/bin/bash
# round 5, GPU call 12: offset / mask convs on the fp32-patch kernel (default) vs the halo kernel on piece inputs, with two fp16 pieces
mkdir -p gpurun_out/r5l
for rep in 1 2; do
for v in 1 0; do
  DEFT_OFFSET_FP32=$v timeout 200 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('OFFSET_FP32=$v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')
"
  python - <<PY
import json, collections
d=json.load(open('gpurun_out/bench_ops.json'))
r=collections.OrderedDict(); up=0.0
for c in d['calls']:
    if ' N=32 ' in c[3] and c[0]=='deft_conv2d_nhwc':
        q=r.setdefault(c[3],[0,0.0]); q[0]+=1; q[1]+=c[2]
    if c[0]=='deft_upsample_add': up+=c[2]
print('     offset convs %.3f ms (upsample_add %.3f): ' % (sum(v[1] for v in r.values()), up) + '; '.join('%s %.3f' % (k.split(' 3x3')[0]+k.split('split')[-1], v[1]) for k,v in r.items()))
PY
done
done > gpurun_out/r5l/offset_ab.log 2>&1
cat gpurun_out/r5l/offset_ab.log
