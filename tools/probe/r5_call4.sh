#!/bin/bash
# round 5, GPU call 4: tile A/B of the im2col kernel on the deep layers with two fp16 pieces (the one-stage 64x128 tile was tuned for three bf16 pieces)
mkdir -p gpurun_out/r5d
for t in "" 128x128:1 128x128:2 128x64:1 64x64:1; do
  DEFT_P3_IM2COL_TILE=$t timeout 200 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('tile [$t]', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')
" 
  python - <<PY
import json
d=json.load(open('gpurun_out/bench_ops.json'))
import collections
r=collections.OrderedDict()
for c in d['calls']:
    if ' x3' in c[3]:
        k=c[3]; q=r.setdefault(k,[0,0.0]); q[0]+=1; q[1]+=c[2]
for k,(n,ms) in r.items(): print('    ', k, n, round(ms,3), 'ms')
PY
done > gpurun_out/r5d/im2col_tiles.log 2>&1
cat gpurun_out/r5d/im2col_tiles.log
