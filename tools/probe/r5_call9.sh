#!/bin/bash
# round 5, GPU call 9: more MFMAs per barrier -- 64x128 im2col tile in 2 / 3 stages, 64-column halo tile on a filter row per interval
mkdir -p gpurun_out/r5i
run() {
  env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('[$*]', d['value'], 'frames/s', d['ms_per_step'], 'ms/step')
"
  python - <<PY
import json, collections
d=json.load(open('gpurun_out/bench_ops.json'))
r=collections.OrderedDict()
for c in d['calls']:
    if (' x3' in c[3]) or ('N=64 K=576 3x3 s1 @152x272 split halo' in c[3]):
        q=r.setdefault(c[3],[0,0.0]); q[0]+=1; q[1]+=c[2]
print('     ' + '; '.join('%s: %.3f' % (k.split(' 3x3')[0] + (' halo' if 'halo' in k else ' x3'), v[1]) for k,v in r.items()))
PY
}
for rep in 1 2; do
run A=0
run DEFT_P3H_TPI3_64=1
run DEFT_P3_IM2COL_TILE=64x128:2
run DEFT_P3_IM2COL_TILE=64x128:3
done > gpurun_out/r5i/barrier_ab.log 2>&1
cat gpurun_out/r5i/barrier_ab.log
