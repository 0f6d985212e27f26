#!/bin/bash
# round 5, GPU call 10: the full-resolution layers (direct.hip) on two fp16 pieces against the build that kept three bf16 pieces there
mkdir -p gpurun_out/r5j
timeout 600 python -m pytest tests/test_gpu_parity.py -q --maxfail=8 -k "direct or forward_embed or full_size_properties or golden or out_of_range or identity" > gpurun_out/r5j/pytest.log 2>&1
tail -3 gpurun_out/r5j/pytest.log
for rep in 1 2; do
for v in hip dold; do
  DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$v.so timeout 200 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step', json.dumps(d['config']['parity']['max_err']), d['config']['parity']['pass_up_to_roundoff_ties'])
"
  python - <<PY
import json, collections
d=json.load(open('gpurun_out/bench_ops.json'))
r=collections.OrderedDict()
for c in d['calls']:
    if c[0]=='deft_conv_direct':
        q=r.setdefault(c[3],[0,0.0]); q[0]+=1; q[1]+=c[2]
print('     direct total %.3f ms: ' % sum(v[1] for v in r.values()) + '; '.join('%s %.3f' % (k.split(' @')[0], v[1]) for k,v in r.items()))
PY
done
done > gpurun_out/r5j/direct_ab.log 2>&1
cat gpurun_out/r5j/direct_ab.log
