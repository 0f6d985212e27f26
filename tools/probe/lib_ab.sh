# A/B of whole-step throughput between builds of the library:  bash tools/probe/lib_ab.sh <variant> [<variant> ...]   (libdeft_<variant>.so; "hip" = the product build)
for rep in $(seq 1 ${REPS:-2}); do
for v in "$@"; do
  DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_$v.so timeout 200 python bench.py --no-extras --no-check --no-cpu-baseline --steps 40 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$v', d['value'], 'frames/s', d['ms_per_step'], 'ms/step  frac', r['frac'], 'dominant', r['dominant_kernel']['name'][:60], r['dominant_kernel']['avg_us'])
"
done
done
