run() { timeout 900 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --no-check $1 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.readline()
try:
    j = json.loads(l); print('[$1]', j['value'], 'frames/s', j['ms_per_step'], 'ms/step')
except Exception: print('[$1] failed:', l[:200])"; }
for rep in 1 2; do
run "--batch 64 --streams 2"
run "--batch 96 --streams 3"
run "--batch 128 --streams 4"
run "--batch 64 --streams 4"
done
