# round 6, call 13: the 128 x 128 two-stage im2col tile on the deep layers: new default (where >= 512 tiles remain) vs off vs forced everywhere
for rep in 1 2; do
for v in "DEFT_P3_IM2COL_WIDE=1" "DEFT_P3_IM2COL_WIDE=0" "DEFT_P3_IM2COL_TILE=128x128:2"; do
  env $v timeout 600 python bench.py --steps 30 --warmup 3 --no-extras --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('[$v]', j['value'], 'frames/s', j['ms_per_step'], 'ms/step')"
done; done
