# round 6, call 12: tile of the deep im2col layers (256->256 @38x68, 512->512 @19x34), A/B inside one call
for rep in 1 2; do
for t in "" "128x128:2" "128x128:1" "64x128:2"; do
  DEFT_P3_IM2COL_TILE=$t timeout 600 python bench.py --steps 30 --warmup 3 --no-extras --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('tile [%s]' % '$t', j['value'], 'frames/s', j['ms_per_step'], 'ms/step')"
done; done
