# round 6, call 14: knob A/Bs on the two-piece kernels, inside one call (config B, 30 steps)
run() { env $1 timeout 600 python bench.py --steps 30 --warmup 3 --no-extras --no-cpu-baseline --no-check $2 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('[$1 $2]', j['value'], 'frames/s', j['ms_per_step'], 'ms/step')"; }
for rep in 1 2; do
run "A=0" ""
run "DEFT_P3_MIN_TILES=320" ""
run "DEFT_P3_STRIDE2=1" ""
run "DEFT_OFFSET_FP32_MIN_HW=2000" ""
run "DEFT_DCN_PATCH_MIN_TILES=128" ""
run "A=0" "--batch 48"
run "A=0" "--batch 64"
run "A=0" "--batch 48 --streams 3"
done
