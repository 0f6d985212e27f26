# round 6, call 5: float-error seed sweep on the timed plans, both arithmetics
for c in "B 64" "A 16" "D 16" "E 16"; do set -- $c; timeout 1500 python tools/probe/float_sweep.py $1 $2 > gpurun_out/r6c5_sweep_$1.log 2>&1; tail -1 gpurun_out/r6c5_sweep_$1.log | cut -c1-1500; done
