# round 6, call 10: where a tracked frame of config B goes now (fused pair MLP in the tracker's chain): cProfile of update(), kernel stats of the loop
python tools/probe/r5_e2e_profile.py B 2>&1 | grep -v amdgpu.ids | head -60 > gpurun_out/r6_e2e_profile_B.log; head -45 gpurun_out/r6_e2e_profile_B.log
ROOT=$PWD; mkdir -p gpurun_out/r6t; cd /tmp && export TMPDIR=/tmp
DEFT_E2E_REPORTED_MODE_ONLY=1 timeout 170 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r6t/stats -o e2eB --output-format csv -- python $ROOT/bench.py --e2e-only B --e2e-frames 100 > $ROOT/gpurun_out/r6t/run.log 2>&1
cd $ROOT; grep '^{' gpurun_out/r6t/run.log | cut -c1-500
f=$(find gpurun_out/r6t/stats -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r6_e2e_B_kernel_stats.csv; find gpurun_out/r6t/stats -name '*kernel_trace.csv' -delete
head -22 gpurun_out/r6_e2e_B_kernel_stats.csv | cut -c1-150
