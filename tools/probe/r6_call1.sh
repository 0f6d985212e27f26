# round 6, call 1: LDS / issue counters of the 64-column DCN launch alone (design input for the producer / consumer form)
export OFFSET_SIGMA=1.5
python tools/probe/dcnp_one.py 152 272 64 64 16 64 20 > gpurun_out/r6c1_time.log 2>&1
bash tools/pmc.sh r6c1 'dcn_patch' -- python tools/probe/dcnp_one.py 152 272 64 64 16 64 5 > gpurun_out/r6c1_pmc.log 2>&1
cat gpurun_out/r6c1_time.log; cat gpurun_out/r6c1/pmc*.txt
