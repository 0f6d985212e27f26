#!/bin/bash
# round 5, GPU call 18: frames per lookahead pass of the end-to-end loop (config B is device-bound now: a larger pass is a more efficient pass)
mkdir -p gpurun_out/r5r
for pp in 8 6 4; do
  DEFT_E2E_PER_PASS=$pp timeout 200 python tools/probe/r5_e2e_ab.py B 2>/dev/null | grep '^{' | sed "s/^/per_pass=$pp /" | tee -a gpurun_out/r5r/per_pass.log
done
