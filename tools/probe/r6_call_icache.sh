#!/bin/bash
# round 6: is the 64-column DCN kernel (≈ 6 500 lines of ISA, the K loop unrolled 18 steps) short of INSTRUCTIONS?  Instruction-cache and
# instruction-fetch counters of one launch shape, beside conv3h (a compact loop) as the control.
mkdir -p gpurun_out/r6i
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQC_" | head -40 > gpurun_out/r6i/avail.txt
PMC_SETS="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES;SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_STALL" \
  OFFSET_SIGMA=1.5 bash tools/pmc.sh r6i "dcn_patch_kernel<2" -- python tools/probe/dcnp_one.py 152 272 64 64 16 64 5 > gpurun_out/r6i/dcn.txt 2>&1
cat gpurun_out/r6i/dcn.txt | head -40
PMC_SETS="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES" \
  bash tools/pmc.sh r6i2 "conv3h|igemm3|dcn_patch|pair_mlp|direct_conv" -- python bench.py --no-cpu-baseline --no-extras --no-check --steps 2 --warmup 1 --serialize > gpurun_out/r6i/step.txt 2>&1
head -120 gpurun_out/r6i/step.txt
