#!/bin/bash
# rocprofv3 passes for one tag, run ON THE GPU BOX (through gpurun) from the repo root:
#     tools/prof.sh <tag> [cmd...]        default cmd: python bench.py --no-cpu-baseline --no-extras --no-check --steps 6 --warmup 2
# Writes gpurun_out/<tag>/{bench.log, stats/, pmc_sq/, pmc_fetch/, pmc_write/}; tools/summarize_prof.py <tag>
# turns them into profiles/<tag>_summary.md.  Counters are collected in their own passes with
# --kernel-trace only (never together with sys/hip/hsa tracing).
set -u
TAG=$1; shift
ONE=""
if [ $# -eq 0 ]; then set -- python bench.py --no-cpu-baseline --no-extras --no-check --steps 6 --warmup 2; ONE="--serialize"; fi
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
"$@" > "$OUT/bench.log" 2>&1
cp gpurun_out/bench_ops.json "$OUT/" 2>/dev/null
cd /tmp
# kernel durations are taken with the launches serialised on ONE stream (the same condition as bench.py's
# per-launch HIP-event roofline measurement); with 2 streams concurrent kernels share the chip and every
# duration inflates
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o "$TAG" --output-format csv -- bash -c "cd $ROOT && $* $ONE" > "$OUT/stats.log" 2>&1
if [ "${PROF_ONLY_STATS:-0}" != "1" ]; then
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_sq" -o p --output-format csv -- bash -c "cd $ROOT && $* $ONE" > "$OUT/pmc_sq.log" 2>&1
if [ "${PROF_HBM:-1}" = "1" ]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p --output-format csv -- bash -c "cd $ROOT && $* $ONE" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p --output-format csv -- bash -c "cd $ROOT && $* $ONE" > "$OUT/pmc_write.log" 2>&1
fi
fi
cd "$ROOT"
# rocprofv3 may nest its output under <dir>/<host>/<pid>: flatten
for d in stats pmc_sq pmc_fetch pmc_write; do
    [ -d "$OUT/$d" ] && find "$OUT/$d" -mindepth 2 -name "*.csv" -exec mv {} "$OUT/$d/" \;
done
# keep only what summarize_prof.py reads (the merge-back limit is 64 MiB)
find "$OUT" -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*kernel_trace.csv" -delete
find "$OUT" -type f -size +30M -delete
ls -R "$OUT" | head -40
