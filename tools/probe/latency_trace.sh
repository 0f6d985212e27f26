# kernel timeline of the one-frame-per-step mode: rocprofv3 --kernel-trace of tools/probe/latency_ab.py, then tools/probe/latency_gaps.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/lat
mkdir -p $OUT
for df in ${DFS:-2}; do
  rm -rf /tmp/lt_$df
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_$df -o t -- python $GRAFT_REPO_ROOT/tools/probe/latency_ab.py DATAFLOW=$df > $OUT/run_$df.log 2>&1
  f=$(find /tmp/lt_$df -name '*kernel_trace.csv' | head -1)
  python $GRAFT_REPO_ROOT/tools/probe/latency_gaps.py $f all > $OUT/gaps_$df.txt 2>&1
  tail -1 $OUT/run_$df.log; head -14 $OUT/gaps_$df.txt
done
