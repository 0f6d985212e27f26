# hipStreamEndCapture segfault hunt: the multi-stream capture test (small plan) in a loop; the first failing log is printed
for i in $(seq 1 ${N:-24}); do
  timeout 100 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -x -s -k "dataflow and 96" > /tmp/td_$i.log 2>&1
  rc=$?
  echo "iter $i rc=$rc $(grep -o 'ops per stream.*' /tmp/td_$i.log | head -1) $(tail -1 /tmp/td_$i.log | cut -c1-50)"
  if [ $rc -ne 0 ]; then grep -n -A8 "Fatal Python error\|Error\|assert" /tmp/td_$i.log | cut -c1-160 | head -24; break; fi
done
