# hipStreamEndCapture segfault hunt: the multi-stream capture test in a loop; the first failing log is printed
for i in $(seq 1 ${N:-24}); do
  timeout 100 python -X faulthandler -m pytest tests/test_gpu_parity.py -q -x -k "dataflow" > /tmp/td_$i.log 2>&1
  rc=$?
  echo "iter $i rc=$rc $(tail -1 /tmp/td_$i.log | cut -c1-80)"
  if [ $rc -ne 0 ]; then grep -n -A12 "Fatal Python error\|Error\|assert" /tmp/td_$i.log | cut -c1-200 | head -40; break; fi
done
