mkdir -p gpurun_out/r5a
cd /root/repo
./tools/probe/f16_denorm.bin > gpurun_out/r5a/f16_denorm.log 2>&1
timeout 120 ./tools/probe/pkf32_min.bin 20 > gpurun_out/r5a/pkf32_min.log 2>&1
REPS=2 bash tools/probe/lib_ab.sh hip abl3 > gpurun_out/r5a/abl3_ab.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "afe_and_lstm or motion_bank_shared or bench_parity_gate or bit_exact_beside_another" > gpurun_out/r5a/pytest_new.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/bench_full.json 2> gpurun_out/r5a/bench_full.err
tail -3 gpurun_out/r5a/pytest_new.log; cat gpurun_out/r5a/f16_denorm.log; cat gpurun_out/r5a/abl3_ab.log; tail -8 gpurun_out/r5a/pkf32_min.log
