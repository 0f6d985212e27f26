# round 6, call 6: the fused pair MLP on the hardware: parity, then the affinity chain of the 32-frame step fused vs the four-launch chain
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "pair_mlp or affinity or frame_pipeline or track_similarity" > gpurun_out/r6c6_tests.log 2>&1; tail -3 gpurun_out/r6c6_tests.log
cat > /tmp/ring.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from deft_amd import engine, hiplib, synth
lib = hiplib.get_lib()
sd = synth.synth_state_dict("mot")
afe = engine.AfePlan(sd, 100, "cuda", lib)
R, K, Bc, H = 37, 100, 32, 5
ring = (torch.rand(R, K, afe.D, generator=torch.Generator().manual_seed(0)) * 3).cuda().contiguous()
for _ in range(3): afe.affinity_ring(ring, H, Bc, H)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): afe.affinity_ring(ring, H, Bc, H)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
fl = engine.AfePlan.affinity_flops(H * K, K, afe.D) * Bc
print("affinity chain of a 32-frame step (1.6 M pairs): %s  %.3f ms  %.1f TFLOP/s" % ("FUSED deft_pair_mlp" if afe._pair_mlp is not None else "four-launch chain", ms, fl / ms / 1e9))
PY
for rep in 1 2; do DEFT_PAIR_MLP=1 python /tmp/ring.py; DEFT_PAIR_MLP=0 python /tmp/ring.py; done 2>&1 | grep -v amdgpu.ids
for v in 1 0; do DEFT_PAIR_MLP=$v timeout 900 python bench.py --steps 30 --warmup 3 --no-extras --no-cpu-baseline --no-check 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('DEFT_PAIR_MLP=$v', j['value'], j['ms_per_step'], j['roofline']['frac'])"; done
