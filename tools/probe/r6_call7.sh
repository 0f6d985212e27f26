python tools/probe/r6_det.py 2>&1 | grep -v amdgpu.ids
cat > /tmp/ring.py <<PY
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
print("affinity chain of a 32-frame step: %s  %.3f ms" % ("FUSED" if afe._pair_mlp is not None else "chain", ms))
PY
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -x -k "pair_mlp or affinity or frame_pipeline or track_similarity or co_resid or seam or detector or tracks_against or fused_run" > gpurun_out/r6c7_tests.log 2>&1; tail -3 gpurun_out/r6c7_tests.log
for rep in 1 2; do DEFT_PAIR_MLP=1 python /tmp/ring.py; DEFT_PAIR_MLP=0 python /tmp/ring.py; done 2>&1 | grep -v amdgpu.ids
