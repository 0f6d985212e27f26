"""In-kernel phase timing of the halo conv (conv3h_kernel built with -DP3H_TIMING: tools/build_variant.sh p3htime igemm3.hip -DP3H_TIMING).
    DEFT_HIP_LIB=$PWD/deft_amd/lib/libdeft_p3htime.so python tools/probe/p3h_timing.py
Per shape: mean cycles per K interval and wave in each phase (DMA wait, barrier, DMA issue, fragment reads, MFMA issue) and the loop total."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from deft_amd import engine, hiplib  # noqa: E402

T = lambda bm, bn: (bm << 16) | bn
W16 = 1 << 28
lib = hiplib.get_lib()


def case(name, B, H, W, Ci, Co, tile, waves):
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    wp, K = engine.pack_conv_weight(w)
    sc, sh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    plan = engine._Plan("cuda", lib)
    xv = plan.alloc(B, H, W, Ci)
    xv.buf.copy_(torch.randn(B * H * W * Ci, generator=g).cuda())
    plan.conv("c", xv, plan.dev(wp), K, 3, 3, 1, 1, Co, plan.dev(sc), plan.dev(sh), True, tile=tile, p3="halo")
    plan.finalize_p3()
    plan.run(); torch.cuda.synchronize()
    d = plan._gemms[-1][2]
    th, tw, bn = (tile >> 16) & 0xfff, 16 if tile & W16 else 32, tile & 0xffff
    grid = B * -(-H // th) * -(-W // tw) * -(-Co // bn)
    ws = torch.zeros(grid * waves * 8, dtype=torch.int64, device="cuda")
    d.ws = ws.data_ptr()
    plan.ops = [plan.ops[-1]]
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
    t = ws.view(grid * waves, 8).double().cpu()
    nI = t[:, 6].mean().item()
    per = (t[:, :5] / t[:, 6:7]).mean(0).tolist()
    tot = (t[:, 5] / t[:, 6]).mean().item()
    span = (t[:, 7].max() - t[:, 7].min()).item()
    print("%-34s %4d WGs x %d waves, %3d intervals, %.1f us | per interval: DMA wait %4.0f  barrier %4.0f  DMA issue %4.0f  fragment reads %4.0f  MFMA issue %4.0f  = %4.0f of %4.0f cycles | first-to-last wave start %.0f cycles"
          % (name, grid, waves, nI, e0.elapsed_time(e1) * 1e3, per[0], per[1], per[2], per[3], per[4], sum(per), tot, span), flush=True)


case("128->128 @76x136  x16 frames", 16, 76, 136, 128, 128, T(8, 128) | W16, 4)
case("128->128 @76x136  x1 frame", 1, 76, 136, 128, 128, T(8, 128) | W16, 4)
case("64->64 @152x272   x16 frames", 16, 152, 272, 64, 64, T(8, 64) | W16, 4)
case("64->64 @152x272   x1 frame", 1, 152, 272, 64, 64, T(8, 64) | W16, 4)
case("64->256 @152x272  x16 frames", 16, 152, 272, 64, 256, T(8, 128) | W16, 4)
