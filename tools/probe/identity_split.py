import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import parity_checks as pc
from deft_amd import engine, hiplib
lib = hiplib.get_lib()
g = torch.Generator().manual_seed(5)
C = 64
x = torch.randn(1, C, 6, 8, generator=g) * torch.exp2(torch.randint(-30, 30, (1, C, 6, 8), generator=g).float())
x[0, 0, 0, 0] = 1.0 + 2.0 ** -23; x[0, 1, 0, 0] = -(2.0 - 2.0 ** -23); x[0, 2, 0, 0] = 0.0
plan = engine._Plan("cuda", lib)
xv = plan.alloc(1, 6, 8, C); pc.fill_view(xv, x)
wp, K = engine.pack_conv_weight(torch.eye(C).view(C, C, 1, 1), C)
out = plan.conv("id", xv, plan.dev(wp), K, 1, 1, 1, 0, C, None, None, False, tile=pc.T(64, 64))
plan.run(); torch.cuda.synchronize()
y = out.to_nchw().cpu()
print("prec", plan._gemms[-1][2].prec, "bit-exact identity:", torch.equal(y, x), "max rel err", float(((y - x).abs() / x.abs().clamp_min(1e-30)).max()))
