"""GPU micro-benchmark of the pre-split (P3) conv kernel (igemm3.hip) against the in-loop split (igemm.hip, prec 1) on
the layer shapes of config B.   python tools/bench_p3.py [batch]
Per (shape, tile): ms and fp32-equivalent TFLOP/s of the conv launch alone (the input's P3 form is prepared once,
outside the timed region), plus a bit-exactness check against the igemm.hip result.  Tuning aid, not product path."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deft_amd import engine, hiplib  # noqa: E402

T = lambda bm, bn: (bm << 16) | bn
S3 = 1 << 29
S1 = 1 << 30
lib = hiplib.get_lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ONLY = sys.argv[2] if len(sys.argv) > 2 else ""          # run only the cases whose name contains this
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10


def timeit(plan, n=None):
    n = REPS if n is None else n
    for _ in range(2):
        plan.run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        plan.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(name, H, W, Ci, Co, k, stride, tiles, res=False):
    if ONLY and ONLY not in name:
        return
    g = torch.Generator().manual_seed(0)
    w = torch.randn(Co, Ci, k, k, generator=g) * 0.05
    wp, K = engine.pack_conv_weight(w)
    sc, sh = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    ref = None
    x0 = torch.randn(B * H * W * Ci, generator=g).cuda()
    for tile in [None] + tiles:
        if tile is not None and not isinstance(tile, tuple) and -(-Co // (tile & 0xffff)) * (tile & 0xffff) > -(-Co // 128) * 128:
            continue                                  # tile wider than the padded weight matrix
        plan = engine._Plan("cuda", lib)
        xv = plan.alloc(B, H, W, Ci)
        xv.buf.copy_(x0)
        rv = None
        if res:
            rv = plan.alloc(B, (H - 1) // stride + 1, (W - 1) // stride + 1, Co); rv.buf.fill_(0.25)
        halo = isinstance(tile, tuple)
        out = plan.conv("c", xv, plan.dev(wp), K, k, k, stride, k // 2, Co, plan.dev(sc), plan.dev(sh), True, res=rv,
                        tile=0 if tile is None else (tile[1] if halo else tile), p3=False if tile is None else ("halo" if halo else "im2col"))
        plan.finalize_p3()                       # nobody reads the P3 output here: fp32 epilogue only, like the igemm.hip launch
        plan.run(); torch.cuda.synchronize()
        if tile is None:
            ref = out.buf.clone()
        same = "ref" if tile is None else ("bit-identical" if torch.equal(out.buf, ref) else "max diff %.3e" % float((out.buf - ref).abs().max()))
        plan.ops = [plan.ops[-1]]
        fl = plan.ops[-1][3]
        ms = timeit(plan)
        d = plan._gemms[-1][2]
        if halo:
            label = "P3 halo %dx%d x %d" % ((tile[1] >> 16) & 0xfff, 16 if tile[1] >> 28 & 1 else 32, tile[1] & 0xffff)
        else:
            label = "igemm.hip (in-loop split)" if tile is None else "P3 %3dx%-3d %dst" % ((tile >> 16) & 0x1fff, tile & 0xffff, 1 if tile & S1 else (3 if tile & S3 else 2)) if tile else "P3 auto"
        print("%-30s %-28s %7.3f ms %6.1f TF/s  %s" % (name, label, ms, fl / ms / 1e9, same), flush=True)


W16 = 1 << 28
if ONLY == "small":                                     # offset convs of the small maps: halo tiles against the intra-workgroup split-K tiles
    ONLY = ""
    case("offset 3x3 256->32 @38x68", 38, 68, 256, 32, 3, 1, [("h", T(4, 32)), ("h", T(8, 32) | (1 << 28))])
    case("offset 3x3 512->32 @19x34", 19, 34, 512, 32, 3, 1, [("h", T(4, 32)), ("h", T(8, 32) | (1 << 28))])
    case("3x3 256->256 @38x68", 38, 68, 256, 256, 3, 1, [T(64, 128) | (1 << 30), ("h", T(8, 128) | (1 << 28))], res=True)
    sys.exit(0)
if ONLY == "tw16":                                      # 8 x 16 against 4 x 32 pixel halo tiles on the maps whose width is 8 mod 16
    ONLY = ""
    case("3x3 128->128 @76x136", 76, 136, 128, 128, 3, 1, [("h", T(4, 128)), ("h", T(8, 128) | W16)], res=True)
    case("head 3x3 64->256 @152x272", 152, 272, 64, 256, 3, 1, [("h", T(4, 128)), ("h", T(8, 128) | W16)])
    case("offset 3x3 64->32 @152x272", 152, 272, 64, 32, 3, 1, [("h", T(4, 32)), ("h", T(8, 32) | W16)])
    case("offset 3x3 128->32 @76x136", 76, 136, 128, 32, 3, 1, [("h", T(4, 32)), ("h", T(8, 32) | W16)])
    case("3x3 64->64 @152x272", 152, 272, 64, 64, 3, 1, [T(128, 64) | (1 << 30), ("h", T(4, 64)), ("h", T(8, 64) | W16)], res=True)
    sys.exit(0)
ALL = [T(256, 128), T(128, 128), T(128, 256), T(128, 128) | S1, T(64, 128) | S1, T(128, 64) | S1]
N64 = [T(256, 64), T(128, 64), T(64, 64), T(128, 64) | S1, T(64, 64) | S1]
H128 = [("h", T(4, 128)), ("h", T(8, 128))]
H64 = [("h", T(4, 64)), ("h", T(8, 64))]
case("3x3 64->64 @152x272", 152, 272, 64, 64, 3, 1, N64 + H64, res=True)
case("3x3 128->128 @76x136", 76, 136, 128, 128, 3, 1, ALL + H128, res=True)
case("3x3 256->256 @38x68", 38, 68, 256, 256, 3, 1, ALL + H128, res=True)
case("3x3 512->512 @19x34", 19, 34, 512, 512, 3, 1, ALL + H128)
case("head 3x3 64->256 @152x272", 152, 272, 64, 256, 3, 1, ALL + H128)
case("offset 3x3 64->32 @152x272", 152, 272, 64, 32, 3, 1, [("h", T(4, 32))])
case("offset 3x3 128->32 @76x136", 76, 136, 128, 32, 3, 1, [("h", T(4, 32))])
case("1x1 1280->512 @19x34", 19, 34, 1280, 512, 1, 1, ALL)
case("1x1 448->128 @76x136", 76, 136, 448, 128, 1, 1, ALL)
case("1x1 128->64 @152x272", 152, 272, 128, 64, 1, 1, N64)
case("3x3s2 64->128 @152x272", 152, 272, 64, 128, 3, 2, ALL)
