"""What corrupts a sub-batch plan when the other sub-batch plan starts beside it?  Plan 0 of config A (16 frames) runs on stream s0 from an
idle GPU; on stream s1, a moment later: nothing / a generic HBM-streaming load / the first k launches of plan 1.  Every launch output of
plan 0 is compared BIT FOR BIT with a clean run of plan 0 alone (the kernels are deterministic)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools", "probe"))
import bench  # noqa: E402
from batch_plan_parity import record_writes  # noqa: E402
from deft_amd import engine, hiplib, synth  # noqa: E402


def outputs(p):
    return [[v.buf.clone() for v in vs] for _, _, vs in p._wv]


def first_diff(p, clean):
    for i, ((kind, name, vs), cs) in enumerate(zip(p._wv, clean)):
        for v, c in zip(vs, cs):
            if not torch.equal(v.buf, c):
                d = p._op_desc.get(i)
                bad = int((v.buf != c).sum())
                return "op %d %s %s (%d words differ%s)" % (i, kind, name, bad, "" if d is None else ", splitk=%d tile=%#x p3_kernel=%d" % (d.splitk, d.tile, d.p3_kernel))
    return None


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "A"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    lib = hiplib.get_lib()
    record_writes()
    cfg = bench.CONFIGS[name]
    H, W, ds = cfg["H"], cfg["W"], cfg["dataset"]
    sd = synth.synth_state_dict(ds)
    x = torch.randn(32, 3, H, W, generator=torch.Generator().manual_seed(1000)).cuda()
    p0 = engine.DlaSegPlan(sd, 16, H, W, ds, K=100, device="cuda", lib=lib)
    p1 = engine.DlaSegPlan(sd, 16, H, W, ds, K=100, device="cuda", lib=lib)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    p0.forward(x[:16]); p1.forward(x[16:]); torch.cuda.synchronize()
    clean = outputs(p0)
    p0.forward(x[:16]); torch.cuda.synchronize()
    assert first_diff(p0, clean) is None, "plan 0 alone is not deterministic"
    big_a, big_b = torch.empty(1 << 28, device="cuda"), torch.empty(1 << 28, device="cuda")     # 1 GiB each

    def run(variant, k=None, delay=0.0):
        torch.cuda.synchronize(); time.sleep(0.02)
        with torch.cuda.stream(s0):
            p0.image.copy_(x[:16], non_blocking=True)
            p0.run()
        if delay:
            time.sleep(delay)
        with torch.cuda.stream(s1):
            if variant == "load":
                for _ in range(6):
                    big_b.copy_(big_a, non_blocking=True)
            elif variant == "plan1":
                p1.image.copy_(x[16:], non_blocking=True)
                p1._stream_cache = hiplib.stream_ptr(p1.device)
                for _, _, fn, _ in p1.ops[:k]:
                    fn()
                p1._stream_cache = None
        torch.cuda.synchronize()
        return first_diff(p0, clean)

    def run_window(a, b, reps=1):
        """plan 0 complete on s0; plan 1's launches [a, b) (`reps` times) on s1, released when plan 0 reaches launch 43."""
        torch.cuda.synchronize(); time.sleep(0.02)
        ev = torch.cuda.Event()
        with torch.cuda.stream(s0):
            p0.image.copy_(x[:16], non_blocking=True)
            p0._stream_cache = hiplib.stream_ptr(p0.device)
            for i, (_, _, fn, _) in enumerate(p0.ops):
                if i == 43:
                    ev.record(s0)
                fn()
            p0._stream_cache = None
        with torch.cuda.stream(s1):
            s1.wait_event(ev)
            p1._stream_cache = hiplib.stream_ptr(p1.device)
            for _ in range(reps):
                for _, _, fn, _ in p1.ops[a:b]:
                    fn()
            p1._stream_cache = None
        torch.cuda.synchronize()
        return first_diff(p0, clean)

    p1.forward(x[16:]); torch.cuda.synchronize()          # plan 1's buffers hold a complete pass: any window of it can be replayed
    p0.forward(x[:16]); torch.cuda.synchronize()
    assert first_diff(p0, clean) is None
    heavy = [i for i, (k, nm, _, _) in enumerate(p1.ops) if nm == "base.level4.tree1.tree1.conv2"][0]

    def run_single(i, co):
        """ONE launch of plan 0 (its inputs are the clean pass's buffers) beside `co` repetitions of a heavy launch of plan 1."""
        torch.cuda.synchronize(); time.sleep(0.005)
        ev = torch.cuda.Event()
        with torch.cuda.stream(s1):
            p1._stream_cache = hiplib.stream_ptr(p1.device)
            for _ in range(co):
                p1.ops[heavy][2]()
            p1._stream_cache = None
        with torch.cuda.stream(s0):
            p0._stream_cache = hiplib.stream_ptr(p0.device)
            p0.ops[i][2]()
            p0._stream_cache = None
        torch.cuda.synchronize()
        bad = 0
        for v, c in zip(p0._wv[i][2], clean[i]):
            bad += int((v.buf != c).sum())
        return bad

    for i in range(len(p0.ops) if "--quick" not in sys.argv else 0):
        if not p0._wv[i][2] or p0.ops[i][0] != "deft_dcn_v2_nhwc":
            continue
        res = [run_single(i, 6) for _ in range(n)]
        d = p0._op_desc.get(i)
        extra = "" if d is None else " splitk=%d tile=%#x p3_kernel=%d Cin=%d Cout=%d H=%d y3=%d" % (d.splitk, d.tile, d.p3_kernel, d.Cin, d.Cout, d.H, int(bool(d.y3)))
        if any(res):
            v = p0._wv[i][2][0]
            diff = (v.buf != clean[i][0]).nonzero().flatten()
            vals = [(int(j), float(v.buf[j]), float(clean[i][0][j])) for j in diff[:6]]
            print("plan 0 op %2d %-20s %-34s%s -> words differing per attempt %s; e.g. (index, got, want) %s" % (i, p0.ops[i][0], p0.ops[i][1], extra, res, vals), flush=True)
            p0.ops[i][2](); torch.cuda.synchronize()           # restore the clean output for the next launch's inputs
            assert all(torch.equal(v.buf, c) for v, c in zip(p0._wv[i][2], clean[i])), "launch %d is not deterministic even alone" % i
    print("done: every other launch of plan 0 reproduced its clean output beside the heavy launch", flush=True)
    # variants of one susceptible launch: arithmetic and tile
    i = [k for k, (_, nm, _, _) in enumerate(p0.ops) if nm == "dla_up.ida_0.node_1.dcn"][0]
    d = p0._op_desc[i]
    v = p0._wv[i][2][0]
    for prec, tile in (((1, 0), (1, 0x800040)) if "--quick" in sys.argv else ((1, 0), (0, 0), (1, 0x400040), (0, 0x400040), (1, 0x800040), (1, 0x800080))):
        d.prec, d.tile = prec, tile
        try:
            p0.ops[i][2](); torch.cuda.synchronize()
        except Exception as e:
            print("variant prec %d tile %#x: %s" % (prec, tile, str(e)[:80])); continue
        ref = v.buf.clone()
        res, where = [], None
        for _ in range(n):
            run_single(i, 6)
            diff = (v.buf != ref)
            res.append(int(diff.sum()))
            if where is None and res[-1]:
                pix = diff.view(-1, d.ldy).any(1).nonzero().flatten()          # bad pixels (GEMM rows m)
                bm = 64 if tile == 0 else (tile >> 16)
                rows = sorted(set(int(m) % bm for m in pix))
                tiles = sorted(set(int(m) // bm for m in pix))
                where = "bad pixels %d in %d tiles (of %d); rows inside the tile: %s" % (len(pix), len(tiles), d.M // bm, rows[:40])
        print("variant prec %d tile %#x: words differing %s  %s" % (prec, tile, res, where or ""), flush=True)


if __name__ == "__main__":
    main()
