"""Mine well-conditioned frames for bench.py's DECIDABLE gate stream -- from the oracle alone (no GPU, no device result is looked at).

    python tools/mine_gate_seeds.py CONFIG [want] [max_seeds] [threads]     ->  tests/golden/gate_seeds.json (merged)

A frame is `torch.randn(1, 3, H, W, generator=manual_seed(seed))` through the config's synthetic net (the frames of SURVEY.md 8(d)); its
margin is bench.oracle_margin of the ORACLE's heat map: every heat map within margin / 2 of the oracle's decodes to the same ORDERED top-K
(class, index) list.  A random-weight net draws K = 100 "detections" out of noise, and the smallest of their ~300 gaps / NMS margins is
usually below the ~1e-4 logit error any two correct fp32 implementations have against each other (profiles/r6_gate_margins.md: median margin
7e-5 over 40 frames): ordered equality is a decidable question only on the few frames whose margin clears twice that error.  Those are what
this script looks for (margin >= MINE_MARGIN, comfortably above bench.GATE_MARGIN); bench.py re-derives the margin from its own oracle run and
demands ordered equality only where it still holds.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import deft_oracle as O  # noqa: E402

MINE_MARGIN = 3.5e-4
SEED0 = 5000


def main():
    name = sys.argv[1]
    want = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    torch.set_num_threads(int(sys.argv[4]) if len(sys.argv) > 4 else 2)
    cfg = bench.CONFIGS[name]
    H, W, ds = cfg["H"], cfg["W"], cfg["dataset"]
    sd = O.synth_state_dict(ds)
    found, margins, t0 = [], [], time.time()
    for seed in range(SEED0, SEED0 + nmax):
        x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(seed))
        with torch.no_grad():
            out, _ = O.dlaseg_forward(x, sd, ds)
        m = bench.oracle_margin(out["hm"][0], bench.KDET)
        margins.append(m)
        if m >= MINE_MARGIN:
            found.append({"seed": seed, "margin": round(m, 7)})
            print(name, "seed", seed, "margin %.2e" % m, "(%d of %d seeds, %.0f s)" % (len(found), seed - SEED0 + 1, time.time() - t0), flush=True)
            if len(found) >= want:
                break
    path = os.path.join(ROOT, "tests", "golden", "gate_seeds.json")
    allj = json.load(open(path)) if os.path.exists(path) else {}
    ms = sorted(margins)
    allj[name] = {"frames": found, "H": H, "W": W, "dataset": ds, "mine_margin": MINE_MARGIN, "seeds_tried": len(margins), "first_seed": SEED0,
                  "margin_median_of_tried": ms[len(ms) // 2], "margin_p90_of_tried": ms[int(0.9 * len(ms))],
                  "generator": "tools/mine_gate_seeds.py (oracle only); frame = torch.randn(1,3,H,W, generator=manual_seed(seed))"}
    json.dump(allj, open(path, "w"), indent=1)
    print(name, "done:", found)


if __name__ == "__main__":
    main()
