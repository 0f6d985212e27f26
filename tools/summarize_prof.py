"""Turn the rocprofv3 CSVs of one gpurun (gpurun_out/<tag>/{stats,pmc_sq,pmc_fetch,pmc_write}) into
profiles/<tag>_summary.md + copies of the small CSVs.  Usage: python tools/summarize_prof.py r1"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:60]


def counters(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return per, {k: len(v) for k, v in disp.items()}


out = ["# rocprofv3 summary `%s`\n" % tag]
b = os.path.join(src, "bench.log")
if os.path.exists(b):
    line = open(b).read().strip().splitlines()[-1]
    out += ["## bench.py JSON line (same build)\n", "```", line, "```\n"]
    shutil.copy(b, os.path.join(dst, tag + "_bench.log"))
st = os.path.join(src, "stats", tag + "_kernel_stats.csv")
if os.path.exists(st):
    shutil.copy(st, os.path.join(dst, tag + "_kernel_stats.csv"))
    out += ["## `rocprofv3 --kernel-trace --stats` (kernel_stats.csv, top 14)\n",
            "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for i, r in enumerate(csv.DictReader(open(st))):
        if i >= 14:
            break
        out.append("| `%s` | %s | %.2f | %.1f | %s |" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                    float(r["AverageNs"]) / 1e3, r["Percentage"]))
    ig = [(int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(st)) if any(t in r["Name"] for t in ("igemm", "conv3h", "direct_conv", "dcn_patch", "dcn_pc", "pair_mlp"))]
    if ig:
        out.append("")
        out.append("All `igemm*` / `conv3h` / `direct_conv` / `dcn_patch` / `pair_mlp` kernels together: %d launches, average %.1f us (compare `roofline.avg_launch_us` of the bench line; "
                   "this pass runs the launches serialised on one stream, like bench.py's per-launch HIP-event measurement)."
                   % (sum(c for c, _ in ig), sum(t for _, t in ig) / sum(c for c, _ in ig) / 1e3))
    out.append("")
sq = os.path.join(src, "pmc_sq", "p_counter_collection.csv")
if os.path.exists(sq):
    per, n = counters(sq)
    out += ["## SQ counters per kernel (separate `--pmc` pass, 1 stream)\n",
            "wait_any = s_waitcnt/barrier parked; wait_inst = issue stalls (MFMA pipe busy / dependencies); "
            "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs).\n",
            "| kernel | launches | waves/launch | wait_any | wait_inst | active | mfma_busy | clk GHz |", "|---|---|---|---|---|---|---|---|"]
    tr = {}
    ktp = os.path.join(src, "pmc_sq", "p_kernel_trace.csv")
    dur = collections.defaultdict(float)
    for r in csv.DictReader(open(ktp)):
        dur[short(r["Kernel_Name"])] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k in sorted(per, key=lambda k: -per[k].get("SQ_WAVE_CYCLES", 0))[:12]:
        v = per[k]
        wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
        gui = max(v.get("GRBM_GUI_ACTIVE", 0), 1) / 8.0
        clk = gui / max(dur[k], 1)
        out.append("| `%s` | %d | %.0f | %.2f | %.2f | %.2f | %.2f | %.2f |" % (
            k, n[k], v.get("SQ_WAVES", 0) / n[k], v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc,
            v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024), clk))
    out.append("")
    # counters of the kernel with the largest total duration: bench.py's roofline.dominant_kernel.mfma_busy reads this file
    k0 = max((k for k in per if k), key=lambda k: dur[k])
    v = per[k0]
    gui = max(v.get("GRBM_GUI_ACTIVE", 0), 1) / 8.0
    allk = {}
    for k in per:
        vv = per[k]
        g_ = max(vv.get("GRBM_GUI_ACTIVE", 0), 1) / 8.0
        wc_ = max(vv.get("SQ_WAVE_CYCLES", 0), 1)
        allk[k] = {"launches": n[k], "mfma_busy": round(vv.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g_ * 1024), 4),
                   "wait_any": round(vv.get("SQ_WAIT_ANY", 0) / wc_, 4), "wait_inst": round(vv.get("SQ_WAIT_INST_ANY", 0) / wc_, 4)}
    json.dump({"kernel": k0, "launches": n[k0], "per_kernel": allk, "mfma_busy": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024), 4),
               "wait_any": round(v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 0), 1), 4),
               "wait_inst": round(v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 0), 1), 4),
               "source": "rocprofv3 --pmc pass of tools/prof.sh (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs))"},
              open(os.path.join(dst, tag + "_dominant_counters.json"), "w"), indent=1)
fe = os.path.join(src, "pmc_fetch", "p_counter_collection.csv")
wr = os.path.join(src, "pmc_write", "p_counter_collection.csv")
if os.path.exists(fe) and os.path.exists(wr):
    pf, nf = counters(fe)
    pw, nw = counters(wr)
    out += ["## HBM traffic per launch (FETCH_SIZE and WRITE_SIZE in separate `--pmc` passes)\n",
            "FETCH_SIZE/WRITE_SIZE are in KiB; per MI355X_MICROARCH.md §HBM the gfx950 FETCH_SIZE reads half of a wide "
            "coalesced stream, so `fetch x2` is the corrected figure; WRITE_SIZE is uncalibrated.\n",
            "| kernel | launches | fetch raw MB | fetch x2 MB | write MB |", "|---|---|---|---|---|"]
    for k in sorted(pf, key=lambda k: -pf[k]["FETCH_SIZE"])[:12]:
        out.append("| `%s` | %d | %.1f | %.1f | %.1f |" % (k, nf[k], pf[k]["FETCH_SIZE"] / nf[k] / 1024, 2 * pf[k]["FETCH_SIZE"] / nf[k] / 1024,
                                                     pw.get(k, {}).get("WRITE_SIZE", 0) / max(1, nw.get(k, 1)) / 1024))
    out.append("")
if os.path.exists(fe) and os.path.exists(wr):
    # dominant kernel family for bench.py's roofline.traffic: HBM bytes per launch, averaged over all igemm launches
    fk = [k for k in pf if any(t in k for t in ("igemm", "conv3h", "direct_conv", "dcn_patch", "dcn_pc", "pair_mlp"))]
    nl = sum(nf[k] for k in fk)
    fetch = sum(pf[k]["FETCH_SIZE"] for k in fk) * 1024.0 * 2.0          # KiB -> B, gfx950 wide-stream correction x2
    write = sum(pw.get(k, {}).get("WRITE_SIZE", 0.0) for k in fk) * 1024.0
    # the whole step: every kernel's bytes over the number of steps the run made (two sub-batch plans per step: one image-layer launch each)
    first = [k for k in pf if "direct_conv_kernel<7" in k] or [k for k in pf if "nchw_to_nhwc" in k or "preprocess_u8" in k]      # one launch per sub-batch plan
    nsteps = max(1, sum(nf[k] for k in first) // 2)
    all_bytes = sum(v["FETCH_SIZE"] for v in pf.values()) * 1024.0 * 2.0 + sum(v.get("WRITE_SIZE", 0.0) for v in pw.values()) * 1024.0
    json.dump({"kernel": "igemm* + conv3h + direct_conv + dcn_patch + pair_mlp", "launches": nl, "fetch_bytes_per_launch": fetch / max(nl, 1), "write_bytes_per_launch": write / max(nl, 1),
               "traffic_bytes_per_launch": (fetch + write) / max(nl, 1),
               "steps_in_the_run": nsteps, "traffic_bytes_per_step": all_bytes / nsteps,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/prof.sh), KiB units, FETCH_SIZE x2 per "
                         "MI355X_MICROARCH.md HBM section; WRITE_SIZE uncalibrated"},
              open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
ops = os.path.join(src, "bench_ops.json")
if os.path.exists(ops):
    d = json.load(open(ops))
    out += ["## bench.py per-entry-point HIP-event timing of one step (serialized on one stream)\n",
            "| C-ABI entry | ms/step | launches | algorithmic GFLOP | TFLOP/s |", "|---|---|---|---|---|"]
    for k, v in sorted(d["by_entry_ms_launches_flops"].items(), key=lambda kv: -kv[1][0]):
        out.append("| `%s` | %.3f | %d | %.1f | %s |" % (k, v[0], v[1], v[2] / 1e9, ("%.1f" % (v[2] / v[0] / 1e9)) if v[2] else "-"))
    out.append("")
open(os.path.join(dst, tag + "_summary.md"), "w").write("\n".join(out))
print("\n".join(out))
