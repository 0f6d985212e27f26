"""Kernel timeline of the last steps of a latency-mode run (rocprofv3 --kernel-trace CSV): per step the span, the sum of kernel
durations, the time no kernel runs, and the largest gaps with the kernels on either side."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48],
                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)) *
                 max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1)), int(r.get("LDS_Block_Size", 0) or 0)))
rows.sort()
# steps: split at the first kernel of the plan (the layout kernel of the input image)
first = [i for i, r in enumerate(rows) if "nchw_to_nhwc" in r[2]]
steps = [rows[a:b] for a, b in zip(first[-12:-1], first[-11:])]
tot = []
for st in steps:
    span = st[-1][1] - st[0][0] if st else 0
    busy, idle, cur_end = 0, 0, st[0][0]
    for s, e, *_ in st:
        if s > cur_end:
            idle += s - cur_end
        cur_end = max(cur_end, e)
    ksum = sum(e - s for s, e, *_ in st)
    tot.append((len(st), span / 1e3, ksum / 1e3, idle / 1e3))
print("kernels/step, span us (first start -> last end), sum of kernel durations us, idle us (no kernel running)")
for t in tot:
    print("%4d %9.1f %9.1f %9.1f" % t)
st = steps[-1]
gaps = []
cur_end, cur_name = st[0][1], st[0][2]
for s, e, n, *_ in st[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n))
    if e > cur_end:
        cur_end, cur_name = e, n
gaps.sort(reverse=True)
print("largest gaps of the last step (us, kernel before, kernel after):")
for g, a, b in gaps[:25]:
    print("%7.1f  %-48s -> %s" % (g / 1e3, a, b))
import collections
hist = collections.Counter(min(int(g / 1e3), 20) for g, _, _ in gaps)
print("gap histogram (us -> count):", sorted(hist.items()))
print("step-to-step: start of step k+1 minus end of step k (us):", [round((b[0][0] - a[-1][1]) / 1e3, 1) for a, b in zip(steps[:-1], steps[1:])])

if len(sys.argv) > 2:
    print("every launch of the last step: start us (from the step's first kernel), duration us, workgroups, LDS bytes, kernel")
    t0 = st[0][0]
    for s, e, n, wg, lds in st:
        print("%8.1f %7.1f %6d %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, wg, lds, n))
