#!/bin/bash
# rocprofv3 counter passes for one command, ON THE GPU BOX:  tools/pmc.sh <tag> <kernel-name regex> -- cmd...
# Writes gpurun_out/<tag>/pmc{1,2,3}.csv (per-kernel means of the counters, kernels matching the regex).  Counter passes run with
# --kernel-trace only.
set -u
TAG=$1; RE=$2; shift 3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
if [ -n "${PMC_SETS:-}" ]; then IFS=';' read -ra SETS <<< "$PMC_SETS"; else SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM"); fi
for set in "${SETS[@]}"; do
    i=$((i+1))
    rm -rf "$OUT/p$i"
    rocprofv3 --kernel-trace --pmc $set -d "$OUT/p$i" -o p --output-format csv -- bash -c "cd $ROOT && $*" > "$OUT/p$i.log" 2>&1
    f=$(find "$OUT/p$i" -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then
        python3 - "$f" "$RE" > "$OUT/pmc$i.txt" <<'PY'
import csv, re, sys, collections
f, rx = sys.argv[1], re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if not rx.search(k): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k[:90])
    for c, v in d.items():
        print("   %-28s %16.1f  (mean of %d dispatches)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
    else
        tail -5 "$OUT/p$i.log" > "$OUT/pmc$i.txt"
    fi
    rm -rf "$OUT/p$i"
done
cd "$ROOT"; cat "$OUT"/pmc*.txt
