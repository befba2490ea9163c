#!/bin/bash
# Counter passes of one command on the GPU box, one rocprofv3 --pmc run per counter group (never combined with
# trace domains other than the kernel trace):  tools/pmc_passes.sh <out-dir> -- <program> [args...]
# Groups: 1 SQ occupancy/wait split, 2 SQ instruction mix + LDS conflicts, 3 FETCH_SIZE, 4 WRITE_SIZE, 5 L2 hit/miss,
# 6 clocks / TA busy.  PMC_ONLY="3 4 5" restricts the run to those groups.
set -o pipefail
out=$1; shift; shift
mkdir -p "$out"
export TMPDIR=/tmp
i=0
for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
    i=$((i + 1))
    if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi      # PMC_ONLY="3 4 5": these groups only
    rocprofv3 --pmc $group --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i ($group) failed"; tail -5 "$out/pass$i.log"; }
done
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:120]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: {"avg": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in agg.items()}
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for k, cs in res.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %.6g  (n=%d)" % (c, v["avg"], v["n"]))
PY
