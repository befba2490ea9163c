#!/bin/bash
# Counter passes of one command on the GPU box, one rocprofv3 --pmc run per counter group (never combined with
# trace domains other than the kernel trace):  tools/pmc_passes.sh <out-dir> -- <program> [args...]
# Groups: 1 SQ occupancy/wait split, 2 SQ instruction mix + LDS conflicts, 3 FETCH_SIZE, 4 WRITE_SIZE, 5 L2 hit/miss,
# 6 clocks / TA busy.  PMC_ONLY="3 4 5" restricts the run to those groups.
# <program> must be the program ITSELF -- `python3 script.py ...` or a compiled binary: under --pmc the profiler's preloaded
# library initialises the GPU before the program starts, so a wrapper in that place (env, bash -c, taskset, numactl, a
# `#!/usr/bin/env` script) is an exec AFTER GPU initialisation, which takes the machine down on this pool.  Pass settings
# through exported environment variables instead (SGX_PROBE_ONLY=lds tools/pmc_passes.sh out -- python3 tools/xw_sparse_probe.py).
# Exits non-zero when any requested pass fails: a partial summary must not be joined into profiles/*.jsonl.
set -o pipefail
out=$1; shift
if [ "$1" != "--" ] || [ -z "$2" ]; then echo "usage: tools/pmc_passes.sh <out-dir> -- <program> [args...]" >&2; exit 64; fi
shift
prog=$(basename -- "$1")
case "$prog" in
    env|bash|sh|dash|zsh|taskset|numactl|nice|timeout|stdbuf|nohup|time|xargs|sudo)
        echo "pmc_passes.sh: '$1' is a wrapper, not the profiled program: put the program itself after -- (python3 <script> or a binary)" >&2; exit 64;;
esac
resolved=$(command -v -- "$1" || true)
if [ -z "$resolved" ]; then echo "pmc_passes.sh: '$1' not found" >&2; exit 64; fi
if [ "$(head -c 4 "$resolved" | od -An -c | tr -d ' ')" != '177ELF' ]; then
    echo "pmc_passes.sh: '$1' is not an ELF program (a script would be started through an interpreter hop): use python3 <script>" >&2; exit 64
fi
mkdir -p "$out"
export TMPDIR=/tmp
i=0
failed=""
for group in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
    i=$((i + 1))
    if [ -n "$PMC_ONLY" ] && ! echo " $PMC_ONLY " | grep -q " $i "; then continue; fi      # PMC_ONLY="3 4 5": these groups only
    rocprofv3 --pmc $group --output-format csv -d "$out/pass$i" -- "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i ($group) failed"; tail -5 "$out/pass$i.log"; failed="$failed $i"; }
done
if [ -n "$failed" ]; then echo "pmc_passes.sh: passes$failed failed -- no summary written" >&2; exit 1; fi
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
