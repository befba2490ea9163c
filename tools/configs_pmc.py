#!/usr/bin/env python3
"""Joins the timings of tools/bench_configs.py with the counter passes of the same aggregation launches
(tools/pmc_passes.sh over `bench_configs.py --pmc-launches 3`, groups 3-5: FETCH_SIZE, WRITE_SIZE, L2 hit / miss) into
one record per configuration: algorithmic bytes and the fraction of the 8 TB/s HBM roofline they give, next to the
bytes the L2 actually requested from the memory side and THEIR fraction -- SURVEY 8d: when the measured bytes are below
the algorithmic ones (cache-resident table, hub rows served by L2) the measured-bytes fraction is the binding number.

    python tools/configs_pmc.py gpurun_out/configs_r02.jsonl gpurun_out/pmc_cfg_uniform gpurun_out/pmc_cfg_rmat > profiles/r02_configs.jsonl

FETCH_SIZE correction (MI355X_MICROARCH.md "HBM" + tools/pmc_calibrate.py): 16-byte-per-lane reads (the row gathers)
are tallied at half their bytes, the narrow streams (4-byte column index, 2-byte value, row pointers) and all writes
1:1; FETCH counts L2 -> fabric requests, Infinity-Cache hits included (DRAM bytes are not observable).
"""
import collections
import csv
import glob
import json
import sys

PEAK = 8.0e12


def dispatches(pmc_dir, group):
    files = glob.glob(f"{pmc_dir}/pass{group}/**/*_counter_collection.csv", recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if "GLOBAL__N_1" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def per_config(pmc_dir):
    """{config key: {counter: bytes or count per aggregation call}} -- calls are told apart by their order: c3's three
    A.H launches, c4's three, then c5's GAT calls (one head x 3, eight heads x 3) and its three plain aggregates.  A
    call = its opening kernel (spmm_kernel | gat_scores_kernel | gat_scores_heads_kernel) and the helper launches behind
    it (split tasks, finalize)."""
    out = collections.defaultdict(dict)
    for group, scale in ((3, 1024.0), (4, 1024.0), (5, 1.0)):
        by_dispatch = collections.OrderedDict()
        for did, name, counter, value in dispatches(pmc_dir, group):
            by_dispatch.setdefault(did, (name, {}))[1][counter] = value * scale
        calls = collections.defaultdict(list)
        cur = None
        for _did, (name, counters) in by_dispatch.items():
            if "spmm_kernel" in name:
                fam = "spmm"
            elif "gat_scores_heads" in name:
                fam = "gat8"
            elif "gat_scores_kernel" in name:
                fam = "gat1"
            elif "spmm_split_finalize" in name or "gat_" in name:
                fam = None                               # a helper of the call that is open
            else:
                cur = None                               # another kernel (X.W ...): closes the call
                continue
            if fam is not None:
                cur = collections.defaultdict(float)
                calls[fam].append(cur)
            if cur is not None:
                for c, v in counters.items():
                    cur[c] += v
        spmm = calls.get("spmm", [])
        groups = {"c3": spmm[0:3], "c4": spmm[3:6], "c5_plain": spmm[6:9], "c5_gat1": calls.get("gat1", []),
                  "c5_gat8": calls.get("gat8", [])}
        for key, lst in groups.items():
            if lst:
                for counter in lst[0]:
                    out[key][counter] = sum(d[counter] for d in lst) / len(lst)
    return out


def main():
    timings = [json.loads(ln) for ln in open(sys.argv[1]) if ln.startswith("{")]
    pmc = {"uniform": per_config(sys.argv[2]), "rmat": per_config(sys.argv[3])}
    for t in timings:
        name = t["config"]
        gen = "rmat" if "(rmat)" in name else "uniform" if "(uniform)" in name else None
        if gen is None:
            print(json.dumps(t))
            continue
        e, n = t["edges"], t["nodes"]
        stream = e * 6 + (n + 1) * 4

        def measured(c):
            if not c:
                return None
            read = 2 * (c.get("FETCH_SIZE", 0.0) - stream) + stream
            return {"FETCH_SIZE_bytes_raw": c.get("FETCH_SIZE"), "WRITE_SIZE_bytes": c.get("WRITE_SIZE"),
                    "L2_hit_requests": c.get("TCC_HIT_sum"), "L2_miss_requests": c.get("TCC_MISS_sum"),
                    "memory_side_read_bytes_corrected": read, "memory_side_bytes": read + c.get("WRITE_SIZE", 0.0)}
        if name.startswith("c3") or name.startswith("c4"):
            key = name[:2]
            m = measured(pmc[gen].get(key))
            ms = t["ms_agg1"]
            b_alg = t.get("agg1_algorithmic_bytes") or (e * (6 + t["hidden"] * 2) + (n + 1) * 4 + n * t["hidden"] * 2)
            t["agg1"] = {"ms": ms, "algorithmic_bytes": b_alg, "frac_of_8TBps_algorithmic": b_alg / (ms * 1e-3) / PEAK}
            if m:
                t["agg1"].update(m)
                t["agg1"]["frac_of_8TBps_measured_bytes"] = m["memory_side_bytes"] / (ms * 1e-3) / PEAK
                t["agg1"]["measured_over_algorithmic"] = m["memory_side_bytes"] / b_alg
                t["agg1"]["binding"] = ("measured bytes (below the algorithmic count: table rows served by L2 / Infinity Cache)"
                                        if m["memory_side_bytes"] < b_alg else "algorithmic bytes")
        elif name.startswith("c5 ogbn-arxiv"):
            P = t["width"]
            for key, ms_key, label in (("c5_gat1", "ms_gat_aggregate", "gat_one_head"), ("c5_gat8", "ms_gat_aggregate_8_heads", "gat_8_heads"),
                                       ("c5_plain", "ms_gcn_aggregate_same_shape", "plain_aggregate")):
                m = measured(pmc[gen].get(key))
                ms = t[ms_key]
                b_alg = e * (6 + (8 if "gat" in label else 0) + P * 2) + (n + 1) * 4 + n * P * 2
                rec = {"ms": ms, "algorithmic_bytes": b_alg, "frac_of_8TBps_algorithmic": b_alg / (ms * 1e-3) / PEAK}
                if m:
                    rec.update(m)
                    rec["frac_of_8TBps_measured_bytes"] = m["memory_side_bytes"] / (ms * 1e-3) / PEAK
                    rec["measured_over_algorithmic"] = m["memory_side_bytes"] / b_alg
                t[label] = rec
        print(json.dumps(t))


if __name__ == "__main__":
    main()
