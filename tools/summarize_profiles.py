#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into the two files kept under
profiles/: the per-kernel duration statistics and the HBM traffic per launch
(2 * FETCH_SIZE + WRITE_SIZE, KiB -> bytes; FETCH_SIZE counts half of the bytes of wide
coalesced reads on gfx950, see MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import glob
import json
import os
import re
import sys

out, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)


def find(d, pattern):
    hits = glob.glob(os.path.join(out, d, "**", pattern), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s/%s" % (pattern, out, d))
    return hits[0]


# 1. kernel statistics: copy the rocprofv3 summary as it is (names shortened)
rows = list(csv.reader(open(find("prof_stats", "*kernel_stats.csv"))))
dst = os.path.join(root, "profiles", "r04_bench_%s_kernel_stats.csv" % workload)
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        w.writerow([short(r[0])] + r[1:])
print("wrote", dst)

# 2. counters
raw = {}
for d, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    acc = {}
    with open(find(d, "*counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    raw[counter] = {k: {"launches": n, "avg_KiB": s / n} for k, (n, s) in acc.items()}
summary = {}
for k in raw["FETCH_SIZE"]:
    fe = raw["FETCH_SIZE"][k]["avg_KiB"]
    wr = raw["WRITE_SIZE"].get(k, {"avg_KiB": 0.0})["avg_KiB"]
    summary[k] = {"traffic_bytes_per_launch": (2.0 * fe + wr) * 1024.0,
                  "read_bytes_per_launch": 2.0 * fe * 1024.0, "write_bytes_per_launch": wr * 1024.0}
spmm = [k for k in summary if k.startswith("k_spmm")]
spmm_pick = next((k for k in spmm if "gram" in k), spmm[0] if spmm else None)
bench = json.loads(open(os.path.join(out, "prof_stats_bench.json")).read().strip().splitlines()[-1])
doc = {
    "command": "tools/profile_bench.sh: rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace "
               "--output-format csv -- python3 bench.py --workload %s --steps 10 --warmup 2 --no-cpu --spmm-reps 10 --survey-nparts 0 --phase-iters 0" % workload,
    "workload": bench["config"]["workload"],
    "units": "counter values are KiB; FETCH_SIZE is doubled (gfx950 reports half of the bytes of wide coalesced "
             "reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE is exact",
    "calibration": "k_probe_copy moves 1 GiB in and 1 GiB out per launch, k_update_z reads 3 panels and writes 1: "
                   "compare their rows below with those byte counts",
    # the solver's product (with the Gram block when the run formed it) is the one bench.py's roofline block times
    "k_spmm": summary[spmm_pick] if spmm else None,
    "k_spmm_name": spmm_pick if spmm else None,
    "k_spmm_plain": next((dict(summary[k], name=k) for k in spmm if k != spmm_pick), None),
    # the two kinds of block-solve launch separately: plain, and the solver's (which also forms [AP | AP_prev]^T Z)
    "k_bj": next((dict(summary[k], name=k) for k in summary if (k.startswith("k_bj_g4<") and "true" not in k) or k.startswith("k_bj_apply")), None),
    "k_bj_gram": next((dict(summary[k], name=k) for k in summary if k.startswith("k_bj_g4<") and "true" in k), None),
    "per_kernel": summary,
    "raw": raw,
    "bench_line_of_the_profiled_run": bench,
}
dst = os.path.join(root, "profiles", "r04_pmc_hbm_traffic_%s.json" % workload)
with open(dst, "w") as f:
    json.dump(doc, f, indent=1)
print("wrote", dst)
