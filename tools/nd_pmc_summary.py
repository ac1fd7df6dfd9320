#!/usr/bin/env python3
"""HBM bytes fetched by the sparse block solve (k_nd_forward / k_nd_backward) per apply, from a
rocprofv3 --pmc FETCH_SIZE run of tools/bj_bench.py (2 * FETCH_SIZE KiB, see summarize_profiles.py)."""
import csv, re, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != "FETCH_SIZE":
        continue
    k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
    if not k.startswith("k_nd_"):
        continue
    acc[k][0] += 1
    acc[k][1] += float(r["Counter_Value"])
tot = 0.0
for k, (n, s) in sorted(acc.items()):
    print("%-28s %6d launches, %10.1f MB fetched per launch on average" % (k, n, 2.0 * s * 1024 / n / 1e6))
    tot += 2.0 * s * 1024
napply = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print("total %.1f MB over the run; per apply (%d applies): %.1f MB" % (tot / 1e6, napply, tot / 1e6 / napply))
