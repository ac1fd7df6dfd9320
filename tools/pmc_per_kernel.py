#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc run collected:
    tools/pmc_per_kernel.py <counter_collection.csv> [kernel-name prefix]
One line per kernel: launches, then every counter's sum over the device divided by the launches."""
import csv, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
prefix = sys.argv[2] if len(sys.argv) > 2 else "k_"
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
    if not k.startswith(prefix):
        continue
    c = acc[k][r["Counter_Name"]]
    c[0] += 1
    c[1] += float(r["Counter_Value"])
for k in sorted(acc):
    parts = []
    n = 0
    for name, (cnt, s) in sorted(acc[k].items()):
        n = max(n, cnt)
        parts.append("%s %.0f" % (name, s / cnt))
    print("%-34s %5d launches  %s" % (k, n, "  ".join(parts)))
