#!/usr/bin/env python3
"""Gaps between consecutive kernels of the steady iteration loop, from a rocprofv3 kernel trace
(`*kernel_trace.csv`): per pair of kernels the mean / min / max idle time between the end of one and the start of
the next, and the span per iteration.  usage: r4_graph_gaps.py trace.csv [anchor-kernel-substring]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_update_z"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    return re.sub(r"[<(].*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
idx = [i for i, r in enumerate(rows) if anchor in nm(r)]
seg = rows[idx[len(idx) // 4]:idx[-len(idx) // 8]]
gaps = collections.defaultdict(list)
for a, b in zip(seg, seg[1:]):
    gaps[(nm(a), nm(b))].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
tot = 0.0
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) * 4 < len(idx) // 2:
        continue
    v.sort()
    med = v[len(v) // 2]
    tot += med
    print("%-50s n=%4d median %6.2f us  mean %6.2f  max %7.2f" % (k[0] + " -> " + k[1], len(v), med / 1e3, sum(v) / len(v) / 1e3, v[-1] / 1e3))
n = sum(anchor in nm(r) for r in seg)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
print("sum of median gaps %.2f us; per iteration: span %.1f us, kernels %.1f us, idle %.1f us (%d iterations)" % (tot / 1e3, span / 1e3 / n, busy / 1e3 / n, (span - busy) / 1e3 / n, n))
