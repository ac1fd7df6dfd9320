#!/bin/bash
# rocprofv3 kernel-trace statistics of one python script (run on the GPU box from the repo root):
#   tools/trace_stats.sh <out-dir-under-gpurun_out> <script.py> [args...]
# prints the top of the per-kernel summary and leaves the csv under gpurun_out/<dir>/.
set -o pipefail
export TMPDIR=/tmp
D=gpurun_out/$1; shift
mkdir -p "$D"
rocprofv3 --kernel-trace --stats -d "$D"/prof -o s --output-format csv -- python3 "$@" > "$D"/run.log 2>&1
tail -3 "$D"/run.log
f=$(find "$D"/prof -name "*kernel_stats.csv" | head -1)
cp "$f" "$D"/kernel_stats.csv
python3 - "$f" <<'PY'
import csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))
print(rows[0])
for r in rows[1:16]:
    print(re.sub(r"\(anonymous namespace\)::|void ", "", r[0])[:70], r[1:8])
PY
