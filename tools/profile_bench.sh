#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline block refers to (run on the GPU box, from
# the repo root):  kernel-trace statistics of the bench command, then HBM traffic counters in
# two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) as MI355X_MICROARCH.md prescribes.
# Raw output goes to gpurun_out/prof_*; tools/summarize_profiles.py turns it into the files
# committed under profiles/.
set -e -o pipefail
export TMPDIR=/tmp
OUT=${1:-gpurun_out}
WORKLOAD=${2:-elasticity}
rm -rf "$OUT"/prof_stats "$OUT"/prof_fetch "$OUT"/prof_write
rocprofv3 --kernel-trace --stats -d "$OUT"/prof_stats -o stats --output-format csv -- \
  python3 bench.py --workload "$WORKLOAD" --steps 50 --warmup 5 --no-cpu --survey-nparts 0 > "$OUT"/prof_stats_bench.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT"/prof_fetch -o fetch --output-format csv -- \
  python3 bench.py --workload "$WORKLOAD" --steps 10 --warmup 2 --no-cpu --spmm-reps 10 --survey-nparts 0 --phase-iters 0 > "$OUT"/prof_fetch_bench.json
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT"/prof_write -o write --output-format csv -- \
  python3 bench.py --workload "$WORKLOAD" --steps 10 --warmup 2 --no-cpu --spmm-reps 10 --survey-nparts 0 --phase-iters 0 > "$OUT"/prof_write_bench.json
python3 tools/summarize_profiles.py "$OUT" "$WORKLOAD"
