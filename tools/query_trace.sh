#!/bin/bash
# Kernel durations of tools/gpu_query_rate.py under rocprofv3: bash tools/query_trace.sh <tag> [args of the tool]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/q_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $R/tools/gpu_query_rate.py "$@" > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
grep "^batch\|^rays" "$OUT/run.log" | tail -3
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
grep k_query "$F" | cut -c1-30,200-400 | head -3
python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_query" in r["Name"]:
        print("k_query calls", r["Calls"], "avg ms", float(r["AverageNs"]) / 1e6, "min ms", float(r["MinNs"]) / 1e6)
PY
