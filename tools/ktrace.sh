#!/bin/bash
# Kernel-trace of a short tools/gpu_perf.py run: per-kernel durations of one frame.
# Usage (GPU box, repo root): bash tools/ktrace.sh <tag> <scene> [ENV=VALUE ...]
set -o pipefail
TAG=$1; SCENE=$2; shift 2
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kt_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $R/tools/gpu_perf.py --scenes $SCENE > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
