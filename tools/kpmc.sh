#!/bin/bash
# One PMC pass over a short tools/gpu_perf.py run; prints per-kernel counter averages.
# Usage (GPU box, repo root): bash tools/kpmc.sh <tag> <scene> "<counters>" [ENV=VALUE ...]
set -o pipefail
TAG=$1; SCENE=$2; CTRS=$3; shift 3
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -- python3 $R/tools/gpu_perf.py --scenes $SCENE --reps 3 > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
F=$(find "$OUT" -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:28]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if not (k.startswith('void k_p') or k.startswith('k_p')): continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, 'launches', len(next(iter(d.values()))))
PY
