#!/bin/bash
# WRITE_SIZE of k_paths per launch for a given library: bash tools/kpmc_write.sh <tag> <library> [bench args]
set -o pipefail
TAG=$1; LIB=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/kw_$TAG
mkdir -p "$OUT"
export PT_AMD_LIBRARY=$R/$LIB
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT" -- python3 $R/bench.py --steps 5 --warmup 1 --repeats 1 --only-timed --no-cpu-baseline --frames-in-flight 1 "$@" > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
vals = {}
for r in csv.DictReader(open(f)):
    if "k_paths" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
        vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print(sys.argv[1].split("/")[-1], [round(v) for v in vals.values()])
PY
