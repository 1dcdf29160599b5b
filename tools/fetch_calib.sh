#!/bin/bash
# FETCH_SIZE against known byte counts: a streamed table and random 64-byte gathers (tools/micro/fetch_calib.hip).
# Usage (GPU box, repo root): bash tools/fetch_calib.sh > gpurun_out/fetch_calib.txt   (the binary is built beforehand by __graft_entry__.build())
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
BIN=$R/tools/micro/fetch_calib
OUT=$R/gpurun_out/fetch_calib
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # tag, args...
    local tag=$1; shift
    rm -rf "$OUT/$tag"
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$tag" -- "$BIN" "$@" > "$OUT/$tag.log" 2>&1 || { echo "$tag: pass failed"; tail -3 "$OUT/$tag.log"; return 1; }
    python3 - "$OUT/$tag" "$OUT/$tag.log" "$tag" <<'PY'
import csv, glob, json, sys
d, log, tag = sys.argv[1:4]
asked = None
for line in open(log):
    if line.startswith("{"):
        asked = json.loads(line)["bytes_asked_for"]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and ("k_stream" in r["Kernel_Name"] or "k_gather64" in r["Kernel_Name"])]
kib = sum(float(r["Counter_Value"]) for r in rows)
print(f"{tag:28s} asked for {asked / 1e6:10.1f} MB   FETCH_SIZE {kib * 1024 / 1e6:10.1f} MB   asked / reported = {asked / (kib * 1024):.3f}")
PY
}
run stream_2048mib stream 2048 && run stream_64mib stream 64 && run gather64_2048mib gather64 2048 64 && run gather64_512mib gather64 512 64 && run gather64_11mib gather64 11 256
