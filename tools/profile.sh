#!/bin/bash
# rocprofv3 passes over bench.py: two kernel traces (frames in flight as the bench runs them, and one frame at a time)
# + separate PMC passes (SQ x3, TCC fetch, TCC write).
# Usage (on the GPU box, from the repo root): bash tools/profile.sh <tag> [bench args...]
# Results land under gpurun_out/prof_<tag>/ ; tools/summarize_profile.py condenses them into profiles/.
# Every pass runs `bench.py --only-timed --repeats 1`: the last 20 frames of a trace are the bench's timed region.
set -o pipefail
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH3="python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --only-timed --no-cpu-baseline $*"           # frames in flight: the bench's default
BENCH="python3 $R/bench.py --steps 5 --warmup 3 --repeats 1 --only-timed --no-cpu-baseline --frames-in-flight 1 $*"   # one frame at a time: kernels alone on the chip
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_fif" -- $BENCH3 > "$OUT/trace_fif.log" 2>&1 || { echo "trace (frames in flight) pass failed"; tail -5 "$OUT/trace_fif.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F32 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1 || { echo "sq2 pass failed (optional)"; tail -3 "$OUT/pmc_sq2.log"; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INST_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_sq3" -- $BENCH > "$OUT/pmc_sq3.log" 2>&1 || { echo "sq3 pass failed (optional)"; tail -3 "$OUT/pmc_sq3.log"; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
find "$OUT" -name "*.csv" | sort
