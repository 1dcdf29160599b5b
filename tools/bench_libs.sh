#!/bin/bash
# bench.py --only-timed on one workload for several builds of the library (PT_AMD_LIBRARY); usage: bench_libs.sh <workload> <extra bench args> -- lib...
wl=$1; shift
extra=()
while [ "$1" != "--" ]; do extra+=("$1"); shift; done
shift
for lib in "$@"; do
  out=gpurun_out/libs_${wl}_$(basename $lib .so).json
  PT_AMD_LIBRARY=$PWD/$lib timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --only-timed "${extra[@]}" > $out 2> ${out%.json}.err || { echo "FAILED $lib"; tail -5 ${out%.json}.err; exit 1; }
  echo "$(basename $lib) $(tail -1 $out)"
done
