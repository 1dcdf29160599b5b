#!/bin/bash
# Three workloads through bench.py (short form), results into gpurun_out/<tag>_<workload>.json; prints the step times.
tag=${1:-x}
set -e
timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 5 > gpurun_out/${tag}_boxes.json 2> gpurun_out/${tag}_boxes.err
timeout -k 10 200 python bench.py --workload scene_glb_textured --no-cpu-baseline --repeats 3 > gpurun_out/${tag}_glb.json 2> gpurun_out/${tag}_glb.err
timeout -k 10 200 python bench.py --workload instanced_boxes --no-cpu-baseline --repeats 1 --steps 10 > gpurun_out/${tag}_inst.json 2> gpurun_out/${tag}_inst.err
python - "$tag" <<'PY'
import json, sys
for w in ("boxes", "glb", "inst"):
    d = json.loads(open(f"gpurun_out/{sys.argv[1]}_{w}.json").read().strip().splitlines()[-1])
    print(w, d["ms_per_step"], d.get("ms_per_step_min"), "one", d.get("ms_per_step_one_in_flight"), "nohint", d.get("ms_per_step_no_hints"), "mean", d.get("image_mean"))
PY
