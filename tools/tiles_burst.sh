#!/bin/bash
# Development aid (GPU box): a short run (20 steps after 5) on the whole frame as one region and as two or three regions side by side.
for tiles in 1 2 3 1 2; do
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --tiles $tiles --no-cpu-baseline --no-photon-bench --no-extra-legs > gpurun_out/tiles.json 2> gpurun_out/tiles.err || { echo "$tiles FAILED"; continue; }
    python3 - "$tiles" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/tiles.json").read().strip().splitlines()[-1])
print(f"regions {sys.argv[1]}: {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s  passes of {d['samples_per_pass']} x {d['samples_in_flight']}")
PY
done
