#!/bin/bash
# Development aid (GPU box): the default bench with instances folded in pairs, singly and not at all, in turn, twice.
for round in 1 2; do
for mode in 1 single 0; do
    RT_SCENE_FOLD=$mode timeout -k 10 200 python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-photon-bench --no-extra-legs > gpurun_out/fold_$mode.json 2> gpurun_out/fold_$mode.err || { echo "$mode FAILED"; continue; }
    python3 - "$mode" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/fold_{v}.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print(f"fold={v:8s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s  trace {s['trace']:.3f}  shade {s['shade']:.3f}  nodes/ray {d['roofline']['nodes_per_ray']:.2f} tris/ray {d['roofline']['triangles_per_ray']:.2f} frac {d['roofline']['frac']:.3f}")
PY
done
done
