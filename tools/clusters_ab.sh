#!/bin/bash
# Development aid (GPU box): the default bench with the upper-level rebuild cut at several cluster counts.
for clusters in 8192 32768 131072 2048; do
    RT_BLAS_CLUSTERS=$clusters timeout -k 10 200 python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-photon-bench --no-extra-legs > gpurun_out/clusters.json 2> gpurun_out/clusters.err || { echo "$clusters FAILED"; tail -3 gpurun_out/clusters.err; continue; }
    python3 - "$clusters" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/clusters.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print(f"clusters {sys.argv[1]:7s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s  trace {s['trace']:.3f}  nodes/ray {d['roofline']['nodes_per_ray']:.2f} tris/ray {d['roofline']['triangles_per_ray']:.2f}")
PY
done
