#!/bin/bash
# Development aid (GPU box): times the default bench on each variant named on the command line.
for v in "$@"; do
    RT_BENCH_ROOT=variants/$v timeout -k 10 200 python bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-photon-bench > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/var_$v.err; continue; }
    python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.load(open(f"gpurun_out/var_{v}.json"))
s = d["stage_ms_per_step"]
print(f"{v:16s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s  trace {s['trace']:.3f}  shade {s['shade']:.3f}  nodes/ray {d['roofline']['nodes_per_ray']:.2f} tris/ray {d['roofline']['triangles_per_ray']:.2f}")
PY
done
