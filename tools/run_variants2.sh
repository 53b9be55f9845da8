#!/bin/bash
# Development aid (GPU box): the default bench on the tree's own build and on each variant named, twice round.
for round in 1 2; do
for v in base "$@"; do
    root=variants/$v; [ $v = base ] && root=.
    RT_BENCH_ROOT=$root timeout -k 10 200 python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-photon-bench --no-extra-legs > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/var_$v.err; continue; }
    python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/var_{v}.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print(f"{v:12s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s  trace {s['trace']:.3f}  shade {s['shade']:.3f}  nodes/ray {d['roofline']['nodes_per_ray']:.2f} tris/ray {d['roofline']['triangles_per_ray']:.2f}")
PY
done
done
