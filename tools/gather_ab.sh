#!/bin/bash
# Development aid (GPU box): the photon gather of the default bench and the caustic-lab step on the tree's own build
# and on each variant named (tools/variants.sh), twice round.
for round in 1 2; do
for v in base "$@"; do
    root=variants/$v; [ $v = base ] && root=.
    RT_BENCH_ROOT=$root timeout -k 10 200 python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extra-legs > gpurun_out/gab_$v.json 2> gpurun_out/gab_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/gab_$v.err; continue; }
    RT_BENCH_ROOT=$root timeout -k 10 200 python3 bench.py --workload caustic-lab --steps 64 --warmup 32 --no-cpu-baseline --no-photon-bench > gpurun_out/gab_lab_$v.json 2> gpurun_out/gab_lab_$v.err || { echo "$v lab FAILED"; tail -3 gpurun_out/gab_lab_$v.err; continue; }
    python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/gab_{v}.json").read().strip().splitlines()[-1])
lab = json.loads(open(f"gpurun_out/gab_lab_{v}.json").read().strip().splitlines()[-1])
p = d["photon"]
s = lab["stage_ms_per_step"]
frame = lab.get("gathers_per_s_in_frame") or lab.get("caustic_lab", {}).get("gathers_per_s_in_frame", 0.0)
print(f"{v:10s} gather {p['gather_ms']:.3f} ms  {p['gathers_per_s'] / 1e6:.0f} M/s  build {p['grid_build_ms']:.3f} ms | lab {lab['ms_per_step']:.3f} ms/step  shade {s['shade']:.3f}  {frame / 1e6:.0f} M gathers/s in frame")
PY
done
done
