#!/bin/bash
# Development aid (GPU box): the caustic-lab bench on the tree's own build and on each variant named, with the
# duration of one photon pass.
for v in base "$@"; do
    root=variants/$v; [ $v = base ] && root=.
    RT_BENCH_ROOT=$root timeout -k 10 200 python3 bench.py --workload caustic-lab --steps 64 --warmup 32 --no-cpu-baseline --no-photon-bench > gpurun_out/lab_$v.json 2> gpurun_out/lab_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/lab_$v.err; continue; }
    python3 - "$v" <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads(open(f"gpurun_out/lab_{v}.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print(f"{v:12s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s  trace {s['trace']:.3f}  shade {s['shade']:.3f}  photon pass {d.get('photon_pass_ms', 0):.3f} ms")
PY
done
