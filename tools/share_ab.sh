#!/bin/bash
# Development aid (GPU box): the default bench with the ray launches at several shares of the wave slots.
for share in 5/5 4/5 3/5 5/5 4/5; do
    RT_BENCH_RAY_SHARE=$share timeout -k 10 200 python3 bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-photon-bench --no-extra-legs > gpurun_out/share.json 2> gpurun_out/share.err || { echo "$share FAILED"; tail -3 gpurun_out/share.err; continue; }
    python3 - "$share" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/share.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print(f"share {sys.argv[1]:5s} {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s  trace {s['trace']:.3f}  shade {s['shade']:.3f}")
PY
done
