#!/bin/bash
# Development aid (GPU box): passes in flight against samples per pass, on the whole frame (bench) and on an eighth of
# it (share rehearsal, steady state and a burst of twenty samples).
for cut in "16 1" "16 2" "32 1" "8 2" "32 1" "16 2"; do
    set -- $cut
    timeout -k 10 200 python3 bench.py --steps 64 --warmup 16 --batch $1 --in-flight $2 --no-cpu-baseline --no-photon-bench --no-extra-legs > gpurun_out/inflight.json 2> gpurun_out/inflight.err || { echo "$cut FAILED"; continue; }
    python3 - "$cut" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/inflight.json").read().strip().splitlines()[-1])
print(f"whole frame, samples per pass and passes in flight {sys.argv[1]}: {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s")
PY
done
for cut in "64 1" "64 2" "32 2"; do
    set -- $cut
    timeout -k 10 100 python3 tools/share_rehearsal.py --batch $1 --in-flight $2 8 | grep "dragon-cornell 1/"
    timeout -k 10 100 python3 tools/share_rehearsal.py --steps 20 --warmup 5 --batch $1 --in-flight $2 8 | grep "dragon-cornell 1/" | sed 's/^/burst: /'
done
