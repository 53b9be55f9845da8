#!/bin/bash
# Development aid (GPU box): caustic-lab with 16 x 2 and 32 x 1 samples per pass x passes in flight; bursts on the whole frame.
for cut in "16 2" "32 1" "16 2" "32 1"; do
    set -- $cut
    timeout -k 10 200 python3 bench.py --workload caustic-lab --steps 64 --warmup 32 --batch $1 --in-flight $2 --no-cpu-baseline --no-photon-bench > gpurun_out/inflight.json 2> gpurun_out/inflight.err || { echo "$cut FAILED"; tail -2 gpurun_out/inflight.err; continue; }
    python3 - "$cut" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/inflight.json").read().strip().splitlines()[-1])
print(f"caustic-lab, samples per pass and passes in flight {sys.argv[1]}: {d['ms_per_step']:.3f} ms/step  {d['value']:.0f} Mrays/s")
PY
done
for cut in "16 2" "32 1" "16 1"; do
    set -- $cut
    timeout -k 10 100 python3 tools/share_rehearsal.py --steps 20 --warmup 5 --batch $1 --in-flight $2 1 | grep "dragon-cornell 1/" | sed 's/^/burst: /'
done
for cut in "32 2" "64 1"; do
    set -- $cut
    timeout -k 10 100 python3 tools/share_rehearsal.py --batch $1 --in-flight $2 2 | grep "dragon-cornell 1/"
    timeout -k 10 100 python3 tools/share_rehearsal.py --steps 20 --warmup 5 --batch $1 --in-flight $2 2 | grep "dragon-cornell 1/" | sed 's/^/burst: /'
done
