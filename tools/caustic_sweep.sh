#!/bin/bash
# caustic-lab (final profile, 1080p) over samples per pass x passes in flight: one line each
# (optionally parts of the frame side by side in front): tools/caustic_sweep.sh "1x4 4x1 2x4x1" [more bench flags]
for combo in ${1:-1x4 2x2 4x1 4x2}; do
  IFS=x read -r t b d <<< "$combo"
  if [ -z "$d" ]; then d=$b; b=$t; t=1; fi
  timeout -k 10 200 python bench.py --workload caustic-lab --steps 32 --warmup 8 --no-cpu-baseline --no-photon-bench \
      --tiles $t --batch $b --in-flight $d $2 > /tmp/sweep_out.json 2>/tmp/sweep_err.txt || { echo "$combo FAILED"; tail -3 /tmp/sweep_err.txt; continue; }
  python - "$combo" <<'PY'
import json,sys
d=json.loads(open('/tmp/sweep_out.json').read().strip().split('\n')[-1])
print(f"{sys.argv[1]:6s} {d['ms_per_step']:.3f} ms/sample {d['value']:.0f} Mrays/s roofline {d['roofline']['frac']:.3f} launches {d['roofline']['launches']} avg {d['roofline']['avg_launch_ms']:.3f} ms stages {({k: round(v,2) for k,v in d['stage_ms_per_step'].items()})}")
PY
done
