#!/bin/bash
# Turns the trees tools/profile_round3.sh left under gpurun_out/ into the committed summaries under profiles/.
set -e
TAG=${1:-r3a}
python3 profiles/summarize.py gpurun_out/prof_$TAG > profiles/${TAG}_bench_summary.txt
for s in 1000000:1m 10000000:10m 100000000:100m; do
    n=${s%%:*}; t=${s##*:}
    { python3 profiles/summarize.py gpurun_out/prof_soup_${TAG}_$n; echo; echo "== bench line of this size =="; cat gpurun_out/prof_soup_${TAG}_$n/lines.json; } > profiles/${TAG}_soup_${t}_summary.txt
done
{ python3 profiles/summarize.py gpurun_out/prof_lab_$TAG; echo; echo "== line =="; cat gpurun_out/prof_lab_$TAG/line.json; echo; echo "== share rehearsal =="; cat gpurun_out/prof_lab_$TAG/share_caustic_lab.txt gpurun_out/prof_lab_$TAG/share_dragon.txt; } > profiles/${TAG}_caustic_lab_summary.txt
{ echo "# SQ / TCP / TA counters per launch (means), tools/pmc_sq.sh (bench --steps 32 --batch 32 --in-flight 1) on the library of profiles/${TAG}_bench_summary.txt"; grep Kernel gpurun_out/prof_sq_$TAG.txt; } > profiles/${TAG}_sq_counters.txt
