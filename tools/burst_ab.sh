#!/bin/bash
# Development aid (GPU box): a burst of 20 render() calls after 5 (what a short benchmark run makes) on 1/8 .. 1/1 of the
# frame, as one pass at the renderer's defaults and cut into passes of 10 and 5 samples.
for share in 8 4 2 1; do
    timeout -k 10 100 python3 tools/share_rehearsal.py --steps 20 --warmup 5 $share | grep "dragon-cornell 1/"
    for cut in "10 2" "5 2" "5 4" "7 3"; do
        set -- $cut
        timeout -k 10 100 python3 tools/share_rehearsal.py --steps 20 --warmup 5 --batch $1 --in-flight $2 $share | grep "dragon-cornell 1/"
    done
done
