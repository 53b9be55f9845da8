#!/bin/bash
# Development aid (GPU box): soup_profile.py on the package itself and on each variant named.
SIZES=${SIZES:-"1000000 10000000"}
echo "== package =="; timeout -k 10 200 python tools/soup_profile.py $SIZES | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['triangles'], round(d['mrays_per_s']), 'Mrays/s', round(d['nodes_per_ray'],2), round(d['triangles_per_ray'],2), round(d['build_ms'],1),'ms build')"
for v in "$@"; do
echo "== $v =="; RT_BENCH_ROOT=variants/$v timeout -k 10 200 python tools/soup_profile.py $SIZES | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['triangles'], round(d['mrays_per_s']), 'Mrays/s', round(d['nodes_per_ray'],2), round(d['triangles_per_ray'],2), round(d['build_ms'],1),'ms build')"
done
echo "== package, RT_BLAS_REFINE=0 =="; RT_BLAS_REFINE=0 timeout -k 10 200 python tools/soup_profile.py $SIZES | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['triangles'], round(d['mrays_per_s']), 'Mrays/s', round(d['nodes_per_ray'],2), round(d['triangles_per_ray'],2), round(d['build_ms'],1),'ms build')"
