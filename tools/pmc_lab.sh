#!/bin/bash
# SQ counters of the caustic-lab bench's kernels (one pass of <= 8 SQ counters), means per launch of the largest launches
export TMPDIR=/tmp
OUT=gpurun_out/prof_sq_lab
rm -rf $OUT; mkdir -p $OUT
BENCH="python3 bench.py --workload caustic-lab --steps 16 --warmup 16 --no-cpu-baseline --no-photon-bench"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
python3 - <<'PY'
import csv,glob,collections
rows=collections.defaultdict(lambda: collections.defaultdict(dict))
for path in glob.glob('gpurun_out/prof_sq_lab/*/**/*_counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k=row['Kernel_Name']
        for key in ('shadeKernel','causticGatherKernel','photonShadeKernel'):
            if key in k and 'Heavy' not in k:
                rows[key][row['Dispatch_Id']][row['Counter_Name']]=float(row['Counter_Value'])
for k,d in rows.items():
    big=sorted(d.values(), key=lambda c:-c.get('SQ_WAVES',0))[:4]
    for c in big:
        w=c['SQ_WAVES']; print(k, 'waves', int(w), 'valu/wave', round(c['SQ_INSTS_VALU']/w,1), 'lane use', round(c['SQ_THREAD_CYCLES_VALU']/(c['SQ_ACTIVE_INST_VALU']*64),3), 'wave-cycles/wave', round(c['SQ_WAVE_CYCLES']/w), 'wait frac', round(c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES'],3), 'vmem rd/wave', round(c['SQ_INSTS_VMEM_RD']/w,1), 'busy', int(c['SQ_BUSY_CYCLES']))
PY
