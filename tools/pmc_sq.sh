#!/bin/bash
# SQ counters of the bench's kernels (two passes of <= 8 SQ counters each)
export TMPDIR=/tmp
OUT=gpurun_out/prof_sq
mkdir -p $OUT
BENCH="python3 bench.py --steps 32 --warmup 32 --no-cpu-baseline --no-photon-bench --no-extra-legs --tiles 1 --in-flight 1 --batch 32"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $OUT/c -- $BENCH > $OUT/c.log 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/prof_sq/*/**/*_counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k=row['Kernel_Name']
        for key in ('traceKernel','shadeKernel','accumulateKernel'):
            if key in k:
                acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
for k,c in acc.items():
    print(k, {n: round(sum(v)/len(v),1) for n,v in sorted(c.items())}, 'launches', max(len(v) for v in c.values()))
PY
