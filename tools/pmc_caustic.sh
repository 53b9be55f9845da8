#!/bin/bash
# SQ / TCP / TA counters of the renderer's gather kernels on the caustic-lab frame (separate passes,
# program directly after --); prints the mean per launch over the first-bounce launches (the longest
# launch of every sample) of causticGatherKernel
export TMPDIR=/tmp
OUT=gpurun_out/prof_caustic
mkdir -p $OUT
BENCH="python3 bench.py --workload caustic-lab --tiles 1 --steps 2 --warmup 1 --no-cpu-baseline --no-photon-bench"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $OUT/c -- $BENCH > $OUT/c.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob('gpurun_out/prof_caustic/*/**/*_counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        if 'causticGatherKernel' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for name, values in sorted(acc.items()):
    top = sorted(values)[-4:]            # the first-bounce launches carry the largest counts
    print(f"{name:40s} {sum(top) / len(top):16.1f}   ({len(values)} launches)")
PY
rm -rf $OUT/a $OUT/b $OUT/c
