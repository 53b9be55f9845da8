#!/bin/bash
# SQ / TCP / TA counters of the photon kernels (separate passes, program directly after --)
export TMPDIR=/tmp
OUT=gpurun_out/prof_gather
mkdir -p $OUT
BENCH="python3 tools/gather_order.py"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- $BENCH > $OUT/a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- $BENCH > $OUT/b.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $OUT/c -- $BENCH > $OUT/c.log 2>&1
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob('gpurun_out/prof_gather/*/**/*_counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k=row['Kernel_Name']
        if 'photonGather' in k:
            acc['gather'][row['Counter_Name']].append(float(row['Counter_Value']))
for k,c in acc.items():
    for n,v in sorted(c.items()):
        # launches: 4 orders x 6 launches; print the mean of the first 6 (row-major) and of the last 6 (random)
        print(f"{n:40s} row-major {sum(v[:6])/6:14.1f}   tiles {sum(v[6:12])/6:14.1f}   random {sum(v[18:24])/6:14.1f}")
PY
tail -3 $OUT/c.log
