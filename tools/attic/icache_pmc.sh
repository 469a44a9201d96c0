#!/bin/bash
# Dev tool (GPU box): instruction-fetch counters of the fused launch (tools/exp_run.py prod): requests / hits / misses of the instruction cache
# and the accumulated number of fetches in flight (SQ_IFETCH_LEVEL / SQ_IFETCH = average fetch latency in cycles of the counter).
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache; mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/tools/exp_run.py prod > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -o p -- python $GRAFT_REPO_ROOT/tools/exp_run.py prod > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/p3 -o p -- python $GRAFT_REPO_ROOT/tools/exp_run.py prod > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'fused' in r['Kernel_Name'] and int(r['Grid_Size']) >= 256 * 512:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    v = sorted(acc[k])[len(acc[k]) // 2:]
    print(k, '%.4e' % v[len(v)//2], len(acc[k]))
PY
