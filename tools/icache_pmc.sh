cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_[A-Z_]*IFETCH[A-Z_]*" | sort -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache; mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $OUT/p1 -o p -- python $GRAFT_REPO_ROOT/tools/exp_run.py prod > $OUT/p1.log 2>&1
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
tail -3 $OUT/p1.log
