# Dev tool (GPU box): the headline pair on one box -- rocprofv3 kernel-trace stats of the default bench command, then the default bench line
OUT=$PWD/gpurun_out/pair_${1:-x}
mkdir -p $OUT
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-modes none > $OUT/prof.log 2>&1
cp $(find $OUT/prof -name '*kernel_stats.csv' | head -1) $OUT/fused_kernel_stats.csv
cd $ROOT
python bench.py > $OUT/bench_wave.json 2> $OUT/bench_wave.err
python bench.py --config plate > $OUT/bench_plate.json 2> $OUT/bench_plate.err
head -2 $OUT/fused_kernel_stats.csv | cut -c1-160
python - <<PY
import json
for n in ('wave','plate'):
    for line in open('$OUT/bench_%s.json' % n):
        if line.startswith('{'):
            d=json.loads(line); print(n, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']); break
PY
