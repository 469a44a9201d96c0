#!/bin/bash
# Dev tool (GPU box): the round's rocprofv3 evidence for profiles/ -- kernel-trace stats of the default bench command and the
# PMC passes of the fused launch (build/exp/prod/libpinn_hip.so = a copy of the production library).
#   tools/collect_profiles.sh TAG     -> gpurun_out/TAG_fused_kernel_stats.csv, gpurun_out/TAG_fused_pmc_summary.json
TAG=${1:-r02}
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-modes none > $OUT/prof_$TAG.log 2>&1
cp $(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_fused_kernel_stats.csv
cd $ROOT
bash tools/pmc_collect.sh prod > $OUT/pmc_prod.log 2>&1
cp $OUT/pmc_prod/summary.json $OUT/${TAG}_fused_pmc_summary.json
# the reference's own net widths through the LDS-operand layouts: kernel-trace stats of short bench runs
cd /tmp
for cfg in "wave 80" "wave 100" "plate 70"; do
    set -- $cfg
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$1$2 -o bench -- python $ROOT/bench.py --config $1 --width $2 --points-per-gpu 1000000 --steps 10 --warmup 3 --ramp-steps 10 --no-cpu-baseline --extra-modes none > $OUT/prof_${TAG}_$1$2.log 2>&1
    head -4 $(find $OUT/prof_${TAG}_$1$2 -name '*kernel_stats.csv' | head -1) | sed "s/^/$1$2,/" >> $OUT/${TAG}_wide_kernel_stats.csv.tmp
done
cd $ROOT
mv $OUT/${TAG}_wide_kernel_stats.csv.tmp $OUT/${TAG}_wide_kernel_stats.csv
head -8 $OUT/${TAG}_fused_kernel_stats.csv
cat $OUT/${TAG}_wide_kernel_stats.csv
