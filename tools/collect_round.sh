#!/bin/bash
# Dev tool (GPU box): everything a round commits under profiles/ -- rocprofv3 stats + PMC (tools/collect_profiles.sh) and the bench lines.
#   tools/collect_round.sh TAG      -> gpurun_out/TAG_*.{csv,json}
TAG=${1:-r03}
OUT=gpurun_out
mkdir -p $OUT
bash tools/collect_profiles.sh $TAG > $OUT/collect_$TAG.log 2>&1
bash tools/pmc_collect.sh prod 80 > $OUT/pmc80.log 2>&1; cp $OUT/pmc_prod_80/summary.json $OUT/${TAG}_wide80_pmc_summary.json
bash tools/pmc_collect.sh prod 100 > $OUT/pmc100.log 2>&1; cp $OUT/pmc_prod_100/summary.json $OUT/${TAG}_wide100_pmc_summary.json
bash tools/pmc_collect.sh prod nc3d > $OUT/pmc_nc3d.log 2>&1; cp $OUT/pmc_prod_nc3d/summary.json $OUT/${TAG}_nc3d_pmc_summary.json
bash tools/pmc_collect.sh prod plate > $OUT/pmc_plate.log 2>&1; cp $OUT/pmc_prod_plate/summary.json $OUT/${TAG}_plate_pmc_summary.json
# the bench lines below quote the counters of THIS tree (bench.py refuses a summary whose kernel_source_sha differs): the fresh summaries
# take the place of the committed ones on the box, and are copied into profiles/ from gpurun_out/ afterwards
cp $OUT/${TAG}_fused_pmc_summary.json $OUT/${TAG}_wide80_pmc_summary.json $OUT/${TAG}_wide100_pmc_summary.json $OUT/${TAG}_nc3d_pmc_summary.json $OUT/${TAG}_plate_pmc_summary.json profiles/
python bench.py > $OUT/${TAG}_bench_wave.json 2> $OUT/bench_wave.err
python bench.py --config plate > $OUT/${TAG}_bench_plate.json 2> $OUT/bench_plate.err
python bench.py --config plate --width 70 --no-cpu-baseline > $OUT/${TAG}_bench_plate70.json 2> $OUT/bench_plate70.err
python bench.py --config nc3d > $OUT/${TAG}_bench_nc3d.json 2> $OUT/bench_nc3d.err
python bench.py --width 80 --points-per-gpu 1000000 --no-cpu-baseline --extra-modes none > $OUT/${TAG}_bench_wave80.json 2> $OUT/bench_wave80.err
python bench.py --width 100 --points-per-gpu 1000000 --no-cpu-baseline --extra-modes none > $OUT/${TAG}_bench_wave100.json 2> $OUT/bench_wave100.err
python bench.py --points-per-gpu 250000 --no-cpu-baseline --extra-modes none > $OUT/${TAG}_bench_250k.json 2> $OUT/bench_250k.err
python tools/conf_time.py > $OUT/${TAG}_conf_time.txt 2>&1
python bench.py --points-per-gpu 250000 --no-cpu-baseline --extra-modes none --no-small-config --always-reduce > $OUT/${TAG}_bench_250k_rccl.json 2> $OUT/bench_250k_rccl.err
tail -c 600 $OUT/${TAG}_bench_*.json; cat $OUT/${TAG}_conf_time.txt
