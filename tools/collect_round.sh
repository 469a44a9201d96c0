#!/bin/bash
# Dev tool (GPU box): everything a round commits under profiles/ -- rocprofv3 stats + PMC (tools/collect_profiles.sh) and the bench lines.
#   tools/collect_round.sh TAG [bench-only]     -> gpurun_out/TAG_*.{csv,json}
TAG=${1:-r05}
OUT=gpurun_out
mkdir -p $OUT build/exp/prod
cp pinn_elastodynamics_amd/lib/libpinn_hip.so build/exp/prod/libpinn_hip.so
if [ "$2" != "bench-only" ]; then       # (bench-only: the PMC summaries committed under profiles/ are of these kernel sources already)
bash tools/collect_profiles.sh $TAG > $OUT/collect_$TAG.log 2>&1
for k in 80 100 nc3d plate plate70 conf; do
    bash tools/pmc_collect.sh prod $k > $OUT/pmc_$k.log 2>&1
done
cp $OUT/pmc_prod_80/summary.json $OUT/${TAG}_wide80_pmc_summary.json
cp $OUT/pmc_prod_100/summary.json $OUT/${TAG}_wide100_pmc_summary.json
cp $OUT/pmc_prod_nc3d/summary.json $OUT/${TAG}_nc3d_pmc_summary.json
cp $OUT/pmc_prod_plate/summary.json $OUT/${TAG}_plate_pmc_summary.json
cp $OUT/pmc_prod_plate70/summary.json $OUT/${TAG}_plate70_pmc_summary.json
cp $OUT/pmc_prod_conf/summary.json $OUT/${TAG}_conf_pmc_summary.json
# the bench lines below quote the counters of THIS tree (bench.py refuses a summary whose kernel_source_sha differs): the fresh summaries
# take the place of the committed ones on the box, and are copied into profiles/ from gpurun_out/ afterwards
cp $OUT/${TAG}_*_pmc_summary.json profiles/
fi
python bench.py > $OUT/${TAG}_bench_wave.json 2> $OUT/bench_wave.err
python bench.py --config plate > $OUT/${TAG}_bench_plate.json 2> $OUT/bench_plate.err
python bench.py --config plate --width 70 --no-cpu-baseline > $OUT/${TAG}_bench_plate70.json 2> $OUT/bench_plate70.err
python bench.py --config nc3d > $OUT/${TAG}_bench_nc3d.json 2> $OUT/bench_nc3d.err
python bench.py --width 80 --points-per-gpu 1000000 --no-cpu-baseline --extra-modes none > $OUT/${TAG}_bench_wave80.json 2> $OUT/bench_wave80.err
python bench.py --width 100 --points-per-gpu 1000000 --no-cpu-baseline --extra-modes none > $OUT/${TAG}_bench_wave100.json 2> $OUT/bench_wave100.err
python tools/conf_time.py > $OUT/${TAG}_conf_time.txt 2>&1
# what ONE of 8 ranks of the 2 M-point strong-scaling run executes per step (1/8 of every set), without and with the collective branch (RCCL; P2P)
python bench.py --global-points 2000000 --rank-share 8 --no-cpu-baseline --extra-modes none --no-small-config > $OUT/${TAG}_bench_250k.json 2> $OUT/bench_250k.err
python bench.py --global-points 2000000 --rank-share 8 --always-reduce --no-cpu-baseline --extra-modes none --no-small-config > $OUT/${TAG}_bench_250k_rccl.json 2> $OUT/bench_250k_rccl.err
python bench.py --global-points 2000000 --rank-share 8 --always-reduce --collective p2p --no-cpu-baseline --extra-modes none --no-small-config > $OUT/${TAG}_bench_250k_p2p.json 2> $OUT/bench_250k_p2p.err
python tools/phase_trace_3d.py > $OUT/${TAG}_nc3d_phase_trace.txt 2>&1
for f in $OUT/${TAG}_bench_*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); r = d.get('roofline', {})
    print(sys.argv[1].split('/')[-1], 'pts/s %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'launch', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'GHz', d.get('shader_clock_ghz'), 'allreduce', d.get('allreduce_ms'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
cat $OUT/${TAG}_conf_time.txt
