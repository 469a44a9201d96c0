# round-5 GPU call e: suite on the plate-step tree; bench lines (wave, plate, rank share with RCCL / P2P); 3-D phase stamps
mkdir -p gpurun_out/r5e
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r5e/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5e/gputests.log)
tail -5 gpurun_out/r5e/gputests.log
show() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][0]); r=d['roofline']
print('$2', 'pts/s %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'launch %.4f' % r['avg_launch_ms'], 'frac %.4f' % r['frac'], 'GHz', d['shader_clock_ghz'], 'allreduce', d['allreduce_ms'])"; }
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r5e/bench_wave.json 2> gpurun_out/r5e/bench_wave.err; show gpurun_out/r5e/bench_wave.json wave
timeout 300 python bench.py --config plate --no-cpu-baseline > gpurun_out/r5e/bench_plate.json 2> gpurun_out/r5e/bench_plate.err; show gpurun_out/r5e/bench_plate.json plate
for c in rccl p2p; do
timeout 200 python bench.py --global-points 2000000 --rank-share 8 --always-reduce --collective $c --no-cpu-baseline --extra-modes none --no-small-config > gpurun_out/r5e/bench_share8_$c.json 2> gpurun_out/r5e/bench_share8_$c.err; show gpurun_out/r5e/bench_share8_$c.json share8_$c
done
timeout 200 python tools/phase_trace_3d.py > gpurun_out/r5e/nc3d_phase_trace.txt 2>&1; tail -16 gpurun_out/r5e/nc3d_phase_trace.txt
