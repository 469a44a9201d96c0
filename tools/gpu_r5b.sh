# round-5 GPU call b: tests on the step-launch tree, low-part policy A/B, bench lines (step launch on)
mkdir -p gpurun_out/r5b
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r5b/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5b/gputests.log)
tail -8 gpurun_out/r5b/gputests.log
EXP_ROUNDS=5 timeout 300 python tools/exp_run.py lo2 lo3 lo4 lo8 > gpurun_out/r5b/exp.log 2>&1; tail -5 gpurun_out/r5b/exp.log
show() { python -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][0]); r=d['roofline']
print('$2', 'pts/s %.4g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'launch %.4f' % r['avg_launch_ms'], 'frac %.4f' % r['frac'], 'GHz', d['shader_clock_ghz'], 'allreduce', d['allreduce_ms'], r.get('step_decomposition_ms'))"; }
timeout 300 python bench.py > gpurun_out/r5b/bench_wave.json 2> gpurun_out/r5b/bench_wave.err; show gpurun_out/r5b/bench_wave.json wave
timeout 200 python bench.py --global-points 2000000 --rank-share 8 --always-reduce --no-cpu-baseline --extra-modes none --no-small-config > gpurun_out/r5b/bench_share8.json 2> gpurun_out/r5b/bench_share8.err; show gpurun_out/r5b/bench_share8.json share8_rccl
timeout 200 python bench.py --global-points 2000000 --rank-share 8 --no-cpu-baseline --extra-modes none --no-small-config > gpurun_out/r5b/bench_share8_nored.json 2> gpurun_out/r5b/bench_share8_nored.err; show gpurun_out/r5b/bench_share8_nored.json share8_noreduce
