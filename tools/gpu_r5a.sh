mkdir -p gpurun_out/r5a
(timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r5a/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5a/gputests.log)
tail -5 gpurun_out/r5a/gputests.log
EXP_ROUNDS=5 timeout 300 python tools/exp_run.py base lo3 lo4 lo8 > gpurun_out/r5a/exp.log 2>&1; cat gpurun_out/r5a/exp.log | tail -6
timeout 300 python bench.py > gpurun_out/r5a/bench_wave.json 2> gpurun_out/r5a/bench_wave.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r5a/bench_wave.json') if l.startswith('{')][0]); r=d['roofline']
print('wave', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], d['shader_clock_ghz'], r['step_decomposition_ms'])"
timeout 200 python bench.py --global-points 2000000 --rank-share 8 --always-reduce --no-cpu-baseline --extra-modes none --no-small-config > gpurun_out/r5a/bench_share8.json 2> gpurun_out/r5a/bench_share8.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r5a/bench_share8.json') if l.startswith('{')][0]); r=d['roofline']
print('share8', d['value'], d['ms_per_step'], r['avg_launch_ms'], d['shader_clock_ghz'], d['allreduce_ms'], r['step_decomposition_ms'])"
