# Dev tool (GPU box): A / B of the one-launch step (pinn_wave2d_step) against the separate calls on ONE box, interleaved: the 2 M-point step and
# one of 8 ranks' share with the RCCL branch.   bash tools/step_ab.sh  ->  gpurun_out/step_ab.txt
mkdir -p gpurun_out
: > gpurun_out/step_ab.txt
for rep in 1 2 3; do
for flag in "" "--no-step-call"; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --extra-modes none --no-small-config $flag 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('2M      step_call=%s  ms/step %.4f  GHz %.4f  Mcycles/step %.4f' % (d['config']['step_call'], d['ms_per_step'], d['shader_clock_ghz'], d['ms_per_step']*d['shader_clock_ghz']))" >> gpurun_out/step_ab.txt
  python bench.py --global-points 2000000 --rank-share 8 --always-reduce --no-cpu-baseline --extra-modes none --no-small-config $flag 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('share8  step_call=%s  ms/step %.4f  GHz %.4f  Mcycles/step %.4f' % (d['config']['step_call'], d['ms_per_step'], d['shader_clock_ghz'], d['ms_per_step']*d['shader_clock_ghz']))" >> gpurun_out/step_ab.txt
done; done
sort gpurun_out/step_ab.txt
(timeout 600 python -m pytest tests/test_gpu_bench_contract.py -q -p no:cacheprovider 2>&1 | tail -15)
