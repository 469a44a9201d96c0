#!/bin/bash
mkdir -p gpurun_out/r4c2
python -m pytest tests -m gpu -x -q > gpurun_out/r4c2/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c2/pytest.log
tail -8 gpurun_out/r4c2/pytest.log
python tools/step_cycles.py > gpurun_out/r4c2/step_cycles.log 2>&1; cat gpurun_out/r4c2/step_cycles.log | tail -5
python bench.py --steps 20 --warmup 5 > gpurun_out/r4c2/bench_wave.json 2> gpurun_out/r4c2/bench_wave.err
python bench.py --points-per-gpu 250000 --no-cpu-baseline --extra-modes none --no-small-config --always-reduce > gpurun_out/r4c2/bench_250k_rccl.json 2> gpurun_out/r4c2/bench_250k_rccl.err
python -c "
import json
for f in ('bench_wave','bench_250k_rccl'):
    try:
        d=json.loads([l for l in open('gpurun_out/r4c2/%s.json'%f) if l.startswith('{')][0]); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('side_sets_launch_ms'), d.get('other_precision_modes'))
    except Exception as e: print(f, 'FAILED', e)
"
