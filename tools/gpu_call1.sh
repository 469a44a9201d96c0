#!/bin/bash
# round-4 GPU call 1: new contract / boundary tests, opcode probe, clock vs bytes, nc3d PMC baseline
mkdir -p gpurun_out/r4c1
python -m pytest tests -m gpu -x -q > gpurun_out/r4c1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c1/pytest.log
tail -5 gpurun_out/r4c1/pytest.log
tools/probes/opcode_cost_probe > gpurun_out/r4c1/opcode_probe.log 2>&1
tail -3 gpurun_out/r4c1/opcode_probe.log
python tools/clock_vs_bytes.py 3 > gpurun_out/r4c1/clock.log 2>&1
cat gpurun_out/r4c1/clock.log | tail -6
python bench.py --steps 20 --warmup 5 > gpurun_out/r4c1/bench_wave.json 2> gpurun_out/r4c1/bench_wave.err
python bench.py --points-per-gpu 250000 --no-cpu-baseline --extra-modes none --no-small-config --always-reduce > gpurun_out/r4c1/bench_250k_rccl.json 2> gpurun_out/r4c1/bench_250k_rccl.err
python bench.py --points-per-gpu 250000 --no-cpu-baseline --extra-modes none --no-small-config > gpurun_out/r4c1/bench_250k.json 2> gpurun_out/r4c1/bench_250k.err
python -c "
import json
for f in ('bench_wave','bench_250k_rccl','bench_250k'):
    try:
        d=json.load(open('gpurun_out/r4c1/%s.json'%f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('side_sets_launch_ms'))
    except Exception as e: print(f, 'FAILED', e)
"
mkdir -p build/exp/prod && cp pinn_elastodynamics_amd/lib/libpinn_hip.so build/exp/prod/
python tools/nc3d_time.py prod > gpurun_out/r4c1/nc3d_time.log 2>&1; tail -2 gpurun_out/r4c1/nc3d_time.log
bash tools/pmc_collect.sh prod nc3d > gpurun_out/r4c1/pmc_nc3d.log 2>&1
cp gpurun_out/pmc_prod_nc3d/summary.json gpurun_out/r4c1/nc3d_pmc_summary.json; tail -30 gpurun_out/r4c1/pmc_nc3d.log
