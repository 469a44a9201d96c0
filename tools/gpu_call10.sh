mkdir -p gpurun_out
EXP_ROUNDS=10 timeout 600 python tools/exp_run.py base s1p2 s1p4 2>&1 | grep -v Warn > gpurun_out/s1wg.log
cat gpurun_out/s1wg.log
