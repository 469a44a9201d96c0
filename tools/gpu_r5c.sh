# round-5 GPU call c: clean A/B of the per-layer low-part policy (order shuffled per round), bench contract tests
mkdir -p gpurun_out/r5c
EXP_ROUNDS=10 timeout 400 python tools/exp_run.py lo2 lo3 lo4 > gpurun_out/r5c/exp.log 2>&1; tail -4 gpurun_out/r5c/exp.log
(timeout 600 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_paths.py -q -p no:cacheprovider > gpurun_out/r5c/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5c/gputests.log)
tail -4 gpurun_out/r5c/gputests.log
