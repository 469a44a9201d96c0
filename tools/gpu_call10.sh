mkdir -p gpurun_out/r4c10
python -m pytest tests -m gpu -q > gpurun_out/r4c10/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c10/pytest.log
grep -E "^E  |FAILED|passed|failed|pytest rc" gpurun_out/r4c10/pytest.log | head -40
bash tools/collect_round.sh r04 2>&1 | tail -5
