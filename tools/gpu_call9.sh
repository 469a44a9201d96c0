mkdir -p gpurun_out/r4c9
python -m pytest tests -m gpu -q > gpurun_out/r4c9/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c9/pytest.log
grep -E "^E  |FAILED|passed|failed" gpurun_out/r4c9/pytest.log | head -40
python tools/wide_time.py 80 2>&1 | grep fused | head -1
python tools/wide_time.py 100 2>&1 | grep fused | head -1
python tools/nc3d_time.py - 2>&1 | tail -1
python tools/conf_time.py 2>&1 | tail -3
python tools/plate_time.py 70 2>&1 | tail -3
