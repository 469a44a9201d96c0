# round-5 GPU call d: the P2P collective on two ranks of one GPU (first, under its own timeout), then the whole suite on the LO_FROM = 3 tree
mkdir -p gpurun_out/r5d
(timeout 300 python -m pytest tests/test_gpu_dp.py -q -p no:cacheprovider -k p2p > gpurun_out/r5d/p2p.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5d/p2p.log)
tail -25 gpurun_out/r5d/p2p.log
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_dp.py::test_p2p_one_shot_allreduce_two_ranks_on_one_gpu > gpurun_out/r5d/gputests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5d/gputests.log)
tail -6 gpurun_out/r5d/gputests.log
