"""Worker of tests/test_gpu_dp.py::test_rccl_collective_branch_on_one_gpu: a process group of ONE rank over the "nccl" backend (RCCL
on ROCm).  The model classes run their step's all-reduce on the HIP-written [gradient | loss sums] buffer in stream order
(``always_reduce=True``), Adam behind it -- the exact call sequence of the 8-GPU run, on the one GPU a test box has."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from pinn_elastodynamics_amd.elastic_wave import DeepHPM
from pinn_elastodynamics_amd.hip_engine import HipEngine

sys.path.insert(0, ".")
from tests.test_gpu_dp import LAYERS, LB, UB, sets        # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
# the collective itself on a device buffer: sum over one rank is the identity
probe = torch.arange(1000, dtype=torch.float32, device=dev)
dist.all_reduce(probe)
torch.cuda.synchronize()
assert torch.equal(probe.cpu(), torch.arange(1000, dtype=torch.float32))
Collo, SRC, IC, UP = sets(12001)
out = {}
for tag, flag in (("plain", False), ("rccl", True)):
    eng = HipEngine(LAYERS, precision="f16x3", device=dev, max_points=1 << 14)
    m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="infinite", engine=eng, verbose=False, seed=9, always_reduce=flag)
    assert m._reduce == flag and m.world == 1
    losses = m.train(6, 1e-3, 2)
    out[tag] = (m.theta.cpu().numpy(), np.array(losses[4]))
np.savez(sys.argv[1], theta_plain=out["plain"][0], theta_rccl=out["rccl"][0], loss_plain=out["plain"][1], loss_rccl=out["rccl"][1])
dist.barrier()
dist.destroy_process_group()
