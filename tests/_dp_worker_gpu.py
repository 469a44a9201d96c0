"""Worker of tests/test_gpu_dp.py: ONE rank of a 2-process data-parallel run on a single GPU (both ranks on cuda:0, gloo for the
collective) with the real HipEngine -- the product's multi-process path short of RCCL itself."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from pinn_elastodynamics_amd.elastic_wave import DeepHPM
from pinn_elastodynamics_amd.hip_engine import HipEngine

sys.path.insert(0, ".")
from tests.test_gpu_dp import LAYERS, LB, UB, sets        # noqa: E402

dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
Collo, SRC, IC, UP = sets()
eng = HipEngine(LAYERS, precision="f16x3", device=dev, max_points=1 << 15)
m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="infinite", engine=eng, verbose=False, seed=9)
losses = m.train(8, 1e-3, 2)
th = [torch.zeros(m.n_params) for _ in range(dist.get_world_size())]
dist.all_gather(th, m.theta.cpu())
if dist.get_rank() == 0:
    np.savez(sys.argv[1], theta0=th[0].numpy(), theta1=th[1].numpy(), loss=np.array(losses[4]), rows=np.array([m._collo[0].numel()]))
dist.barrier()
dist.destroy_process_group()
