"""Worker of test_data_parallel_two_ranks_matches_single (gloo, CPU)."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from pinn_elastodynamics_amd.elastic_wave import DeepHPM
from tests._oracle_engine import OracleEngine
from tests.test_host_logic import LAYERS, LB, UB, small_sets

dist.init_process_group("gloo")
rank = dist.get_rank()
Collo, SRC, IC, UP = small_sets(n=257)
m = DeepHPM(Collo, SRC, IC, UP, LAYERS, LB, UB, case="semi_infinite", engine=OracleEngine(LAYERS), verbose=False, seed=9)
losses = m.train(3, 1e-3, 2)
th = [torch.zeros_like(m.theta) for _ in range(2)]
dist.all_gather(th, m.theta)
if rank == 0:
    np.savez(sys.argv[1], theta0=th[0].numpy(), theta1=th[1].numpy(), loss=np.array(losses[4]))
dist.destroy_process_group()
