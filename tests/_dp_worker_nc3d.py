"""Worker of test_nc3d_data_parallel_two_ranks (gloo, CPU)."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from pinn_elastodynamics_amd.navier_cauchy_3d import NavierCauchy3D, halfspace_case
from tests._oracle_engine import OracleEngine

dist.init_process_group("gloo")
c = halfspace_case(n_collo=301, n_ic=40, n_top=40, n_src=(6, 5), seed=4, width=16, depth=2)
m = NavierCauchy3D(c["Collo"], c["SRC"], c["IC"], c["TOP"], c["uv_layers"], c["lb"], c["ub"], engine=OracleEngine(c["uv_layers"]), verbose=False, seed=9)
assert m._collo_full is None and sum(m._rows(0, 301)[0].numel() for _ in range(1)) in (150, 151)      # only this rank's rows are on the device
losses = m.train(3, 1e-3, 2)
th = [torch.zeros_like(m.theta) for _ in range(2)]
dist.all_gather(th, m.theta)
if dist.get_rank() == 0:
    np.savez(sys.argv[1], theta0=th[0].numpy(), theta1=th[1].numpy(), loss=np.array(losses[4]))
dist.destroy_process_group()
