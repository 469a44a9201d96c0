"""Worker of test_plate_data_parallel_two_ranks (gloo, CPU): plate model, Adam steps then a short L-BFGS stage."""
import sys

import numpy as np
import torch
import torch.distributed as dist

from tests.test_plate_host import make_model

dist.init_process_group("gloo")
rank = dist.get_rank()
m, _ = make_model(4)
hist = m.train(2, 1e-3)
m.train_bfgs(options=dict(maxiter=3, maxfun=5))
th = [torch.zeros_like(m.theta["uv"]) for _ in range(2)]
dist.all_gather(th, m.theta["uv"])
if rank == 0:
    np.savez(sys.argv[1], theta0=th[0].numpy(), theta1=th[1].numpy(), loss=np.array(hist[3]))
dist.destroy_process_group()
