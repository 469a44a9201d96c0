"""Worker of tests/test_gpu_dp.py::test_two_ranks_on_one_gpu_plate_and_nc3d: one rank of a 2-process data-parallel run on a single GPU
(both ranks on cuda:0, gloo for the collective) with the real HipEngine, for the plate class and the 3-D class.  ``run_model`` is also
what the test calls in one process for the comparison."""
import sys

import numpy as np
import torch


def run_model(model, dev, collective="rccl"):
    """Returns (theta as numpy, per-step loss list) after a few Adam steps (+ a short L-BFGS stage for the plate)."""
    if model == "plate":
        from pinn_elastodynamics_amd.plate_hole import PINN
        from tests.test_plate_host import plate_sets
        rng = np.random.default_rng(12)
        sets = plate_sets(rng, n=6001)
        lN, lD, lP = [3] + 4 * [32] + [5], [3] + 3 * [20] + [5], [3] + 3 * [20] + [5]
        m = PINN(*sets, lN, lD, lP, [0.0, 0.0, 0.0], [0.5, 0.5, 10.0], verbose=False, seed=5, collective=collective)
        hist = m.train(4, 1e-3)
        m.train_bfgs(options=dict(maxiter=3, maxfun=5))
        return m.theta["uv"].cpu().numpy(), np.array(hist[3])
    from pinn_elastodynamics_amd.navier_cauchy_3d import NavierCauchy3D, halfspace_case
    c = halfspace_case(n_collo=8001, n_ic=400, n_top=400, n_src=(12, 9), seed=4, width=32, depth=4)
    m = NavierCauchy3D(c["Collo"], c["SRC"], c["IC"], c["TOP"], c["uv_layers"], c["lb"], c["ub"], verbose=False, seed=9, collective=collective)
    losses = m.train(5, 1e-3, 2)
    return m.theta.cpu().numpy(), np.array(losses[4])


if __name__ == "__main__":
    import torch.distributed as dist
    sys.path.insert(0, ".")
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    theta, loss = run_model(sys.argv[1], torch.device("cuda:0"), sys.argv[3] if len(sys.argv) > 3 else "rccl")
    th = [torch.zeros(theta.size) for _ in range(dist.get_world_size())]
    dist.all_gather(th, torch.from_numpy(theta))
    if dist.get_rank() == 0:
        np.savez(sys.argv[2], theta0=th[0].numpy(), theta1=th[1].numpy(), loss=loss)
    dist.barrier()
    dist.destroy_process_group()
