"""Driver for the 3-D Navier-Cauchy half-space case (BASELINE configs[4]; a build-side extension -- the reference has no 3-D script),
shaped like examples/elastic_wave.py: point sets, model, Adam and/or L-BFGS, save, a probe of the displacement amplitude on the free
surface at a few times.

    python examples/navier_cauchy_3d.py --iters 200 --n-f 100000 --bfgs-iters 20
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/navier_cauchy_3d.py        # data parallel
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinn_elastodynamics_amd.navier_cauchy_3d import NavierCauchy3D, halfspace_case      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-f", type=int, default=200000, help="collocation points")
    ap.add_argument("--width", type=int, default=128)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--batch-num", type=int, default=1)
    ap.add_argument("--bfgs-iters", type=int, default=0)
    ap.add_argument("--load", default="", help="weights to start from (.npz or [W_list, b_list] pickle)")
    ap.add_argument("--save", default="uv3d_NN.npz")
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()

    if "RANK" in os.environ:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        torch.distributed.init_process_group("nccl")
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0

    c = halfspace_case(n_collo=a.n_f, seed=1111, width=a.width, depth=a.depth)
    model = NavierCauchy3D(c["Collo"], c["SRC"], c["IC"], c["TOP"], c["uv_layers"], c["lb"], c["ub"], ExistModel=int(bool(a.load)), modelDir=a.load,
                           precision=a.precision, verbose=rank == 0)
    t0 = time.time()
    if a.iters:
        hist = model.train(iter=a.iters, learning_rate=a.lr, batch_num=a.batch_num)
        if rank == 0:
            print("Adam: loss %.4e -> %.4e" % (hist[-1][0], hist[-1][-1]))
    if a.bfgs_iters:
        model.train_bfgs(batch_num=a.batch_num, options=dict(maxiter=a.bfgs_iters, maxfun=a.bfgs_iters))
    final = model.getloss()          # every rank: the evaluation all-reduces across the data-parallel group
    if rank == 0:
        print("--- %.1f seconds ---" % (time.time() - t0))
        model.save_NN(a.save)
        print("loss terms on the full sets:", " ".join("%.4e" % v for v in final))
        lb, ub = np.asarray(c["lb"], float), np.asarray(c["ub"], float)
        g = np.linspace(0.0, 1.0, 41)
        xs, ys = np.meshgrid(lb[0] + (ub[0] - lb[0]) * g, lb[1] + (ub[1] - lb[1]) * g)
        xs, ys = xs.reshape(-1, 1), ys.reshape(-1, 1)
        top = np.full_like(xs, ub[2])                        # the free surface z = ub[2]
        for tt in np.linspace(lb[3], ub[3], 5):
            out = model.predict(xs, ys, top, np.full_like(xs, tt))
            amp = np.sqrt(out[0] ** 2 + out[1] ** 2 + out[2] ** 2)
            print("t=%6.2f  max |(u,v,w)| on the free surface = %.4e" % (tt, float(amp.max())))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
