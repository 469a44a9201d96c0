"""Driver with the structure of the reference's ``__main__`` blocks (INF:634-770, SEMI:667-800, CONF:881-1010): build the
point sets, construct the model, Adam and/or L-BFGS, save the weights, predict frames and -- where FEM frames are given --
print the relative L2 error per field (the reference only plots them).

    python examples/elastic_wave.py --case infinite --iters 200 --n-f 20000 --bfgs-iters 50
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/elastic_wave.py --case infinite   # data parallel
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinn_elastodynamics_amd import pointsets as ps                                      # noqa: E402
from pinn_elastodynamics_amd.elastic_wave import DeepHPM, DeepHPMConfined               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", choices=["infinite", "semi", "confined"], default="infinite")
    ap.add_argument("--max-t", type=float, default=None)
    ap.add_argument("--n-f", type=int, default=120000)
    ap.add_argument("--width", type=int, default=None, help="hidden width (reference: 80 / 100 / 140)")
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--batch-num", type=int, default=1)
    ap.add_argument("--bfgs-iters", type=int, default=0, help="0 = skip the L-BFGS stage; reference default maxiter is 100000")
    ap.add_argument("--load", default="", help="weights to start from (reference pickle or .npz)")
    ap.add_argument("--save", default="uv_NN.npz")
    ap.add_argument("--fem", default="", help="FEM ProbeData-<i>.mat file pattern with {i}, compared at the predicted frames")
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()

    if "RANK" in os.environ:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        torch.distributed.init_process_group("nccl")
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0

    kw = {} if a.width is None else {"width": a.width}
    if a.case == "infinite":
        c = ps.infinite_case(MAX_T=a.max_t or 20.0, N_f=a.n_f, **kw)
        model = DeepHPM(c["Collo"], c["SRC"], c["IC"], c["UP"], c["uv_layers"], c["lb"], c["ub"], ExistModel=int(bool(a.load)), modelDir=a.load,
                        case="infinite", precision=a.precision, verbose=rank == 0)
    elif a.case == "semi":
        c = ps.semi_infinite_case(MAX_T=a.max_t or 16.0, N_f=a.n_f, **kw)
        model = DeepHPM(c["Collo"], c["SRC"], c["IC"], c["UP"], c["uv_layers"], c["lb"], c["ub"], ExistModel=int(bool(a.load)), modelDir=a.load,
                        case="semi_infinite", precision=a.precision, verbose=rank == 0)
    else:
        c = ps.confined_case(MAX_T=a.max_t or 14.0, N_f=a.n_f, **kw)
        model = DeepHPMConfined(c["Collo"], c["SRC"], c["IC"], c["FIXED"], None, c["uv_layers"], None, None, c["lb"], c["ub"], uvDir=a.load,
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
        xc, yc, r = c["source"]
        x_star, y_star = ps.probe_points(c["lb"][0], c["ub"][0], c["lb"][1], c["ub"][1], 201, xc, yc, r)
        times = ps.frame_times(c["ub"][2])
        for i in range(0, times.size, 10):
            if a.fem and os.path.exists(a.fem.format(i=i)):
                xs, ys, u, v, amp, s11, s22, s12, mis = ps.preprocess(a.fem.format(i=i))
                pred = model.predict(xs, ys, np.full_like(xs, times[i]))
                errs = [ps.relative_l2(p, f) for p, f in zip(pred[:5], (u, v, s11, s22, s12))]
                print("frame %3d t=%5.2f  rel-L2 u %.3f v %.3f s11 %.3f s22 %.3f s12 %.3f" % (i, times[i], *errs))
            else:
                u, v = model.predict(x_star, y_star, np.full_like(x_star, times[i]))[:2]
                print("frame %3d t=%5.2f  max |(u,v)| = %.4f" % (i, times[i], float(np.sqrt(u ** 2 + v ** 2).max())))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
