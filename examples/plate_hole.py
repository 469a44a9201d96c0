"""Driver with the structure of the plate script's ``__main__`` (PLATE:870-1000): distance net -> particular net -> composite
net (L-BFGS and/or Adam), weights saved per net, composite fields predicted at the FEM frames when available.

    python examples/plate_hole.py --pre-iters 200 --iters 100 --bfgs-iters 100 --n-collo 20000
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pinn_elastodynamics_amd import pointsets as ps                    # noqa: E402
from pinn_elastodynamics_amd.plate_hole import PINN                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-collo", type=int, default=70000)
    ap.add_argument("--n-refine", type=int, default=40000)
    ap.add_argument("--pre-iters", type=int, default=20000, help="L-BFGS iterations of the distance / particular stages (PLATE:220-235)")
    ap.add_argument("--iters", type=int, default=0, help="Adam iterations of the composite stage (PLATE:963, commented out upstream)")
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--bfgs-iters", type=int, default=70000)
    ap.add_argument("--part", default="")
    ap.add_argument("--dist", default="")
    ap.add_argument("--uv", default="")
    ap.add_argument("--fem", default="", help="FEM ProbeData-<i>.mat pattern with {i}")
    a = ap.parse_args()

    c = ps.plate_case(n_collo=a.n_collo, n_refine=a.n_refine)
    model = PINN(c["Collo"], c["HOLE"], c["IC"], c["LF"], c["RT"], c["UP"], c["LW"], c["DIST"], c["uv_layers"], c["dist_layers"], c["part_layers"],
                 c["lb"], c["ub"], partDir=a.part, distDir=a.dist, uvDir=a.uv)
    if not a.dist:
        model.train_bfgs_dist(options=dict(maxiter=a.pre_iters, maxfun=a.pre_iters))
        model.count = 0
    if not a.part:
        model.train_bfgs_part(options=dict(maxiter=a.pre_iters, maxfun=a.pre_iters))
        model.count = 0
    t0 = time.time()
    if a.iters:
        model.train(iter=a.iters, learning_rate=a.lr)
    if a.bfgs_iters:
        model.train_bfgs(options=dict(maxiter=a.bfgs_iters, maxfun=a.bfgs_iters))
    print("--- %.1f seconds ---" % (time.time() - t0))
    model.save_NN("uvNN.npz", TYPE="UV")
    model.save_NN("distNN.npz", TYPE="DIST")
    model.save_NN("partNN.npz", TYPE="PART")
    model.getloss()
    times = ps.frame_times(10.0, 8)                                     # N_t = MAX_T*8+1 (PLATE:888)
    for i in range(0, times.size, 10):
        if a.fem and os.path.exists(a.fem.format(i=i)):
            xs, ys, u, v, s11, s22, s12 = ps.preprocess(a.fem.format(i=i), case="plate")
            pred = model.predict(xs, ys, np.full_like(xs, times[i]))
            errs = [ps.relative_l2(p, f) for p, f in zip(pred[:5], (u, v, s11, s22, s12))]
            print("frame %3d t=%5.2f  rel-L2 u %.3f v %.3f s11 %.3f s22 %.3f s12 %.3f" % (i, times[i], *errs))


if __name__ == "__main__":
    main()
