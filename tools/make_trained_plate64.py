"""Fixture generator (GPU): a TRAINED plate 8x64 composite net for the parity tests of BASELINE configs[2]'s own kernel.

Round-5 review: `Fused<OpF16,3,64,8,5>` -- the five-stream register-state kernel with the one-part weight gradient (ZDB), one-byte parked
low parts (LO8) and `S1_HI_BY_WG` -- had been compared with the oracle at fresh Xavier weights only; every trained-weight plate test loads the
reference's 8 x 70 net, i.e. the LDS-operand kernel.  This script trains the BASELINE net (8 x 64 uv net; the FROZEN distance / particular
nets are the reference's own trained 4 x 20 nets, tests/golden/weights_plate_{dist,part}.npz, converted from its pickles) with THIS framework
on the plate problem (PLATE:870-929 point sets; Adam, then L-BFGS on the device) until the PDE residuals are small, and writes the weights.
They are the framework's own product, not reference data: what they provide is a point in weight space where the residuals are differences of
O(1) terms -- the regime in which operand rounding is amplified.  The float64 oracle's outputs at these weights (oracle/make_golden.py
`plate64`) are the golden vectors.
  gpurun -- python tools/make_trained_plate64.py gpurun_out/plate64     then copy weights_plate64_uv.npz to tests/golden/ and run make_golden plate64
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pinn_elastodynamics_amd import pointsets as ps                  # noqa: E402
from pinn_elastodynamics_amd.elastic_wave import unpack_params       # noqa: E402
from pinn_elastodynamics_amd.plate_hole import PINN                  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/plate64"
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 420.0
start = sys.argv[3] if len(sys.argv) > 3 else ""          # continue from a weights file of an earlier run
backend = sys.argv[4] if len(sys.argv) > 4 else "torch"   # "scipy": L-BFGS-B in float64 on the host (PLATE:220-247), for the last decade: torch's strong-Wolfe search gives up at ~3e-4
os.makedirs(out, exist_ok=True)
gd = os.path.join(ROOT, "tests", "golden")
c = ps.plate_case(seed=1111, n_collo=70000, n_refine=40000, uv_width=64)
m = PINN(c["Collo"], c["HOLE"], c["IC"], c["LF"], c["RT"], c["UP"], c["LW"], c["DIST"], c["uv_layers"], c["dist_layers"], c["part_layers"], c["lb"], c["ub"],
         partDir=os.path.join(gd, "weights_plate_part.npz"), distDir=os.path.join(gd, "weights_plate_dist.npz"), uvDir=start, precision="f16x3", seed=1111,
         verbose=False)
assert m.eng["uv"].path("plate") == "fused-registers", m.eng["uv"].path("plate")
log = []
t0 = time.time()


def note(tag):
    g = m.getloss()
    log.append(f"{tag}: loss {g['loss']:.3e}  f_uv {g['loss_f_uv']:.3e}  f_s {g['loss_f_s']:.3e}  HOLE {g['loss_HOLE']:.3e}  ({time.time() - t0:.0f} s)")
    print(log[-1], flush=True)
    return g


note(("continuing from " + start) if start else "fresh Xavier uv net, the reference's trained distance / particular nets")
if not start:
    m.train(3000, 1e-3)
    note("Adam 3000 steps lr 1e-3")
    m.train(2000, 2e-4)
    note("Adam 2000 steps lr 2e-4")
stage = 0
last, stalls, lr = None, 0, 1e-4
while time.time() - t0 < budget and stage < 400:
    m.train_bfgs(options=dict(maxiter=5000, maxfun=5500), backend=backend)
    g = note(f"L-BFGS stage {stage} ({backend})")
    stage += 1
    if g["loss_f_uv"] + g["loss_f_s"] < 1.5e-5:
        break
    if last is not None and g["loss"] > 0.999 * last:
        # the line search gives up (the f16x3 loss / gradient pair is consistent to ~1e-3 only): a stretch of Adam at a small rate moves the
        # point and resets the curvature pairs; the rate falls as the stalls repeat
        stalls += 1
        if stalls > 60:
            break
        lr = (1e-4, 7e-5, 5e-5)[stalls % 3]
        m.train(3000, lr)
        g = note(f"Adam 3000 steps lr {lr:.1e} (stall {stalls})")
    last = g["loss"]
# Off the optimiser's own optimum (as tools/make_trained64.py): at the weights L-BFGS converged to IN f16x3 ARITHMETIC the f16x3 gradient is ~0 by
# construction and a comparison "f16x3 vs float64" there is biased against the mode that did the training.  A seeded relative perturbation of 1e-4
# leaves the residual losses where they are and favours no arithmetic.
flat64 = m.theta["uv"].cpu().numpy().astype(np.float64)
flat64 = flat64 * (1.0 + 1e-4 * np.random.default_rng(78).standard_normal(flat64.size))
W, b = unpack_params(flat64, c["uv_layers"])
np.savez_compressed(os.path.join(out, "weights_plate64_uv.npz"), layers=np.array(c["uv_layers"]), **{f"W{i}": w for i, w in enumerate(W)},
                    **{f"b{i}": x.reshape(-1) for i, x in enumerate(b)})
open(os.path.join(out, "train_log.txt"), "w").write("\n".join(log) + "\n")
