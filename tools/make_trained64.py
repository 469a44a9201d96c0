"""Fixture generator (GPU): a TRAINED 8x64 wave net for the parity tests of the headline kernel.

The reference's trained nets are 8x80, 8x100, 6x140 and 8x70 wide -- none runs through the width-64 fused kernel that BASELINE configs[1]
and the bench measure, so that kernel was only ever compared with the oracle at fresh (Xavier) weights, where nothing cancels.  This
script trains the BASELINE net with THIS framework on the infinite-domain problem (INF:634-705 point sets at a 10 s horizon; Adam, then
L-BFGS on the device) until the PDE residuals are small, and writes the weights.  They are the framework's own product, not reference
data: what they provide is a point in weight space where the residuals are differences of O(1) terms -- the regime in which operand
rounding is amplified.  The float64 oracle's outputs at these weights (oracle/make_golden.py, `trained64`) are the
golden vectors.
  gpurun -- python tools/make_trained64.py gpurun_out/trained64       then copy weights_wave64.npz to tests/golden/ and run make_golden
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pinn_elastodynamics_amd import pointsets as ps                  # noqa: E402
from pinn_elastodynamics_amd.elastic_wave import DeepHPM, unpack_params  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/trained64"
os.makedirs(out, exist_ok=True)
c = ps.infinite_case(MAX_T=10.0, N_f=60000, N_ext=5000, seed=1111, width=64)
m = DeepHPM(c["Collo"], c["SRC"], c["IC"], c["UP"], c["uv_layers"], c["lb"], c["ub"], case="infinite", precision="f16x3", seed=1111, verbose=False)
log = []
t0 = time.time()
hist = m.train(3000, 1e-3, 1)
log.append(f"Adam 3000 steps lr 1e-3: loss {hist[4][0]:.3e} -> {hist[4][-1]:.3e}  ({time.time() - t0:.0f} s)")
hist = m.train(2000, 2e-4, 1)
log.append(f"Adam 2000 steps lr 2e-4: loss -> {hist[4][-1]:.3e}  ({time.time() - t0:.0f} s)")
for stage in range(3):
    m.train_bfgs(1, options=dict(maxiter=4000, maxfun=4400), backend="torch")
    g = m.getloss()
    log.append(f"L-BFGS stage {stage}: loss {g[0]:.3e}  f_uv {g[1]:.3e}  f_s {g[2]:.3e}  IC {g[3]:.3e}  SRC {g[4]:.3e}  ({time.time() - t0:.0f} s)")
    print(log[-1], flush=True)
# Off the optimiser's own optimum: at the weights L-BFGS converged to IN f16x3 ARITHMETIC the f16x3 gradient is ~0 by construction, so
# the float64 gradient there consists of f16x3's own error and any comparison "f16x3 vs float64" is biased against the mode that did
# the training (measured: error ratios f16x3 / host-fp32 of 4-9 per layer at the raw optimum, 0.9-1.9 after this perturbation; the
# reference's nets, trained by TF1 in fp32, carry the same selection effect against fp32).  A seeded relative perturbation of 1e-4
# leaves the residual losses where they are (2.06e-5 -> 2.10e-5) and favours no arithmetic.
flat64 = m.theta.cpu().numpy().astype(np.float64)
flat64 = flat64 * (1.0 + 1e-4 * np.random.default_rng(77).standard_normal(flat64.size))
W, b = unpack_params(flat64, c["uv_layers"])
np.savez_compressed(os.path.join(out, "weights_wave64.npz"), layers=np.array(c["uv_layers"]), **{f"W{i}": w for i, w in enumerate(W)},
                    **{f"b{i}": x.reshape(-1) for i, x in enumerate(b)})
open(os.path.join(out, "train_log.txt"), "w").write("\n".join(log) + "\n")
print("\n".join(log))
