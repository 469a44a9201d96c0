"""Dev tool (GPU): train the BASELINE configs[1] net (8x64) on the infinite-domain case for a while (Adam, then L-BFGS on the device) and
store the weights as a fixture: tests/golden/weights_c2_trained.npz.  Trained weights are where cancellation makes the gradient
sensitive to the precision of the parked states; the reference ships trained weights only for its own widths."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pinn_elastodynamics_amd import pointsets as ps
from pinn_elastodynamics_amd.elastic_wave import DeepHPM
c = ps.infinite_case(N_f=40000, N_ext=4000, seed=7, width=64)
m = DeepHPM(c["Collo"], c["SRC"], c["IC"], c["UP"], c["uv_layers"], c["lb"], c["ub"], case="infinite", precision="f16x3", seed=11, verbose=False)
m.engine.lib.set_fused(False)                     # the accurate path for the training itself
for lr, it in ((1e-3, 3000), (3e-4, 3000), (1e-4, 2000)):
    rec = m.train(it, lr, 1)
    print('adam lr', lr, 'loss', rec[-1][-1], flush=True)
m.train_bfgs(1, options={"maxiter": int(sys.argv[1]) if len(sys.argv) > 1 else 1500, "maxfun": 4000}, backend="torch")
print('after bfgs', m.getloss(), flush=True)
flat = m.theta.detach().cpu().numpy().astype(np.float64)
out = os.path.join(ROOT, 'gpurun_out', 'weights_c2_trained.npz')
np.savez(out, flat=flat.astype(np.float32), layers=np.asarray(c["uv_layers"]), lb=c["lb"], ub=c["ub"])
print('saved', out, flat.shape)
