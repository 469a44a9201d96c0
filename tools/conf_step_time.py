"""Dev tool (GPU, round 6): one Adam step of the reference's confined-domain case at ITS OWN set sizes (CONF:901-947: ~185 k collocation points,
IC 6 000, FIX 4 x 7 000, SRC ~56 k -- value-only side sets are a third of the step's points), 6 x 140 net (CONF:891), through the model class:
the one-launch step (pinn_wave2d_step -> fused_step_kernel<..,160,6,4>) against the separate calls (step_call=False).
   python tools/conf_step_time.py [steps]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pinn_elastodynamics_amd.elastic_wave import DeepHPM
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(5)
lb, ub = np.array([-15.0, -15.0, 0.0]), np.array([15.0, 15.0, 14.0])
box = lambda n: lb + (ub - lb) * rng.random((n, 3))
Collo = box(185_000)
IC = box(6000) * [1, 1, 0]
FIX = np.concatenate([box(7000) for _ in range(4)], 0)
SRC = np.concatenate([box(56_000), 0.01 * rng.standard_normal((56_000, 2))], 1)       # (x, y, t, u, v) rows of the source ring
layers = [3] + 6 * [140] + [7]
for step_call in (True, False, True, False):
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18)
    m = DeepHPM(Collo, SRC, IC, None, layers, lb, ub, case='confined', FIX=FIX, engine=eng, seed=3, verbose=False, step_call=step_call)
    m.train(10, 1e-3, 1)
    torch.cuda.synchronize()
    eng.lib.path_counts(reset=True)
    t0 = time.perf_counter()
    m.train(steps, 1e-3, 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    print(f"CONF 6x140 step ({Collo.shape[0]} collocation + {IC.shape[0] + FIX.shape[0] + SRC.shape[0]} side points), step_call={step_call}: {dt:.3f} ms per Adam step; paths {eng.lib.path_counts(reset=True)}", flush=True)
