"""Dev tool (GPU): chain vs weight-gradient time of the two-kernel path (HIP events of pinn_wave2d_loss_grad_profile) per 1 M points."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
n = 1_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
for width, depth in ((64, 8), (80, 8), (100, 8), (140, 6)):
    layers = [3] + depth * [width] + [7]
    rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng)
    theta = torch.from_numpy(po.pack_params(Ws, bs).astype(np.float32)).to(dev)
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18)
    eng.lib.set_fused(False)
    tw = np.ones(7) / n
    eng.wave_loss_grad_profile(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    ms = eng.wave_loss_grad_profile(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    eng.lib.set_fused(True)
    print(f'{depth}x{width}: ' + '  '.join(f'{k} {v:7.2f} ms' for k, v in ms.items()), flush=True)
