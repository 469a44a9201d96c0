import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else '.')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0'); n = 1_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
layers = [3] + 8 * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng)
theta = torch.from_numpy(po.pack_params(Ws, bs).astype(np.float32)).to(dev)
for name in sys.argv[1:]:
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, lib_path=os.path.join('build/exp', name, 'libpinn_hip.so'))
    eng.lib.set_fused(False)
    tw = np.ones(7) / n
    for _ in range(3): eng.wave_loss_grad_profile(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    ms = eng.wave_loss_grad_profile(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    print(name, '  '.join(f'{k} {v:7.2f} ms' for k, v in ms.items()), flush=True)
