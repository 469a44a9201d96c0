"""Dev tool (GPU): HIP-event breakdown (repack / chain / wgrad / reduce) of the two-kernel path per width, 1M points."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
import os
libkw = {'lib_path': os.path.join('build/exp', sys.argv[1], 'libpinn_hip.so')} if len(sys.argv) > 1 else {}
dev = torch.device('cuda:0'); n = 1_000_000
rng = np.random.default_rng(0)
xs = [torch.rand(n, device=dev) * s for s in (30.0, 30.0, 20.0)]
for width, depth in (((64, 8),) if libkw else ((64, 8), (80, 8), (100, 8), (140, 6))):
    lw = [3] + depth * [width] + [7]
    W, b = po.xavier_init(lw, rng); th = torch.from_numpy(po.pack_params(W, b).astype(np.float32)).to(dev)
    e = HipEngine(lw, device=dev, max_points=1 << 18, **libkw)
    e.lib.set_fused(False)
    e.wave_loss_grad_profile(th, *xs, [0, 0, 0], [30, 30, 20], True, np.ones(7) / n)
    p = e.wave_loss_grad_profile(th, *xs, [0, 0, 0], [30, 30, 20], True, np.ones(7) / n)
    print(f'{depth}x{width:3d}: ' + '  '.join(f'{k} {v:7.2f}' for k, v in p.items()), ' ws GB', e.ws_bytes / 1e9, flush=True)
