"""Dev tool (GPU): time the fused kernel with the persistent grid limited through the workspace size, to see whether the
parked-state footprint (grid x 252 KB) matters (L2/Infinity-Cache residency vs HBM streaming)."""
import sys, numpy as np, torch, time
sys.path.insert(0, '.')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
from pinn_elastodynamics_amd.capi import PinnLib
dev = torch.device('cuda:0'); layers = [3] + 8 * [64] + [7]; prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
lib = PinnLib()
min_ws = lib.min_workspace_bytes(layers, prec)
np_ = 2 if prec.endswith('x3') else 1
per_tile = 2 * (4 * np_ * 32 * (16 + 8 * 64)) * 2
fixed = min_ws - 64 * per_tile
per_wg = 4 * 7 * ((4 * 16 * 136 + 1023) // 1024 * 1024)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n
for grid in (32, 64, 128, 192, 256):
    ws = fixed + grid * per_wg + 1024
    eng = HipEngine(layers, precision=prec, device=dev, workspace_bytes=max(ws, 0))
    if eng.ws_bytes != ws:
        print('grid', grid, 'workspace raised to minimum ->', (eng.ws_bytes - fixed) // per_wg)
    ms = [eng.wave_loss_grad_profile(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)['chain'] for _ in range(4)][1:]
    print(f'grid {min(256, (eng.ws_bytes - fixed) // per_wg):4d}: fused kernel {np.mean(ms):7.3f} ms  -> per-WG step {np.mean(ms) * 1e3 / (n / 64 / min(256, (eng.ws_bytes - fixed) // per_wg)):7.2f} us')
