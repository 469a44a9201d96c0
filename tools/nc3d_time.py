"""Dev tool (GPU): launches of the fused 3-D kernel (10x128, four inputs, five first-order streams: BASELINE configs[4]) on 1,000,000 points,
for tools/pmc_collect.sh NAME nc3d and for timing.   python tools/nc3d_time.py [NAME of build/exp/NAME/libpinn_hip.so] [launches]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
layers = [4] + 10 * [128] + [12]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 1_000_000
lb, ub = [0.0, 0.0, 0.0, 0.0], [30.0, 30.0, 30.0, 20.0]
X = np.random.default_rng(1).random((n, 4)) * np.array(ub)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(4)]
libp = os.path.join(ROOT, 'build/exp', sys.argv[1], 'libpinn_hip.so') if len(sys.argv) > 1 and sys.argv[1] != '-' else None
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 17, **({'lib_path': libp} if libp else {}))
tw = np.ones(12) / n
for _ in range(3):
    eng.nc3d_loss_grad(theta, *xs, lb, ub, True, tw)
torch.cuda.synchronize()
eng.lib.profile_ring_arm(256)
for _ in range(reps):
    eng.nc3d_loss_grad(theta, *xs, lb, ub, True, tw)
torch.cuda.synchronize()
ms, tags = eng.lib.profile_ring_read()
ms = np.sort(ms[tags >= 4])
print(f'3-D 10x128 fused launch, 1 M points: ms min {ms[0]:.2f} median {ms[len(ms) // 2]:.2f} max {ms[-1]:.2f}  ({len(ms)} launches)', flush=True)
