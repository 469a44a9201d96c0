"""Dev experiment (GPU): the XCD-aware step assignment (FusedArgs::n_plain, pinn_debug_set_xcd_bonus) -- launch time of the 2 M-point collocation
kernel for several skews (extra steps of the even-XCD workgroups in 1/1000), interleaved in shuffled order on one box; the gradient of every
setting against the unskewed one.   python tools/xcd_bonus_ab.py [permille ...]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
layers = [3] + 8 * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18)
settings = [int(v) for v in sys.argv[1:]] or [0, 8, 12, 16, 20, 30]
ref = None
ts = {k: [] for k in settings}
for k in settings:
    eng.lib.lib.pinn_debug_set_xcd_bonus(k)
    for _ in range(3):
        _, g = eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    g = g.cpu().numpy().astype(np.float64)
    if ref is None: ref = g
    print(f'permille {k:3d}: gradient vs the unskewed assignment {np.linalg.norm(g - ref) / np.linalg.norm(ref):.1e}', flush=True)
order = np.random.default_rng(3)
for rnd in range(int(os.environ.get("AB_ROUNDS", "8"))):
    for k in order.permutation(settings):
        eng.lib.lib.pinn_debug_set_xcd_bonus(int(k))
        for _ in range(6):
            ts[int(k)].append(eng.wave_loss_grad_profile(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)['chain'])
for k in settings:
    t = sorted(ts[k])
    print(f'permille {k:3d}: launch ms min {t[0]:.3f} q1 {t[len(t) // 4]:.3f} med {t[len(t) // 2]:.3f}', flush=True)
eng.lib.lib.pinn_debug_set_xcd_bonus(0)
