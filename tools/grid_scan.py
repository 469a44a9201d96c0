"""Dev experiment (GPU, round 6): launch time of a fused LDS-operand kernel against the size of its persistent grid (pinn_debug_set_fused_grid_cap):
fewer workgroups = a smaller memory footprint (profiles/r06_footprint_and_cache_policy.txt), but fewer compute units at work.
   python tools/grid_scan.py WIDTH [DEPTH=8] [caps ...]        (wave head, 1 M points; the 3-D net: tools/nc3d_grid_scan.py)"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
width = int(sys.argv[1]) if len(sys.argv) > 1 else 100
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
caps = [int(v) for v in sys.argv[3:]] or [256, 248, 240, 224, 208, 192]
layers = [3] + depth * [width] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 1_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18)
eng.lib.lib.pinn_debug_set_fused_grid_cap.argtypes = [ctypes.c_int]
tw = np.ones(7) / n
for _ in range(20):
    eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
for rep in range(2):
    for cap in caps:
        eng.lib.lib.pinn_debug_set_fused_grid_cap(cap)
        for _ in range(3):
            eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
        prof = eng.lib.set_profile_buffer(True)
        ms = []
        for _ in range(12):
            eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
            ms.append(float(prof[:3].sum()))
        eng.lib.set_profile_buffer(False)
        ms = np.sort(np.array(ms))
        print(f'{depth}x{width} grid cap {cap:3d}: kernel ms per 1 M points median {ms[6]:.3f} (min {ms[0]:.3f})', flush=True)
eng.lib.lib.pinn_debug_set_fused_grid_cap(0)
