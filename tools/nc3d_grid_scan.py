"""Dev experiment (GPU, round 6): does the fused 3-D kernel's memory footprint fit the Infinity Cache?  Per workgroup it keeps 2 x 364 KB of parked
state images and 4 x 136 KB of in-memory weight-gradient sums = 1.27 MB; 256 workgroups = 325 MB against 256 MB of MALL.  Launch time per 1 M points
with the persistent grid capped (pinn_debug_set_fused_grid_cap): fewer workgroups = a smaller footprint, but also fewer compute units at work.
   python tools/nc3d_grid_scan.py [NAME of build/exp/NAME/libpinn_hip.so | -] [caps ...]"""
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
caps = [int(v) for v in sys.argv[2:]] or [256, 240, 224, 208, 192, 176, 160, 128]
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 17, **({'lib_path': libp} if libp else {}))
eng.lib.lib.pinn_debug_set_fused_grid_cap.argtypes = [__import__('ctypes').c_int]
tw = np.ones(12) / n
for rep in range(2):
    for cap in caps:
        eng.lib.lib.pinn_debug_set_fused_grid_cap(cap)
        for _ in range(2):
            eng.nc3d_loss_grad(theta, *xs, lb, ub, True, tw)
        torch.cuda.synchronize()
        eng.lib.profile_ring_arm(64)
        for _ in range(5):
            eng.nc3d_loss_grad(theta, *xs, lb, ub, True, tw)
        torch.cuda.synchronize()
        ms, tags = eng.lib.profile_ring_read()
        ms = np.sort(ms[tags >= 4])
        steps = -(-n // 32)
        print(f'grid cap {cap:3d}: launch ms median {ms[len(ms) // 2]:.2f} (min {ms[0]:.2f})  = {1e3 * ms[len(ms) // 2] / (-(-steps // cap)):.1f} us per workgroup step,'
              f' footprint {cap * 1.27:.0f} MB', flush=True)
eng.lib.lib.pinn_debug_set_fused_grid_cap(0)
