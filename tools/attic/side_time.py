"""Dev tool (GPU): the one-stream fused launch (value-only side sets, 8x64 net) against the number of workgroup steps -- fixed cost and per-step cost."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
layers = [3] + 8 * [int(sys.argv[1]) if len(sys.argv) > 1 else 64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
libp = os.path.join(ROOT, 'build/exp', sys.argv[2], 'libpinn_hip.so') if len(sys.argv) > 2 else None
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, **({'lib_path': libp} if libp else {}))
nmax = 16384 * 8
X = np.random.default_rng(1).random((nmax, 3)) * np.array([30, 30, 20.])
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tg = torch.from_numpy(np.random.default_rng(2).standard_normal((7, nmax)).astype(np.float32)).to(dev)
ow = np.ones(7)
grad = torch.empty(eng.n_params, dtype=torch.float32, device=dev)
for n in (64, 16384, 32768, 49152, 65536, 80601, 81920, 98304, 131072):
    a = [v[:n].contiguous() for v in xs]; t = tg[:, :n].contiguous()
    for _ in range(3):
        eng.data_loss_grad(theta, *a, [0, 0, 0], [30, 30, 20], True, t, ow / n, grad_out=grad)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    for i in range(20):
        ev[i].record()
        eng.data_loss_grad(theta, *a, [0, 0, 0], [30, 30, 20], True, t, ow / n, grad_out=grad)
    ev[20].record(); torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
    print(f'n {n:7d}  steps/WG {-(-n // 64) / 256:5.2f}  call ms (repack + fused + reduce): min {ts[0]:.4f} med {ts[10]:.4f}', flush=True)
