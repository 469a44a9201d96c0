"""Dev tool (GPU): per-layer gradient error of an experiment library's fused launch against the float64 oracle."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
layers = [3] + 8 * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
m = int(os.environ.get('N', 4096))
X = np.random.default_rng(1).random((m, 3)) * np.array([30, 30, 20.])
ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], [0, 0, 0], [30, 30, 20], True, term_weights=np.ones(7) / m)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
for name in sys.argv[1:]:
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, lib_path=os.path.join(ROOT, 'build/exp', name, 'libpinn_hip.so'))
    for rep in range(2):
        l, gr = eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, np.ones(7) / m)
        gr = gr.cpu().numpy().astype(np.float64)
        gW, gb = po.unpack_params(gr, layers); oW, ob = po.unpack_params(g, layers)
        print(f'{name} rep {rep}: loss {rel(l.cpu().numpy(), ss):.1e} grad {rel(gr, g):.1e} | W:', ' '.join(f'{rel(a, b):.0e}' for a, b in zip(gW, oW)), '| b:', ' '.join(f'{rel(a, b):.0e}' for a, b in zip(gb, ob)))
