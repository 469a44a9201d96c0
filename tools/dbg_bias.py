import os, sys, numpy as np, torch
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
layers = [3] + 8 * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
m = 64
X = np.random.default_rng(1).random((m, 3)) * np.array([30, 30, 20.])
ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], [0, 0, 0], [30, 30, 20], True, term_weights=np.ones(7) / m)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
oW, ob = po.unpack_params(g, layers)
np.set_printoptions(linewidth=250, precision=1)
for name in sys.argv[1:]:
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, lib_path=os.path.join(ROOT, 'build/exp', name, 'libpinn_hip.so'))
    for rep in range(3):
        l, gr = eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, np.ones(7) / m)
        gW, gb = po.unpack_params(gr.cpu().numpy().astype(np.float64), layers)
        for L in (4, 3):
            e = (gb[L] - ob[L]) / np.abs(ob[L]).max()
            print(name, rep, 'layer', L, 'bias err x1e4 per feature:', np.array2string(e * 1e4))
