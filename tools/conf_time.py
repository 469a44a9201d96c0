"""Dev tool (GPU): loss+gradient launch time of the reference's confined-domain net (6 x 140, CONF:891; padded width 160) on 1,000,000 points,
fused LDS-operand kernel vs the two-kernel path (library profiling hook, HIP events), with the gradient checked against the oracle."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
layers = [3] + 6 * [140] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 1_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 14.]) - np.array([15, 15, 0.])
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
lb, ub = [-15, -15, 0], [15, 15, 14]
m = 4096
ss_o, g_o, _ = po.wave2d_loss_grad(flat, layers, X[:m, 0], X[:m, 1], X[:m, 2], lb, ub, False, term_weights=np.ones(7) / m)
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18)
for fused in (True, False):
    eng.lib.set_fused(fused)
    ss, g = eng.wave_loss_grad(theta, *(v[:m].contiguous() for v in xs), lb, ub, False, np.ones(7) / m)
    err = float(np.linalg.norm(g.cpu().numpy() - g_o) / np.linalg.norm(g_o))
    ts = []
    for i in range(8):
        ms = eng.wave_loss_grad_profile(theta, *xs, lb, ub, False, np.ones(7) / n)
        if i >= 2: ts.append(sum(ms.values()))
    print(f"6x140, 1M points, fused={fused}: {np.median(ts):.2f} ms per loss+grad (kernels {ms}), grad err vs oracle {err:.1e}")
eng.lib.set_fused(True)
