"""Dev tool (GPU): loss + gradient of the reference's 8x80 net (padded width 96), fused LDS-operand kernel against the two-kernel path."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
width = int(sys.argv[1]) if len(sys.argv) > 1 else 80
layers = [3] + 8 * [width] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 1_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
libp = os.path.join(ROOT, 'build/exp', sys.argv[2], 'libpinn_hip.so') if len(sys.argv) > 2 else None
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, **({'lib_path': libp} if libp else {}))
modes = (True, False, True) if libp is None else (True, True)
m = 20000
ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:m, 0], X[:m, 1], X[:m, 2], [0, 0, 0], [30, 30, 20], True, term_weights=np.ones(7) / m)
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
tw = np.ones(7) / n
for _ in range(30):       # clock ramp
    eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
for fused in modes:
    eng.lib.set_fused(fused)
    l, gr = eng.wave_loss_grad(theta, *(v[:m].contiguous() for v in xs), [0, 0, 0], [30, 30, 20], True, np.ones(7) / m)
    e = (rel(l.cpu().numpy(), ss), rel(gr.cpu().numpy(), g))
    prof = eng.lib.set_profile_buffer(True)
    ms = []
    for _ in range(40):
        eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
        ms.append(float(prof[:3].sum()))
    eng.lib.set_profile_buffer(False)
    ms = np.sort(np.array(ms))
    print(f"8x{width} {'fused' if fused else 'two-kernel'}: kernel ms per 1 M points: min {ms[0]:.2f} median {ms[20]:.2f} max {ms[-1]:.2f}; loss err {e[0]:.1e} grad err {e[1]:.1e} ({m} points vs oracle)", flush=True)
eng.lib.set_fused(True)
