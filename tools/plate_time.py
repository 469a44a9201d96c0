"""Dev tool (GPU): plate loss + gradient (five streams, 8x64 net) per 1 M points, fused kernel against the two-kernel path."""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po, plate_oracle as pl
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
lN = [3] + 8 * [int(sys.argv[1]) if len(sys.argv) > 1 else 64] + [5]
rng = np.random.default_rng(5)
W, b = po.xavier_init(lN, rng); fN = po.pack_params(W, [0.2 * rng.standard_normal(x.shape) for x in b])
n = 1_000_000
C = np.stack([rng.random(n) * 0.5, rng.random(n) * 0.5, rng.random(n) * 10], 1)
xs = [torch.from_numpy(C[:, k].astype(np.float32)).to(dev) for k in range(3)]
frozen = torch.from_numpy(rng.standard_normal((2, 5, 5, n)).astype(np.float32)).to(dev)
th = torch.from_numpy(fN.astype(np.float32)).to(dev)
libp = os.path.join(ROOT, 'build/exp', sys.argv[2], 'libpinn_hip.so') if len(sys.argv) > 2 else None
eng = HipEngine(lN, precision='f16x3', device=dev, max_points=1 << 18, **({'lib_path': libp} if libp else {}))
tw = [10.0 / n] * 5
m = 8192
fz = frozen[:, :, :, :m].contiguous().cpu().numpy().astype(np.float64)
ss, g, _ = pl.plate_loss_grad(fN, lN, C[:m, 0], C[:m, 1], C[:m, 2], fz[0], fz[1], term_weights=np.full(5, 10.0 / m))
rel = lambda a, b: float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
for fused in (True, False):
    eng.lib.set_fused(fused)
    l_s, g_s = eng.plate_loss_grad(th, *(v[:m].contiguous() for v in xs), [0, 0, 0], [0.5, 0.5, 10], False, frozen[:, :, :, :m].contiguous(), [10.0 / m] * 5)
    e = (rel(l_s.cpu().numpy(), ss), rel(g_s.cpu().numpy(), g))
    for _ in range(3):
        eng.plate_loss_grad(th, *xs, [0, 0, 0], [0.5, 0.5, 10], False, frozen, tw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        eng.plate_loss_grad(th, *xs, [0, 0, 0], [0.5, 0.5, 10], False, frozen, tw)
    torch.cuda.synchronize()
    print(f"{'fused' if fused else 'two-kernel'}: {(time.perf_counter() - t0) * 100:.2f} ms per 1 M points; loss err {e[0]:.1e} grad err {e[1]:.1e} (8192 points vs oracle)", flush=True)
eng.lib.set_fused(True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(12)]
for i in range(11):
    ev[i].record()
    eng.plate_loss_grad(th, *xs, [0, 0, 0], [0.5, 0.5, 10], False, frozen, tw)
ev[11].record(); torch.cuda.synchronize()
print('fused, per-call event times (ms):', ' '.join(f'{ev[i].elapsed_time(ev[i+1]):.2f}' for i in range(11)))
for nn in (250_000, 500_000):
    a = [v[:nn].contiguous() for v in xs]; fz2 = frozen[:, :, :, :nn].contiguous()
    for _ in range(3): eng.plate_loss_grad(th, *a, [0, 0, 0], [0.5, 0.5, 10], False, fz2, tw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.plate_loss_grad(th, *a, [0, 0, 0], [0.5, 0.5, 10], False, fz2, tw)
    torch.cuda.synchronize(); print(f'fused {nn} points: {(time.perf_counter() - t0) * 100:.2f} ms')
