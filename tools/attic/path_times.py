"""Dev tool (GPU): wall time per loss+grad evaluation of the two-kernel path families at 1M points (plate 5-stream, wide wave nets)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
n = 1_000_000
rng = np.random.default_rng(0)
xs = [torch.rand(n, device=dev) * s for s in (0.5, 0.5, 10.0)]
def net(layers):
    W, b = po.xavier_init(layers, rng); return torch.from_numpy(po.pack_params(W, b).astype(np.float32)).to(dev)
def timeit(f, reps=3):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.time() - t) / reps * 1e3
for width, depth in ((64, 8), (70, 8), (80, 8), (100, 8), (140, 6)):
    lw = [3] + depth * [width] + [7]
    e = HipEngine(lw, device=dev, max_points=1 << 18); th = net(lw)
    e.lib.set_fused(False)
    ms2 = timeit(lambda: e.wave_loss_grad(th, *xs, [0, 0, 0], [30, 30, 20], True, np.ones(7) / n))
    e.lib.set_fused(True)
    msf = timeit(lambda: e.wave_loss_grad(th, *xs, [0, 0, 0], [30, 30, 20], True, np.ones(7) / n))
    print(f'wave  {depth}x{width:3d}: two-kernel {ms2:7.2f} ms  default {msf:7.2f} ms per 1M points', flush=True)
for width in (64, 70):
    lp = [3] + 8 * [width] + [5]
    e = HipEngine(lp, device=dev, max_points=1 << 18); th = net(lp)
    frozen = torch.rand((2, 5, 5, n), device=dev)
    ms = timeit(lambda: e.plate_loss_grad(th, *xs, [0, 0, 0], [0.5, 0.5, 10], False, frozen, [10.0 / n] * 5))
    mss = timeit(lambda: e.net_streams(th, *xs, [0, 0, 0], [0.5, 0.5, 10], False))
    print(f'plate 8x{width:3d}: loss+grad {ms:7.2f} ms  streams (forward only) {mss:7.2f} ms per 1M points', flush=True)
