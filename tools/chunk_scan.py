"""Dev tool (GPU): loss+gradient time of the two-kernel path per 1 M points as a function of the workspace chunk (points per pass) --
do the spill panels pay off staying inside the 256 MB Infinity Cache?"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
n = 1_000_000
for width, depth in ((80, 8), (100, 8), (140, 6)):
    layers = [3] + depth * [width] + [7]
    rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
    X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
    theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
    xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
    tw = np.ones(7) / n
    for chunk in (1 << 12, 1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 18):
        eng = HipEngine(layers, precision='f16x3', device=dev, max_points=chunk)
        for _ in range(2): eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
        torch.cuda.synchronize()
        print(f'{depth}x{width} chunk {chunk:7d} points ({eng.ws_bytes / 1e6:8.1f} MB workspace): {(time.perf_counter() - t0) / 3 * 1e3:7.2f} ms per 1M points', flush=True)
        del eng
