"""Dev experiment (GPU): the XCD-aware tail (pinn_debug_set_xcd_bonus) on the LDS-operand kernels -- the 3-D net (1 M points) and the reference's
8 x 100 net (1 M points): launches with the tail off / on interleaved on one box.   python tools/xcd_tail_wide_ab.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
n = 1_000_000
def case(layers, din):
    rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
    ub = [30.0] * (din - 1) + [20.0]
    X = np.random.default_rng(1).random((n, din)) * np.array(ub)
    theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
    xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(din)]
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 17)
    call = (lambda: eng.nc3d_loss_grad(theta, *xs, [0.0] * 4, ub, True, np.ones(12) / n)) if din == 4 else \
           (lambda: eng.wave_loss_grad(theta, *xs, [0.0] * 3, ub, True, np.ones(7) / n))
    ts = {0: [], 16: []}
    for _ in range(3): call()
    for rnd in range(6):
        for pm in ((0, 16) if rnd % 2 == 0 else (16, 0)):
            eng.lib.lib.pinn_debug_set_xcd_bonus(pm)
            call(); torch.cuda.synchronize()
            eng.lib.profile_ring_arm(64)
            for _ in range(4): call()
            torch.cuda.synchronize()
            ms, tags = eng.lib.profile_ring_read()
            ts[pm] += list(ms[tags >= 4])
    eng.lib.lib.pinn_debug_set_xcd_bonus(16)
    for pm in (0, 16):
        t = np.sort(ts[pm]); print(f'{layers[1]} x {len(layers) - 2}, {din} inputs: tail {pm:2d} permille: ms min {t[0]:.3f} med {t[len(t) // 2]:.3f} max {t[-1]:.3f}', flush=True)
case([4] + 10 * [128] + [12], 4)
case([3] + 8 * [100] + [7], 3)
case([3] + 8 * [80] + [7], 3)


def plate_case():
    """the plate's five-stream 8 x 64 kernel (configs[2]) at 2 M points"""
    n2 = 2_000_000
    lN = [3] + 8 * [64] + [5]
    rng = np.random.default_rng(5)
    W, b = po.xavier_init(lN, rng); fN = po.pack_params(W, [0.2 * rng.standard_normal(x.shape) for x in b])
    C = np.stack([rng.random(n2) * 0.5, rng.random(n2) * 0.5, rng.random(n2) * 10], 1)
    xs = [torch.from_numpy(C[:, k].astype(np.float32)).to(dev) for k in range(3)]
    frozen = torch.from_numpy(rng.standard_normal((2, 5, 5, n2)).astype(np.float32)).to(dev)
    th = torch.from_numpy(fN.astype(np.float32)).to(dev)
    eng = HipEngine(lN, precision='f16x3', device=dev, max_points=1 << 18)
    call = lambda: eng.plate_loss_grad(th, *xs, [0, 0, 0], [0.5, 0.5, 10], False, frozen, [10.0 / n2] * 5)
    ts = {0: [], 16: []}
    for _ in range(3): call()
    for rnd in range(8):
        for pm in ((0, 16) if rnd % 2 == 0 else (16, 0)):
            eng.lib.lib.pinn_debug_set_xcd_bonus(pm)
            call(); torch.cuda.synchronize()
            eng.lib.profile_ring_arm(64)
            for _ in range(4): call()
            torch.cuda.synchronize()
            ms, tags = eng.lib.profile_ring_read()
            ts[pm] += list(ms[tags >= 4])
    eng.lib.lib.pinn_debug_set_xcd_bonus(16)
    for pm in (0, 16):
        t = np.sort(ts[pm]); print(f'plate 8 x 64, five streams, 2 M points: tail {pm:2d} permille: ms min {t[0]:.3f} med {t[len(t) // 2]:.3f} max {t[-1]:.3f}', flush=True)


plate_case()
