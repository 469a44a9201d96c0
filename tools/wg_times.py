"""Dev experiment (GPU; library built with -DPINN_X_WGTIMES: tools/exp_build.sh wgtimes -DPINN_X_WGTIMES): the lifetime of EVERY workgroup of the
collocation launch (device wall clock, first to last step) -- is the persistent grid balanced?   python tools/wg_times.py wgtimes"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
layers = [3] + 8 * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, lib_path=os.path.join(ROOT, 'build/exp', sys.argv[1], 'libpinn_hip.so'))
khz = eng.lib.lib.pinn_debug_wall_clock_khz()
stamps = torch.zeros(1024, dtype=torch.int64, device=dev)
for _ in range(30):
    eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
eng.lib.set_stamp_buffer(stamps.data_ptr())
for rep in range(3):
    eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    torch.cuda.synchronize()
    t = stamps.cpu().numpy()[256:256 + 512].reshape(256, 2).astype(np.float64)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / khz, (t[:, 1] - t0) / khz          # ms
    life = end - start
    steps = np.array([len(range(b, -(-n // 64), 256)) for b in range(256)])
    per = life / steps * 1e3                                       # us per step
    print(f'rep {rep}: workgroup start spread {start.max() * 1e3:.1f} us; end min / median / max {end.min():.3f} / {np.median(end):.3f} / {end.max():.3f} ms; '
          f'us per step min / median / max {per.min():.2f} / {np.median(per):.2f} / {per.max():.2f}; workgroup 0: end {end[0]:.3f} ms, {per[0]:.2f} us per step')
    xcd = np.arange(256) % 8
    print('   median us per step by blockIdx % 8 (XCD):', ' '.join(f'{np.median(per[xcd == k]):.2f}' for k in range(8)))
    print('   slowest 8 workgroups:', [(int(b), round(float(per[b]), 2)) for b in np.argsort(per)[-8:]])
eng.lib.set_stamp_buffer(None)
