"""Dev tool (GPU): run the fused kernel with the timestamp hook and print per-phase cycles of workgroup 0."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo' if __import__('os').path.exists('/root/repo/bench.py') else '.')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
prec = 'f16x3'
import os
libp = os.path.join('build/exp', sys.argv[1], 'libpinn_hip.so') if len(sys.argv) > 1 else None
dev = torch.device('cuda:0'); NL = 8
width = int(sys.argv[2]) if len(sys.argv) > 2 else 64
layers = [3] + NL * [width] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
eng = HipEngine(layers, precision=prec, device=dev, max_points=1 << 18, **({'lib_path': libp} if libp else {}))
eng.lib.lib.pinn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(128, dtype=torch.int64, device=dev)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n
eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
eng.lib.lib.pinn_debug_set_stamp_buffer(stamps.data_ptr())
eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
torch.cuda.synchronize()
eng.lib.lib.pinn_debug_set_stamp_buffer(None)
t = stamps.cpu().numpy()
c = t[:64]; w = t[64:]
print('chain wave (cycles, last step of WG0):')
print(f'  forward {c[1]-c[0]}  head {c[2]-c[1]}')
if width > 64:
    print('  forward layer ends rel. step start:', [int(c[32 + l] - c[0]) for l in range(NL)])
tot = 0
for i, L in enumerate(range(NL, -1, -1)):
    a, b, e = c[3 + 3 * i], c[4 + 3 * i], c[5 + 3 * i]
    prev_end = c[2] if i == 0 else c[5 + 3 * (i - 1)]
    print(f'  L={L}: wait@A {a-prev_end:6d}  put+wait@B {b-a:6d}  bwd {e-b if L>0 else 0:6d}')
print(f'  step total ~ {c[4+3*NL]-c[0]}')
print(f'wgrad wave forward: start {t[100]-c[0]:+d} after chain start, duration {t[101]-t[100]}')
print('wgrad stamps rel. chain start: fwd start %d, fwd end %d, round2 start %d, round2 end %d, saved %d; chain step end %d' % tuple(int(v - c[0]) for v in (t[100], t[101], t[102], t[103], t[104], c[4 + 3 * NL])))
print('wgrad wave:')
for i, L in enumerate(range(NL, -1, -1)):
    a, b, e = w[3 * i], w[1 + 3 * i], w[2 + 3 * i]
    prev_end = w[2 + 3 * (i - 1)] if i > 0 else a
    print(f'  L={L}: wait@A {a-prev_end:6d}  wait@B {b-a:6d}  wgrad {e-b:6d}')
if len(sys.argv) > 3:
    print('window detail (cycles after barrier A): chain half_store done | wgrad ops issued, counted wait done')
    for i, L in enumerate(range(NL, -1, -1)):
        print(f'  L={L}: chain {c[44 + i] - c[3 + 3 * i]:6d} | wgrad issued {t[96 + 2 * i] - w[3 * i]:6d}  waited {t[97 + 2 * i] - w[3 * i]:6d}  barrier {w[1 + 3 * i] - w[3 * i]:6d}')
