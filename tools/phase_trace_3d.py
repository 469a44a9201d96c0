"""Dev tool (GPU): per-phase cycles of one workgroup step of the fused 3-D kernel (10 x 128, four inputs, five streams); stamps as tools/phase_trace.py.
   python tools/phase_trace_3d.py [NAME]      (library build/exp/NAME/libpinn_hip.so, default the production library)"""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else '.')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
libp = os.path.join('build/exp', sys.argv[1], 'libpinn_hip.so') if len(sys.argv) > 1 else None
dev = torch.device('cuda:0'); NL = 10
layers = [4] + NL * [128] + [12]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 1_000_000
lb, ub = [0, 0, 0, 0], [30, 30, 30, 20]
X = np.random.default_rng(1).random((n, 4)) * np.array(ub, float)
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 17, **({'lib_path': libp} if libp else {}))
eng.lib.lib.pinn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(128, dtype=torch.int64, device=dev)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(4)]
tw = np.ones(12) / n
eng.nc3d_loss_grad(theta, *xs, lb, ub, True, tw)
prof = eng.lib.set_profile_buffer(True)
eng.nc3d_loss_grad(theta, *xs, lb, ub, True, tw)
print('launch ms per 1 M points:', float(prof[1]))
eng.lib.set_profile_buffer(False)
eng.lib.lib.pinn_debug_set_stamp_buffer(stamps.data_ptr())
eng.nc3d_loss_grad(theta, *xs, lb, ub, True, tw)
torch.cuda.synchronize()
eng.lib.lib.pinn_debug_set_stamp_buffer(None)
t = stamps.cpu().numpy()
c = t[:64]; w = t[64:]
print(f'chain wave: forward {c[1]-c[0]}  head {c[2]-c[1]}   (32-point workgroup step)')
print('  forward layer ends rel. step start (layers 4..9; the slots of 0..3 are overwritten by the reverse):', [int(c[32 + l] - c[0]) for l in range(4, NL)])
for i, L in enumerate(range(NL, -1, -1)):
    a, b, e = c[3 + 3 * i], c[4 + 3 * i], c[5 + 3 * i]
    prev_end = c[2] if i == 0 else c[5 + 3 * (i - 1)]
    wa, wb, we = w[3 * i], w[1 + 3 * i], w[2 + 3 * i]
    wprev = w[2 + 3 * (i - 1)] if i > 0 else wa
    print(f'  L={L:2d}: chain wait@A {a-prev_end:6d} put+wait@B {b-a:6d} bwd {e-b if L>0 else 0:6d}   |  wgrad wait@A {wa-wprev:6d} wait@B {wb-wa:6d} wgrad {we-wb:6d}')
print(f'  step total ~ {c[4+3*NL]-c[0]}')

if t[44] > 0:
    print('forward detail (PINN_X_FSTAMP build): per layer l: gemm | epilogue | store+barrier ; weight-gradient wave: park duration, barrier wait')
    for l in range(1, NL):
        g, e = t[44 + 2 * (l - 1)], t[45 + 2 * (l - 1)]
        start = c[36 + l - 5] if l >= 5 else 0
        end = c[32 + l] if l >= 4 else 0
        pb, pe = t[106 + 2 * l], t[107 + 2 * l]
        print(f'  l={l}: gemm end {g - c[0]:7d}  epilogue end {e - c[0]:7d}  layer end {end - c[0] if end else -1:7d} | wgrad: barrier passed {pb - c[0]:7d} park issued {pe - pb:6d}')
    print('  k-step starts of layer 6 / 7 rel. step start:', [int(t[97 + i] - c[0]) for i in range(8)])
