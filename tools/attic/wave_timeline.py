"""Dev tool (GPU): phase stamps of ALL eight waves of workgroup 0 for one steady-state step of the fused kernel (needs a build whose
fused_stamp writes per-wave records: tools/exp_build.sh with the all-wave stamp patch).  Prints, per reverse layer, when each wave
passes barrier A / barrier B / finishes its layer work, relative to chain wave 0's step start."""
import sys, ctypes, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
libp = os.path.join('build/exp', sys.argv[1], 'libpinn_hip.so')
dev = torch.device('cuda:0'); NL = 8
layers = [3] + NL * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, lib_path=libp)
eng.lib.lib.pinn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(512, dtype=torch.int64, device=dev)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n
eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
eng.lib.lib.pinn_debug_set_stamp_buffer(stamps.data_ptr())
eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
torch.cuda.synchronize()
eng.lib.lib.pinn_debug_set_stamp_buffer(None)
t = stamps.cpu().numpy().reshape(8, 64)
t0 = t[0, 0]
print('chain waves: step start, forward end, head end (cycles rel. chain wave 0 start)')
for w in range(4):
    print(f'  chain {w}: start {t[w,0]-t0:7d}  fwd end {t[w,1]-t0:7d}  head end {t[w,2]-t0:7d}')
print('per layer: [A passed, B passed, layer work done] for chain 0-3 | wgrad 0-3')
for i, L in enumerate(range(NL, -1, -1)):
    ch = ['%6d %6d %6d' % tuple(int(t[w, 3 + 3 * i + k] - t0) for k in range(3)) for w in range(4)]
    wg = ['%6d %6d %6d' % tuple(int(t[4 + w, 3 * i + k] - t0) for k in range(3)) for w in range(4)]
    print(f' L={L}: ' + ' | '.join(ch) + ' || ' + ' | '.join(wg))
