"""Dev tool (GPU): shader cycles of one steady-state workgroup step of the fused 8x64 kernel (in-kernel stamps of workgroup 0) next to the
launch's HIP-event time, for the default and the fp16-state variant: cycles per step x steps per workgroup / launch time = the clock the
kernel actually ran at (the counter follows the shader clock) -- "do fewer bytes buy clock?" (VERDICT r3, item 1b).
   python tools/step_cycles.py [NAME of build/exp/NAME/libpinn_hip.so]"""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0'); NL = 8
libp = os.path.join(ROOT, 'build/exp', sys.argv[1], 'libpinn_hip.so') if len(sys.argv) > 1 and sys.argv[1] != '-' else None
BRIEF = len(sys.argv) > 2
layers = [3] + NL * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n
for name, kw in ((('default', {}),) if BRIEF else (('default', {}), ('fp16 states', dict(fast_state=True)))):
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, **kw, **({'lib_path': libp} if libp else {}))
    eng.lib.lib.pinn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    stamps = torch.zeros(128, dtype=torch.int64, device=dev)
    for _ in range(40):
        eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    eng.lib.lib.pinn_debug_set_stamp_buffer(stamps.data_ptr())
    eng.lib.profile_ring_arm(64)
    for _ in range(20):
        eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    torch.cuda.synchronize()
    ms, tags = eng.lib.profile_ring_read()
    eng.lib.lib.pinn_debug_set_stamp_buffer(None)
    t = stamps.cpu().numpy(); c = t[:64]
    step = int(c[4 + 3 * NL] - c[0])
    nsteps = -(-n // 64)
    per_wg = nsteps / 256.0
    msm = float(np.median(ms[tags >= 4]))
    fwd, head = int(c[1] - c[0]), int(c[2] - c[1])
    lay = []
    for i, L in enumerate(range(NL, -1, -1)):
        a, b, e = c[3 + 3 * i], c[4 + 3 * i], c[5 + 3 * i]
        prev_end = c[2] if i == 0 else c[5 + 3 * (i - 1)]
        lay.append((L, int(a - prev_end), int(b - a), int(e - b) if L > 0 else 0))
    if BRIEF:
        rev = sum(a_ + b_ + e_ for _, a_, b_, e_ in lay)
        print(f'{sys.argv[1]:14s} step {step:6d} fwd {fwd:6d} head {head:5d} reverse {rev:6d}  launch {msm:.3f} ms  rev layers ' + ' '.join(str(a_ + b_ + e_) for _, a_, b_, e_ in lay), flush=True)
        continue
    print(f'{name:12s}: step {step} cycles (forward {fwd}, head {head}), {per_wg:.1f} steps per workgroup, launch {msm:.3f} ms  ->  {step * per_wg / (msm * 1e-3) / 1e9:.2f} GHz if the step is typical')
    print('     reverse layers (L: before first stamp | window | chain): ' + '  '.join(f'{L}:{a}|{b}|{e}' for L, a, b, e in lay), flush=True)
    del eng
