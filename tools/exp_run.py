"""Dev tool (GPU): time the fused loss+grad launch (2M points, 8x64, f16x3) of every build/exp/*/libpinn_hip.so."""
import glob, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0'); NL = 8
layers = [3] + NL * [64] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n
m = 4096
ss_o, g_o, _ = po.wave2d_loss_grad(flat, layers, X[:m, 0], X[:m, 1], X[:m, 2], [0, 0, 0], [30, 30, 20], True, term_weights=np.ones(7) / m)
names = sys.argv[1:] or sorted(os.path.basename(os.path.dirname(p)) for p in glob.glob(os.path.join(ROOT, 'build/exp/*/libpinn_hip.so')))
engs = {}
for name in names:
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, lib_path=os.path.join(ROOT, 'build/exp', name, 'libpinn_hip.so'))
    ss, g = eng.wave_loss_grad(theta, *(v[:m].contiguous() for v in xs), [0, 0, 0], [30, 30, 20], True, np.ones(7) / m)
    err = float(np.linalg.norm(g.cpu().numpy() - g_o) / np.linalg.norm(g_o))
    for _ in range(2):
        eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    engs[name] = (eng, err, [])
order_rng = np.random.default_rng(7)
for rnd in range(int(os.environ.get("EXP_ROUNDS", "4"))):                        # interleaved rounds: box-to-box and clock drift hit every variant alike
    for name in order_rng.permutation(names):                                    # (in a new order every round: a variant that always runs behind a lighter one inherits its clock)
        eng, err, ts = engs[name]
        for _ in range(6):
            ts.append(eng.wave_loss_grad_profile(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)['chain'])
for name in names:
    eng, err, ts = engs[name]
    ts = sorted(ts)
    print(f'{name:28s} fused launch ms: min {ts[0]:.3f} q1 {ts[len(ts)//4]:.3f} med {ts[len(ts)//2]:.3f}   grad err vs oracle {err:.1e}', flush=True)
