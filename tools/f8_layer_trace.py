import sys, ctypes, os, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0'); layers = [3] + 8 * [80] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, bs)
n = 2_000_000
X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 20.])
eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 18, lib_path='build/exp/f8stamps/libpinn_hip.so')
eng.lib.lib.pinn_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
stamps = torch.zeros(128, dtype=torch.int64, device=dev)
theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
tw = np.ones(7) / n
for rep in range(3):
    eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    eng.lib.lib.pinn_debug_set_stamp_buffer(stamps.data_ptr())
    eng.wave_loss_grad(theta, *xs, [0, 0, 0], [30, 30, 20], True, tw)
    torch.cuda.synchronize()
    eng.lib.lib.pinn_debug_set_stamp_buffer(None)
    t = stamps.cpu().numpy(); c = t
    b2, b3, b4 = c[32 + 2], c[32 + 3], c[32 + 4]          # chain: barrier behind layers 2, 3, 4
    print('chain layer 3: gemm done +%d, epilogue+store done +%d, barrier done +%d   | layer 4: +%d +%d +%d' % (c[53] - b2, c[54] - b2, b3 - b2, c[56] - b3, c[57] - b3, b4 - b3))
    w2, w3, w4 = t[64 + 50], t[64 + 51], t[64 + 52]
    print('wgrad layer 3: gemm done +%d, store done +%d, park issued +%d, barrier done +%d | layer 4: +%d +%d +%d +%d   (wgrad barrier2 vs chain barrier2: %+d)' % (
        t[105] - w2, t[106] - w2, t[107] - w2, w3 - w2, t[108] - w3, t[109] - w3, t[110] - w3, w4 - w3, w2 - b2))
