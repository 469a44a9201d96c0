"""Dev tool (GPU): time of the value-only side-set launch (IC 10201 + SRC 70400 points, 8x64 and 4x32 nets) per experiment library."""
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import bench
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
SRC, IC = bench.ricker_source(), bench.ic_grid()
def d(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
for name in sys.argv[1:]:
    for layers in ([3] + 8 * [64] + [7], [3] + 4 * [32] + [7]):
        if len(layers) == 6 and name != 'main':
            continue                                  # experiment libraries hold the 64-wide variant only
        kw = {} if name == 'main' else {'lib_path': os.path.join('build/exp', name, 'libpinn_hip.so')}
        e = HipEngine(layers, device=dev, max_points=1 << 17, **kw)
        W, b = po.xavier_init(layers, np.random.default_rng(0)); th = d(po.pack_params(W, b))
        tg = torch.zeros((7, SRC.shape[0]), device=dev); tg[0] = d(SRC[:, 3]); tg[1] = d(SRC[:, 4])
        lo = torch.zeros(16, device=dev); g = torch.zeros(th.numel(), device=dev)
        sets = [(d(IC[:, 0]), d(IC[:, 1]), d(IC[:, 2]), None, [1e-4] * 4 + [0] * 3, lo[:8]), (d(SRC[:, 0]), d(SRC[:, 1]), d(SRC[:, 2]), tg.contiguous(), [1e-5, 1e-5] + [0] * 5, lo[8:])]
        for _ in range(5): e.data_loss_grad_multi(th, sets, bench.LB, bench.UB, True, g)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(50): e.data_loss_grad_multi(th, sets, bench.LB, bench.UB, True, g)
        en.record(); torch.cuda.synchronize()
        print(f'{name:8s} {len(layers)-2}x{layers[1]}: side-set call {st.elapsed_time(en) / 50 * 1e3:.1f} us  loss {lo[:2].cpu().numpy()} {lo[8:10].cpu().numpy()}', flush=True)
