import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import pinn_oracle as po
from pinn_elastodynamics_amd.hip_engine import HipEngine
dev = torch.device('cuda:0')
layers = [3] + 6 * [140] + [7]
rng = np.random.default_rng(0); Ws, bs = po.xavier_init(layers, rng); flat = po.pack_params(Ws, [0.2*rng.standard_normal(b.shape) for b in bs])
for n in (32, 4096):
    X = np.random.default_rng(1).random((n, 3)) * np.array([30, 30, 14.]) - np.array([15, 15, 0.])
    theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
    xs = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(3)]
    lb, ub = [-15, -15, 0], [15, 15, 14]
    ss_o, g_o, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, False, term_weights=np.ones(7) / n)
    eng = HipEngine(layers, precision='f16x3', device=dev, max_points=1 << 14, **({'lib_path': sys.argv[1]} if len(sys.argv) > 1 else {}))
    ss, g = eng.wave_loss_grad(theta, *xs, lb, ub, False, np.ones(7) / n)
    g = g.cpu().numpy().astype(np.float64)
    print('n', n, 'loss err', np.linalg.norm(ss.cpu().numpy() - ss_o) / np.linalg.norm(ss_o))
    Wd, bd = po.unpack_params(g, layers); Wo, bo = po.unpack_params(g_o, layers)
    for l in range(len(layers) - 1):
        e = np.abs(Wd[l] - Wo[l]); sc = np.abs(Wo[l]).max()
        # which 16x16 blocks are wrong
        bad = [(i // 16, o // 16) for i in range(0, Wo[l].shape[0], 16) for o in range(0, Wo[l].shape[1], 16) if e[i:i+16, o:o+16].max() > 1e-3 * sc]
        if bad and l in (1, 3):
            i0, o0 = bad[0]; blk_d = Wd[l][16*i0:16*i0+16, 16*o0:16*o0+16]; blk_o = Wo[l][16*i0:16*i0+16, 16*o0:16*o0+16]
            print('    block', bad[0], 'dev row0', np.round(blk_d[0,:6]/sc, 4), 'ora row0', np.round(blk_o[0,:6]/sc, 4), 'ratio median', np.median(blk_d[:12]/blk_o[:12]))
            print('    dev-ora row sums / sc', np.round((blk_d-blk_o)[:12].sum(1)/sc, 4))
        print('  layer', l, 'W err %.1e b err %.1e' % (np.linalg.norm(Wd[l]-Wo[l])/np.linalg.norm(Wo[l]), np.linalg.norm(bd[l]-bo[l])/np.linalg.norm(bo[l])), 'bad blocks', bad[:12], len(bad))
