"""Dev experiment: forward in f16x3 (split), backward with the PARKED state S rounded to fp16 (hi only) while the
adjoint chain and the weights stay split -- gradient error vs float64 oracle at the reference's trained weights."""
import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from oracle import pinn_oracle as po
from precision_study import mm, r_f16, r_bf16, rel

def run(X, Ws, bs, lb, ub, normalize, tw, state_round, mode='f16x3s'):
    X = np.asarray(X, np.float32); N = X.shape[0]
    Ws = [np.asarray(W, np.float32) for W in Ws]; bs = [np.asarray(b, np.float32) for b in bs]
    if normalize:
        lb = np.asarray(lb, np.float32); ub = np.asarray(ub, np.float32); sc = 2 / (ub - lb); h = 2 * (X - lb) / (ub - lb) - 1
    else:
        sc = np.ones(3, np.float32); h = X
    z = h @ Ws[0] + bs[0]; dz = [np.tile(sc[k] * Ws[0][k], (N, 1)) for k in range(3)]
    h = np.tanh(z); s = 1 - h * h; dh = [s * d for d in dz]
    cache = [(h, dh)]
    for l in range(1, len(Ws) - 1):
        z = mm(h, Ws[l], mode) + bs[l]; dz = [mm(d, Ws[l], mode) for d in dh]
        h = np.tanh(z); s = 1 - h * h; dh = [s * d for d in dz]; cache.append((h, dh))
    Y = mm(h, Ws[-1], mode) + bs[-1]; dY = [mm(d, Ws[-1], mode) for d in dh]
    f = po.wave2d_residuals(Y, dY)
    g = (2 * f * tw[None, :]).astype(np.float32); Yb, dYb = po.wave2d_residual_adjoint(g)
    rs = state_round
    cache = [(rs(h_), [rs(d) for d in dh_]) for (h_, dh_) in cache[:-1]] + [cache[-1]]   # top state stays in registers (exact)
    L = len(Ws); Wbar = [None] * L; bbar = [None] * L
    h, dh = cache[-1]
    Wbar[-1] = mm(h.T, Yb, mode) + sum(mm(dh[k].T, dYb[k], mode) for k in range(3)); bbar[-1] = Yb.sum(0)
    hb = mm(Yb, Ws[-1].T, mode); dhb = [mm(dYb[k], Ws[-1].T, mode) for k in range(3)]
    for l in range(L - 2, 0, -1):
        h, dh = cache[l]; hin, dhin = cache[l - 1]
        s = 1 - h * h
        zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
        Wbar[l] = mm(hin.T, zb, mode) + sum(mm(dhin[k].T, dzb[k], mode) for k in range(3)); bbar[l] = zb.sum(0)
        hb = mm(zb, Ws[l].T, mode); dhb = [mm(dzb[k], Ws[l].T, mode) for k in range(3)]
    h, dh = cache[0]; s = 1 - h * h
    zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
    Xn = (2 * (X - lb) / (ub - lb) - 1) if normalize else X
    Wbar[0] = Xn.T @ zb + np.stack([sc[k] * dzb[k].sum(0) for k in range(3)]); bbar[0] = zb.sum(0)
    return po.pack_params(Wbar, bbar, np.float32)

ident = lambda a: a
def r_f16_then_lo_bf16(a):   # hi fp16 + lo as 8-bit-mantissa (3 bytes per value)
    hi = r_f16(a); return hi + r_bf16((a - hi) * 2048.0) / 2048.0
for case in ['inf20s', 'semi16s', 'conf14s']:
    w = np.load(f'tests/golden/weights_{case}.npz'); g = np.load(f'tests/golden/golden_{case}.npz')
    layers = list(w['layers']); L = len(layers) - 1
    Ws = [w[f'W{i}'] for i in range(L)]; bs = [w[f'b{i}'] for i in range(L)]
    X = g['X']; N = X.shape[0]; tw = np.ones(7) / N
    ref = g['grad'].astype(np.float64)
    for name, fn in [('exact state', ident), ('state fp16', r_f16), ('state bf16', r_bf16), ('state f16+8bit lo', r_f16_then_lo_bf16)]:
        gr = run(X, Ws, bs, g['lb'], g['ub'], bool(g['normalize']), tw, fn)
        print(f'{case:8s} {name:20s} grad rel err {rel(gr, ref):.2e}')
rng = np.random.default_rng(1); layers = [3] + 8 * [64] + [7]
Ws, bs = po.xavier_init(layers, rng); lb = [0, 0, 0]; ub = [30, 30, 20]
X = po.collocation_points(2048, lb, ub, rng); flat = po.pack_params(Ws, bs); tw = np.ones(7) / 2048
ss, gg, ff = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, True, term_weights=tw)
for name, fn in [('exact state', ident), ('state fp16', r_f16)]:
    print('xavier 8x64', name, f'{rel(run(X, Ws, bs, lb, ub, True, tw, fn), gg):.2e}')
