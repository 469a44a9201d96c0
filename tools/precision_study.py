"""Dev experiment (not product, not oracle): emulate MFMA operand rounding modes in numpy and
measure field / residual / gradient error against the float64 oracle on the reference's trained weights."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import pinn_oracle as po

def r_bf16(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)
def r_f16(a): return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)

def mm(A, B, mode):
    A = np.asarray(A, np.float32); B = np.asarray(B, np.float32)
    if mode == 'f32': return A @ B
    r = r_bf16 if 'bf16' in mode else r_f16
    if mode in ('bf16', 'f16'): return r(A) @ r(B)
    Ah, Bh = r(A), r(B)
    if mode.endswith('x3s'):   # scaled lo (f16 only): lo' = (a-hi)*2^11
        Al, Bl = r((A - Ah) * 2048.0), r((B - Bh) * 2048.0)
        return Ah @ Bh + (Ah @ Bl + Al @ Bh) * np.float32(1 / 2048.0)
    Al, Bl = r(A - Ah), r(B - Bh)
    return Ah @ Bh + (Ah @ Bl + Al @ Bh)

def fwd(X, Ws, bs, lb, ub, normalize, mode, bwd_mode=None, tw=None):
    X = np.asarray(X, np.float32); N = X.shape[0]
    Ws = [np.asarray(W, np.float32) for W in Ws]; bs = [np.asarray(b, np.float32) for b in bs]
    if normalize:
        lb = np.asarray(lb, np.float32); ub = np.asarray(ub, np.float32)
        sc = 2 / (ub - lb); h = 2 * (X - lb) / (ub - lb) - 1
    else:
        sc = np.ones(3, np.float32); h = X
    # first layer fp32 VALU
    z = h @ Ws[0] + bs[0]; dz = [np.tile(sc[k] * Ws[0][k], (N, 1)) for k in range(3)]
    h = np.tanh(z); s = 1 - h * h; dh = [s * d for d in dz]
    cache = [(h, dh)]
    for l in range(1, len(Ws) - 1):
        z = mm(h, Ws[l], mode) + bs[l]; dz = [mm(d, Ws[l], mode) for d in dh]
        h = np.tanh(z); s = 1 - h * h; dh = [s * d for d in dz]
        cache.append((h, dh))
    Y = mm(h, Ws[-1], mode) + bs[-1]; dY = [mm(d, Ws[-1], mode) for d in dh]
    f = po.wave2d_residuals(Y, dY)
    if bwd_mode is None: return Y, dY, f, None
    g = (2 * f * tw[None, :]).astype(np.float32)
    Yb, dYb = po.wave2d_residual_adjoint(g)
    L = len(Ws); Wbar = [None] * L; bbar = [None] * L
    h, dh = cache[-1]
    Wbar[-1] = mm(h.T, Yb, bwd_mode) + sum(mm(dh[k].T, dYb[k], bwd_mode) for k in range(3)); bbar[-1] = Yb.sum(0)
    hb = mm(Yb, Ws[-1].T, bwd_mode); dhb = [mm(dYb[k], Ws[-1].T, bwd_mode) for k in range(3)]
    for l in range(L - 2, 0, -1):
        h, dh = cache[l]; hin, dhin = cache[l - 1]
        s = 1 - h * h
        zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
        Wbar[l] = mm(hin.T, zb, bwd_mode) + sum(mm(dhin[k].T, dzb[k], bwd_mode) for k in range(3)); bbar[l] = zb.sum(0)
        hb = mm(zb, Ws[l].T, bwd_mode); dhb = [mm(dzb[k], Ws[l].T, bwd_mode) for k in range(3)]
    h, dh = cache[0]; s = 1 - h * h
    zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
    Xn = (2 * (X - lb) / (ub - lb) - 1) if normalize else X
    Wbar[0] = Xn.T @ zb + np.stack([sc[k] * dzb[k].sum(0) for k in range(3)]); bbar[0] = zb.sum(0)
    return Y, dY, f, po.pack_params(Wbar, bbar, np.float32)

def rel(a, b): return np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b)

for case in ['inf20s', 'semi16s', 'conf14s']:
    w = np.load(f'tests/golden/weights_{case}.npz'); g = np.load(f'tests/golden/golden_{case}.npz')
    layers = list(w['layers']); L = len(layers) - 1
    Ws = [w[f'W{i}'] for i in range(L)]; bs = [w[f'b{i}'] for i in range(L)]
    X = g['X']; N = X.shape[0]; tw = np.ones(7) / N
    print(case, 'max|dY|', np.abs(g['dY']).max(), 'rms f', np.sqrt((g['f']**2).mean()))
    for mode, bm in [('f32', 'f32'), ('bf16', 'bf16'), ('f16', 'f16'), ('bf16x3', 'bf16x3'), ('f16x3', 'f16x3'), ('f16x3s', 'f16x3s'), ('bf16x3', 'bf16'), ('f16x3s', 'f16'), ('f16x3s','bf16')]:
        Y, dY, f, gr = fwd(X, Ws, bs, g['lb'], g['ub'], bool(g['normalize']), mode, bm, tw)
        print(f'  fwd {mode:7s} bwd {bm:7s} fields {rel(Y, g["Y"]):.2e} jac {rel(np.stack(dY), g["dY"]):.2e} resid {rel(f, g["f"]):.2e} grad {rel(gr, g["grad"].astype(np.float64)):.2e}')
# fresh xavier 8x64
rng = np.random.default_rng(1); layers = [3] + 8 * [64] + [7]
Ws, bs = po.xavier_init(layers, rng); lb = [0, 0, 0]; ub = [30, 30, 20]
X = po.collocation_points(2048, lb, ub, rng); flat = po.pack_params(Ws, bs); tw = np.ones(7) / 2048
out = po.wave2d_fields(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, True)
ss, gg, ff = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, True, term_weights=tw)
print('xavier 8x64')
for mode, bm in [('f32', 'f32'), ('bf16', 'bf16'), ('f16', 'f16'), ('bf16x3', 'bf16x3'), ('f16x3s', 'f16x3s'), ('f16x3s','f16')]:
    Y, dY, f, gr = fwd(X, Ws, bs, lb, ub, True, mode, bm, tw)
    print(f'  fwd {mode:7s} bwd {bm:7s} fields {rel(Y, out["Y"]):.2e} resid {rel(f, ff):.2e} grad {rel(gr, gg):.2e}')
