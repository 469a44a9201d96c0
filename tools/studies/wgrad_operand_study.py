"""Dev experiment (round 4): what the WEIGHT GRADIENT's MFMA operands need.  The fused kernel's arithmetic emulated in numpy (forward
f16x3 with unscaled activation low parts, reverse chain with 2^11-scaled adjoint low parts, full-precision states in the activation
reverse) with the weight gradient  Wbar_l = S_{l}^T Z_l  computed four ways:
  full   : S (hi + lo) x Z (hi + lo)              (what the LDS-operand layouts and the two-kernel path do)
  cur    : S hi only   x Z (hi + scaled lo)        (the narrow layouts today: 2 MFMAs per product)
  zhi    : S hi only   x Z hi only, Z loss-scaled by 2^k into fp16's normal range  (1 MFMA per product)
  zhi_ns : the same without the loss scale         (shows why the scale is needed: tiny adjoints are fp16 subnormals)
Per weight layer: error against the float64 oracle as a multiple of the error of a host fp32 evaluation -- the bound the GPU tests use
(tests/test_gpu_parity.py: <= 6x per layer).   python tools/studies/wgrad_operand_study.py [case ...]"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import pinn_oracle as po, golden_points as gp

f32 = np.float32
def r16(a): return np.asarray(a, f32).astype(np.float16).astype(f32)
WS = f32(32.0)
def mm_fwd(A, W):
    A = np.asarray(A, f32); V = np.asarray(W, f32) * WS
    Ah = r16(A); Al = r16(A - Ah); Vh = r16(V); Vl = r16(V - Vh)
    return (Ah @ Vh + Al @ Vh + Ah @ Vl) / WS
def mm_bwd(Z, Wt):
    Z = np.asarray(Z, f32); V = np.asarray(Wt, f32) * WS
    Zh = r16(Z); Zl = r16((Z - Zh) * f32(2048)); Vh = r16(V); Vl = r16(V - Vh); W2 = r16(Vh / f32(2048))
    return (Zh @ Vh + Zl @ W2 + Zh @ Vl) / WS
def wg(S, Z, mode, zscale):
    S = np.asarray(S, f32); Z = np.asarray(Z, f32)
    Sh = r16(S)
    if mode == 'full':
        Sl = r16(S - Sh); Zh = r16(Z); Zl = r16((Z - Zh) * f32(2048))
        return Sh.T @ Zh + (Sh.T @ Zl) / f32(2048) + Sl.T @ Zh
    if mode == 'cur':
        Zh = r16(Z); Zl = r16((Z - Zh) * f32(2048))
        return Sh.T @ Zh + (Sh.T @ Zl) / f32(2048)
    if mode.startswith('zhi') and mode != 'zhi_ns':      # (also 'zhi+lo8' / 'zhi+hi': the state-precision rows)
        return (Sh.T @ r16(Z * f32(zscale))) / f32(zscale)
    if mode == 'zhi_ns':
        return Sh.T @ r16(Z)
    raise ValueError(mode)

def bsum(Z, mode, zscale):
    Z = np.asarray(Z, f32)
    if mode in ('full', 'cur'):
        Zh = r16(Z); Zl = r16((Z - Zh) * f32(2048))
        return Zh.sum(0) + Zl.sum(0) / f32(2048)
    if mode == 'zhi_ns': return r16(Z).sum(0)
    return r16(Z * f32(zscale)).sum(0) / f32(zscale)

STATE_MODE = 'full'      # what the ACTIVATION REVERSE sees of the parked states: 'full' (hi + fp16 lo), 'lo8' (hi + the low part's top byte, truncated: e5m2), 'hi'
def parked(v):
    v = np.asarray(v, f32)
    if STATE_MODE == 'full': return v
    hi = r16(v)
    if STATE_MODE == 'hi': return hi
    lo = (v - hi).astype(np.float16)
    lo8 = (lo.view(np.uint16) & np.uint16(0xff00)).view(np.float16).astype(f32)      # top byte of the fp16 low part
    return hi + lo8

def run(X, Ws, bs, lb, ub, normalize, tw, mode, zscale):
    X = np.asarray(X, f32); N = X.shape[0]
    Ws = [np.asarray(W, f32) for W in Ws]; bs = [np.asarray(b, f32) for b in bs]
    if normalize:
        lb = np.asarray(lb, f32); ub = np.asarray(ub, f32); sc = 2 / (ub - lb); h0 = 2 * (X - lb) / (ub - lb) - 1
    else:
        sc = np.ones(3, f32); h0 = X
    z = h0 @ Ws[0] + bs[0]; dz = [np.tile(sc[k] * Ws[0][k], (N, 1)) for k in range(3)]
    h = np.tanh(z); s = 1 - h * h; dh = [s * d for d in dz]
    cache = [(h, dh)]
    for l in range(1, len(Ws) - 1):
        z = mm_fwd(h, Ws[l]) + bs[l]; dz = [mm_fwd(d, Ws[l]) for d in dh]
        h = np.tanh(z); s = 1 - h * h; dh = [s * d for d in dz]
        cache.append((h, dh))
    Y = mm_fwd(h, Ws[-1]) + bs[-1]; dY = [mm_fwd(d, Ws[-1]) for d in dh]
    f = po.wave2d_residuals(Y, dY)
    g = (2 * f * tw[None, :]).astype(f32)
    Yb, dYb = po.wave2d_residual_adjoint(g)
    L = len(Ws); Wbar = [None] * L; bbar = [None] * L
    h, dh = cache[-1]
    Wbar[-1] = wg(h, Yb, mode, zscale) + sum(wg(dh[k], dYb[k], mode, zscale) for k in range(3)); bbar[-1] = bsum(Yb, mode, zscale)
    hb = mm_bwd(Yb, Ws[-1].T); dhb = [mm_bwd(dYb[k], Ws[-1].T) for k in range(3)]
    for l in range(L - 2, 0, -1):
        h, dh = cache[l]; hin, dhin = cache[l - 1]
        if l < L - 2: h, dh = parked(h), [parked(d) for d in dh]       # (the top state comes from the forward's registers)
        s = 1 - h * h
        zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
        Wbar[l] = wg(hin, zb, mode, zscale) + sum(wg(dhin[k], dzb[k], mode, zscale) for k in range(3)); bbar[l] = bsum(zb, mode, zscale)
        hb = mm_bwd(zb, Ws[l].T); dhb = [mm_bwd(dzb[k], Ws[l].T) for k in range(3)]
    h, dh = cache[0]; s = 1 - h * h      # (S_1 is recomputed in full)
    zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
    Wbar[0] = h0.T @ zb + np.stack([sc[k] * dzb[k].sum(0) for k in range(3)]); bbar[0] = zb.sum(0)
    zmax = max(float(np.abs(zb).max()), 0.0)
    return po.pack_params(Wbar, bbar, f32)

def layer_errs(gv, g64, layers, bias=False):
    W, b = po.unpack_params(np.asarray(gv, np.float64), layers); W64, b64 = po.unpack_params(g64, layers)
    if bias: W, W64 = b, b64
    return np.array([np.linalg.norm(W[l] - W64[l]) / np.linalg.norm(W64[l]) for l in range(len(W))])

gd = '/root/repo/tests/golden'
cases = sys.argv[1:] or ['inf20s', 'wave64']
for case in cases:
    w = np.load(f'{gd}/weights_{case}.npz'); g = np.load(f'{gd}/golden_{case}_32k.npz')
    layers = [int(v) for v in w['layers']]; L = len(layers) - 1
    Ws = [w[f'W{i}'] for i in range(L)]; bs = [w[f'b{i}'] for i in range(L)]
    flat = po.pack_params(Ws, bs)
    lb, ub, norm, n = g['lb'], g['ub'], bool(g['normalize']), int(g['n'])
    X = gp.wave_points(lb, ub, tuple(g['src']), n)
    for m in (4096, n):
        Xm = X[:m]; tw = np.ones(7) / m
        _, g64, _ = po.wave2d_loss_grad(flat, layers, Xm[:, 0], Xm[:, 1], Xm[:, 2], lb, ub, norm, term_weights=tw)
        _, g32, _ = po.wave2d_loss_grad(flat.astype(f32), layers, Xm[:, 0], Xm[:, 1], Xm[:, 2], lb, ub, norm, term_weights=tw, dtype=f32)
        e32 = layer_errs(g32, g64, layers)
        zscale = 2.0 ** np.round(np.log2(m))          # host-side loss scale: the term weights carry 1/N
        print(f'{case} n={m}: host fp32 rel err per W layer ' + ' '.join(f'{e:.1e}' for e in e32))
        eb32 = layer_errs(g32, g64, layers, True)
        for mode, sc, stm in (('full', 1, 'full'), ('cur', 1, 'full'), ('zhi/64', 1 / 64., 'full'), ('zhi/8', 1 / 8., 'full'), ('zhi', 1, 'full'), ('zhi*16', 16, 'full'),
                              ('zhi*256', 256, 'full'), ('zhi_ns', 1, 'full'), ('zhi+lo8', 16, 'lo8'), ('zhi+hi', 16, 'hi')):
            globals()['STATE_MODE'] = stm
            gv = run(Xm, Ws, bs, lb, ub, norm, tw, mode, zscale * sc)
            e = layer_errs(gv, g64, layers); eb = layer_errs(gv, g64, layers, True)
            print(f'   {mode:8s} W x fp32: ' + ' '.join(f'{a / b:5.1f}' for a, b in zip(e, e32)) + ' | b x fp32: ' + ' '.join(f'{a / b:5.1f}' for a, b in zip(eb, eb32)) + f' | all {np.linalg.norm(gv - g64) / np.linalg.norm(g64):.1e}')
