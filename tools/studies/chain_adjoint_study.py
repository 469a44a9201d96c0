"""Dev experiment (round 5): does the REVERSE CHAIN need the adjoints' low part?  Round 4 showed the weight gradient does not (ZDB: one MFMA per
product there), provided the adjoints sit in fp16's normal range (normalised term weights x 16).  The reverse chain still multiplies
V_hi z_hi + (V_hi / 2^11) z_lo' + V_lo z_hi: three MFMAs per product, and emits a scaled low part per adjoint value.  Here the kernel's arithmetic
(tools/studies/wgrad_operand_study.py + the shipped low-part policy) with the chain's adjoint operand as
   'hi+lo' : what ships            'hi' : fp16 high part only (two MFMAs per product: V_hi z_hi + V_lo z_hi)
all adjoints carried in the kernel's units (normalised term weights x 16), per weight layer and bias: error against float64 as a multiple of
host-fp32's (the GPU tests' bound: 6).     python tools/studies/chain_adjoint_study.py [case ...]"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
argv, sys.argv = sys.argv[1:], sys.argv[:1] + ['__none__']
import importlib.util
spec = importlib.util.spec_from_file_location('wos', '/root/repo/tools/studies/wgrad_operand_study.py')
wos = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(wos)
except Exception:
    pass
from oracle import pinn_oracle as po, golden_points as gp
f32 = np.float32
r16 = wos.r16
WS = f32(32.0)

def parked(v, l, lo_from=3):
    v = np.asarray(v, f32); hi = r16(v)
    if 1 <= l and l + 1 < lo_from and l >= 1:       # cache index l <-> state S_{l+1}: S_2 travels without its record
        return hi
    lo = (v - hi).astype(np.float16)
    return hi + (lo.view(np.uint16) & np.uint16(0xff00)).view(np.float16).astype(f32)

def mm_bwd(Z, Wt, chain):
    Z = np.asarray(Z, f32); V = np.asarray(Wt, f32) * WS
    Zh = r16(Z); Vh = r16(V); Vl = r16(V - Vh)
    if chain == 'hi':
        return (Zh @ Vh + Zh @ Vl) / WS
    Zl = r16((Z - Zh) * f32(2048)); W2 = r16(Vh / f32(2048))
    return (Zh @ Vh + Zl @ W2 + Zh @ Vl) / WS

def run(X, Ws, bs, lb, ub, normalize, tw, chain):
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
        z = wos.mm_fwd(h, Ws[l]) + bs[l]; dz = [wos.mm_fwd(d, Ws[l]) for d in dh]
        h = np.tanh(z); s = 1 - h * h; dh = [s * d for d in dz]
        cache.append((h, dh))
    Y = wos.mm_fwd(h, Ws[-1]) + bs[-1]; dY = [wos.mm_fwd(d, Ws[-1]) for d in dh]
    f = po.wave2d_residuals(Y, dY)
    SC = f32(16.0 / float(np.max(tw)))                       # the kernel's units: term weights normalised by their maximum, seeds x 16
    g = (2 * f * (tw * SC)[None, :]).astype(f32)
    Yb, dYb = po.wave2d_residual_adjoint(g)
    L = len(Ws); Wbar = [None] * L; bbar = [None] * L
    wg = lambda S, Z: r16(np.asarray(S, f32)).T @ r16(np.asarray(Z, f32))          # ZDB: both factors as high parts
    bsum = lambda Z: r16(np.asarray(Z, f32)).sum(0)
    h, dh = cache[-1]
    Wbar[-1] = wg(h, Yb) + sum(wg(dh[k], dYb[k]) for k in range(3)); bbar[-1] = bsum(Yb)
    hb = mm_bwd(Yb, Ws[-1].T, chain); dhb = [mm_bwd(dYb[k], Ws[-1].T, chain) for k in range(3)]
    for l in range(L - 2, 0, -1):
        h, dh = cache[l]; hin, dhin = cache[l - 1]
        if l < L - 2: h, dh = parked(h, l), [parked(d, l) for d in dh]
        s = 1 - h * h
        zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
        Wbar[l] = wg(hin, zb) + sum(wg(dhin[k], dzb[k]) for k in range(3)); bbar[l] = bsum(zb)
        hb = mm_bwd(zb, Ws[l].T, chain); dhb = [mm_bwd(dzb[k], Ws[l].T, chain) for k in range(3)]
    h, dh = cache[0]; s = 1 - h * h
    zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
    Wbar[0] = h0.T @ zb + np.stack([sc[k] * dzb[k].sum(0) for k in range(3)]); bbar[0] = zb.sum(0)
    amin = float(np.percentile(np.abs(zb[zb != 0]), 1))
    return po.pack_params(Wbar, bbar, f32) / SC, amin

gd = '/root/repo/tests/golden'
for case in (argv or ['inf20s', 'semi16s', 'conf14s', 'wave64']):
    w = np.load(f'{gd}/weights_{case}.npz'); g = np.load(f'{gd}/golden_{case}_32k.npz')
    layers = [int(v) for v in w['layers']]; L = len(layers) - 1
    Ws = [w[f'W{i}'] for i in range(L)]; bs = [w[f'b{i}'] for i in range(L)]
    flat = po.pack_params(Ws, bs)
    lb, ub, norm, n = g['lb'], g['ub'], bool(g['normalize']), int(g['n'])
    X = gp.wave_points(lb, ub, tuple(g['src']), n)
    for m, off in ((64, 0), (1024, 3000), (4096, 0), (32768, 0)):
        Xm = X[off:off + m]; tw = np.ones(7) / m
        _, g64, _ = po.wave2d_loss_grad(flat, layers, Xm[:, 0], Xm[:, 1], Xm[:, 2], lb, ub, norm, term_weights=tw)
        _, g32, _ = po.wave2d_loss_grad(flat.astype(f32), layers, Xm[:, 0], Xm[:, 1], Xm[:, 2], lb, ub, norm, term_weights=tw, dtype=f32)
        e32 = wos.layer_errs(g32, g64, layers); eb32 = wos.layer_errs(g32, g64, layers, True)
        print(f'{case} n={m}: multiples of host-fp32 error per weight layer | bias')
        for chain in ('hi+lo', 'hi'):
            gv, amin = run(Xm, Ws, bs, lb, ub, norm, tw, chain)
            e = wos.layer_errs(gv, g64, layers) / e32; eb = wos.layer_errs(gv, g64, layers, True) / eb32
            print(f'   chain {chain:6s} W: ' + ' '.join(f'{a:5.1f}' for a in e) + ' | b: ' + ' '.join(f'{a:5.1f}' for a in eb) +
                  f' | max {max(e.max(), eb.max()):5.1f} | all {np.linalg.norm(gv - g64) / np.linalg.norm(g64):.1e} | 1st pct of |z_1| in kernel units {amin:.1e}', flush=True)
