"""Dev experiment (round 6): which LOW parts the weight gradient of the LDS-operand layouts needs.  The kernel's arithmetic in numpy (as
tools/studies/wgrad_operand_study.py: forward f16x3, reverse chain with 2^11-scaled adjoint low parts, exact states in the activation reverse)
with  Wbar_l = S_l^T Z_l  computed as
  full  : S (hi + lo) x Z (hi + lo)      3 MFMAs per product (what the LDS-operand layouts do today)
  s_lo8 : S (hi + lo8) x Z (hi + lo)      3 MFMAs, the state's low part as the one-byte e5m2 the narrow kernels park (would cut the parked images by a quarter)
  s_hi  : S hi        x Z (hi + lo)      2 MFMAs; the state image the weight gradient reads needs no low part
  z_hi  : S (hi + lo) x Z hi             2 MFMAs; the adjoint image needs no low part
  hi    : S hi        x Z hi             1 MFMA (round 4: fails at inf10s)
on the 1024-point golden sets (the GPU tests' per-layer bound: error <= 6 x host-fp32's error).
   python tools/studies/wgrad_split_study.py [case ...]"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import pinn_oracle as po

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
def wg(S, Z, mode):
    S = np.asarray(S, f32); Z = np.asarray(Z, f32)
    Sh = r16(S); Sl = r16(S - Sh); Zh = r16(Z); Zl = r16((Z - Zh) * f32(2048))
    if mode == 's_lo8':          # the parked image's low part as ONE byte (the top byte of the fp16: e5m2, three significant bits) -- the narrow kernels' LO8 format
        Sl = (Sl.astype(np.float16).view(np.uint16) & np.uint16(0xff00)).view(np.float16).astype(f32)
    out = Sh.T @ Zh
    if mode in ('full', 's_hi', 's_lo8'): out = out + (Sh.T @ Zl) / f32(2048)
    if mode in ('full', 'z_hi', 's_lo8'): out = out + Sl.T @ Zh
    return out
def bsum(Z, mode):
    Z = np.asarray(Z, f32); Zh = r16(Z); Zl = r16((Z - Zh) * f32(2048))
    return Zh.sum(0) + (Zl.sum(0) / f32(2048) if mode in ('full', 's_hi', 's_lo8') else 0)

def run(X, Ws, bs, lb, ub, normalize, tw, mode, seed_scale):
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
    g = (2 * f * tw[None, :] * seed_scale).astype(f32)
    Yb, dYb = po.wave2d_residual_adjoint(g)
    L = len(Ws); Wbar = [None] * L; bbar = [None] * L
    h, dh = cache[-1]
    Wbar[-1] = wg(h, Yb, mode) + sum(wg(dh[k], dYb[k], mode) for k in range(3)); bbar[-1] = bsum(Yb, mode)
    hb = mm_bwd(Yb, Ws[-1].T); dhb = [mm_bwd(dYb[k], Ws[-1].T) for k in range(3)]
    for l in range(L - 2, 0, -1):
        h, dh = cache[l]; hin, dhin = cache[l - 1]
        s = 1 - h * h
        zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
        Wbar[l] = wg(hin, zb, mode) + sum(wg(dhin[k], dzb[k], mode) for k in range(3)); bbar[l] = bsum(zb, mode)
        hb = mm_bwd(zb, Ws[l].T); dhb = [mm_bwd(dzb[k], Ws[l].T) for k in range(3)]
    h, dh = cache[0]; s = 1 - h * h
    zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
    Wbar[0] = h0.T @ zb + np.stack([sc[k] * dzb[k].sum(0) for k in range(3)]); bbar[0] = zb.sum(0)
    return po.pack_params(Wbar, bbar, f32) / f32(seed_scale)

def layer_errs(gv, g64, layers, bias=False):
    W, b = po.unpack_params(np.asarray(gv, np.float64), layers); W64, b64 = po.unpack_params(g64, layers)
    if bias: W, W64 = b, b64
    return np.array([np.linalg.norm(W[l] - W64[l]) / np.linalg.norm(W64[l]) for l in range(len(W))])

gd = '/root/repo/tests/golden'
for case in sys.argv[1:] or ['inf10s', 'inf20s', 'semi16s', 'conf14s']:
    w = np.load(f'{gd}/weights_{case}.npz'); g = np.load(f'{gd}/golden_{case}.npz')
    layers = [int(v) for v in w['layers']]; L = len(layers) - 1
    Ws = [w[f'W{i}'] for i in range(L)]; bs = [w[f'b{i}'] for i in range(L)]
    flat = po.pack_params(Ws, bs)
    X, lb, ub, norm = g['X'], g['lb'], g['ub'], bool(g['normalize'])
    m = X.shape[0]; tw = np.ones(7) / m
    _, g64, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, norm, term_weights=tw)
    _, g32, _ = po.wave2d_loss_grad(flat.astype(f32), layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, norm, term_weights=tw, dtype=f32)
    e32 = layer_errs(g32, g64, layers); eb32 = layer_errs(g32, g64, layers, True)
    print(f'{case} layers {layers[1]}x{L - 1} n={m}: host fp32 rel err per W layer ' + ' '.join(f'{e:.1e}' for e in e32))
    for mode in ('full', 's_lo8', 's_hi', 'z_hi', 'hi'):
        for ssc in (1.0, float(m)):
            gv = run(X, Ws, bs, lb, ub, norm, tw, mode, ssc)
            e = layer_errs(gv, g64, layers); eb = layer_errs(gv, g64, layers, True)
            print(f'   {mode:5s} seed x{ssc:6.0f} W x fp32: ' + ' '.join(f'{a / b:5.1f}' for a, b in zip(e, e32)) + ' | b x fp32: ' + ' '.join(f'{a / b:5.1f}' for a, b in zip(eb, eb32)) + f' | worst {max((e / e32).max(), (eb / eb32).max()):.1f}')
