"""Dev experiment (round 5): WHICH layers' parked low parts does the activation reverse need?  Round 4 parks the low part of every state
S_2..S_{NL-1} of the narrow collocation kernel as one byte per value (LO8) and fp16-only states are 8 % faster.  Here the kernel's arithmetic
(tools/studies/wgrad_operand_study.py: forward f16x3, reverse chain with scaled adjoint low parts, weight gradient from high parts) is run
with the low part dropped layer by layer: per weight layer and bias, error against float64 as a multiple of host-fp32's (the GPU tests'
bound: <= 6x per layer, tests/test_gpu_parity.py).   python tools/studies/lo_policy_study.py [case ...]
cache index l = 1..NL-2 <-> kernel state S_{l+1} (S_1 is recomputed in full, S_NL comes from the forward's registers)."""
import sys, itertools, numpy as np
sys.path.insert(0, '/root/repo')
sys.argv, argv = sys.argv[:1] + ['__none__'], sys.argv[1:]
import importlib.util
spec = importlib.util.spec_from_file_location('wos', '/root/repo/tools/studies/wgrad_operand_study.py')
wos = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(wos)          # (its __main__ part runs over sys.argv cases: '__none__' has no files -> stops there)
except Exception:
    pass
from oracle import pinn_oracle as po, golden_points as gp
f32 = np.float32
r16 = wos.r16

POLICY = {}          # cache index -> 'lo8' | 'hi' | 'full'
def parked_l(v, l):
    v = np.asarray(v, f32); mode = POLICY.get(l, 'lo8')
    if mode == 'full': return v
    hi = r16(v)
    if mode == 'hi': return hi
    lo = (v - hi).astype(np.float16)
    return hi + (lo.view(np.uint16) & np.uint16(0xff00)).view(np.float16).astype(f32)

def run(X, Ws, bs, lb, ub, normalize, tw, zscale):
    """wos.run with a per-layer parked() and the shipped weight-gradient operands (mode 'zhi', scale 16)"""
    mode = 'zhi'
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
    g = (2 * f * tw[None, :]).astype(f32)
    Yb, dYb = po.wave2d_residual_adjoint(g)
    L = len(Ws); Wbar = [None] * L; bbar = [None] * L
    h, dh = cache[-1]
    Wbar[-1] = wos.wg(h, Yb, mode, zscale) + sum(wos.wg(dh[k], dYb[k], mode, zscale) for k in range(3)); bbar[-1] = wos.bsum(Yb, mode, zscale)
    hb = wos.mm_bwd(Yb, Ws[-1].T); dhb = [wos.mm_bwd(dYb[k], Ws[-1].T) for k in range(3)]
    for l in range(L - 2, 0, -1):
        h, dh = cache[l]; hin, dhin = cache[l - 1]
        if l < L - 2: h, dh = parked_l(h, l), [parked_l(d, l) for d in dh]
        s = 1 - h * h
        zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
        Wbar[l] = wos.wg(hin, zb, mode, zscale) + sum(wos.wg(dhin[k], dzb[k], mode, zscale) for k in range(3)); bbar[l] = wos.bsum(zb, mode, zscale)
        hb = wos.mm_bwd(zb, Ws[l].T); dhb = [wos.mm_bwd(dzb[k], Ws[l].T) for k in range(3)]
    h, dh = cache[0]; s = 1 - h * h
    zb = s * hb - 2 * h * sum(dhb[k] * dh[k] for k in range(3)); dzb = [s * dhb[k] for k in range(3)]
    Wbar[0] = h0.T @ zb + np.stack([sc[k] * dzb[k].sum(0) for k in range(3)]); bbar[0] = zb.sum(0)
    return po.pack_params(Wbar, bbar, f32)

gd = '/root/repo/tests/golden'
cases = argv or ['inf20s', 'semi16s', 'conf14s', 'wave64']
MS = (4096, 32768)
for case in cases:
    w = np.load(f'{gd}/weights_{case}.npz'); g = np.load(f'{gd}/golden_{case}_32k.npz')
    layers = [int(v) for v in w['layers']]; L = len(layers) - 1
    Ws = [w[f'W{i}'] for i in range(L)]; bs = [w[f'b{i}'] for i in range(L)]
    flat = po.pack_params(Ws, bs)
    lb, ub, norm, n = g['lb'], g['ub'], bool(g['normalize']), int(g['n'])
    X = gp.wave_points(lb, ub, tuple(g['src']), n)
    park = list(range(1, L - 2))                      # cache indices of the parked states
    for m in MS:
        Xm = X[:m]; tw = np.ones(7) / m
        _, g64, _ = po.wave2d_loss_grad(flat, layers, Xm[:, 0], Xm[:, 1], Xm[:, 2], lb, ub, norm, term_weights=tw)
        _, g32, _ = po.wave2d_loss_grad(flat.astype(f32), layers, Xm[:, 0], Xm[:, 1], Xm[:, 2], lb, ub, norm, term_weights=tw, dtype=f32)
        e32 = wos.layer_errs(g32, g64, layers); eb32 = wos.layer_errs(g32, g64, layers, True)
        zscale = 2.0 ** np.round(np.log2(m)) * 16
        def ratios(policy):
            POLICY.clear(); POLICY.update(policy)
            gv = run(Xm, Ws, bs, lb, ub, norm, tw, zscale)
            e = wos.layer_errs(gv, g64, layers) / e32; eb = wos.layer_errs(gv, g64, layers, True) / eb32
            return e, eb
        def show(name, policy):
            e, eb = ratios(policy)
            print(f'   {name:22s} W: ' + ' '.join(f'{a:5.1f}' for a in e) + ' | b: ' + ' '.join(f'{a:5.1f}' for a in eb) + f' | max {max(e.max(), eb.max()):5.1f}', flush=True)
            return max(e.max(), eb.max())
        print(f'{case} n={m} layers={layers[1]}x{L - 1}: multiples of host-fp32 error (bound in the tests: 6)')
        show('all lo8 (shipped)', {})
        show('all hi (fp16 states)', {l: 'hi' for l in park})
        for l in park:
            show(f'drop S_{l + 1} only', {l: 'hi'})
        for l in park:
            show(f'keep S_{l + 1} only', {k: 'hi' for k in park if k != l})
        for k in range(1, len(park)):
            show(f'drop the {k} lowest', {l: 'hi' for l in park[:k]})
            show(f'drop the {k} highest', {l: 'hi' for l in park[-k:]})
