"""Dev experiment (round 2): the fused kernel's operand formats emulated in numpy -- activations as fp16 hi + UNSCALED fp16 lo,
weights as V = WS*w split into T(V) + T(V - T(V)) (and T(V)/2^11 for the reverse pass), adjoints as hi + 2^11-scaled lo -- against the
float64 oracle at the reference's trained weights, for several weight scales WS.  Result: WS = 32 ... 2048 are equally accurate
(fields 3e-6, residuals 2e-4...6e-4, gradient 1e-3...3e-3 = the plain f16x3 numbers); an unscaled low part of the WEIGHTS is not."""
import sys, numpy as np, io, contextlib
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
from oracle import pinn_oracle as po
with contextlib.redirect_stdout(io.StringIO()):
    import precision_study as ps
r_f16 = ps.r_f16; rel = ps.rel
WS = 2048.0
def mm_fwd(A, B):   # activations A: hi + unscaled lo; weights B: V = WS*w, V_hi, V_lo (unscaled remainder)
    A = np.asarray(A, np.float32); B = np.asarray(B, np.float32) * np.float32(WS)
    Ah, Bh = r_f16(A), r_f16(B)
    Al = r_f16(A - Ah); Bl = r_f16(B - Bh)
    return (Ah @ Bh + Al @ Bh + Ah @ Bl) / np.float32(WS)
def mm_bwd(Z, Wt):  # adjoints Z: hi + scaled lo' (2^11); weights: V_hi, V_lo, w2 = f16(V/2048)
    Z = np.asarray(Z, np.float32); V = np.asarray(Wt, np.float32) * np.float32(WS)
    Zh = r_f16(Z); Zl = r_f16((Z - Zh) * 2048.0)
    Vh = r_f16(V); Vl = r_f16(V - Vh); W2 = r_f16(Vh / 2048.0)
    return (Zh @ Vh + Zl @ W2 + Zh @ Vl) / np.float32(WS)
orig_mm = ps.mm
def mm(A, B, mode):
    if mode == 'F': return mm_fwd(A, B)
    if mode == 'R': return mm_bwd(A, B)
    return orig_mm(A, B, mode)
ps.mm = mm
for ws in [2048.0, 64.0, 32.0, 8.0]:
    WS = ws
    for case in ['inf20s', 'conf14s']:
        w = np.load(f'tests/golden/weights_{case}.npz'); g = np.load(f'tests/golden/golden_{case}.npz')
        layers = list(w['layers']); L = len(layers) - 1
        Ws = [w[f'W{i}'] for i in range(L)]; bs = [w[f'b{i}'] for i in range(L)]
        X = g['X']; N = X.shape[0]; tw = np.ones(7) / N
        Y, dY, f, gr = ps.fwd(X, Ws, bs, g['lb'], g['ub'], bool(g['normalize']), 'F', 'R', tw)
        print(f'WS {ws:6.0f} {case} fields {rel(Y, g["Y"]):.2e} jac {rel(np.stack(dY), g["dY"]):.2e} resid {rel(f, g["f"]):.2e} grad {rel(gr, g["grad"].astype(np.float64)):.2e}  max|W| {max(np.abs(W).max() for W in Ws):.2f}', flush=True)
