"""CPU tests: the UNMODIFIED device + host sources of libpinn_hip.so, compiled for x86 against the
host SIMT emulator (tools/emu), checked against the float64 oracle.  This validates the MFMA
fragment index math, the spill-panel layouts, the chunked-workspace walk and the launch geometry
without a GPU.  (The emulator is test tooling; it is not a fallback of the product.)"""
import os
import subprocess

import numpy as np
import pytest

from oracle import pinn_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LB, UB = [0.0, 0.0, 0.0], [30.0, 30.0, 20.0]


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-C", os.path.join(ROOT, "pinn_elastodynamics_amd", "csrc"), "-j", str(min(16, os.cpu_count() or 1)), "emu"],
                   check=True, capture_output=True)
    from pinn_elastodynamics_amd.capi import PinnLib
    return PinnLib(os.path.join(ROOT, "build", "emu", "libpinn_emu.so"))


def aligned(nbytes):
    raw = np.zeros(nbytes + 256, dtype=np.uint8)
    off = (-raw.ctypes.data) % 256
    return raw[off:off + nbytes]


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))


def run_wave(emu, layers, n, prec, normalize=True, min_ws=False, seed=0, fused=False):
    emu.set_fused(fused)
    rng = np.random.default_rng(seed)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.3 * rng.standard_normal(b.shape) for b in bs]
    X = po.collocation_points(n, LB, UB, rng)
    flat = po.pack_params(Ws, bs)
    tw = np.array([1, 2, 3, 1, 0.5, 1, 2.0]) / n
    ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, normalize, term_weights=tw)
    p32 = flat.astype(np.float32)
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    wsb = emu.min_workspace_bytes(layers, prec) if min_ws else emu.workspace_bytes(layers, n, prec)
    ws = aligned(wsb)
    loss = np.full(8, np.nan, np.float32)
    grad = np.full(p32.size, np.nan, np.float32)
    emu.wave2d_loss_grad(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, normalize, 2.5, 0.25, 1.0, True,
                         tw, loss.ctypes.data, grad.ctypes.data, False, prec, ws.ctypes.data, wsb)
    return rel(loss[:7], ss), rel(grad, g)


@pytest.mark.parametrize("layers,n,prec,tol", [
    ([3] + 4 * [32] + [7], 100, "f16x3", 2e-6),        # BASELINE configs[0] net
    ([3] + 4 * [32] + [7], 100, "bf16", 2e-2),
    ([3] + 8 * [64] + [7], 70, "f16x3", 2e-6),         # BASELINE configs[1] net
    ([3] + 2 * [80] + [7], 33, "f16x3", 2e-6),         # reference INF width (padded to 96, 16-point tiles)
    ([3] + 2 * [140] + [7], 20, "f16x3", 2e-6),        # reference CONF width (padded to 160)
])
def test_wave_loss_grad_emulated(emu, layers, n, prec, tol):
    """two-kernel path (chain_kernel + wgrad_kernel): everything split, fp32-class agreement"""
    e_loss, e_grad = run_wave(emu, layers, n, prec, fused=False)
    assert e_loss < tol and e_grad < tol


@pytest.mark.parametrize("layers,n,prec,tol_loss,tol_grad", [
    ([3] + 4 * [32] + [7], 130, "f16x3", 2e-6, 1e-4),   # fused kernel parks the state in fp16 for the reverse pass:
    ([3] + 8 * [64] + [7], 200, "f16x3", 2e-6, 1e-4),   # loss exact, gradient ~5e-4/sqrt(n) (tools/precision_study2.py)
    ([3] + 8 * [64] + [7], 100, "bf16", 2e-2, 2e-2),
    ([3] + 4 * [64] + [7], 70, "bf16x3", 1e-4, 1e-3),
    ([3] + 4 * [64] + [7], 200, "f16x3", 2e-6, 1e-4),   # four layers of padded width 64: S_1 from the weight-gradient wave in TWO pieces (S1_BY_WG, NL < 6)
])
def test_fused_kernel_emulated(emu, layers, n, prec, tol_loss, tol_grad):
    """fused persistent kernel (role-specialised waves, LDS transpose hand-off, LDS-DMA state reload)"""
    e_loss, e_grad = run_wave(emu, layers, n, prec, fused=True)
    assert e_loss < tol_loss and e_grad < tol_grad


def test_fused_persistent_accumulation_emulated(emu):
    """minimum workspace => fewer workgroups than steps: accumulators persist across steps of a workgroup"""
    e_loss, e_grad = run_wave(emu, [3] + 4 * [32] + [7], 9000, "f16x3", min_ws=True, fused=True)
    assert e_loss < 2e-6 and e_grad < 2e-5


def test_fused_narrow_several_steps_per_workgroup_emulated(emu):
    """The 8 x 64 and 4 x 64 collocation kernels with the minimum workspace: several steps per workgroup, so that what the weight-gradient
    wave carries from step to step (its tile's inputs for the recomputation of S_1, the in-register sums) is exercised."""
    for layers, n in (([3] + 8 * [64] + [7], 520), ([3] + 4 * [64] + [7], 390)):
        e_loss, e_grad = run_wave(emu, layers, n, "f16x3", min_ws=True, fused=True, seed=3)
        assert e_loss < 2e-6 and e_grad < 6e-5, (layers, e_loss, e_grad)


def test_fp16_state_flag_emulated(emu):
    """PINN_FLAG_STATE_FP16: the 8-layer collocation kernel with fp16-only parked states (opt-in; the default parks the low parts as well).
    At fresh weights both agree with the oracle; the default is the closer one."""
    layers = [3] + 8 * [64] + [7]
    e_loss, e_fast = run_wave(emu, layers, 70, "f16x3+fp16state", fused=True)
    e_loss2, e_acc = run_wave(emu, layers, 70, "f16x3", fused=True)
    assert e_loss < 2e-6 and e_loss2 < 2e-6 and e_acc < e_fast < 1e-4


@pytest.mark.parametrize("width", [80, 100])
def test_fused_wide_emulated(emu, width):
    """Padded widths 96 / 128 (the reference's 8 x 80 and 8 x 100 nets, INF:645, SEMI:679) through the LDS-operand layout of the fused
    kernel: the tile's state lives in its LDS image and is read one k-step at a time, two waves share a tile's chain (three / four
    feature blocks each), six / eight blocks per side in the weight gradient, two tiles per workgroup (width 128: one state slot).
    Against the oracle (both state parts are kept: the two-kernel path's accuracy) and the two-kernel path."""
    layers = [3] + 8 * [width] + [7]
    e_loss, e_grad = run_wave(emu, layers, 70, "f16x3", fused=True)
    assert e_loss < 2e-6 and e_grad < 3e-6
    e_loss, e_grad = run_wave(emu, layers, 70, "f16x3", fused=False)
    assert e_loss < 2e-6 and e_grad < 2e-6
    e_loss, e_grad = run_wave(emu, layers, 300 if width == 80 else 130, "f16x3", min_ws=True, fused=True, normalize=False, seed=4)      # several steps per workgroup
    assert e_loss < 2e-6 and e_grad < 5e-6


def test_chunked_workspace_emulated(emu):
    """2100 points with the minimum workspace (64 tiles of 32 points) -> two passes of the two-kernel path."""
    e_loss, e_grad = run_wave(emu, [3] + 2 * [32] + [7], 2100, "f16x3", min_ws=True, fused=False)
    assert e_loss < 2e-6 and e_grad < 2e-6


def test_data_term_and_fields_emulated(emu):
    layers = [3] + 3 * [32] + [7]
    rng = np.random.default_rng(5)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.3 * rng.standard_normal(b.shape) for b in bs]
    n = 75
    X = -15 + 30 * rng.random((n, 3))
    flat = po.pack_params(Ws, bs)
    tgt = rng.standard_normal((n, 7))
    ow = np.array([1, 1, 0, 0, 0, 2, 0.5]) / n
    ss, g, _ = po.data_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, False, tgt, ow)
    p32 = flat.astype(np.float32)
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    tg = np.ascontiguousarray(tgt.T.astype(np.float32))
    wsb = emu.workspace_bytes(layers, n, "f16x3")
    ws = aligned(wsb)
    loss = np.zeros(8, np.float32)
    grad = np.zeros(p32.size, np.float32)
    emu.data_loss_grad(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, False, tg.ctypes.data, ow,
                       loss.ctypes.data, grad.ctypes.data, False, "f16x3", ws.ctypes.data, wsb)
    assert rel(loss[:7], ss) < 2e-6 and rel(grad, g) < 2e-6
    out = po.wave2d_fields(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, False)
    fo = np.zeros((28, n), np.float32)
    emu.wave2d_fields(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, False, fo.ctypes.data, "f16x3",
                      ws.ctypes.data, wsb)
    ref = np.concatenate([out["Y"].T] + [d.T for d in out["dY"]])
    assert rel(fo, ref) < 2e-6


def test_fp32_mode_emulated(emu):
    """PINN_PREC_FP32: the same entry points in plain fp32 arithmetic (no matrix pipe, no 16-bit operand) -- the on-device third leg
    of the parity tests.  Loss sums, gradient (several workspace passes, accumulate flag), data term and fields against the oracle."""
    layers = [3] + 3 * [20] + [7]
    e_loss, e_grad = run_wave(emu, layers, 150, "fp32")
    assert e_loss < 5e-6 and e_grad < 5e-6
    e_loss, e_grad = run_wave(emu, layers, 700, "fp32", min_ws=True, normalize=False)      # 256-point passes
    assert e_loss < 5e-6 and e_grad < 5e-6
    rng = np.random.default_rng(5)
    Ws, bs = po.xavier_init(layers, rng)
    flat = po.pack_params(Ws, [0.3 * rng.standard_normal(b.shape) for b in bs])
    n = 75
    X = -15 + 30 * rng.random((n, 3))
    tgt = rng.standard_normal((n, 7))
    ow = np.array([1, 1, 0, 0, 0, 2, 0.5]) / n
    ss, g, _ = po.data_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, False, tgt, ow)
    p32 = flat.astype(np.float32)
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    tg = np.ascontiguousarray(tgt.T.astype(np.float32))
    wsb = emu.workspace_bytes(layers, n, "fp32")
    ws = aligned(wsb)
    loss = np.zeros(8, np.float32)
    grad = np.ones(p32.size, np.float32)
    emu.data_loss_grad(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, False, tg.ctypes.data, ow,
                       loss.ctypes.data, grad.ctypes.data, True, "fp32", ws.ctypes.data, wsb)          # accumulate onto ones
    assert rel(loss[:7], ss) < 5e-6 and rel(grad - 1.0, g) < 5e-6
    out = po.wave2d_fields(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, False)
    fo = np.zeros((28, n), np.float32)
    emu.wave2d_fields(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, False, fo.ctypes.data, "fp32",
                      ws.ctypes.data, wsb)
    assert rel(fo, np.concatenate([out["Y"].T] + [d.T for d in out["dY"]])) < 5e-6


def test_adam_emulated(emu):
    rng = np.random.default_rng(6)
    P = 1000
    th, m, v = rng.standard_normal(P), np.zeros(P), np.zeros(P)
    a, b, c = th.astype(np.float32), m.astype(np.float32), v.astype(np.float32)
    for step in range(1, 4):
        g = rng.standard_normal(P)
        th, m, v = po.adam_tf1_step(th, g, m, v, step, 1e-3)
        g32 = g.astype(np.float32)
        emu.adam_step(a.ctypes.data, b.ctypes.data, c.ctypes.data, g32.ctypes.data, P, 1e-3, step)
    assert rel(a, th) < 1e-6


# ---- plate family (5-stream chain kernels) -------------------------------------------------------------------------
@pytest.mark.parametrize("lN,lD,n", [([3, 20, 20, 20, 5], [3, 10, 10, 5], 50),          # reference dist/part widths (PLATE:700-702)
                                     ([3] + 3 * [70] + [5], [3] + 4 * [20] + [5], 24)])  # reference uv width (padded to 96)
def test_emulated_plate_entry_points(emu, lN, lD, n):
    from oracle import plate_oracle as pl
    prec, LBp, UBp = "f16x3", [0, 0, 0], [0.5, 0.5, 10]
    rng = np.random.default_rng(0)

    def mk(l):
        W, b = po.xavier_init(l, rng)
        return po.pack_params(W, [0.2 * rng.standard_normal(x.shape) for x in b])

    fN, fD, fP = mk(lN), mk(lD), mk(lD)
    C = np.stack([rng.random(n) * 0.5, rng.random(n) * 0.5, rng.random(n) * 10], 1)
    x, y, t = (C[:, k].astype(np.float32).copy() for k in range(3))
    wsb = max(emu.workspace_bytes(lN, n, prec), emu.workspace_bytes(lD, n, prec))
    ws = aligned(wsb)
    Dref, Pref = pl.net_streams(fD, lD, C[:, 0], C[:, 1], C[:, 2]), pl.net_streams(fP, lD, C[:, 0], C[:, 1], C[:, 2])
    out = np.full((5, 5, n), np.nan, np.float32)
    pD = fD.astype(np.float32)
    emu.net_streams(pD.ctypes.data, lD, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, out.ctypes.data, prec, ws.ctypes.data, wsb)
    for s in range(5):
        assert rel(out[s], Dref[s]) < 2e-6, s
    tw = np.array([10, 7, 13, 9, 11.0]) / n
    ss, g, _ = pl.plate_loss_grad(fN, lN, C[:, 0], C[:, 1], C[:, 2], Dref, Pref, term_weights=tw)
    frozen = np.ascontiguousarray(np.stack([Dref, Pref]).astype(np.float32))
    pN = fN.astype(np.float32)
    loss, grad = np.full(8, np.nan, np.float32), np.full(pN.size, np.nan, np.float32)
    emu.plate2d_loss_grad(pN.ctypes.data, lN, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, frozen.ctypes.data, 20.0, 0.25, 1.0,
                          tw, loss.ctypes.data, grad.ctypes.data, False, prec, ws.ctypes.data, wsb)
    assert rel(loss[:5], ss) < 2e-6 and rel(grad, g) < 2e-6
    th = rng.random(n) * np.pi / 2
    H = np.stack([0.1 * np.cos(th), 0.1 * np.sin(th), rng.random(n) * 10], 1)
    hx, hy, ht = (H[:, k].astype(np.float32).copy() for k in range(3))
    DH, PH = pl.net_streams(fD, lD, H[:, 0], H[:, 1], H[:, 2])[0], pl.net_streams(fP, lD, H[:, 0], H[:, 1], H[:, 2])[0]
    ssh, gh = pl.traction_loss_grad(fN, lN, H[:, 0], H[:, 1], H[:, 2], DH, PH, weight=10.0 / n)
    aux = np.ascontiguousarray(np.concatenate([DH, PH, (-H[:, 0] / 0.1)[None], (-H[:, 1] / 0.1)[None]]).astype(np.float32))
    emu.plate2d_traction_loss_grad(pN.ctypes.data, lN, hx.ctypes.data, hy.ctypes.data, ht.ctypes.data, n, LBp, UBp, False, aux.ctypes.data,
                                   [10.0 / n] * 2, loss.ctypes.data, grad.ctypes.data, False, prec, ws.ctypes.data, wsb)
    assert rel(loss[:2], ssh) < 2e-6 and rel(grad, gh) < 2e-6
    tg = rng.standard_normal((5, 5, n))
    w = np.zeros((5, 5))
    w[0, :] = 1000.0 / n
    w[3, 0] = w[3, 1] = 500.0 / n
    s3, g3 = pl.stream_loss_grad(fD, lD, C[:, 0], C[:, 1], C[:, 2], tg, w)
    tg32 = np.ascontiguousarray(tg.astype(np.float32))
    gradD = np.full(pD.size, np.nan, np.float32)
    emu.stream_loss_grad(pD.ctypes.data, lD, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, tg32.ctypes.data, w, loss.ctypes.data,
                         gradD.ctypes.data, False, prec, ws.ctypes.data, wsb)
    assert rel(loss[:5], ((w / w.max()) * s3).sum(0)) < 2e-6 and rel(gradD, g3) < 2e-6


@pytest.mark.parametrize("lN,n", [([3] + 4 * [24] + [5], 70),       # five streams, all layer states in LDS
                                  ([3] + 8 * [30] + [5], 130),      # parked states + LDS-DMA, several workgroup steps
                                  ([3] + 4 * [40] + [5], 40),       # padded width 64: constants from memory (LDS is full)
                                  ([3] + 8 * [48] + [5], 140),      # BASELINE configs[2]'s layout (8 layers of padded width 64): double-buffered adjoints, S_1's
                                                                    # high parts from the weight-gradient wave, several workgroup steps
                                  ([3] + 8 * [70] + [5], 75)])      # the reference's plate net (PLATE:885): padded width 96, LDS-operand layout
def test_emulated_plate_fused(emu, lN, n):
    """The plate's loss + gradient through the five-stream instantiation of the fused kernel (second time derivative carried as a
    fifth stream, composite head PLATE:358-439) against the float64 oracle, and the two-kernel path for the same call."""
    from oracle import plate_oracle as pl
    prec, LBp, UBp = "f16x3", [0, 0, 0], [0.5, 0.5, 10]
    rng = np.random.default_rng(5)
    lD = [3, 10, 10, 5]

    def mk(l):
        W, b = po.xavier_init(l, rng)
        return po.pack_params(W, [0.2 * rng.standard_normal(x.shape) for x in b])

    fN, fD, fP = mk(lN), mk(lD), mk(lD)
    C = np.stack([rng.random(n) * 0.5, rng.random(n) * 0.5, rng.random(n) * 10], 1)
    x, y, t = (C[:, k].astype(np.float32).copy() for k in range(3))
    wsb = emu.workspace_bytes(lN, n, prec)
    ws = aligned(wsb)
    Dref, Pref = pl.net_streams(fD, lD, C[:, 0], C[:, 1], C[:, 2]), pl.net_streams(fP, lD, C[:, 0], C[:, 1], C[:, 2])
    tw = np.array([10, 7, 13, 9, 11.0]) / n
    ss, g, _ = pl.plate_loss_grad(fN, lN, C[:, 0], C[:, 1], C[:, 2], Dref, Pref, term_weights=tw)
    frozen = np.ascontiguousarray(np.stack([Dref, Pref]).astype(np.float32))
    pN = fN.astype(np.float32)
    res = {}
    for fused in (True, False):
        emu.set_fused(fused)
        loss, grad = np.full(8, np.nan, np.float32), np.full(pN.size, np.nan, np.float32)
        emu.plate2d_loss_grad(pN.ctypes.data, lN, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, frozen.ctypes.data, 20.0, 0.25, 1.0,
                              tw, loss.ctypes.data, grad.ctypes.data, False, prec, ws.ctypes.data, wsb)
        res[fused] = (loss[:5].copy(), grad.copy())
        # the fused kernel parks the layer states as fp16 high parts (DESIGN_HISTORY.md section 6): a 2^-12 rounding noise per state element
        # that averages out as 1/sqrt(points) in the gradient -- 3e-5 at ~100 points, 5e-6 at 4096 (GPU tests use real sizes)
        # (the width-96 layout keeps both state parts in LDS: the two-kernel path's accuracy)
        assert rel(loss[:5], ss) < 3e-6 and rel(grad, g) < (1e-4 if fused and lN[1] <= 64 else 2e-6), fused
    emu.set_fused(True)
    assert rel(res[True][1], res[False][1].astype(np.float64)) < 1e-4


def test_empty_and_ragged_batches_emulated(emu):
    """n = 0 is a valid empty batch (zero sums; gradient zeroed, or left alone when accumulating); sizes around the 16-point tile
    and the 64-point workgroup step go through both paths."""
    layers = [3] + 4 * [32] + [7]
    rng = np.random.default_rng(0)
    Ws, bs = po.xavier_init(layers, rng)
    flat = po.pack_params(Ws, bs)
    p32 = flat.astype(np.float32)
    wsb = emu.workspace_bytes(layers, 256, "f16x3")
    ws = aligned(wsb)
    for accumulate in (False, True):
        loss = np.full(8, np.nan, np.float32)
        grad = np.full(p32.size, 3.0, np.float32)
        emu.wave2d_loss_grad(p32.ctypes.data, layers, 0, 0, 0, 0, LB, UB, True, 2.5, 0.25, 1.0, True, np.ones(7), loss.ctypes.data, grad.ctypes.data,
                             accumulate, "f16x3", ws.ctypes.data, wsb)
        assert np.all(loss[:7] == 0) and np.all(grad == (3.0 if accumulate else 0.0))
    for fused, tol in ((1, 2e-4), (0, 2e-6)):
        emu.set_fused(fused)
        for n in (1, 15, 17, 63, 65):
            X = po.collocation_points(n, LB, UB, rng)
            x, y, t = (np.ascontiguousarray(X[:, k], dtype=np.float32) for k in range(3))
            tw = np.ones(7) / n
            ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True, term_weights=tw)
            loss = np.full(8, np.nan, np.float32)
            grad = np.full(p32.size, np.nan, np.float32)
            emu.wave2d_loss_grad(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, True, 2.5, 0.25, 1.0, True, tw,
                                 loss.ctypes.data, grad.ctypes.data, False, "f16x3", ws.ctypes.data, wsb)
            assert rel(loss[:7], ss) < 2e-6 and rel(grad, g) < tol, (fused, n)
    emu.set_fused(1)


@pytest.mark.parametrize("n", [1, 15, 17, 33, 47])
def test_ragged_batches_wide_layout_emulated(emu, n):
    """Sizes around the 16-point tile and the 32-point workgroup step of the LDS-operand layout (padded width 96, two waves per tile):
    masked points of a partial tile contribute nothing, a missing second tile neither."""
    e_loss, e_grad = run_wave(emu, [3] + 8 * [80] + [7], n, "f16x3", fused=True, seed=n)
    assert e_loss < 2e-6 and e_grad < 3e-6, n


@pytest.mark.parametrize("layers,n,fused", [([3] + 4 * [32] + [7], 150, 1), ([3] + 8 * [64] + [7], 90, 1), ([3] + 8 * [64] + [7], 90, 0),
                                            ([3] + 8 * [80] + [7], 100, 1), ([3] + 8 * [100] + [7], 75, 1), ([3] + 6 * [140] + [7], 75, 1)])
def test_data_terms_fused_and_two_kernel_emulated(emu, layers, n, fused):
    """value-only side sets (loss_IC / loss_SRC / ...) through the fused kernel's 1-stream instantiation and the two-kernel path; the
    reference's 8 x 80 / 8 x 100 nets through the one-stream LDS-operand layout with all layer states in LDS (round 3); round 6: the
    confined-domain net's 6 x 140 (CONF:891; its IC / FIX / SRC sets are ~90 k of ~240 k points per step, CONF:901-947) likewise --
    padded width 160, seven 10 KB state slots per tile: the 160 KB exactly, constants from memory"""
    emu.path_counts(reset=True)
    emu.set_fused(fused)
    rng = np.random.default_rng(8)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.3 * rng.standard_normal(b.shape) for b in bs]
    X = -15 + 30 * rng.random((n, 3))
    flat = po.pack_params(Ws, bs)
    p32 = flat.astype(np.float32)
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    wsb = emu.workspace_bytes(layers, n, "f16x3")
    ws = aligned(wsb)
    for tgt, ow in ((rng.standard_normal((n, 7)), np.array([1, 1, 0, 0, 0, 2, 0.5]) / n), (None, np.array([1, 1, 1, 1, 0, 0, 0.0]) / n)):
        ss, g, _ = po.data_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, False, tgt, ow)
        tg = None if tgt is None else np.ascontiguousarray(tgt.T.astype(np.float32))
        loss = np.full(8, np.nan, np.float32)
        grad = np.full(p32.size, np.nan, np.float32)
        emu.data_loss_grad(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, False, 0 if tg is None else tg.ctypes.data, ow,
                           loss.ctypes.data, grad.ctypes.data, False, "f16x3", ws.ctypes.data, wsb)
        assert rel(loss[:7], ss) < 2e-6 and rel(grad, g) < ((2e-4 if layers[1] <= 64 else 3e-6) if fused else 2e-6)
    pc = emu.path_counts(reset=True)
    assert pc["two-kernel" if not fused else ("fused-registers" if layers[1] <= 64 else "fused-lds")] == 2 and sum(pc.values()) == 2, pc
    emu.set_fused(1)


def test_packed_weights_flag_emulated(emu):
    """PINN_FLAG_WEIGHTS_PACKED: a second call on the same workspace with the same parameters may skip the repack and must give
    the same numbers; with different parameters (flag misuse) it would reuse the OLD weights -- shown here as the documented hazard."""
    layers = [3] + 4 * [32] + [7]
    rng = np.random.default_rng(2)
    Ws, bs = po.xavier_init(layers, rng)
    flat = po.pack_params(Ws, bs)
    p32 = flat.astype(np.float32)
    n = 80
    X = po.collocation_points(n, LB, UB, rng)
    x, y, t = (np.ascontiguousarray(X[:, k], dtype=np.float32) for k in range(3))
    wsb = emu.workspace_bytes(layers, n, "f16x3")
    ws = aligned(wsb)
    ow = np.array([1, 1, 1, 1, 0, 0, 0.0]) / n

    def data(params, mode):
        loss, grad = np.full(8, np.nan, np.float32), np.full(p32.size, np.nan, np.float32)
        emu.data_loss_grad(params.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, True, 0, ow, loss.ctypes.data,
                           grad.ctypes.data, False, mode, ws.ctypes.data, wsb)
        return loss[:7].copy(), grad

    l0, g0 = data(p32, "f16x3")
    l1, g1 = data(p32, "f16x3+packed")
    np.testing.assert_array_equal(l0, l1)
    np.testing.assert_array_equal(g0, g1)
    other = (p32 * 1.5).astype(np.float32)
    l2, _ = data(other, "f16x3+packed")           # misuse: still the first parameters' fragments
    np.testing.assert_array_equal(l0, l2)
    l3, _ = data(other, "f16x3")
    assert not np.allclose(l0, l3)


@pytest.mark.parametrize("layers,fused", [([3] + 4 * [32] + [7], 1), ([3] + 2 * [48] + [7], 1), ([3] + 8 * [80] + [7], 1)])
def test_data_loss_grad_multi_emulated(emu, layers, fused):
    """pinn_data_loss_grad_multi: three value-only sets (one of them empty, one with targets) in one call equal three single calls;
    4x32 takes the fused kernel's set table (one launch), 2x48 the two-kernel loop."""
    emu.set_fused(fused)
    rng = np.random.default_rng(9)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.3 * rng.standard_normal(b.shape) for b in bs]
    flat = po.pack_params(Ws, bs)
    p32 = flat.astype(np.float32)
    sizes = (70, 0, 130)
    sets, refs, keep = [], [], []
    g_ref = np.zeros_like(flat)
    for k, n in enumerate(sizes):
        X = -15 + 30 * rng.random((max(n, 1), 3))[:n]
        tgt = rng.standard_normal((n, 7)) if k == 2 else None
        ow = (np.array([1, 1, 0, 0, 0, 2, 0.5]) if k == 2 else np.array([1, 1, 1, 1, 0, 0, 0.0])) / max(n, 1)
        x, y, t = (np.ascontiguousarray(X[:, j], dtype=np.float32) for j in range(3))
        tg = None if tgt is None else np.ascontiguousarray(tgt.T.astype(np.float32))
        lo = np.full(8, np.nan, np.float32)
        keep.append((x, y, t, tg, lo))
        sets.append((x.ctypes.data if n else 0, y.ctypes.data if n else 0, t.ctypes.data if n else 0, n, 0 if tg is None else tg.ctypes.data, ow, lo.ctypes.data))
        if n:
            ss, g, _ = po.data_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, False, tgt, ow)
            g_ref += g
            refs.append(ss)
        else:
            refs.append(np.zeros(7))
    wsb = emu.workspace_bytes(layers, max(sizes), "f16x3")
    ws = aligned(wsb)
    grad = np.full(p32.size, np.nan, np.float32)
    emu.data_loss_grad_multi(p32.ctypes.data, layers, sets, LB, UB, False, grad.ctypes.data, False, "f16x3", ws.ctypes.data, wsb)
    for k in range(3):
        if sizes[k]:
            assert rel(keep[k][4][:7], refs[k]) < 2e-6, k
        else:
            assert np.all(keep[k][4][:7] == 0)
    assert rel(grad, g_ref) < 2e-4
    emu.set_fused(1)


# ---- 3-D Navier-Cauchy extension (4 inputs, 12 outputs, value + 4 first-order streams; oracle/nc3d_oracle.py) ----
@pytest.mark.parametrize("layers,n,normalize", [([4] + 2 * [32] + [12], 45, True), ([4] + 3 * [48] + [12], 21, False), ([4] + 2 * [128] + [12], 17, True)])
def test_nc3d_entry_points_emulated(emu, layers, n, normalize):
    """pinn_nc3d_loss_grad / pinn_nc3d_fields / pinn_nc3d_data_loss_grad on the emulator vs the float64 oracle"""
    from oracle import nc3d_oracle as n3
    lb, ub = [0.0, 0.0, -20.0, 0.0], [30.0, 30.0, 0.0, 15.0]
    rng = np.random.default_rng(21)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.3 * rng.standard_normal(b.shape) for b in bs]
    X = n3.halfspace_points(n, lb, ub, rng) if normalize else rng.uniform(-1.0, 1.0, (n, 4))
    flat = po.pack_params(Ws, bs)
    tw = (0.5 + rng.random(12)) / n
    ss, g, _ = n3.nc3d_loss_grad(flat, layers, *X.T, lb, ub, normalize, term_weights=tw)
    p32 = flat.astype(np.float32)
    cols = [X[:, k].astype(np.float32).copy() for k in range(4)]
    ptr = [v.ctypes.data for v in cols]
    wsb = emu.workspace_bytes(layers, n, "f16x3")
    assert wsb > 0
    ws = aligned(wsb)
    loss = np.full(16, np.nan, np.float32)
    grad = np.full(p32.size, np.nan, np.float32)
    emu.nc3d_loss_grad(p32.ctypes.data, layers, *ptr, n, lb, ub, normalize, 2.5, 0.25, 1.0, tw, loss.ctypes.data, grad.ctypes.data, False,
                       "f16x3", ws.ctypes.data, wsb)
    assert rel(loss[:12], ss) < 2e-6 and rel(grad, g) < 2e-6, (rel(loss[:12], ss), rel(grad, g))
    # forward streams
    ref = n3.nc3d_fields(flat, layers, *X.T, lb, ub, normalize)
    out = np.full((5, 12, n), np.nan, np.float32)
    emu.nc3d_fields(p32.ctypes.data, layers, *ptr, n, lb, ub, normalize, out.ctypes.data, "f16x3", ws.ctypes.data, wsb)
    assert rel(out[0].T, ref["Y"]) < 2e-6
    for k in range(4):
        assert rel(out[1 + k].T, ref["dY"][k]) < 2e-6
    # value-only side term (e.g. the traction-free surface: s33, s13, s23 -> 0; initial state u, v, w, ut, vt, wt -> targets)
    tgt = rng.standard_normal((n, 12))
    ow = np.array([1, 1, 1, 0.5, 0.5, 0.5, 0, 0, 2, 0, 2, 2.0]) / n
    ss_d, g_d, _ = n3.nc3d_data_loss_grad(flat, layers, *X.T, lb, ub, normalize, tgt, ow)
    tg = np.ascontiguousarray(tgt.T.astype(np.float32))
    emu.nc3d_data_loss_grad(p32.ctypes.data, layers, *ptr, n, lb, ub, normalize, tg.ctypes.data, ow, loss.ctypes.data, grad.ctypes.data, False,
                            "f16x3", ws.ctypes.data, wsb)
    assert rel(loss[:12], ss_d) < 2e-6 and rel(grad, g_d) < 2e-6
    # accumulate + empty batch
    emu.nc3d_loss_grad(p32.ctypes.data, layers, *ptr, n, lb, ub, normalize, 2.5, 0.25, 1.0, tw, loss.ctypes.data, grad.ctypes.data, True,
                       "f16x3", ws.ctypes.data, wsb)
    assert rel(grad, g + g_d) < 2e-6
    keep = grad.copy()
    emu.nc3d_loss_grad(p32.ctypes.data, layers, 0, 0, 0, 0, 0, lb, ub, normalize, 2.5, 0.25, 1.0, tw, loss.ctypes.data, grad.ctypes.data, True,
                       "f16x3", ws.ctypes.data, wsb)
    assert np.array_equal(grad, keep) and np.all(loss[:12] == 0)


def test_nc3d_argument_checks_emulated(emu):
    layers = [4, 32, 32, 12]
    assert emu.workspace_bytes([4, 32, 32, 17], 10, "f16x3") == 0          # at most 16 outputs
    assert emu.workspace_bytes([5, 32, 32, 12], 10, "f16x3") == 0
    assert emu.workspace_bytes(layers, 10, "f16x3+packed") == emu.workspace_bytes(layers, 10, "f16x3")     # flag bits are ignored when sizing
    p = np.zeros(po.param_count(layers), np.float32)
    ws = aligned(emu.workspace_bytes(layers, 16, "f16x3"))
    z = np.zeros(16, np.float32)
    from pinn_elastodynamics_amd.capi import PinnLibError
    with pytest.raises(PinnLibError):        # the 3-input entry refuses a 4-input net
        emu.wave2d_loss_grad(p.ctypes.data, layers, z.ctypes.data, z.ctypes.data, z.ctypes.data, 16, LB, UB, True, 2.5, 0.25, 1.0, True,
                             np.ones(7), z.ctypes.data, p.ctypes.data, False, "f16x3", ws.ctypes.data, ws.size)
    with pytest.raises(PinnLibError):        # unsplit modes have no 5-stream kernels
        emu.nc3d_loss_grad(p.ctypes.data, layers, *[z.ctypes.data] * 4, 16, [0] * 4, [1] * 4, True, 2.5, 0.25, 1.0, np.ones(12), z.ctypes.data,
                           p.ctypes.data, False, "bf16", ws.ctypes.data, ws.size)


def test_fp32_mode_plate_and_nc3d_emulated(emu):
    """PINN_PREC_FP32 for the plate family (five streams with the second time derivative, composite head, hole traction, stream-wise
    data head, pinn_net_streams) and for the 4-input heads: every entry point against the float64 oracles, workspace passes of 256
    points included."""
    from oracle import nc3d_oracle as n3
    from oracle import plate_oracle as pl
    LBp, UBp = [0, 0, 0], [0.5, 0.5, 10]
    rng = np.random.default_rng(3)

    def mk(l):
        W, b = po.xavier_init(l, rng)
        return po.pack_params(W, [0.2 * rng.standard_normal(x.shape) for x in b])

    lN, lD, n = [3] + 3 * [24] + [5], [3, 20, 20, 5], 300
    fN, fD, fP = mk(lN), mk(lD), mk(lD)
    C = np.stack([rng.random(n) * 0.5, rng.random(n) * 0.5, rng.random(n) * 10], 1)
    x, y, t = (C[:, k].astype(np.float32).copy() for k in range(3))
    wsb = emu.min_workspace_bytes(lN, "fp32")            # 256 points per pass: two passes
    ws = aligned(wsb)
    Dref, Pref = pl.net_streams(fD, lD, C[:, 0], C[:, 1], C[:, 2]), pl.net_streams(fP, lD, C[:, 0], C[:, 1], C[:, 2])
    Nref = pl.net_streams(fN, lN, C[:, 0], C[:, 1], C[:, 2])
    pN = fN.astype(np.float32)
    out = np.full((5, 5, n), np.nan, np.float32)
    emu.net_streams(pN.ctypes.data, lN, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, out.ctypes.data, "fp32", ws.ctypes.data, wsb)
    for s in range(5):
        assert rel(out[s], Nref[s]) < 5e-6, s
    tw = np.array([10, 7, 13, 9, 11.0]) / n
    ss, g, _ = pl.plate_loss_grad(fN, lN, C[:, 0], C[:, 1], C[:, 2], Dref, Pref, term_weights=tw)
    frozen = np.ascontiguousarray(np.stack([Dref, Pref]).astype(np.float32))
    loss, grad = np.full(8, np.nan, np.float32), np.full(pN.size, np.nan, np.float32)
    emu.plate2d_loss_grad(pN.ctypes.data, lN, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, frozen.ctypes.data, 20.0, 0.25, 1.0,
                          tw, loss.ctypes.data, grad.ctypes.data, False, "fp32", ws.ctypes.data, wsb)
    assert rel(loss[:5], ss) < 5e-6 and rel(grad, g) < 5e-6, (rel(loss[:5], ss), rel(grad, g))
    th = rng.random(n) * np.pi / 2
    H = np.stack([0.1 * np.cos(th), 0.1 * np.sin(th), rng.random(n) * 10], 1)
    hx, hy, ht = (H[:, k].astype(np.float32).copy() for k in range(3))
    DH, PH = pl.net_streams(fD, lD, H[:, 0], H[:, 1], H[:, 2])[0], pl.net_streams(fP, lD, H[:, 0], H[:, 1], H[:, 2])[0]
    ssh, gh = pl.traction_loss_grad(fN, lN, H[:, 0], H[:, 1], H[:, 2], DH, PH, weight=10.0 / n)
    aux = np.ascontiguousarray(np.concatenate([DH, PH, (-H[:, 0] / 0.1)[None], (-H[:, 1] / 0.1)[None]]).astype(np.float32))
    emu.plate2d_traction_loss_grad(pN.ctypes.data, lN, hx.ctypes.data, hy.ctypes.data, ht.ctypes.data, n, LBp, UBp, False, aux.ctypes.data,
                                   [10.0 / n] * 2, loss.ctypes.data, grad.ctypes.data, True, "fp32", ws.ctypes.data, wsb)      # accumulates
    assert rel(loss[:2], ssh) < 5e-6 and rel(grad, g + gh) < 5e-6
    tg = rng.standard_normal((5, 5, n))
    w = np.zeros((5, 5))
    w[0, :] = 1000.0 / n
    w[3, 0] = w[3, 1] = 500.0 / n
    pD = fD.astype(np.float32)
    s3, g3 = pl.stream_loss_grad(fD, lD, C[:, 0], C[:, 1], C[:, 2], tg, w)
    tg32 = np.ascontiguousarray(tg.astype(np.float32))
    gradD = np.full(pD.size, np.nan, np.float32)
    wsd = emu.workspace_bytes(lD, n, "fp32")
    wsD = aligned(wsd)
    emu.stream_loss_grad(pD.ctypes.data, lD, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, tg32.ctypes.data, w, loss.ctypes.data,
                         gradD.ctypes.data, False, "fp32", wsD.ctypes.data, wsd)
    assert rel(loss[:5], ((w / w.max()) * s3).sum(0)) < 5e-6 and rel(gradD, g3) < 5e-6
    # ---- 4-input heads
    layers, m = [4] + 3 * [24] + [12], 270
    lb, ub = [0.0, 0.0, -20.0, 0.0], [30.0, 30.0, 0.0, 15.0]
    Ws, bs = po.xavier_init(layers, rng)
    flat = po.pack_params(Ws, [0.3 * rng.standard_normal(b.shape) for b in bs])
    X = n3.halfspace_points(m, lb, ub, rng)
    tw3 = (0.5 + rng.random(12)) / m
    ss3, g3d, _ = n3.nc3d_loss_grad(flat, layers, *X.T, lb, ub, True, term_weights=tw3)
    p32 = flat.astype(np.float32)
    cols = [X[:, k].astype(np.float32).copy() for k in range(4)]
    ptr = [v.ctypes.data for v in cols]
    wsb3 = emu.min_workspace_bytes(layers, "fp32")
    assert wsb3 > 0
    ws3 = aligned(wsb3)
    loss3 = np.full(16, np.nan, np.float32)
    grad3 = np.full(p32.size, np.nan, np.float32)
    emu.nc3d_loss_grad(p32.ctypes.data, layers, *ptr, m, lb, ub, True, 2.5, 0.25, 1.0, tw3, loss3.ctypes.data, grad3.ctypes.data, False, "fp32",
                       ws3.ctypes.data, wsb3)
    assert rel(loss3[:12], ss3) < 5e-6 and rel(grad3, g3d) < 5e-6, (rel(loss3[:12], ss3), rel(grad3, g3d))
    ref = n3.nc3d_fields(flat, layers, *X.T, lb, ub, True)
    fo = np.full((5, 12, m), np.nan, np.float32)
    emu.nc3d_fields(p32.ctypes.data, layers, *ptr, m, lb, ub, True, fo.ctypes.data, "fp32", ws3.ctypes.data, wsb3)
    assert rel(fo[0].T, ref["Y"]) < 5e-6
    for k in range(4):
        assert rel(fo[1 + k].T, ref["dY"][k]) < 5e-6
    tgt = rng.standard_normal((m, 12))
    ow = np.array([1, 1, 1, 0.5, 0.5, 0.5, 0, 0, 2, 0, 2, 2.0]) / m
    ss_d, g_d, _ = n3.nc3d_data_loss_grad(flat, layers, *X.T, lb, ub, True, tgt, ow)
    tgT = np.ascontiguousarray(tgt.T.astype(np.float32))
    emu.nc3d_data_loss_grad(p32.ctypes.data, layers, *ptr, m, lb, ub, True, tgT.ctypes.data, ow, loss3.ctypes.data, grad3.ctypes.data, False, "fp32",
                            ws3.ctypes.data, wsb3)
    assert rel(loss3[:12], ss_d) < 5e-6 and rel(grad3, g_d) < 5e-6


def test_fused_nc3d_emulated(emu):
    """BASELINE configs[4] net shape (10 hidden layers, padded width 128, inputs (x, y, z, t), 12 outputs) through the five-stream
    LDS-operand instantiation of the fused kernel (round 3: 40 KB images, two tiles fill the LDS, net constants from memory, 16-output
    head) against the float64 oracle and against the two-kernel path; several workgroup steps and a ragged point count."""
    from oracle import nc3d_oracle as n3
    layers = [4] + 10 * [100] + [12]
    lb, ub = [0.0, 0.0, -20.0, 0.0], [30.0, 30.0, 0.0, 15.0]
    rng = np.random.default_rng(31)
    Ws, bs = po.xavier_init(layers, rng)
    flat = po.pack_params(Ws, [0.2 * rng.standard_normal(b.shape) for b in bs])
    p32 = flat.astype(np.float32)
    for n, min_ws in ((45, False), (150, True)):
        X = n3.halfspace_points(n, lb, ub, rng)
        tw = (0.5 + rng.random(12)) / n
        ss, g, _ = n3.nc3d_loss_grad(flat, layers, *X.T, lb, ub, True, term_weights=tw)
        cols = [X[:, k].astype(np.float32).copy() for k in range(4)]
        ptr = [v.ctypes.data for v in cols]
        wsb = emu.min_workspace_bytes(layers, "f16x3") if min_ws else emu.workspace_bytes(layers, n, "f16x3")
        ws = aligned(wsb)
        res = {}
        for fused in (True, False):
            emu.set_fused(fused)
            loss = np.full(16, np.nan, np.float32)
            grad = np.full(p32.size, np.nan, np.float32)
            emu.nc3d_loss_grad(p32.ctypes.data, layers, *ptr, n, lb, ub, True, 2.5, 0.25, 1.0, tw, loss.ctypes.data, grad.ctypes.data, False,
                               "f16x3", ws.ctypes.data, wsb)
            res[fused] = (loss[:12].copy(), grad.copy())
            assert rel(loss[:12], ss) < 2e-6 and rel(grad, g) < 5e-6, (fused, n, rel(loss[:12], ss), rel(grad, g))
        emu.set_fused(True)
        assert rel(res[True][1], res[False][1].astype(np.float64)) < 5e-6
        # round 6: the value-only side sets of the 3-D step (source, initial state, traction-free surface) through the ONE-stream instantiation of
        # the same parked layout (Fused<.., 128, 10, 1, false, 4>): no set of a 3-D training step is left on the two-kernel path
        if n != 45:
            continue        # (one size is enough here: the GPU test runs it at 1500 points)
        tgt = rng.standard_normal((n, 12))
        ow = np.array([1, 1, 1, 0.5, 0.5, 0.5, 0, 0, 2, 0, 2, 2.0]) / n
        ss_d, g_d, _ = n3.nc3d_data_loss_grad(flat, layers, *X.T, lb, ub, True, tgt, ow)
        tg = np.ascontiguousarray(tgt.T.astype(np.float32))
        resd = {}
        for fused in (True, False):
            emu.set_fused(fused)
            emu.path_counts(reset=True)
            loss = np.full(16, np.nan, np.float32)
            grad = np.full(p32.size, np.nan, np.float32)
            emu.nc3d_data_loss_grad(p32.ctypes.data, layers, *ptr, n, lb, ub, True, tg.ctypes.data, ow, loss.ctypes.data, grad.ctypes.data, False,
                                    "f16x3", ws.ctypes.data, wsb)
            pc = emu.path_counts(reset=True)
            assert pc["fused-lds" if fused else "two-kernel"] == 1 and sum(pc.values()) == 1, (fused, pc)
            resd[fused] = grad.copy()
            assert rel(loss[:12], ss_d) < 2e-6 and rel(grad, g_d) < 5e-6, (fused, n, rel(loss[:12], ss_d), rel(grad, g_d))
        emu.set_fused(True)
        assert not np.array_equal(resd[True], resd[False])
        # ... and accumulated behind the collocation call, as a training step does
        grad = res[True][1].copy()
        emu.nc3d_data_loss_grad(p32.ctypes.data, layers, *ptr, n, lb, ub, True, 0, ow, loss.ctypes.data, grad.ctypes.data, True, "f16x3", ws.ctypes.data, wsb)
        ss_0, g_0, _ = n3.nc3d_data_loss_grad(flat, layers, *X.T, lb, ub, True, None, ow)
        assert rel(loss[:12], ss_0) < 2e-6 and rel(grad, g + g_0) < 5e-6


def test_fused_width160_emulated(emu):
    """Padded width 160 -- the reference's confined-domain net, 6 x 140 (CONF:891) -- through the LDS-operand layout of the fused kernel
    (round 3): ten blocks per side, five per chain half (two record pairs + a single), 40 KB images with the state in one slot, net
    constants from memory, and the weight gradient STREAMING its running sums in three passes over the out-blocks.  Against the oracle
    and the two-kernel path; several workgroup steps, a ragged count, raw inputs as the script feeds them (CONF:235)."""
    layers = [3] + 6 * [140] + [7]
    e_loss, e_grad = run_wave(emu, layers, 50, "f16x3", fused=True)
    assert e_loss < 2e-6 and e_grad < 3e-6, (e_loss, e_grad)
    e_loss, e_grad = run_wave(emu, layers, 50, "f16x3", fused=False)
    assert e_loss < 2e-6 and e_grad < 2e-6
    e_loss, e_grad = run_wave(emu, layers, 150, "f16x3", min_ws=True, fused=True, normalize=False, seed=4)      # several steps per workgroup
    assert e_loss < 2e-6 and e_grad < 5e-6, (e_loss, e_grad)


def test_weight_beyond_the_fused_format_is_detected_emulated(emu):
    """|w| > 2047 does not fit the fused kernels' weight format (32 w as fp16): the call returns NaN in EVERY gradient entry and loss sum
    (repack flags -> reduction), PINN_FLAG_TWO_KERNEL evaluates the same weights correctly, and weights inside the range are untouched"""
    from pinn_elastodynamics_amd.capi import FLAG_TWO_KERNEL, PREC
    layers, n = [3] + 4 * [32] + [7], 90
    rng = np.random.default_rng(5)
    Ws, bs = po.xavier_init(layers, rng)
    Ws[2][3, 5] = 3000.0                       # one weight out of range (tanh saturates behind it: the oracle's result stays finite)
    X = po.collocation_points(n, LB, UB, rng)
    flat = po.pack_params(Ws, bs)
    tw = np.ones(7) / n
    ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True, term_weights=tw)
    assert abs(emu.fused_weight_limit() - 2047.0) < 1e-3
    p32 = flat.astype(np.float32)
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    wsb = emu.workspace_bytes(layers, n, "f16x3")
    emu.set_fused(True)
    out = {}
    for name, mode in (("fused", PREC["f16x3"]), ("two_kernel", PREC["f16x3"] | FLAG_TWO_KERNEL)):
        ws = aligned(wsb)
        loss = np.zeros(8, np.float32)
        grad = np.zeros(p32.size, np.float32)
        emu.wave2d_loss_grad(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, True, 2.5, 0.25, 1.0, True,
                             tw, loss.ctypes.data, grad.ctypes.data, False, mode, ws.ctypes.data, wsb)
        out[name] = (loss[:7].copy(), grad.copy())
    assert np.isnan(out["fused"][0]).all() and np.isnan(out["fused"][1]).all()
    # (a weight of 3000 carries an absolute rounding error of 3000 * 2^-22 into its pre-activation: the gradient is good to ~1e-4 there)
    assert rel(out["two_kernel"][0], ss) < 2e-6 and rel(out["two_kernel"][1], g) < 5e-4


@pytest.mark.parametrize("lN,n", [([3] + 4 * [24] + [5], 150), ([3] + 8 * [64] + [5], 90), ([3] + 8 * [70] + [5], 70)])
def test_hole_traction_through_the_fused_one_stream_kernel_emulated(emu, lN, n):
    """round 4: the plate's hole-traction set (net_t, PLATE:452-461: composite values P + D N, normals) through the ONE-STREAM instantiation
    of the fused kernel -- its head takes the set's kind from the set table -- against the oracle and the two-kernel path it replaces
    (narrow layouts: all layer states in LDS; 8 x 70: the LDS-operand layout of padded width 96)"""
    from oracle import plate_oracle as pl
    prec, LBp, UBp = "f16x3", [0, 0, 0], [0.5, 0.5, 10]
    lD = [3, 10, 10, 5]
    rng = np.random.default_rng(3)

    def mk(l):
        W, b = po.xavier_init(l, rng)
        return po.pack_params(W, [0.2 * rng.standard_normal(x.shape) for x in b])

    fN, fD, fP = mk(lN), mk(lD), mk(lD)
    th = rng.random(n) * np.pi / 2
    H = np.stack([0.1 * np.cos(th), 0.1 * np.sin(th), rng.random(n) * 10], 1)
    hx, hy, ht = (H[:, k].astype(np.float32).copy() for k in range(3))
    DH, PH = pl.net_streams(fD, lD, H[:, 0], H[:, 1], H[:, 2])[0], pl.net_streams(fP, lD, H[:, 0], H[:, 1], H[:, 2])[0]
    ssh, gh = pl.traction_loss_grad(fN, lN, H[:, 0], H[:, 1], H[:, 2], DH, PH, weight=10.0 / n)
    aux = np.ascontiguousarray(np.concatenate([DH, PH, (-H[:, 0] / 0.1)[None], (-H[:, 1] / 0.1)[None]]).astype(np.float32))
    pN = fN.astype(np.float32)
    wsb = emu.workspace_bytes(lN, n, prec)
    res = {}
    try:
        for fused in (1, 0):
            emu.set_fused(fused)
            ws = aligned(wsb)
            loss, grad = np.full(8, np.nan, np.float32), np.full(pN.size, np.nan, np.float32)
            emu.plate2d_traction_loss_grad(pN.ctypes.data, lN, hx.ctypes.data, hy.ctypes.data, ht.ctypes.data, n, LBp, UBp, False, aux.ctypes.data,
                                           [10.0 / n] * 2, loss.ctypes.data, grad.ctypes.data, False, prec, ws.ctypes.data, wsb)
            res[fused] = (loss[:2].copy(), grad.copy())
            # (the narrow fused layouts take the layer states as fp16 high parts in the weight gradient: 2^-12 / sqrt(n) of noise)
            assert rel(loss[:2], ssh) < 2e-6 and rel(grad, gh) < ((2e-4 if lN[1] <= 64 else 3e-6) if fused else 2e-6), (fused, rel(grad, gh))
    finally:
        emu.set_fused(1)
    assert not np.array_equal(res[1][1], res[0][1])          # two different code paths ran


def _step_case(emu, layers, n, n_side, prec, seed, with_adam):
    """pinn_wave2d_step against pinn_wave2d_loss_grad + pinn_data_loss_grad_multi (+ pinn_adam_step) on the same inputs"""
    emu.set_fused(True)
    rng = np.random.default_rng(seed)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.3 * rng.standard_normal(b.shape) for b in bs]
    flat = po.pack_params(Ws, bs).astype(np.float32)
    X = po.collocation_points(n, LB, UB, rng)
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    tw = np.array([1, 2, 3, 1, 0.5, 1, 2.0]) / n
    sets_np = []
    for k, m in enumerate(n_side):
        S = po.collocation_points(m, LB, UB, rng).astype(np.float32)
        tg = (0.1 * rng.standard_normal((7, m))).astype(np.float32) if k % 2 == 0 else None
        ow = [(1.0 + i) / max(m, 1) if i in ((0, 1), (0, 1, 2, 3), (5, 6))[k % 3] else 0.0 for i in range(7)]
        sets_np.append([np.ascontiguousarray(S[:, j]) for j in range(3)] + [tg, ow, np.full(8, np.nan, np.float32)])
    wsb = emu.workspace_bytes(layers, max(n, 1 << 12), prec)
    out = {}
    for mode in ("step", "calls"):
        ws = aligned(wsb)
        theta = flat.copy()
        m1, v1 = np.full(theta.size, 0.01, np.float32), np.full(theta.size, 0.02, np.float32)
        loss = np.full(8, np.nan, np.float32)
        grad = np.full(theta.size, np.nan, np.float32)
        rows = []
        for sx, sy, st, tg, ow, lo in sets_np:
            lo[:] = np.nan
            rows.append((sx.ctypes.data, sy.ctypes.data, st.ctypes.data, sx.size, 0 if tg is None else tg.ctypes.data, ow, lo.ctypes.data))
        emu.path_counts(reset=True)
        if mode == "step":
            emu.wave2d_step(theta.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, True, 2.5, 0.25, 1.0, True, tw, loss.ctypes.data,
                            rows, grad.ctypes.data, False, (m1.ctypes.data, v1.ctypes.data, 1e-3, 0.9, 0.999, 1e-8, 3) if with_adam else None, prec,
                            ws.ctypes.data, wsb)
        else:
            emu.wave2d_loss_grad(theta.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, True, 2.5, 0.25, 1.0, True, tw,
                                 loss.ctypes.data, grad.ctypes.data, False, prec, ws.ctypes.data, wsb)
            emu.data_loss_grad_multi(theta.ctypes.data, layers, rows, LB, UB, True, grad.ctypes.data, True, prec, ws.ctypes.data, wsb)
            if with_adam:
                emu.adam_step(theta.ctypes.data, m1.ctypes.data, v1.ctypes.data, grad.ctypes.data, theta.size, 1e-3, 3)
        out[mode] = dict(theta=theta, m=m1, v=v1, loss=loss[:7].copy(), grad=grad, side=[r[5][:7].copy() for r in sets_np], counts=emu.path_counts(reset=True))
    return out, (flat, X, tw, sets_np)


@pytest.mark.parametrize("layers,n,n_side,prec,with_adam", [
    ([3] + 4 * [32] + [7], 300, (70, 50), "f16x3", True),           # BASELINE configs[0] net: four-stream part with all states in LDS
    ([3] + 8 * [64] + [7], 200, (70, 0, 130), "f16x3", True),       # configs[1] net: parked states, S_1 from the weight-gradient wave; an empty set
    ([3] + 8 * [64] + [7], 130, (40,), "bf16", False),              # one MFMA per product, no optimizer step (the data-parallel form)
    ([3] + 8 * [80] + [7], 100, (40, 33), "f16x3", True),           # round 6: the LDS-operand layouts take the one-launch step too -- INF:645's net (padded width 96, two state slots)
    ([3] + 8 * [100] + [7], 70, (35,), "f16x3", False),             # SEMI:679's net (padded width 128: one slot, QUAD chain, streamed sums)
])
def test_step_call_is_the_separate_calls_bit_for_bit_emulated(emu, layers, n, n_side, prec, with_adam):
    """pinn_wave2d_step (round 5): collocation set + side sets in ONE persistent launch (fused_step_kernel: the side sets' workgroups behind the
    collocation set's), one reduction with the Adam update in it.  Same partial sums, same order of the final additions, same Adam expression
    as the three calls it replaces: identical bits in loss sums, gradient, parameters and both moments -- and the oracle's numbers."""
    out, (flat, X, tw, sets_np) = _step_case(emu, layers, n, n_side, prec, 5, with_adam)
    a, b = out["step"], out["calls"]
    path = "fused-registers" if layers[1] <= 64 else "fused-lds"
    assert a["counts"][path] == 2 and b["counts"][path] == 2 and sum(a["counts"].values()) == 2, (a["counts"], b["counts"])
    for key in ("loss", "grad", "theta", "m", "v"):
        assert np.array_equal(a[key], b[key]), key
    for sa, sb in zip(a["side"], b["side"]):
        assert np.array_equal(sa, sb)
    assert np.isfinite(a["grad"]).all() and (not with_adam or not np.array_equal(a["theta"], flat))
    # against the float64 oracle: collocation terms + data terms
    ss, g, _ = po.wave2d_loss_grad(flat.astype(np.float64), layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True, term_weights=tw)
    gsum = g.copy()
    for (sx, sy, st, tg, ow, lo), got in zip(sets_np, a["side"]):
        if sx.size == 0:
            assert np.all(got == 0.0)
            continue
        s2, g2, _ = po.data_loss_grad(flat.astype(np.float64), layers, sx, sy, st, LB, UB, True, None if tg is None else tg.T.astype(np.float64), np.asarray(ow))
        gsum += g2
        assert rel(got, s2) < (2e-2 if prec == "bf16" else 2e-6)
    assert rel(a["loss"], ss) < (2e-2 if prec == "bf16" else 2e-6) and rel(a["grad"], gsum) < (3e-2 if prec == "bf16" else 1e-4)


def test_step_call_several_steps_per_workgroup_and_fallbacks_emulated(emu):
    """more steps than workgroups in both parts (the persistent accumulators of both roles carry over), and the cases the one-launch form
    declines -- a depth without fused kernel, PINN_PREC_FP32 -- which make the separate calls inside: same bits as making them outside."""
    out, _ = _step_case(emu, [3] + 4 * [32] + [7], 17000, (16500,), "f16x3", 9, True)
    a, b = out["step"], out["calls"]
    assert a["counts"]["fused-registers"] == 2
    for key in ("loss", "grad", "theta", "m", "v"):
        assert np.array_equal(a[key], b[key]), key
    for layers, prec, path in (([3] + 3 * [32] + [7], "f16x3", "two-kernel"), ([3] + 4 * [32] + [7], "fp32", "fp32")):
        out, _ = _step_case(emu, layers, 150, (40, 30), prec, 2, True)
        a, b = out["step"], out["calls"]
        assert a["counts"][path] >= 2 and a["counts"]["fused-registers"] == 0, a["counts"]
        for key in ("loss", "grad", "theta", "m", "v"):
            assert np.array_equal(a[key], b[key]), (layers, prec, key)


@pytest.mark.parametrize("lN,n,nh", [([3] + 8 * [64] + [5], 150, 70), ([3] + 4 * [32] + [5], 300, 40), ([3] + 8 * [70] + [5], 100, 40)])      # (8 x 70 = PLATE:885's net: the five-stream LDS-operand layout, round 6)
def test_plate_step_call_is_the_separate_calls_bit_for_bit_emulated(emu, lN, n, nh):
    """pinn_plate2d_step: the plate's five-stream collocation set and its hole-traction set in one launch (fused_step_kernel<..., NSC = 5>), one
    reduction with Adam -- identical bits to pinn_plate2d_loss_grad + pinn_plate2d_traction_loss_grad + pinn_adam_step, and the oracle's numbers."""
    from oracle import plate_oracle as pl
    prec, LBp, UBp = "f16x3", [0, 0, 0], [0.5, 0.5, 10]
    lD = [3, 10, 10, 5]
    rng = np.random.default_rng(4)

    def mk(l):
        W, b = po.xavier_init(l, rng)
        return po.pack_params(W, [0.2 * rng.standard_normal(x.shape) for x in b])

    fN, fD, fP = mk(lN), mk(lD), mk(lD)
    X = np.stack([0.12 + 0.38 * rng.random(n), 0.12 + 0.38 * rng.random(n), 10 * rng.random(n)], 1)
    th = rng.random(nh) * np.pi / 2
    H = np.stack([0.1 * np.cos(th), 0.1 * np.sin(th), rng.random(nh) * 10], 1)
    frozen = np.ascontiguousarray(np.stack([pl.net_streams(f, lD, X[:, 0], X[:, 1], X[:, 2]) for f in (fD, fP)]).astype(np.float32))       # [2][5][5][n]
    DH, PH = pl.net_streams(fD, lD, H[:, 0], H[:, 1], H[:, 2])[0], pl.net_streams(fP, lD, H[:, 0], H[:, 1], H[:, 2])[0]
    aux = np.ascontiguousarray(np.concatenate([DH, PH, (-H[:, 0] / 0.1)[None], (-H[:, 1] / 0.1)[None]]).astype(np.float32))
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    hx, hy, ht = (H[:, k].astype(np.float32).copy() for k in range(3))
    tw, hw = [10.0 / n] * 5, [10.0 / nh] * 2
    wsb = emu.workspace_bytes(lN, 1 << 12, prec)
    emu.set_fused(True)
    out = {}
    for mode in ("step", "calls"):
        ws = aligned(wsb)
        theta = fN.astype(np.float32)
        m1, v1 = np.full(theta.size, 0.01, np.float32), np.full(theta.size, 0.02, np.float32)
        loss, hloss, grad = np.full(8, np.nan, np.float32), np.full(8, np.nan, np.float32), np.full(theta.size, np.nan, np.float32)
        emu.path_counts(reset=True)
        if mode == "step":
            emu.plate2d_step(theta.ctypes.data, lN, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, frozen.ctypes.data, 20.0, 0.25, 1.0, tw,
                             loss.ctypes.data, hx.ctypes.data, hy.ctypes.data, ht.ctypes.data, nh, aux.ctypes.data, hw, hloss.ctypes.data, grad.ctypes.data,
                             False, (m1.ctypes.data, v1.ctypes.data, 1e-3, 0.9, 0.999, 1e-8, 2), prec, ws.ctypes.data, wsb)
        else:
            emu.plate2d_loss_grad(theta.ctypes.data, lN, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LBp, UBp, False, frozen.ctypes.data, 20.0, 0.25, 1.0,
                                  tw, loss.ctypes.data, grad.ctypes.data, False, prec, ws.ctypes.data, wsb)
            emu.plate2d_traction_loss_grad(theta.ctypes.data, lN, hx.ctypes.data, hy.ctypes.data, ht.ctypes.data, nh, LBp, UBp, False, aux.ctypes.data, hw,
                                           hloss.ctypes.data, grad.ctypes.data, True, prec, ws.ctypes.data, wsb)
            emu.adam_step(theta.ctypes.data, m1.ctypes.data, v1.ctypes.data, grad.ctypes.data, theta.size, 1e-3, 2)
        out[mode] = dict(theta=theta, m=m1, v=v1, loss=loss[:5].copy(), hloss=hloss[:2].copy(), grad=grad, counts=emu.path_counts(reset=True))
    a, b = out["step"], out["calls"]
    path = "fused-registers" if lN[1] <= 64 else "fused-lds"
    assert a["counts"][path] == 2 and b["counts"][path] == 2 and sum(a["counts"].values()) == 2, (a["counts"], b["counts"])
    for key in ("loss", "hloss", "grad", "theta", "m", "v"):
        assert np.array_equal(a[key], b[key]), key
    ss, g = pl.plate_loss_grad(fN, lN, X[:, 0], X[:, 1], X[:, 2], frozen[0].astype(np.float64), frozen[1].astype(np.float64), term_weights=np.asarray(tw))[:2]
    ssh, gh = pl.traction_loss_grad(fN, lN, H[:, 0], H[:, 1], H[:, 2], DH, PH, weight=10.0 / nh)
    assert rel(a["loss"], ss) < 5e-6 and rel(a["hloss"], ssh) < 5e-6 and rel(a["grad"], g + gh) < 3e-4


@pytest.mark.parametrize("layers,permille", [([3] + 4 * [32] + [7], 16), ([3] + 8 * [64] + [7], 150)])
def test_xcd_aware_tail_takes_every_step_once_emulated(emu, layers, permille):
    """The XCD-aware step assignment (FusedArgs::n_plain, fused_next_step): steps [0, n_plain) one per workgroup and round, the tail to the
    even-numbered workgroups of every group of eight.  The x86 build applies it to small grids (multiples of 8, >= 4 rounds), so the index
    arithmetic runs here: 8 workgroups, 6.2 rounds, a ragged last step -- sums and gradient equal the oracle's and the unskewed assignment's
    to summation noise, and differ from the latter in the last bits (another grouping did run)."""
    prec = "f16x3"
    rng = np.random.default_rng(11)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.3 * rng.standard_normal(b.shape) for b in bs]
    n = 8 * 64 * 6 + 100
    X = po.collocation_points(n, LB, UB, rng)
    flat = po.pack_params(Ws, bs)
    tw = np.array([1, 2, 3, 1, 0.5, 1, 2.0]) / n
    ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True, term_weights=tw)
    p32 = flat.astype(np.float32)
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    emu.set_fused(True)
    wsb = emu.workspace_bytes(layers, 1 << 14, prec)
    out = {}
    emu.lib.pinn_debug_set_fused_grid_cap(8)                    # 8 workgroups: 6.2 rounds of 64 points
    try:
        for pm in (0, permille):
            emu.lib.pinn_debug_set_xcd_bonus(pm)
            ws = aligned(wsb)
            loss = np.full(8, np.nan, np.float32)
            grad = np.full(p32.size, np.nan, np.float32)
            emu.path_counts(reset=True)
            emu.wave2d_loss_grad(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, n, LB, UB, True, 2.5, 0.25, 1.0, True,
                                 tw, loss.ctypes.data, grad.ctypes.data, False, prec, ws.ctypes.data, wsb)
            assert emu.path_counts(reset=True)["fused-registers"] == 1
            out[pm] = (loss[:7].copy(), grad.copy())
    finally:
        emu.lib.pinn_debug_set_xcd_bonus(16)
        emu.lib.pinn_debug_set_fused_grid_cap(0)
    for pm in out:
        assert rel(out[pm][0], ss) < 2e-6 and rel(out[pm][1], g) < 1e-4, pm
    assert rel(out[permille][1], out[0][1]) < 2e-6 and not np.array_equal(out[permille][1], out[0][1])


def test_checked_call_ladder_emulated(emu):
    """pinn_wave2d_loss_grad_checked / pinn_probe_ranges (round 6): the model classes' finite-gradient ladder as library calls.  (a) In-range
    weights: one evaluation, state untouched, the same bits as pinn_wave2d_loss_grad.  (b) A weight of 2100 (> 2047, the fused format's range):
    the plain call returns NaN throughout; the checked call probes, sets state.two_kernel, repeats on the two-kernel path and agrees with the
    float64 oracle.  (c) Output-layer weights x 3000 with the term weights of a 4000-point mean (the case of tests/test_gpu_parity.py): the
    reverse pass overflows fp16, the ladder raises the adjoint shift until the gradient is finite."""
    from pinn_elastodynamics_amd.capi import RangeState
    emu.set_fused(True)
    layers = [3] + 4 * [32] + [7]
    rng = np.random.default_rng(5)
    Ws, bs = po.xavier_init(layers, rng)
    X = po.collocation_points(130, LB, UB, rng)
    x, y, t = (X[:, k].astype(np.float32).copy() for k in range(3))
    tw = np.ones(7) / 130
    wsb = emu.workspace_bytes(layers, 130, "f16x3")
    ws = aligned(wsb)

    def both(flat, tw_):
        p32 = flat.astype(np.float32)
        la, ga = np.full(8, np.nan, np.float32), np.full(p32.size, np.nan, np.float32)
        lb_, gb = la.copy(), ga.copy()
        emu.wave2d_loss_grad(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, 130, LB, UB, True, 2.5, 0.25, 1.0, True, tw_,
                             la.ctypes.data, ga.ctypes.data, False, "f16x3", ws.ctypes.data, wsb)
        st = RangeState()
        rc = emu.wave2d_loss_grad_checked(p32.ctypes.data, layers, x.ctypes.data, y.ctypes.data, t.ctypes.data, 130, LB, UB, True, 2.5, 0.25, 1.0, True, tw_,
                                          lb_.ctypes.data, gb.ctypes.data, "f16x3", ws.ctypes.data, wsb, st)
        ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True, term_weights=tw_)
        return rc, st, (la, ga), (lb_, gb), (ss, g), p32

    rc, st, plain, chk, ref, p32 = both(po.pack_params(Ws, bs), tw)
    assert rc == 0 and (st.adjoint_shift, st.two_kernel, st.attempts) == (0, 0, 1)
    assert np.array_equal(plain[0][:7], chk[0][:7]) and np.array_equal(plain[1], chk[1]) and rel(chk[1], ref[1]) < 1e-4
    fin, wmax = emu.probe_ranges(p32.ctypes.data, chk[1].ctypes.data, p32.size, ws.ctypes.data, wsb)
    assert fin and abs(wmax - np.abs(p32).max()) == 0.0

    big = [W.copy() for W in Ws]
    big[2][3, 5] = 2100.0
    rc, st, plain, chk, ref, p32 = both(po.pack_params(big, bs), tw)
    assert not np.isfinite(plain[1]).any() and not np.isfinite(plain[0][:7]).any()          # beyond the fused format: NaN throughout
    assert rc == 0 and st.two_kernel == 1 and st.attempts == 2 and st.adjoint_shift == 0
    assert rel(chk[0][:7], ref[0]) < 1e-4 and rel(chk[1], ref[1]) < 2e-3          # (one weight of 2100: a saturated unit; the float64 oracle against fp32-class arithmetic)

    out = [W.copy() for W in Ws]
    out[-1] = out[-1] * 3000.0
    rc, st, plain, chk, ref, p32 = both(po.pack_params(out, bs), np.ones(7) / 130)
    assert np.isfinite(plain[0][:7]).all() and not np.isfinite(plain[1]).all()                # sums fine, gradient overflowed
    assert rc == 0 and st.adjoint_shift >= 4 and st.two_kernel == 0 and st.attempts == 1 + st.adjoint_shift // 4
    assert np.isfinite(chk[1]).all() and rel(chk[1], ref[1]) < 1e-3
