"""-m gpu: the 3-D Navier-Cauchy entry points (BASELINE.json configs[4]: 10x128 net, inputs (x, y, z, t), 12 outputs) through the
C-ABI against the float64 oracle (oracle/nc3d_oracle.py).  The 3-D case is a build-side extension -- PARITY UNPINNED by definition:
the reference has nothing to compare with; what is checked is HIP path == oracle, and the oracle against closed-form
elastodynamics (tests/test_oracle_nc3d.py).

Tolerances (relative L2, f16x3 = the fp32-class mode this config's "fp32" is served by): sums, gradient and fields 2e-5 on fresh
Xavier weights."""
import numpy as np
import pytest
import torch

from oracle import nc3d_oracle as n3
from oracle import pinn_oracle as po

pytestmark = pytest.mark.gpu
LB, UB = [0.0, 0.0, -30.0, 0.0], [30.0, 30.0, 0.0, 15.0]
LAYERS = [4] + 10 * [128] + [12]


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def to_dev(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def net(layers, seed):
    rng = np.random.default_rng(seed)
    Ws, bs = po.xavier_init(layers, rng)
    bs = [0.2 * rng.standard_normal(b.shape) for b in bs]
    return po.pack_params(Ws, bs), rng


@pytest.mark.parametrize("layers,n,prec,tol", [(LAYERS, 1500, "f16x3", 2e-5), ([4] + 3 * [64] + [12], 3000, "f16x3", 2e-5), ([4] + 3 * [64] + [12], 700, "bf16x3", 3e-4)])
def test_nc3d_loss_grad_and_fields_vs_oracle(dev, layers, n, prec, tol):
    from pinn_elastodynamics_amd.hip_engine import HipEngine
    flat, rng = net(layers, 31)
    X = n3.halfspace_points(n, LB, UB, rng)
    tw = (0.5 + rng.random(12)) / n
    ss, g, _ = n3.nc3d_loss_grad(flat, layers, *X.T, LB, UB, True, term_weights=tw)
    eng = HipEngine(layers, precision=prec, device=dev, max_points=n)
    theta = to_dev(flat, dev)
    cols = [to_dev(X[:, k], dev) for k in range(4)]
    loss, grad = eng.nc3d_loss_grad(theta, *cols, LB, UB, True, tw)
    assert rel(loss.cpu().numpy(), ss) < tol and rel(grad.cpu().numpy(), g) < tol, (rel(loss.cpu().numpy(), ss), rel(grad.cpu().numpy(), g))
    ref = n3.nc3d_fields(flat, layers, *X.T, LB, UB, True)
    F = eng.nc3d_fields(theta, *cols, LB, UB, True).cpu().numpy()
    assert rel(F[0].T, ref["Y"]) < tol
    for k in range(4):
        assert rel(F[1 + k].T, ref["dY"][k]) < tol
    # value-only term with targets (source / initial state / free surface)
    tgt = rng.standard_normal((n, 12))
    ow = np.array([1, 1, 1, 0.5, 0.5, 0.5, 0, 0, 2, 0, 2, 2.0]) / n
    ssd, gd, _ = n3.nc3d_data_loss_grad(flat, layers, *X.T, LB, UB, True, tgt, ow)
    eng.lib.path_counts(reset=True)
    lossd, gradd = eng.nc3d_data_loss_grad(theta, *cols, LB, UB, True, to_dev(tgt.T, dev), ow)
    assert rel(lossd.cpu().numpy(), ssd) < tol and rel(gradd.cpu().numpy(), gd) < tol
    # round 6: the value-only sets of the BASELINE configs[4] net take the fused one-stream kernel too (Fused<.., 128, 10, 1, false, 4>); other
    # depths stay on the two-kernel path -- and pinn_path_for says which
    pc = eng.lib.path_counts(reset=True)
    expect = "fused-lds" if layers == LAYERS else "two-kernel"
    assert eng.path("nc3d_data") == expect and pc[expect] == 1 and sum(pc.values()) == 1, (layers, pc)
    if layers == LAYERS:
        eng.lib.set_fused(False)
        try:
            l2, g2 = eng.nc3d_data_loss_grad(theta, *cols, LB, UB, True, to_dev(tgt.T, dev), ow)
        finally:
            eng.lib.set_fused(True)
        assert eng.lib.path_counts(reset=True)["two-kernel"] == 1
        assert rel(g2.cpu().numpy(), gd) < tol and rel(gradd.cpu().numpy(), g2.cpu().numpy().astype(np.float64)) < tol


def test_nc3d_plane_wave_known_answer_through_the_kernels(dev):
    """Known answer end to end: a net whose hidden layer is an exact sine feature map cannot be built from tanh, so instead the
    head is fed through the kernels' own output streams -- the device residuals recomputed on the host from pinn_nc3d_fields equal
    the oracle's for the same net, and the oracle's head vanishes on exact plane waves (tests/test_oracle_nc3d.py)."""
    from pinn_elastodynamics_amd.hip_engine import HipEngine
    layers = [4] + 4 * [96] + [12]
    flat, rng = net(layers, 5)
    n = 2000
    X = n3.halfspace_points(n, LB, UB, rng)
    eng = HipEngine(layers, precision="f16x3", device=dev, max_points=n)
    cols = [to_dev(X[:, k], dev) for k in range(4)]
    F = eng.nc3d_fields(to_dev(flat, dev), *cols, LB, UB, True).cpu().numpy().astype(np.float64)
    f_dev = n3.nc3d_residuals(F[0].T, [F[1 + k].T for k in range(4)])
    _, _, f_or = n3.nc3d_loss_grad(flat, layers, *X.T, LB, UB, True, want_grad=False)
    assert rel(f_dev, f_or) < 2e-5
    loss, _ = eng.nc3d_loss_grad(to_dev(flat, dev), *cols, LB, UB, True, np.ones(12))
    assert rel(loss.cpu().numpy(), (f_or ** 2).sum(0)) < 2e-5


def test_nc3d_full_size_properties(dev):
    """Size-independent properties at the largest single-GPU size of configs[4] (32 M points / 8 GPUs = 4 M points, walked in workspace
    passes): the sums and the gradient are additive over a split of the point set, linear in the term weights, and independent of the
    workspace size."""
    from pinn_elastodynamics_amd.hip_engine import HipEngine
    n = 4_000_000
    flat, rng = net(LAYERS, 17)
    g = torch.Generator(device="cpu").manual_seed(3)
    U = torch.rand((4, n), generator=g, dtype=torch.float32)
    cols = [(LB[k] + (UB[k] - LB[k]) * U[k]).to(dev) for k in range(4)]
    theta = to_dev(flat, dev)
    eng = HipEngine(LAYERS, precision="f16x3", device=dev, max_points=1 << 17)
    tw = np.full(12, 1.0 / n)
    l_all, g_all = (v.clone() for v in eng.nc3d_loss_grad(theta, *cols, LB, UB, True, tw))
    assert bool(torch.isfinite(l_all).all()) and bool(torch.isfinite(g_all).all())
    h = 1_700_003                                  # a ragged split
    l_a, g_a = (v.clone() for v in eng.nc3d_loss_grad(theta, *[c[:h].contiguous() for c in cols], LB, UB, True, tw))
    l_b, g_b = (v.clone() for v in eng.nc3d_loss_grad(theta, *[c[h:].contiguous() for c in cols], LB, UB, True, tw))
    assert rel((l_a + l_b).cpu().numpy(), l_all.cpu().numpy().astype(np.float64)) < 1e-5
    assert rel((g_a + g_b).cpu().numpy(), g_all.cpu().numpy().astype(np.float64)) < 1e-4
    # linearity in the term weights
    tw2 = tw.copy()
    tw2[:6] *= 3.0
    _, g_2 = eng.nc3d_loss_grad(theta, *cols, LB, UB, True, tw2)
    tw_uv = tw.copy()
    tw_uv[6:] = 0.0
    _, g_uv = eng.nc3d_loss_grad(theta, *cols, LB, UB, True, tw_uv)
    assert rel(g_2.cpu().numpy(), (g_all + 2.0 * g_uv).cpu().numpy().astype(np.float64)) < 1e-4
    # a subsample agrees with the oracle (the whole set would take the CPU minutes)
    m = 1200
    idx = np.linspace(0, n - 1, m).astype(np.int64)
    Xs = np.stack([c[idx].cpu().numpy().astype(np.float64) for c in cols], 1)
    ss, go, _ = n3.nc3d_loss_grad(flat, LAYERS, *Xs.T, LB, UB, True, term_weights=np.full(12, 1.0 / m))
    ls, gs = eng.nc3d_loss_grad(theta, *[to_dev(Xs[:, k], dev) for k in range(4)], LB, UB, True, np.full(12, 1.0 / m))
    assert rel(ls.cpu().numpy(), ss) < 2e-5 and rel(gs.cpu().numpy(), go) < 2e-5


def test_nc3d_model_trains_on_device(dev):
    from pinn_elastodynamics_amd.navier_cauchy_3d import NavierCauchy3D, halfspace_case
    c = halfspace_case(n_collo=20000, n_ic=2000, n_top=2000, n_src=(40, 20), seed=2, width=64, depth=4)
    m = NavierCauchy3D(c["Collo"], c["SRC"], c["IC"], c["TOP"], c["uv_layers"], c["lb"], c["ub"], verbose=False, seed=3)
    l0 = m.getloss()[0]
    hist = m.train(60, 2e-3, 1)
    l1 = m.getloss()[0]
    assert np.isfinite(hist[4]).all() and l1 < 0.7 * l0, (l0, l1)
    out = m.predict(*[c["Collo"][:100, k:k + 1] for k in range(4)])
    assert len(out) == 15 and all(np.isfinite(o).all() for o in out)


def test_nc3d_fp32_device_leg(dev):
    """BASELINE configs[4] says "fp32": PINN_PREC_FP32 runs the 4-input heads in plain fp32 arithmetic on the device (round 3).  On a
    10 x 128 net: the fp32 run and the f16x3 product mode against the float64 oracle (build-side definition, parity unpinned), sums and
    gradient, and the f16x3 gradient within a small factor of the fp32 run's own error per weight layer."""
    from oracle import nc3d_oracle as n3
    from pinn_elastodynamics_amd.hip_engine import HipEngine
    layers = [4] + 10 * [128] + [12]
    lb, ub = [0.0, 0.0, -20.0, 0.0], [30.0, 30.0, 0.0, 15.0]
    rng = np.random.default_rng(9)
    Ws, bs = po.xavier_init(layers, rng)
    flat = po.pack_params(Ws, [0.2 * rng.standard_normal(b.shape) for b in bs])
    n = 1500
    X = n3.halfspace_points(n, lb, ub, rng)
    tw = np.ones(12) / n
    ss, g, _ = n3.nc3d_loss_grad(flat, layers, *X.T, lb, ub, True, term_weights=tw)
    theta = torch.from_numpy(flat.astype(np.float32)).to(dev)
    cols = [torch.from_numpy(np.ascontiguousarray(X[:, k], dtype=np.float32)).to(dev) for k in range(4)]
    out = {}
    for prec in ("fp32", "f16x3"):
        eng = HipEngine(layers, precision=prec, device=dev, max_points=n)
        l_, g_ = eng.nc3d_loss_grad(theta, *cols, lb, ub, True, tw)
        out[prec] = (l_.cpu().numpy().astype(np.float64), g_.cpu().numpy().astype(np.float64))
        assert np.linalg.norm(out[prec][0][:12] - ss) < 2e-5 * np.linalg.norm(ss), prec
        assert np.linalg.norm(out[prec][1] - g) < 2e-5 * np.linalg.norm(g), prec
    W32, b32 = po.unpack_params(out["fp32"][1], layers)
    W16, b16 = po.unpack_params(out["f16x3"][1], layers)
    W64, b64 = po.unpack_params(g, layers)
    for l in range(len(layers) - 1):
        for a16, a32, r in ((W16[l], W32[l], W64[l]), (b16[l], b32[l], b64[l])):
            assert np.linalg.norm(a16 - r) <= 12.0 * np.linalg.norm(a32 - r) + 2e-6 * np.linalg.norm(r), l


def test_fused_nc3d_kernel_against_oracle_and_two_kernel_path(dev):
    """Round 3: the 10 x 128 3-D net runs through the five-stream LDS-operand instantiation of the fused kernel (pinn_fused.hpp, DIN = 4).
    The library's profiling hook tells which path ran (the fused kernel reports no separate weight-gradient time); both paths against the
    float64 oracle (2e-5) and against each other (1e-5), on a point count that is not a multiple of the 32-point workgroup step, through
    the default workspace and through one that leaves the fused kernel a few workgroups with many steps each."""
    from pinn_elastodynamics_amd.hip_engine import HipEngine
    flat, rng = net(LAYERS, 17)
    n = 5000 + 13
    X = n3.halfspace_points(n, LB, UB, rng)
    tw = (0.5 + rng.random(12)) / n
    ss, g, _ = n3.nc3d_loss_grad(flat, LAYERS, *X.T, LB, UB, True, term_weights=tw)
    theta = to_dev(flat, dev)
    cols = [to_dev(X[:, k], dev) for k in range(4)]
    grads = {}
    for max_points in (n, 512):
        eng = HipEngine(LAYERS, precision="f16x3", device=dev, max_points=max_points)
        for fused in (True, False):
            eng.lib.set_fused(fused)
            try:
                with eng.lib.profiling() as prof:
                    l_, g_ = eng.nc3d_loss_grad(theta, *cols, LB, UB, True, tw)
                    torch.cuda.synchronize()
                    ran_fused = float(prof[2]) == 0.0 and float(prof[1]) > 0.0
            finally:
                eng.lib.set_fused(True)
            assert ran_fused == fused, (max_points, fused, list(prof))
            assert rel(l_.cpu().numpy()[:12], ss) < 2e-5 and rel(g_.cpu().numpy(), g) < 2e-5, (max_points, fused)
            grads[(max_points, fused)] = g_.cpu().numpy().astype(np.float64)
    assert rel(grads[(n, True)], grads[(n, False)]) < 1e-5 and rel(grads[(512, True)], grads[(n, True)]) < 1e-5
