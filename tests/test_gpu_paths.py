"""-m gpu: no silent slow path (round 5).  Every kernel path runs where pinn_path_for says it runs -- asserted with the library's own
per-path call counters --, the model classes warn once when a layer list lands on the two-kernel path, and the small-batch precision
statement of include/pinn_hip.h (PINN_PREC_F16X3 (2): the narrow collocation kernel's weight gradient multiplies fp16 high parts, a
3.5e-4 / sqrt(points) rounding noise) is held at 64 / 256 / 1024 points, fresh and trained weights."""
import warnings

import numpy as np
import pytest
import torch

from oracle import pinn_oracle as po
from tests.test_gpu_parity import LB, UB, engine, make_net, rel, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def net(depth, width, nout=7):
    return [3] + depth * [width] + [nout]


@pytest.mark.parametrize("layers,prec,expected", [(net(8, 64), "f16x3", "fused-registers"), (net(4, 32), "bf16", "fused-registers"),
                                                  (net(8, 80), "f16x3", "fused-lds"), (net(6, 140), "f16x3", "fused-lds"),
                                                  (net(5, 64), "f16x3", "two-kernel"), (net(4, 80), "f16x3", "two-kernel"),
                                                  (net(8, 64), "fp32", "fp32")])
def test_each_path_runs_where_path_for_says(dev, layers, prec, expected):
    """one collocation call and one side-set call per case: the counters of the library name the path that ran, pinn_path_for named it
    beforehand, and the numbers agree with the float64 oracle at the mode's tolerance whatever the path"""
    n = 3000
    Ws, bs, rng = make_net(layers, 11)
    flat = po.pack_params(Ws, bs)
    X = po.collocation_points(n, LB, UB, rng)
    tw = np.ones(7) / n
    ss, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True, term_weights=tw)
    eng = engine(layers, prec, dev, n)
    assert eng.path("wave") == expected
    data_expected = expected       # (round 6: padded width 160 has its one-stream instantiation too -- CONF's IC / FIX / SRC sets no longer run the two-kernel path)
    assert eng.path("data") == data_expected
    theta = to_dev(flat, dev)
    xs = [to_dev(X[:, k], dev) for k in range(3)]
    eng.lib.path_counts(reset=True)
    loss, grad = eng.wave_loss_grad(theta, *xs, LB, UB, True, tw)
    counts = eng.lib.path_counts(reset=True)
    assert counts[expected] == 1 and sum(counts.values()) == 1, counts
    tol = {"f16x3": 2e-5, "bf16": 3e-2, "fp32": 1e-4}[prec]      # (fp32: plain fp32 FMAs, 2e-5 on a good draw)
    assert rel(loss.cpu().numpy(), ss) < tol and rel(grad.cpu().numpy(), g) < tol
    ow = np.array([1.0, 1.0, 0.5, 0.5, 0.0, 2.0, 0.0]) / n
    ssd, gd, _ = po.data_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True, None, ow)
    ld, gdd = eng.data_loss_grad(theta, *xs, LB, UB, True, None, ow.tolist())
    counts = eng.lib.path_counts(reset=True)
    assert counts[data_expected] == 1 and sum(counts.values()) == 1, counts
    tol_d = {"f16x3": 2e-4 if layers[1] <= 64 else 2e-5, "bf16": 3e-2, "fp32": 1e-4}[prec]      # (narrow one-stream kernel: states reach the reverse as fp16)
    assert rel(ld.cpu().numpy()[:7], ssd) < tol and rel(gdd.cpu().numpy(), gd) < tol_d


def test_small_workspace_is_reported_and_counted_as_two_kernel(dev):
    """the advisor's round-4 point: a workspace that holds fewer than 64 scratch images leaves the fused path -- here the plate's 8 x 64 net
    at pinn_min_workspace_bytes(), the one shipped layer list for which the minimum is that small.  pinn_path_for says so beforehand, the
    counters say so afterwards, and the numbers do not depend on it."""
    layers = net(8, 64, 5)
    from pinn_elastodynamics_amd.hip_engine import HipEngine
    big = HipEngine(layers, precision="f16x3", device=dev, max_points=1 << 18)
    small = HipEngine(layers, precision="f16x3", device=dev, workspace_bytes=big.lib.min_workspace_bytes(layers, "f16x3"))
    assert big.path("plate") == "fused-registers" and small.path("plate") == "two-kernel"
    n = 20000
    Ws, bs, rng = make_net(layers, 3)
    theta = to_dev(po.pack_params(Ws, bs), dev)
    X = po.collocation_points(n, [0, 0, 0], [0.5, 0.5, 10.0], rng)
    xs = [to_dev(X[:, k], dev) for k in range(3)]
    frozen = to_dev(0.3 * rng.standard_normal((2, 5, 5, n)), dev)
    out = {}
    for name, eng in (("small", small), ("big", big)):
        eng.lib.path_counts(reset=True)
        _, g = eng.plate_loss_grad(theta, *xs, [0, 0, 0], [0.5, 0.5, 10.0], False, frozen, [1.0 / n] * 5)
        out[name] = (g.cpu().numpy(), eng.lib.path_counts(reset=True))
    assert out["big"][1]["fused-registers"] == 1 and out["small"][1]["two-kernel"] == 1, (out["big"][1], out["small"][1])
    assert rel(out["small"][0], out["big"][0]) < 3e-5


def test_model_classes_warn_once_on_the_two_kernel_path(dev):
    from pinn_elastodynamics_amd.elastic_wave import DeepHPM
    rng = np.random.default_rng(0)
    Collo = po.collocation_points(4000, LB, UB, rng)
    SRC, IC = po.ricker_source_set(n_pt=8, n_time=8), po.ic_grid(num=8)
    with pytest.warns(RuntimeWarning, match="two-kernel path"):
        m = DeepHPM(Collo, SRC, IC, np.zeros((0, 3)), net(5, 64), LB, UB, verbose=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                        # once per engine and family: not again
        m.engine.warn_if_slow_path("wave")
        losses = m.train(2, 1e-3, 1)                                          # ... and the depth still trains
        DeepHPM(Collo, SRC, IC, np.zeros((0, 3)), net(8, 64), LB, UB, verbose=False)      # the compiled depth: no warning
    assert np.isfinite(losses[4]).all()


@pytest.mark.parametrize("n", [64, 256, 1024])
def test_small_batch_precision_statement_fresh_weights(dev, n):
    """include/pinn_hip.h, PINN_PREC_F16X3 (2): the narrow four-stream kernel's weight gradient multiplies fp16 high parts of both factors --
    a random rounding noise of 3.5e-4 / sqrt(points) relative to the gradient.  Held here with margin (6e-4 / sqrt(n) + 5e-6) where it is
    largest, on small batches (batch_num > 1 in INF:292-301 makes them); the two-kernel path (both parts of both factors) stays at 2e-5."""
    layers = net(8, 64)
    Ws, bs, rng = make_net(layers, 21)
    flat = po.pack_params(Ws, bs)
    X = po.collocation_points(n, LB, UB, rng)
    tw = np.ones(7) / n
    _, g, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], LB, UB, True, term_weights=tw)
    eng = engine(layers, "f16x3", dev, 1 << 14)
    theta = to_dev(flat, dev)
    xs = [to_dev(X[:, k], dev) for k in range(3)]
    eng.lib.path_counts(reset=True)
    _, gf = eng.wave_loss_grad(theta, *xs, LB, UB, True, tw)
    assert eng.lib.path_counts(reset=True)["fused-registers"] == 1
    eng.two_kernel = True
    _, g2 = eng.wave_loss_grad(theta, *xs, LB, UB, True, tw)
    assert eng.lib.path_counts(reset=True)["two-kernel"] == 1
    assert rel(gf.cpu().numpy(), g) < 6e-4 / np.sqrt(n) + 5e-6, (n, rel(gf.cpu().numpy(), g))
    assert rel(g2.cpu().numpy(), g) < 2e-5


@pytest.mark.parametrize("n", [64, 256, 1024])
def test_small_batch_precision_statement_trained_weights(dev, golden_dir, n):
    """... and at the trained 8 x 64 net (cancellation regime), per weight layer and bias: the fused kernel's error against float64 is within
    6x host-fp32's own error or 1.5x the error of the two-kernel path (full two-part operands) at the same points, whichever is larger --
    i.e. rounding the weight gradient's operands does not show next to the chain's own error even on 64 points (numpy restatement of the
    kernel's arithmetic: 4.7 / 2.1 / 1.5 / 4.0 x fp32 with one-part operands against 4.7 / 1.6 / 1.5 / 3.3 with two-part ones, two draws each
    of 64 and 256 points).  No point-count threshold is needed below which the operands would have to be two-part."""
    from oracle import golden_points as gp
    w = np.load(f"{golden_dir}/weights_wave64.npz")
    g32k = np.load(f"{golden_dir}/golden_wave64_32k.npz")
    layers = [int(v) for v in w["layers"]]
    L = len(layers) - 1
    flat = po.pack_params([w[f"W{i}"] for i in range(L)], [w[f"b{i}"] for i in range(L)])
    lb, ub, norm = g32k["lb"], g32k["ub"], bool(g32k["normalize"])
    Xall = gp.wave_points(lb, ub, tuple(g32k["src"]), int(g32k["n"]))
    theta = to_dev(flat, dev)
    eng = engine(layers, "f16x3", dev, 1 << 14)
    worst = 0.0
    for off in (0, 5000, 17000):
        X = Xall[off:off + n]
        tw = np.ones(7) / n
        _, g64, _ = po.wave2d_loss_grad(flat, layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, norm, term_weights=tw)
        _, g32, _ = po.wave2d_loss_grad(flat.astype(np.float32), layers, X[:, 0], X[:, 1], X[:, 2], lb, ub, norm, term_weights=tw, dtype=np.float32)
        xs = [to_dev(X[:, k], dev) for k in range(3)]
        eng.two_kernel = False
        _, gf = eng.wave_loss_grad(theta, *xs, lb, ub, norm, tw)
        eng.two_kernel = True
        _, g2 = eng.wave_loss_grad(theta, *xs, lb, ub, norm, tw)
        parts = [po.unpack_params(np.asarray(v, dtype=np.float64), layers) for v in (gf.cpu().numpy(), g2.cpu().numpy(), g32, g64)]
        for l in range(L):
            for k in (0, 1):                                                  # weights, biases
                ef, e2, e32 = (np.linalg.norm(parts[i][k][l] - parts[3][k][l]) for i in range(3))
                bound = max(6.0 * e32, 1.5 * e2) + 1e-6 * np.linalg.norm(parts[3][k][l])
                worst = max(worst, ef / bound)
                assert ef <= bound, (n, off, l, k, ef, e2, e32)
    print(f"n={n}: worst fused error / bound = {worst:.2f}")


def test_neural_net_on_a_net_of_another_output_count(dev):
    """round-4 advisor: neural_net(X, weights, biases) with weights of other layer sizes goes through the model's own value-stream call; a net
    with another OUTPUT COUNT than the model's (the plate's 4 x 20 distance net has 5) must come out right, and bad arguments say what is wrong"""
    from pinn_elastodynamics_amd.elastic_wave import DeepHPM
    rng = np.random.default_rng(0)
    Collo = po.collocation_points(2000, LB, UB, rng)
    m = DeepHPM(Collo, po.ricker_source_set(n_pt=8, n_time=8), po.ic_grid(num=8), np.zeros((0, 3)), net(4, 32), LB, UB, verbose=False)
    layers = [3, 20, 20, 20, 5]
    Ws, bs, _ = make_net(layers, 5)
    X = Collo[:300]
    Y = m.neural_net(X, [w.astype(np.float32) for w in Ws], [b.astype(np.float32) for b in bs])
    Yo, _, _ = po.mlp_forward_tangent(X, Ws, bs, LB, UB, True, n_tangent=0)
    assert Y.shape == (300, 5) and rel(Y, Yo) < 1e-5
    with pytest.raises(ValueError, match="bias"):
        m.neural_net(X, Ws, None)
    with pytest.raises(ValueError, match="input columns"):
        m.neural_net(X[:, :2], [np.zeros((2, 8)), np.zeros((8, 7))], [np.zeros((1, 8)), np.zeros((1, 7))])
    with pytest.raises(ValueError, match="no kernel variant"):
        m.neural_net(X, [np.zeros((3, 200)), np.zeros((200, 7))], [np.zeros((1, 200)), np.zeros((1, 7))])


def test_step_call_is_the_separate_calls_bit_for_bit_on_the_gpu(dev):
    """pinn_wave2d_step against the three calls it replaces ON THE DEVICE (the emulator tests assert the same on the CPU): five Adam steps of the
    8 x 64 model with and without the one-launch step -- parameters, both Adam moments and the recorded loss sums identical bit for bit."""
    from pinn_elastodynamics_amd.elastic_wave import DeepHPM
    rng = np.random.default_rng(1)
    Collo = po.collocation_points(30001, LB, UB, rng)                      # 469 steps over 256 workgroups: two steps on most of them
    SRC, IC = po.ricker_source_set(n_pt=40, n_time=31), po.ic_grid(num=41)
    out = {}
    for step_call in (True, False):
        m = DeepHPM(Collo, SRC, IC, np.zeros((0, 3)), net(8, 64), LB, UB, verbose=False, seed=9, step_call=step_call)
        m.engine.lib.path_counts(reset=True)
        losses = m.train(5, 1e-3, 1)
        out[step_call] = (m.theta.cpu().numpy(), m.adam_m.cpu().numpy(), m.adam_v.cpu().numpy(), np.array(losses), m.engine.lib.path_counts(reset=True))
    for a, b in zip(out[True][:4], out[False][:4]):
        assert np.array_equal(a, b)
    assert out[True][4]["fused-registers"] == out[False][4]["fused-registers"] and out[True][4]["two-kernel"] == 0


def test_plate_step_call_is_the_separate_calls_bit_for_bit_on_the_gpu(dev):
    """pinn_plate2d_step (five-stream collocation set + hole-traction set in one launch, one reduction) against pinn_plate2d_loss_grad +
    pinn_plate2d_traction_loss_grad on the device: gradient and all loss sums identical bit for bit"""
    layers = net(8, 64, 5)
    n, nh = 50000, 3000
    Ws, bs, rng = make_net(layers, 6)
    theta = to_dev(po.pack_params(Ws, bs), dev)
    X = po.collocation_points(n, [0, 0, 0], [0.5, 0.5, 10.0], rng)
    H = po.collocation_points(nh, [0, 0, 0], [0.1, 0.1, 10.0], rng)
    xs = [to_dev(X[:, k], dev) for k in range(3)]
    hs = [to_dev(H[:, k], dev) for k in range(3)]
    frozen = to_dev(0.3 * rng.standard_normal((2, 5, 5, n)), dev)
    aux = to_dev(0.3 * rng.standard_normal((12, nh)), dev)
    eng = engine(layers, "f16x3", dev, 1 << 16)
    lb, ub = [0, 0, 0], [0.5, 0.5, 10.0]
    tw, hw = [10.0 / n] * 5, [10.0 / nh] * 2
    g1, l1, h1 = (torch.full((eng.n_params,), float("nan"), device=dev), torch.zeros(8, device=dev), torch.zeros(8, device=dev))
    eng.plate_step(theta, *xs, lb, ub, False, frozen, tw, (*hs, aux, hw), g1, l1, h1)
    g2, l2, h2 = (torch.full((eng.n_params,), float("nan"), device=dev), torch.zeros(8, device=dev), torch.zeros(8, device=dev))
    eng.plate_loss_grad(theta, *xs, lb, ub, False, frozen, tw, grad_out=g2, accumulate=False, loss_out=l2)
    eng.traction_loss_grad(theta, *hs, lb, ub, False, aux, hw, grad_out=g2, accumulate=True, loss_out=h2, packed=True)
    assert torch.isfinite(g1).all() and torch.equal(g1, g2) and torch.equal(l1[:5], l2[:5]) and torch.equal(h1[:2], h2[:2])


def test_xcd_aware_step_assignment_changes_the_grouping_not_the_sums(dev):
    """The even-XCD workgroups take the launch's tail (FusedArgs::n_plain; 1.6 % more steps for them on full grids of >= 64 rounds): every step is
    still taken exactly once -- loss sums and gradient agree with the unskewed assignment to fp32 summation noise, at a size where the tail exists
    (1.2 M points: 18 750 steps) and for a skew far larger than the shipped one; repeated calls give the same bits (static assignment)."""
    layers = net(8, 64)
    n = 1_200_000
    Ws, bs, rng = make_net(layers, 2)
    theta = to_dev(po.pack_params(Ws, bs), dev)
    X = po.collocation_points(n, LB, UB, rng)
    xs = [to_dev(X[:, k], dev) for k in range(3)]
    eng = engine(layers, "f16x3", dev, 1 << 18)
    out = {}
    try:
        for permille in (0, 16, 150):
            eng.lib.lib.pinn_debug_set_xcd_bonus(permille)
            l1, g1 = eng.wave_loss_grad(theta, *xs, LB, UB, True, np.ones(7) / n)
            l1, g1 = l1.cpu().numpy().copy(), g1.cpu().numpy().copy()
            l2, g2 = eng.wave_loss_grad(theta, *xs, LB, UB, True, np.ones(7) / n)
            assert np.array_equal(g1, g2.cpu().numpy()) and np.array_equal(l1, l2.cpu().numpy())
            out[permille] = (l1, g1)
    finally:
        eng.lib.lib.pinn_debug_set_xcd_bonus(16)
    for permille in (16, 150):
        assert rel(out[permille][0], out[0][0]) < 2e-6 and rel(out[permille][1], out[0][1]) < 2e-6
        assert not np.array_equal(out[permille][1], out[0][1])              # another grouping of the partial sums did run
