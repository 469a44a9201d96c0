"""Host-side mirror of the reference's plate-with-hole model class (PLATE = PlateHoleQuarter/train/train.py).

``PINN(Collo, HOLE, IC, LF, RT, UP, LW, DIST, uv_layers, dist_layers, part_layers, lb, ub, partDir, distDir, uvDir)`` keeps
the reference's constructor (PLATE:28-29), method names and the ``[N,1]`` column convention: three nets (uv, distance,
particular), composite fields ``P + D*N`` (PLATE:358-388), the three-stage schedule ``train_bfgs_dist`` ->
``train_bfgs_part`` -> ``train`` / ``train_bfgs`` (PLATE:953-972), ``predict`` / ``predict_D`` / ``predict_P`` / ``getloss``
and ``save_NN(fileDir, TYPE)`` / ``load_NN``.  All loss / gradient work runs in the HIP kernels behind include/pinn_hip.h
(5-stream family: value, d/dx, d/dy, d/dt, d2/dt2); the streams of the frozen distance and particular nets on the
collocation and hole points are computed once and kept on the device, because those nets do not change while the uv net
trains (PLATE:240-241).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .elastic_wave import _col, all_reduce_sum, evaluate_with_finite_gradient, lbfgs_on_device, pack_params, relax_adjoint_shift, unpack_params, xavier_init  # noqa: F401
from .net_api import NetApi, read_checkpoint, write_checkpoint

_EPS = float(np.finfo(float).eps)
BFGS_OPTIONS = {   # PLATE:220-247
    "dist": dict(maxiter=20000, maxfun=20000, maxcor=50, maxls=50, ftol=0.00001 * _EPS),
    "part": dict(maxiter=20000, maxfun=20000, maxcor=50, maxls=50, ftol=0.00001 * _EPS),
    "uv": dict(maxiter=70000, maxfun=70000, maxcor=50, maxls=50, ftol=0.00001 * _EPS),
}


class PINN(NetApi):
    def __init__(self, Collo, HOLE, IC, LF, RT, UP, LW, DIST, uv_layers, dist_layers, part_layers, lb, ub,
                 partDir='', distDir='', uvDir='', *, precision="f16x3", engines=None, seed=1111, process_group=None, verbose=True,
                 always_reduce=False, collective="rccl", p2p_timeout_s=None):
        self.count = 0
        self._shift_state = {}
        self.lb = np.asarray(lb, dtype=np.float64).reshape(-1)
        self.ub = np.asarray(ub, dtype=np.float64).reshape(-1)
        self.E, self.mu, self.rho, self.hole_r = 20.0, 0.25, 1.0, 0.1          # PLATE:39-42
        self.uv_layers = [int(v) for v in uv_layers]
        self.dist_layers = [int(v) for v in dist_layers]
        self.part_layers = [int(v) for v in part_layers]
        self.verbose = verbose
        self.pg = process_group
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank, self.world = torch.distributed.get_rank(self.pg), torch.distributed.get_world_size(self.pg)
        else:
            self.rank, self.world = 0, 1
        # (always_reduce: the collective branch also at one rank -- see elastic_wave.DeepHPM)
        self._reduce = self.world > 1 or (bool(always_reduce) and torch.distributed.is_available() and torch.distributed.is_initialized())

        if engines is None:
            from .hip_engine import HipEngine
            n_max = max(int(np.asarray(Collo).shape[0]) // self.world + 1, 1 << 14)
            engines = {k: HipEngine(l, precision=precision, max_points=n_max)
                       for k, l in (("uv", self.uv_layers), ("dist", self.dist_layers), ("part", self.part_layers))}
        self.eng = engines
        self.device = engines["uv"].device

        self.engine = engines["uv"]            # (net_api.NetApi's default engine; nets of other sizes find theirs in _engine_for)
        if hasattr(self.engine, "warn_if_slow_path"):
            self.engine.warn_if_slow_path("plate")      # (the frozen 4 x 20 nets are evaluated once: their path does not matter)
        self._init_rng = np.random.default_rng(seed)
        self.theta, self.adam_m, self.adam_v = {}, {}, {}
        for key, layers, d in (("dist", self.dist_layers, distDir), ("part", self.part_layers, partDir), ("uv", self.uv_layers, uvDir)):
            W, b = self.initialize_NN(layers) if d == '' else self.load_NN(d, layers)      # PLATE:96-112
            self.theta[key] = torch.from_numpy(pack_params(W, b)).to(self.device)
        self._shift_state["theta"] = self.theta["uv"]
        self.adam_m = torch.zeros_like(self.theta["uv"])
        self.adam_v = torch.zeros_like(self.theta["uv"])
        self.adam_t = 0

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

        def xyt(A, shard=True):
            A = np.asarray(A, dtype=np.float64)
            s, e = self._shard(0, A.shape[0]) if shard else (0, A.shape[0])
            return A[s:e], tuple(dev(_col(A[s:e, k])) for k in range(3))

        Collo = np.asarray(Collo, dtype=np.float64)
        self.x_c, self.y_c, self.t_c = Collo[:, 0:1], Collo[:, 1:2], Collo[:, 2:3]
        self.n_collo = Collo.shape[0]
        _, self._collo = xyt(Collo)
        HOLE = np.asarray(HOLE, dtype=np.float64)
        self.n_hole = HOLE.shape[0]
        Hs, self._hole = xyt(HOLE)
        self._hole_normals = (dev(-Hs[:, 0] / self.hole_r), dev(-Hs[:, 1] / self.hole_r))      # PLATE:457-458
        # pre-training sets: (points, targets [5 streams, 5 outs, n] or None, weight mask [5][5]); replicated, they are small
        self._dist_sets, self._part_sets = [], []
        DIST = np.asarray(DIST, dtype=np.float64)
        tg = np.zeros((5, 5, DIST.shape[0]), dtype=np.float32)
        tg[0] = DIST[:, 3:8].T
        w = np.zeros((5, 5))
        w[0, :] = 1.0
        self._dist_sets.append((xyt(DIST, False)[1], dev(tg), w, DIST.shape[0]))               # PLATE:194-198
        ICa = np.asarray(IC, dtype=np.float64)
        w = np.zeros((5, 5))
        w[3, 0] = w[3, 1] = 1.0
        self._dist_sets.append((xyt(ICa, False)[1], None, w, ICa.shape[0]))                    # PLATE:199-200 (dt_D_u, dt_D_v at IC)
        w = np.zeros((5, 5))
        w[0, :] = 1.0
        w[3, 0] = w[3, 1] = 1.0
        self._part_sets.append((xyt(ICa, False)[1], None, w, ICa.shape[0]))                    # PLATE:201-207
        for A, cols, tcol in ((LF, (0, 4), None), (RT, (2, 4), 3), (LW, (1, 4), None), (UP, (3, 4), None)):   # PLATE:208-215
            A = np.asarray(A, dtype=np.float64)
            w = np.zeros((5, 5))
            w[0, list(cols)] = 1.0
            tgt = None
            if tcol is not None:
                tg = np.zeros((5, 5, A.shape[0]), dtype=np.float32)
                tg[0, 2] = A[:, tcol]                                                          # s11_RT
                tgt = dev(tg)
            self._part_sets.append((xyt(A, False)[1], tgt, w, A.shape[0]))
        self._buf = torch.zeros(self.theta["uv"].numel() + 16, dtype=torch.float32, device=self.device)
        # collective: "rccl" (torch.distributed.all_reduce) or "p2p" (the library's one-shot all-reduce over IPC-mapped peer buffers; elastic_wave.DeepHPM)
        if collective not in ("rccl", "p2p"):
            raise ValueError("collective must be 'rccl' or 'p2p'")
        self._p2p = None
        if collective == "p2p" and self._reduce:
            if not hasattr(self.eng["uv"], "lib"):
                raise ValueError("collective='p2p' needs the HIP engine")
            from .p2p import P2PAllReduce
            self._p2p = P2PAllReduce(self.eng["uv"].lib, self._buf.numel(), self.pg, timeout_s=p2p_timeout_s)
        self.refresh_frozen()

    def _check_collective(self):
        """collective='p2p': raise if a one-shot all-reduce has failed on this rank (elastic_wave.DeepHPM._check_collective)"""
        if getattr(self, "_p2p", None) is not None:
            self._p2p.check()

    def close(self):
        """Release the P2P communicator (collective: every rank calls it).  Nothing to do for collective='rccl'."""
        p2p, self._p2p = getattr(self, "_p2p", None), None
        if p2p is not None:
            p2p.close()

    # ------------------------------------------------------------------------------------------------------------------
    def _shard(self, lo, hi):
        n = hi - lo
        return lo + n * self.rank // self.world, lo + n * (self.rank + 1) // self.world

    def refresh_frozen(self):
        """(Re)compute the streams of the frozen distance / particular nets on the collocation and hole points."""
        x, y, t = self._collo
        if x.numel():
            D = self.eng["dist"].net_streams(self.theta["dist"], x, y, t, self.lb, self.ub, False)
            P = self.eng["part"].net_streams(self.theta["part"], x, y, t, self.lb, self.ub, False)
            self._frozen_collo = torch.stack([D, P]).contiguous()
        x, y, t = self._hole
        if x.numel():
            D0 = self.eng["dist"].net_streams(self.theta["dist"], x, y, t, self.lb, self.ub, False)[0]
            P0 = self.eng["part"].net_streams(self.theta["part"], x, y, t, self.lb, self.ub, False)[0]
            self._aux_hole = torch.cat([D0, P0, self._hole_normals[0][None], self._hole_normals[1][None]]).contiguous()

    # ---- checkpoints (PLATE:276-306) ----------------------------------------------------------------------------------
    def save_NN(self, fileDir, TYPE=''):
        key = {"UV": "uv", "DIST": "dist", "PART": "part"}.get(TYPE)
        if key is None:
            return
        layers = {"uv": self.uv_layers, "dist": self.dist_layers, "part": self.part_layers}[key]
        W, b = unpack_params(self.theta[key].detach().cpu().numpy(), layers)
        write_checkpoint(fileDir, W, b, layers)          # .npz: plain arrays (documented default); otherwise the reference's pickle
        if self.verbose:
            print("Save " + TYPE + " NN parameters successfully...")

    def load_NN(self, fileDir, layers):
        uv_weights, uv_biases = read_checkpoint(fileDir)       # .npz, or the reference's pickle through the arrays-only unpickler
        assert len(layers) == (len(uv_weights) + 1)                                           # PLATE:299
        return ([np.asarray(w, dtype=np.float32) for w in uv_weights],
                [np.asarray(b, dtype=np.float32).reshape(1, -1) for b in uv_biases])

    # ---- neural_net(X, weights, biases) (PLATE:308-320; PLATE:322-356 runs all three nets through it): net_api.NetApi ------------
    def _current_theta(self):
        return self.theta["uv"]

    def _engine_for(self, layers):
        layers = [int(v) for v in layers]
        for eng in self.eng.values():
            if [int(v) for v in eng.layers] == layers:
                return eng
        return super()._engine_for(layers)

    def _net_fields(self, eng, theta, X):
        xs = [torch.from_numpy(np.ascontiguousarray(_col(X[:, k]), dtype=np.float32)).to(eng.device) for k in range(3)]
        return eng.net_streams(theta, xs[0], xs[1], xs[2], self.lb, self.ub, False)[0]

    # ---- graph pieces on column arrays --------------------------------------------------------------------------------
    def _streams(self, key, x, y, t):
        xs = [torch.from_numpy(np.ascontiguousarray(_col(a), dtype=np.float32)).to(self.device) for a in (x, y, t)]
        return self.eng[key].net_streams(self.theta[key], xs[0], xs[1], xs[2], self.lb, self.ub, False)

    @staticmethod
    def _cols(T):
        return tuple(T[i].detach().cpu().numpy().reshape(-1, 1) for i in range(T.shape[0]))

    def _composite(self, x, y, t):
        N, D, P = (self._streams(k, x, y, t) for k in ("uv", "dist", "part"))
        F = torch.empty_like(N)
        F[0] = P[0] + D[0] * N[0]
        for k in (1, 2, 3):
            F[k] = P[k] + D[k] * N[0] + D[0] * N[k]
        F[4] = P[4] + D[4] * N[0] + 2.0 * D[3] * N[3] + D[0] * N[4]
        return F

    def net_dist(self, x, y, t):                     # PLATE:322-329
        return self._cols(self._streams("dist", x, y, t)[0])

    def net_dist_dt(self, x, y, t):                  # PLATE:331-345
        return self._cols(self._streams("dist", x, y, t)[3])

    def net_part(self, x, y, t):                     # PLATE:347-356
        S = self._streams("part", x, y, t)
        return self._cols(torch.cat([S[0], S[3, 0:2]]))

    def net_uv(self, x, y, t):                       # PLATE:358-388
        return self._cols(self._composite(x, y, t)[0])

    def net_e(self, x, y, t):                        # PLATE:390-396
        F = self._composite(x, y, t)
        return self._cols(torch.stack([F[1, 0], F[2, 1], F[2, 0] + F[1, 1]]))

    def net_vel(self, x, y, t):                      # PLATE:398-402
        F = self._composite(x, y, t)
        return self._cols(F[3, 0:2])

    def net_f_sig(self, x, y, t):                    # PLATE:404-439
        F = self._composite(x, y, t)
        E, mu, rho = self.E, self.mu, self.rho
        e11, e22, e12 = F[1, 0], F[2, 1], F[2, 0] + F[1, 1]
        sp11 = E / (1 - mu * mu) * e11 + E * mu / (1 - mu * mu) * e22
        sp22 = E * mu / (1 - mu * mu) * e11 + E / (1 - mu * mu) * e22
        sp12 = E / (2 * (1 + mu)) * e12
        f_u = F[1, 2] + F[2, 4] - rho * F[4, 0]
        f_v = F[2, 3] + F[1, 4] - rho * F[4, 1]
        return self._cols(torch.stack([f_u, f_v, F[0, 2] - sp11, F[0, 3] - sp22, F[0, 4] - sp12]))

    def net_surf_var(self, x, y, t, nx, ny):         # PLATE:441-450
        u, v, s11, s22, s12 = self.net_uv(x, y, t)
        return s11 * nx + s12 * ny, s12 * nx + s22 * ny

    def net_t(self, x, y, t):                        # PLATE:452-461
        x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
        return self.net_surf_var(x, y, t, -x / self.hole_r, -y / self.hole_r)

    def callback(self, loss):
        self.count = self.count + 1
        if self.verbose and self.rank == 0:
            print('{} th iterations, Loss: {}'.format(self.count, loss))

    callback_dist = callback
    callback_part = callback

    # ---- main stage: loss = 10 (loss_f_uv + loss_f_s + loss_HOLE) wrt the uv net (PLATE:187-193,217) -----------------------
    def _loss_and_grad(self, adam=None):
        """self._buf = [grad | 8 collocation sums | 8 hole sums], all-reduced.  ``adam = (learning_rate, step)``: where the whole evaluation is one
        library call (engine.plate_step) and no collective stands in between, the Adam update rides in its reduction: returns True then."""
        P = self.theta["uv"].numel()
        buf, eng = self._buf, self.eng["uv"]
        grad = buf[:P]
        x, y, t = self._collo
        hx, hy, ht = self._hole
        if x.numel() and hx.numel() and hasattr(eng, "plate_step"):
            fold = adam is not None and not self._reduce
            eng.plate_step(self.theta["uv"], x, y, t, self.lb, self.ub, False, self._frozen_collo, [10.0 / self.n_collo] * 5,
                           (hx, hy, ht, self._aux_hole, [10.0 / self.n_hole] * 2), grad, buf[P:P + 8], buf[P + 8:P + 16], self.E, self.mu, self.rho,
                           adam=(self.adam_m, self.adam_v, adam[0], adam[1]) if fold else None)
            if self._reduce:
                all_reduce_sum(buf, self.pg, getattr(self, "collective_events", None), getattr(self, "_p2p", None))
            return fold
        buf[P:].zero_()
        wrote = False
        if x.numel():
            tw = [10.0 / self.n_collo] * 5
            eng.plate_loss_grad(self.theta["uv"], x, y, t, self.lb, self.ub, False, self._frozen_collo, tw, self.E, self.mu, self.rho,
                                grad_out=grad, accumulate=False, loss_out=buf[P:P + 8])
            wrote = True
        x, y, t = self._hole
        if x.numel():
            w = [10.0 / self.n_hole] * 2
            eng.traction_loss_grad(self.theta["uv"], x, y, t, self.lb, self.ub, False, self._aux_hole, w,
                                   grad_out=grad, accumulate=wrote, loss_out=buf[P + 8:P + 16], packed=wrote)
            wrote = True
        if not wrote:
            grad.zero_()
        if self._reduce:
            all_reduce_sum(buf, self.pg, getattr(self, "collective_events", None), getattr(self, "_p2p", None))

    def _terms(self, sums):
        out = {"loss_f_uv": float(sums[0:2].sum() / self.n_collo), "loss_f_s": float(sums[2:5].sum() / self.n_collo),
               "loss_HOLE": float(sums[8:10].sum() / self.n_hole)}
        out["loss"] = 10.0 * (out["loss_f_uv"] + out["loss_f_s"] + out["loss_HOLE"])
        return out

    def train(self, iter, learning_rate):
        """Adam loop of PLATE:475-506 (whole collocation set every step).  Returns (loss_f_uv, loss_f_s, loss_HOLE, loss) lists;
        as in elastic_wave.DeepHPM.train the recorded values are those the step's gradient was taken at."""
        P = self.theta["uv"].numel()
        rec = torch.empty((iter, 16), dtype=torch.float32, device=self.device)

        def probe():
            self._loss_and_grad()
            return self._buf

        if iter > 0 and getattr(self.eng["uv"], "needs_finite_probe", False) and not self._shift_state.get("probed"):
            # once per model: settle the adjoint shift (E = 20 makes the plate's early residuals large)
            evaluate_with_finite_gradient(self.eng["uv"], probe, P, self._shift_state, check=self._check_collective)
            self._shift_state["probed"] = True
        for it in range(iter):
            self.adam_t += 1
            updated = self._loss_and_grad(adam=(learning_rate, self.adam_t))
            rec[it].copy_(self._buf[P:])
            if not updated:
                self.eng["uv"].adam_step(self.theta["uv"], self.adam_m, self.adam_v, self._buf[:P], learning_rate, self.adam_t)
            if self.verbose and it % 10 == 0 and self.rank == 0:
                print('It: %d, Loss: %.6e' % (it, self._terms(rec[it].detach().cpu().numpy())["loss"]))
        sums = rec.detach().cpu().numpy()
        self._check_collective()                     # (behind the loop's one host synchronisation)
        tms = [self._terms(s) for s in sums]
        return ([t["loss_f_uv"] for t in tms], [t["loss_f_s"] for t in tms], [t["loss_HOLE"] for t in tms], [t["loss"] for t in tms])

    def _bfgs(self, key, fun, options):
        import scipy.optimize
        x0 = self.theta[key].detach().cpu().numpy().astype(np.float64)
        res = scipy.optimize.minimize(fun, x0, jac=True, method='L-BFGS-B', options=options)
        self.theta[key].copy_(torch.from_numpy(res.x.astype(np.float32)).to(self.device))
        return res

    def train_bfgs(self, options: Optional[dict] = None, backend: str = "scipy"):          # PLATE:508-526
        """backend="torch": the optimizer runs on the device too (elastic_wave.lbfgs_on_device)."""
        P = self.theta["uv"].numel()

        def evaluate():
            self._loss_and_grad()
            return self._buf

        def fun(th):
            self.theta["uv"].copy_(torch.from_numpy(th.astype(np.float32)).to(self.device))
            host = evaluate_with_finite_gradient(self.eng["uv"], evaluate, P, self._shift_state, check=self._check_collective)
            loss = self._terms(host[P:])["loss"]
            relax_adjoint_shift(self.eng["uv"], loss, self._shift_state)
            self.callback(loss)
            return loss, host[:P].astype(np.float64)

        opts = dict(BFGS_OPTIONS["uv"], **(options or {}))
        if backend == "torch":
            def loss_and_grad():
                host = evaluate_with_finite_gradient(self.eng["uv"], evaluate, 0, self._shift_state, device_check=P, check=self._check_collective)
                loss = self._terms(host)["loss"]
                relax_adjoint_shift(self.eng["uv"], loss, self._shift_state)
                return loss, self._buf[:P]
            return lbfgs_on_device(self.theta["uv"], loss_and_grad, opts, self.callback)
        return self._bfgs("uv", fun, opts)

    def _pretrain_loss_grad(self, key, sets):
        """sum over sets of mean-square terms (PLATE:194-215); returns (loss, grad) on the host."""
        eng, th = self.eng[key], self.theta[key]
        grad = torch.zeros_like(th)
        total = torch.zeros((), dtype=torch.float32, device=self.device)
        for (x, y, t), tg, w, n in sets:
            wl = (w / n).tolist()
            ls, _ = eng.stream_loss_grad(th, x, y, t, self.lb, self.ub, False, tg, wl, grad_out=grad, accumulate=True)
            total = total + ls.sum() * float(w.max() / n)           # kernel sums are normalised by max|w|
        return float(total.item()), grad.detach().cpu().numpy().astype(np.float64)

    def _pretrain(self, key, sets, cb, options):
        def fun(th):
            self.theta[key].copy_(torch.from_numpy(th.astype(np.float32)).to(self.device))
            loss, g = self._pretrain_loss_grad(key, sets)
            if not (np.isfinite(loss) and np.isfinite(g).all()):
                # the stream losses have no adjoint-shift retry (pinn_stream_loss_grad evaluates small nets on O(1) targets); a
                # non-finite value here means the optimizer was handed a wild point -- say so instead of feeding NaN to scipy
                raise FloatingPointError(f"pre-training of the {key!r} net produced a non-finite loss or gradient "
                                         f"(loss = {loss}); restart from other weights or use precision='bf16x3'")
            cb(loss)                                                 # fetches = [loss_DIST] / [loss_PART]: the callbacks see the unscaled loss (PLATE:527-559)
            return 1000.0 * loss, 1000.0 * g                         # ScipyOptimizerInterface(1000 * loss_X, ...) PLATE:220,230
        res = self._bfgs(key, fun, dict(BFGS_OPTIONS[key], **(options or {})))
        self.refresh_frozen()
        return res

    def train_bfgs_dist(self, options: Optional[dict] = None):     # PLATE:527-544
        return self._pretrain("dist", self._dist_sets, self.callback_dist, options)

    def train_bfgs_part(self, options: Optional[dict] = None):     # PLATE:546-559
        return self._pretrain("part", self._part_sets, self.callback_part, options)

    # ---- inference / diagnostics ----------------------------------------------------------------------------------------
    def predict(self, x_star, y_star, t_star):       # PLATE:561-570
        F = self._composite(x_star, y_star, t_star)
        return self._cols(torch.stack([F[0, 0], F[0, 1], F[0, 2], F[0, 3], F[0, 4], F[1, 0], F[2, 1], F[2, 0] + F[1, 1]]))

    def predict_D(self, x_star, y_star, t_star):     # PLATE:572-578
        return self.net_dist(x_star, y_star, t_star)

    def predict_P(self, x_star, y_star, t_star):     # PLATE:580-586
        return self._cols(self._streams("part", x_star, y_star, t_star)[0])

    def getloss(self):                               # PLATE:588-612
        P = self.theta["uv"].numel()
        self._loss_and_grad()
        tm = self._terms(self._buf[P:].detach().cpu().numpy())
        self._check_collective()
        tm["loss_PART"] = self._pretrain_loss_grad("part", self._part_sets)[0]
        tm["loss_DIST"] = self._pretrain_loss_grad("dist", self._dist_sets)[0]
        if self.verbose and self.rank == 0:
            for k in ("loss_f_uv", "loss_f_s", "loss_HOLE", "loss", "loss_PART", "loss_DIST"):
                print(k, tm[k])
        return tm
