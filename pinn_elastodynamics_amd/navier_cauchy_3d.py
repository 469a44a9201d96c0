"""3-D Navier-Cauchy half-space model (BASELINE.json configs[4]) -- a BUILD-SIDE EXTENSION, not in the reference.

The reference's four scripts are 2-D + time.  This class carries their model-class surface
(``class DeepHPM``, ElasticWaveSemiInfinite/ElasticWave.py = SEMI) over to four inputs (x, y, z, t) and twelve outputs
(u, v, w, ut, vt, wt, s11, s22, s33, s12, s13, s23): same method names and column-tensor convention, the mixed-variable
formulation of SEMI:228-272 written for three dimensions (stated term by term in include/pinn_hip.h and DESIGN.md), the loss layout of
SEMI:112-127 with the traction-free top surface as the ``loss_NB`` term.  Kernels: pinn_nc3d_* of include/pinn_hip.h.
Parity for this case is unpinned by definition (there is nothing in the reference to compare with).

Data parallel exactly like the 2-D classes: every point set is sharded into contiguous row ranges (a rank uploads only its own
rows), partial sums carry the global 1/N, one all-reduce of [gradient | loss sums], identical on-device Adam on every rank.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .elastic_wave import _col, all_reduce_sum, evaluate_with_finite_gradient, pack_params, unpack_params, xavier_init  # noqa: F401
from .net_api import NetApi, read_checkpoint, write_checkpoint

OUT = ("u", "v", "w", "ut", "vt", "wt", "s11", "s22", "s33", "s12", "s13", "s23")
_SLOTS = ("collo", "IC", "SRC", "NB")          # 16 floats each in the loss-sum buffer
LOSS_LAYOUT_3D = dict(f_uv=5.0, f_s=5.0, IC=2.0, SRC=2.0, NB=2.0)        # the semi-infinite script's weights (SEMI:127)


class NavierCauchy3D(NetApi):
    """NavierCauchy3D(Collo[N,4], SRC[Ns,7], IC[Ni,4], TOP[Nt,4], uv_layers, lb[4], ub[4], ExistModel=0, modelDir='')

    Collo: collocation points (x, y, z, t); SRC: points with prescribed displacement (x, y, z, t, u, v, w) -- the source;
    IC: points at t = 0 where u, v, w, ut, vt, wt vanish; TOP: points on the free surface z = ub[2] where s33 = s13 = s23 = 0.
    uv_layers = [4] + depth * [width] + [12]."""

    def __init__(self, Collo, SRC, IC, TOP, uv_layers, lb, ub, ExistModel=0, modelDir='', *, precision="f16x3", engine=None, seed=1111,
                 process_group=None, verbose=True, E=2.5, mu=0.25, rho=1.0, normalize=True, layout: Optional[dict] = None,
                 always_reduce=False, collective="rccl", p2p_timeout_s=None):
        self.count = 0
        self._shift_state = {}
        self.loss_rec = []
        self.layout = dict(LOSS_LAYOUT_3D if layout is None else layout)
        self.normalize = bool(normalize)
        self.lb = np.asarray(lb, dtype=np.float64).reshape(-1)
        self.ub = np.asarray(ub, dtype=np.float64).reshape(-1)
        assert self.lb.size == 4 and self.ub.size == 4, "lb / ub are (x, y, z, t) bounds"
        self.E, self.mu, self.rho = E, mu, rho
        self.uv_layers = [int(v) for v in uv_layers]
        assert self.uv_layers[0] == 4 and self.uv_layers[-1] == 12, "uv_layers = [4] + depth*[width] + [12]"
        self.verbose = verbose
        self.pg = process_group
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank = torch.distributed.get_rank(self.pg)
            self.world = torch.distributed.get_world_size(self.pg)
        else:
            self.rank, self.world = 0, 1
        # (always_reduce: the collective branch also at one rank -- see elastic_wave.DeepHPM)
        self._reduce = self.world > 1 or (bool(always_reduce) and torch.distributed.is_available() and torch.distributed.is_initialized())
        if engine is None:
            from .hip_engine import HipEngine
            n_max = max(int(np.asarray(Collo).shape[0]) // self.world + 1, 1 << 14)
            engine = HipEngine(self.uv_layers, precision=precision, max_points=n_max)
        self.engine = engine
        self.device = engine.device
        if hasattr(engine, "warn_if_slow_path"):
            engine.warn_if_slow_path("nc3d")
        self._init_rng = np.random.default_rng(seed)
        if ExistModel == 0:
            W, b = self.initialize_NN(self.uv_layers)
        else:
            W, b = self.load_NN(modelDir, self.uv_layers)
        self.n_params = sum(w.size for w in W) + sum(x.size for x in b)
        self.theta = torch.from_numpy(pack_params(W, b)).to(self.device)
        self._shift_state["theta"] = self.theta
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_t = 0

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

        Collo = np.asarray(Collo, dtype=np.float64)
        self.x_c, self.y_c, self.z_c, self.t_c = (Collo[:, k:k + 1] for k in range(4))
        self._n_collo = Collo.shape[0]
        self._collo_host = tuple(np.ascontiguousarray(_col(Collo[:, k]), dtype=np.float32) for k in range(4))
        self._collo_full = tuple(dev(a) for a in self._collo_host) if self.world == 1 else None
        self._collo_cache = {}
        self._sides = {}

        def side(name, A, cols, target_cols=None):
            if A is None:
                return
            A = np.asarray(A, dtype=np.float64)
            if A.shape[0] == 0:
                return
            n_glob = A.shape[0]
            s, e = self._shard(0, n_glob)
            A = A[s:e]
            tg = None
            if target_cols is not None and e > s:
                T = np.zeros((12, A.shape[0]), dtype=np.float32)
                for o, cidx in zip(cols, target_cols):
                    T[o] = A[:, cidx]
                tg = dev(T)
            self._sides[name] = tuple(dev(_col(A[:, k])) for k in range(4)) + (tg, tuple(cols), n_glob)

        side("IC", IC, (0, 1, 2, 3, 4, 5))                 # u, v, w, ut, vt, wt = 0 at t = 0      (as SEMI:119-122)
        side("SRC", SRC, (0, 1, 2), (4, 5, 6))             # prescribed source displacement       (as SEMI:123-124)
        side("NB", TOP, (8, 10, 11))                       # s33 = s13 = s23 = 0 on the free top  (as SEMI:125-126)
        self.SRC, self.IC, self.TOP = SRC, IC, TOP
        self._buf = torch.zeros(self.n_params + 16 * len(_SLOTS), dtype=torch.float32, device=self.device)
        # collective: "rccl" (torch.distributed.all_reduce) or "p2p" (the library's one-shot all-reduce over IPC-mapped peer buffers; elastic_wave.DeepHPM)
        if collective not in ("rccl", "p2p"):
            raise ValueError("collective must be 'rccl' or 'p2p'")
        self._p2p = None
        if collective == "p2p" and self._reduce:
            if not hasattr(self.engine, "lib"):
                raise ValueError("collective='p2p' needs the HIP engine")
            from .p2p import P2PAllReduce
            self._p2p = P2PAllReduce(self.engine.lib, self._buf.numel(), self.pg, timeout_s=p2p_timeout_s)

    def _check_collective(self):
        """collective='p2p': raise if a one-shot all-reduce has failed on this rank (elastic_wave.DeepHPM._check_collective)"""
        if getattr(self, "_p2p", None) is not None:
            self._p2p.check()

    def close(self):
        """Release the P2P communicator (collective: every rank calls it).  Nothing to do for collective='rccl'."""
        p2p, self._p2p = getattr(self, "_p2p", None), None
        if p2p is not None:
            p2p.close()

    # ---- checkpoints: the reference's [W_list, b_list] pickle / npz (INF:159-186) ---------------------------------
    def save_NN(self, fileDir, TYPE=''):
        W, b = unpack_params(self.theta.detach().cpu().numpy(), self.uv_layers)
        write_checkpoint(fileDir, W, b, self.uv_layers)        # .npz: plain arrays (documented default); otherwise the reference's pickle

    def load_NN(self, fileDir, layers):
        W, b = read_checkpoint(fileDir)                        # .npz, or the reference's pickle through the arrays-only unpickler
        assert len(layers) == len(W) + 1                   # INF:178
        W = [np.asarray(w, dtype=np.float32) for w in W]
        b = [np.asarray(x, dtype=np.float32).reshape(1, -1) for x in b]
        for i, w in enumerate(W):
            assert w.shape == (layers[i], layers[i + 1]), "stored weight shape does not match uv_layers"
        return W, b

    # ---- graph pieces on column arrays [N,1] ---------------------------------------------------------------------------
    def _fields(self, x, y, z, t):
        xs = [torch.from_numpy(np.ascontiguousarray(_col(a), dtype=np.float32)).to(self.device) for a in (x, y, z, t)]
        return self.engine.nc3d_fields(self.theta, *xs, self.lb, self.ub, self.normalize)        # [5, 12, N]

    @staticmethod
    def _cols(T):
        return tuple(T[i].detach().cpu().numpy().reshape(-1, 1) for i in range(T.shape[0]))

    # neural_net(X, weights, biases): net_api.NetApi, through these two
    def _current_theta(self):
        return self.theta

    def _net_fields(self, eng, theta, X):
        xs = [torch.from_numpy(np.ascontiguousarray(_col(X[:, k]), dtype=np.float32)).to(eng.device) for k in range(4)]
        return eng.nc3d_fields(theta, *xs, self.lb, self.ub, self.normalize)[0]

    def net_uv(self, x, y, z, t):
        """-> (u, v, w, ut, vt, wt, s11, s22, s33, s12, s13, s23), each [N,1]"""
        return self._cols(self._fields(x, y, z, t)[0])

    def net_e(self, x, y, z, t):
        """-> (e11, e22, e33, e12, e13, e23), engineering shear as INF:216-218"""
        F = self._fields(x, y, z, t)
        X, Y, Z = F[1], F[2], F[3]
        return self._cols(torch.stack([X[0], Y[1], Z[2], Y[0] + X[1], Z[0] + X[2], Z[1] + Y[2]]))

    def net_f_sig(self, x, y, z, t):
        """-> the twelve residuals (f_u, f_v, f_w, f_ut, f_vt, f_wt, f_s11, f_s22, f_s33, f_s12, f_s13, f_s23)"""
        F = self._fields(x, y, z, t)
        V, X, Y, Z, T = F[0], F[1], F[2], F[3], F[4]
        E, mu, rho = self.E, self.mu, self.rho
        coef = E / ((1 + mu) * (1 - 2 * mu))
        c1, c2, G = coef * (1 - mu), coef * mu, E / (2 * (1 + mu))
        e11, e22, e33 = X[0], Y[1], Z[2]
        e12, e13, e23 = Y[0] + X[1], Z[0] + X[2], Z[1] + Y[2]
        f = [X[6] + Y[9] + Z[10] - rho * T[3], X[9] + Y[7] + Z[11] - rho * T[4], X[10] + Y[11] + Z[8] - rho * T[5],
             T[0] - V[3], T[1] - V[4], T[2] - V[5],
             V[6] - (c1 * e11 + c2 * (e22 + e33)), V[7] - (c1 * e22 + c2 * (e11 + e33)), V[8] - (c1 * e33 + c2 * (e11 + e22)),
             V[9] - G * e12, V[10] - G * e13, V[11] - G * e23]
        return self._cols(torch.stack(f))

    net_uvp = net_uv
    net_f = net_f_sig

    def callback(self, loss):
        self.count += 1
        self.loss_rec.append(loss)
        if self.verbose and self.rank == 0:
            print('{} th iterations, Loss: {}'.format(self.count, loss))

    # ---- loss + gradient of one collocation block -----------------------------------------------------------------------
    def _shard(self, lo, hi):
        n = hi - lo
        return lo + n * self.rank // self.world, lo + n * (self.rank + 1) // self.world

    def _rows(self, lo, hi):
        s, e = self._shard(lo, hi)
        if self._collo_full is not None:
            return tuple(a[s:e] for a in self._collo_full)
        key = (lo, hi)
        if key in self._collo_cache:
            self._collo_cache[key] = self._collo_cache.pop(key)          # most recently used last
        else:
            # Bounded cache, least recently used out first: TWICE this rank's share of the set stays resident, so that train() walking
            # its blocks and getloss() asking for (0, N) in between -- two batchings of the same rows -- alternate without re-uploading
            # a shard from the host every time, and shifting windows still cannot grow the device footprint beyond that bound.
            self._collo_cache[key] = tuple(torch.from_numpy(h[s:e]).to(self.device) for h in self._collo_host)
            cap = 2 * (-(-self._n_collo // self.world)) + 64
            while len(self._collo_cache) > 1 and sum(v[0].numel() for v in self._collo_cache.values()) > cap:
                del self._collo_cache[next(iter(self._collo_cache))]
        return self._collo_cache[key]

    def _loss_and_grad(self, idx_start, idx_end):
        """self._buf = [grad (P) | 16 floats per slot], this rank's partial sums, then all-reduced."""
        P, lay, eng, buf = self.n_params, self.layout, self.engine, self._buf
        sums = buf[P:]
        sums.zero_()
        grad = buf[:P]
        n_blk = idx_end - idx_start
        s, e = self._shard(idx_start, idx_end)
        tw = [lay["f_uv"] / n_blk] * 6 + [lay["f_s"] / n_blk] * 6
        wrote = False
        if e > s:
            x, y, z, t = self._rows(idx_start, idx_end)
            eng.nc3d_loss_grad(self.theta, x, y, z, t, self.lb, self.ub, self.normalize, tw, self.E, self.mu, self.rho,
                               grad_out=grad, accumulate=False, loss_out=sums[0:16])
            wrote = True
        for k, name in enumerate(_SLOTS[1:], start=1):
            if name not in self._sides or lay[name] == 0.0:
                continue
            x, y, z, t, tg, cols, n = self._sides[name]
            if x.numel() == 0:
                continue
            ow = [0.0] * 12
            for o in cols:
                ow[o] = lay[name] / n
            eng.nc3d_data_loss_grad(self.theta, x, y, z, t, self.lb, self.ub, self.normalize, tg, ow, grad_out=grad, accumulate=wrote,
                                    loss_out=sums[16 * k:16 * k + 16], packed=wrote)
            wrote = True
        if not wrote:
            grad.zero_()
        if self._reduce:
            all_reduce_sum(buf, self.pg, getattr(self, "collective_events", None), getattr(self, "_p2p", None))

    def _terms_from_sums(self, sums, n_blk):
        lay = self.layout
        out = {"loss_f_uv": float(sums[0, :6].sum() / n_blk), "loss_f_s": float(sums[0, 6:12].sum() / n_blk)}
        for k, name in enumerate(_SLOTS[1:], start=1):
            if name in self._sides:
                cols, n = list(self._sides[name][5]), self._sides[name][6]
                out["loss_" + name] = float(sums[k, cols].sum() / n)
            else:
                out["loss_" + name] = 0.0
        out["loss"] = (lay["f_uv"] * out["loss_f_uv"] + lay["f_s"] * out["loss_f_s"] + lay["IC"] * out["loss_IC"]
                       + lay["SRC"] * out["loss_SRC"] + lay["NB"] * out["loss_NB"])
        return out

    # ---- training drivers ---------------------------------------------------------------------------------------------
    def train(self, iter, learning_rate, batch_num):
        """Adam loop with the reference's block-sequential batching (SEMI:290-328): returns the per-step lists
        (loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss) -- the loss each step's gradient was taken at."""
        P, L = self.n_params, len(_SLOTS)
        hist = ([], [], [], [], [])
        for i in range(batch_num):
            lo, hi = int(i * self._n_collo / batch_num), int((i + 1) * self._n_collo / batch_num)
            rec = torch.zeros((iter, 16 * L), dtype=torch.float32, device=self.device)

            def probe():
                self._loss_and_grad(lo, hi)
                return self._buf

            if iter > 0 and getattr(self.engine, "needs_finite_probe", False) and not self._shift_state.get("probed"):
                evaluate_with_finite_gradient(self.engine, probe, P, self._shift_state, check=self._check_collective)
                self._shift_state["probed"] = True
            for it in range(iter):
                self._loss_and_grad(lo, hi)
                rec[it].copy_(self._buf[P:])
                self.adam_t += 1
                self.engine.adam_step(self.theta, self.adam_m, self.adam_v, self._buf[:P], learning_rate, self.adam_t)
            if iter > 0 and not bool(torch.isfinite(self.theta).all()):
                raise FloatingPointError("parameters became non-finite during train(): lower the learning rate or raise engine.adjoint_shift")
            sums = rec.detach().cpu().numpy().reshape(iter, L, 16)
            self._check_collective()                 # (behind the block's one host synchronisation)
            for it in range(iter):
                tm = self._terms_from_sums(sums[it], hi - lo)
                for lst, key in zip(hist, ("loss_f_uv", "loss_f_s", "loss_IC", "loss_SRC", "loss")):
                    lst.append(tm[key])
        return hist

    def train_bfgs(self, batch_num, options: Optional[dict] = None):
        """host L-BFGS-B stage (scipy, float64 flat vector) on the device kernels, as SEMI:330-344"""
        import scipy.optimize
        P, L = self.n_params, len(_SLOTS)
        opts = dict(maxiter=1000, maxfun=1000, maxcor=50, maxls=50, ftol=0.001 * float(np.finfo(float).eps))       # SEMI:130-137
        if options:
            opts.update(options)
        result = None
        for i in range(batch_num):
            lo, hi = int(i * self._n_collo / batch_num), int((i + 1) * self._n_collo / batch_num)

            def evaluate():
                self._loss_and_grad(lo, hi)
                return self._buf

            def fun(theta64):
                self.theta.copy_(torch.from_numpy(theta64.astype(np.float32)).to(self.device))
                host = evaluate_with_finite_gradient(self.engine, evaluate, P, self._shift_state, check=self._check_collective)
                tm = self._terms_from_sums(host[P:].reshape(L, 16), hi - lo)
                self.callback(tm["loss"])
                return tm["loss"], host[:P].astype(np.float64)

            result = scipy.optimize.minimize(fun, self.theta.detach().cpu().numpy().astype(np.float64), jac=True, method='L-BFGS-B', options=opts)
            self.theta.copy_(torch.from_numpy(result.x.astype(np.float32)).to(self.device))
        return result

    # ---- inference / diagnostics ---------------------------------------------------------------------------------------
    def predict(self, x_star, y_star, z_star, t_star):
        """-> (u, v, w, s11, s22, s33, s12, s13, s23, e11, e22, e33, e12, e13, e23), each numpy [M,1] (the 3-D form of INF:337-347)"""
        F = self._fields(x_star, y_star, z_star, t_star)
        V, X, Y, Z = F[0], F[1], F[2], F[3]
        return self._cols(torch.stack([V[0], V[1], V[2], V[6], V[7], V[8], V[9], V[10], V[11],
                                       X[0], Y[1], Z[2], Y[0] + X[1], Z[0] + X[2], Z[1] + Y[2]]))

    probe = predict

    def getloss(self):
        """(loss, loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss_NB) on the full sets (as SEMI:370-385)"""
        self._loss_and_grad(0, self._n_collo)
        tm = self._terms_from_sums(self._buf[self.n_params:].detach().cpu().numpy().reshape(len(_SLOTS), 16), self._n_collo)
        self._check_collective()
        return tm["loss"], tm["loss_f_uv"], tm["loss_f_s"], tm["loss_IC"], tm["loss_SRC"], tm["loss_NB"]


def halfspace_case(n_collo=100_000, n_ic=8000, n_top=8000, n_src=(60, 60), seed=1111, width=128, depth=10,
                   lb=(0.0, 0.0, -30.0, 0.0), ub=(30.0, 30.0, 0.0, 15.0), src_center=(15.0, 15.0, -8.0), src_r=2.0):
    """Seeded point sets of a half space z <= 0 with a buried spherical source driven by a Ricker pulse (the 3-D analogue of
    SEMI:667-769): collocation points in the box minus the source sphere, an initial-state set at t = 0, a traction-free set on
    z = 0, and the source surface at n_src[1] times.  Returns a dict with Collo, IC, TOP, SRC, uv_layers, lb, ub."""
    rng = np.random.default_rng(seed)
    lb = np.asarray(lb, dtype=np.float64)
    ub = np.asarray(ub, dtype=np.float64)
    c = np.asarray(src_center, dtype=np.float64)

    def lhs(n, d):
        return np.stack([(rng.permutation(n) + rng.random(n)) / n for _ in range(d)], axis=1)

    out = np.zeros((0, 4))
    while out.shape[0] < n_collo:
        P = lb + (ub - lb) * lhs(int((n_collo - out.shape[0]) * 1.05) + 64, 4)
        out = np.concatenate([out, P[((P[:, :3] - c) ** 2).sum(1) > src_r ** 2]], 0)
    Collo = out[:n_collo]
    IC = lb + (ub - lb) * lhs(n_ic, 4)
    IC[:, 3] = 0.0
    IC = IC[((IC[:, :3] - c) ** 2).sum(1) > src_r ** 2]
    TOP = lb + (ub - lb) * lhs(n_top, 4)
    TOP[:, 2] = ub[2]
    n_pt, n_time = n_src
    # quasi-uniform points on the sphere (Fibonacci lattice), Ricker amplitude as INF:691-704, radial displacement
    k = np.arange(n_pt) + 0.5
    phi, th = np.arccos(1 - 2 * k / n_pt), np.pi * (1 + 5 ** 0.5) * k
    nrm = np.stack([np.cos(th) * np.sin(phi), np.sin(th) * np.sin(phi), np.cos(phi)], 1)
    tt = np.linspace(0.0, ub[3], n_time + 1)[1:]
    amp = (2 * np.pi ** 2 * (tt - 3.0) ** 2 / 9.0 - 1) * np.exp(-np.pi ** 2 * (tt - 3.0) ** 2 / 9.0)
    pts = np.repeat((c + src_r * nrm)[None], n_time, 0).reshape(-1, 3)
    tcol = np.repeat(tt, n_pt)
    disp = (amp[:, None, None] * nrm[None]).reshape(-1, 3)
    SRC = np.concatenate([pts, tcol[:, None], disp], 1)
    return dict(Collo=Collo, IC=IC, TOP=TOP, SRC=SRC, uv_layers=[4] + depth * [width] + [12], lb=lb, ub=ub, source=(c, src_r))
