"""Host-side mirror of the reference's model class for the three elastic-wave cases.

Same names, argument order and column-tensor convention as ``class DeepHPM`` in
  INF  = ElasticWaveInfinite/ElasticWave.py      (float32, inputs normalised, INF:191)
  SEMI = ElasticWaveSemiInfinite/ElasticWave.py  (raw inputs, loss layout SEMI:127)
  CONF = ElasticWaveConfined/ElasticWave.py      (raw inputs, FIXED edges, CONF:156)
but nothing of TF1's runtime: weights live in one flat fp32 device vector, point sets are device
resident (no per-step feed, cf. INF:297-305), the loss + gradient of every term comes from the
HIP kernels behind ``include/pinn_hip.h`` and Adam runs on the device.

Data-parallel training (one process per GPU, torch.distributed): every point set is sharded into
contiguous row ranges (the reference's own batching precedent, INF:292-297); each rank's kernels
weight their partial sums with the GLOBAL 1/N, one all-reduce(sum) of [flat gradient | loss sums]
follows, and the identical fused Adam step runs on every rank, so the weights stay bit-identical.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from .net_api import NetApi, read_checkpoint, truncated_normal, write_checkpoint

# multipliers of (loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss_NB, loss_FIX) in the total loss
LOSS_LAYOUT = {
    "infinite": dict(f_uv=1.0, f_s=1.0, IC=1.0, SRC=1.0, NB=0.0, FIX=0.0),        # INF:119 (loss_NB excluded)
    "semi_infinite": dict(f_uv=5.0, f_s=5.0, IC=2.0, SRC=2.0, NB=2.0, FIX=0.0),   # SEMI:127
    "confined": dict(f_uv=5.0, f_s=5.0, IC=1.0, SRC=1.0, NB=0.0, FIX=1.0),        # CONF:156
}
# scipy L-BFGS-B options of tf.contrib.opt.ScipyOptimizerInterface (INF:122-129, SEMI:130-137, CONF:159-166)
_EPS = float(np.finfo(float).eps)
BFGS_OPTIONS = {
    "infinite": dict(maxiter=10000, maxfun=10000, maxcor=50, maxls=50, ftol=0.001 * _EPS),
    "semi_infinite": dict(maxiter=1000, maxfun=1000, maxcor=50, maxls=50, ftol=0.001 * _EPS),
    "confined": dict(maxiter=100000, maxfun=100000, maxcor=50, maxls=50, ftol=1.0 * _EPS),
}
_SLOTS = ("collo", "IC", "SRC", "NB", "FIX")   # 8 floats each in the loss-sum buffer


def _col(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


def xavier_init(layers: Sequence[int], rng: np.random.Generator):
    """initialize_NN / xavier_init (INF:141-156) as one function of a layer list and a generator: truncated normal (|z| <= 2) times
    sqrt(2/(in+out)), zero biases of shape [1, out].  The build's own seeded stream (TF1's is
    not reproducible).  The model classes also carry the reference's two METHODS (net_api.NetApi)."""
    Ws, bs = [], []
    for i in range(len(layers) - 1):
        n_in, n_out = layers[i], layers[i + 1]
        Ws.append(truncated_normal(rng, (n_in, n_out), float(np.sqrt(2.0 / (n_in + n_out)))))
        bs.append(np.zeros((1, n_out), dtype=np.float32))
    return Ws, bs


def pack_params(weights, biases) -> np.ndarray:
    """[W_list, b_list] -> flat fp32 vector W0,b0,W1,b1,... (the C-ABI layout)."""
    parts = []
    for W, b in zip(weights, biases):
        parts += [np.asarray(W, dtype=np.float32).reshape(-1), np.asarray(b, dtype=np.float32).reshape(-1)]
    return np.concatenate(parts)


def unpack_params(flat, layers):
    Ws, bs, o = [], [], 0
    flat = np.asarray(flat)
    for i in range(len(layers) - 1):
        n_in, n_out = layers[i], layers[i + 1]
        Ws.append(flat[o:o + n_in * n_out].reshape(n_in, n_out).copy())
        o += n_in * n_out
        bs.append(flat[o:o + n_out].reshape(1, n_out).copy())
        o += n_out
    assert o == flat.size
    return Ws, bs


def evaluate_with_finite_gradient(engine, evaluate, n_params, state, device_check=0, check=None):
    """Run ``evaluate()`` (it fills and returns the device buffer [grad | sums]) and bring the result to the host.  The 16-bit
    reverse pass can overflow when residuals are orders of magnitude above their trained size (include/pinn_hip.h,
    PINN_ADJOINT_SHIFT); the sums of squares are still right then, only the gradient is non-finite.  In that case the adjoint
    shift of the engine is raised by 4 (x1/16) and the evaluation repeated; once the loss has fallen 256-fold below where the shift
    was raised, it is lowered again.  ``state`` is a dict the caller keeps between evaluations.  ``device_check=P``: test the first P
    entries on the device and return only the sums behind them.  Deterministic in every rank of a
    data-parallel job (all ranks see the same reduced buffer).  ``check``: called behind the host synchronisation of every evaluation,
    BEFORE the result is looked at -- the model classes pass the status check of their collective (a failed P2P all-reduce leaves NaN in the
    buffer: that must raise as a collective failure, not climb this ladder)."""
    for _ in range(7):
        buf = evaluate().detach()
        if device_check:         # gradient stays on the device: check it there, bring only the loss sums back
            finite = bool(torch.isfinite(buf[:device_check]).all())
            if check is not None:
                check()
            if finite:
                return buf[device_check:].cpu().numpy()
        else:
            host = buf.cpu().numpy()
            if check is not None:
                check()
            if np.isfinite(host[:n_params]).all():
                return host
        # a weight beyond the fused kernels' format (|w| <= 2047) poisons the whole result with NaN: leave the fused path and repeat
        leave = getattr(engine, "leave_fused_path_if_weights_out_of_range", None)
        if leave is not None and state.get("theta") is not None and leave(state["theta"]):
            continue
        if engine.adjoint_shift >= 24:
            break
        engine.adjoint_shift = min(engine.adjoint_shift + 4, 24)
        state["raised_at"] = None                    # filled with the loss of the next successful evaluation
    raise FloatingPointError("non-finite gradient even with the reverse pass scaled by 2^-24")


def relax_adjoint_shift(engine, loss, state):
    """Companion of evaluate_with_finite_gradient: call with the loss of every successful evaluation."""
    if engine.adjoint_shift == 0:
        return
    if state.get("raised_at") is None:
        state["raised_at"] = loss
    elif loss < state["raised_at"] / 256.0:
        engine.adjoint_shift -= 4
        state["raised_at"] = loss


def all_reduce_sum(buf, group, events=None, p2p=None, adam=None, n_params=0):
    """The step's one collective: all-reduce(sum) of the fused buffer [gradient | loss sums], enqueued in stream order behind the kernels that
    filled it.  Default: torch.distributed.all_reduce (RCCL under the "nccl" backend).  ``p2p`` (a p2p.P2PAllReduce): the library's one-shot
    kernel over IPC-mapped peer buffers instead; with ``adam = (params, m, v, lr, step)`` it also applies the Adam update to the first
    ``n_params`` entries of the sum -- returns True then.  ``events``: a list that receives one (start, end) pair of timing events per call,
    recorded on the current stream around the collective (with RCCL the stream waits for the communicator's own stream, so the pair brackets
    it) -- bench.py's ``allreduce_ms``; None (default): nothing is recorded."""
    if events is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    folded = False
    if p2p is not None:
        p2p.all_reduce(buf, adam=adam, n_params=n_params)
        folded = adam is not None
    else:
        torch.distributed.all_reduce(buf, op=torch.distributed.ReduceOp.SUM, group=group)
    if events is not None:
        ev[1].record()
        events.append(ev)
    return folded


def lbfgs_on_device(theta, loss_and_grad, options, callback=None):
    """The L-BFGS stage without the host in the loop: ``torch.optim.LBFGS`` (two-loop recursion with ``maxcor`` correction pairs and a
    strong-Wolfe line search, all vector work on the GPU in fp32) drives the same kernels.  ``loss_and_grad()`` evaluates at the current
    ``theta`` and returns (loss as float, gradient device tensor).  scipy's L-BFGS-B -- what the reference uses through
    ScipyOptimizerInterface -- spends ~35 ms per iteration on a 35 k-parameter net with 50 pairs, ten times the kernels' time; this is
    the opt-in alternative (``train_bfgs(..., backend="torch")``).  The iterates differ from scipy's (another line search), the
    objective and its gradient do not."""
    p = theta.detach().clone().requires_grad_(True)
    opt = torch.optim.LBFGS([p], lr=1.0, max_iter=int(options.get("maxiter", 1000)), max_eval=int(options.get("maxfun", 1250)),
                            history_size=int(options.get("maxcor", 50)), tolerance_grad=float(options.get("gtol", 1e-10)),
                            tolerance_change=float(options.get("tolerance_change", 1e-14)), line_search_fn="strong_wolfe")
    stats = {"nfev": 0, "fun": None}

    def closure():
        theta.copy_(p.detach())
        loss, grad = loss_and_grad()
        p.grad = grad.detach().clone()
        stats["nfev"] += 1
        stats["fun"] = loss
        if callback is not None:
            callback(loss)
        return torch.tensor(loss, dtype=torch.float32, device=theta.device)

    opt.step(closure)
    theta.copy_(p.detach())
    final = loss_and_grad()[0]                       # the optimizer's last point is not necessarily its last evaluation
    stats["fun"] = final
    stats["nit"] = opt.state[p].get("n_iter", 0)
    return stats


class DeepHPM(NetApi):
    """Drop-in for the reference's model class on the wave cases (INF:21-376)."""

    def __init__(self, Collo, SRC, IC, UP, uv_layers, lb, ub, ExistModel=0, modelDir='', *, case="infinite",
                 FIX=None, precision="f16x3", engine=None, seed=1111, process_group=None, verbose=True,
                 E=2.5, mu=0.25, rho=1.0, always_reduce=False, shard_as=None, collective="rccl", step_call=True, p2p_timeout_s=None):
        self.count = 0                      # callback counter (INF:26)
        self._step_call = bool(step_call)   # False: the step as separate library calls (collocation, side sets, Adam) -- the same bits; for A / B timing
        self._shift_state = {}              # adjoint-shift bookkeeping of evaluate_with_finite_gradient
        self.loss_rec = []                  # SEMI:39
        self.case = case
        self.layout = LOSS_LAYOUT[case]
        self.normalize = case == "infinite"            # INF:191 vs SEMI:198 / CONF:235
        self.lb = np.asarray(lb, dtype=np.float64).reshape(-1)
        self.ub = np.asarray(ub, dtype=np.float64).reshape(-1)
        self.E, self.mu, self.rho = E, mu, rho         # INF:33-35
        self.uv_layers = [int(v) for v in uv_layers]
        self.verbose = verbose

        # ---- data parallel topology
        self.pg = process_group
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank = torch.distributed.get_rank(self.pg)
            self.world = torch.distributed.get_world_size(self.pg)
        else:
            self.rank, self.world = 0, 1
        if shard_as is not None:
            # ``shard_as=(r, w)``: hold and evaluate rank r's share of a w-rank job WITHOUT the other ranks -- every set sharded by _shard, sums
            # weighted with the global 1/N, exactly what that rank executes per step (bench.py --rank-share; the collective is a separate switch)
            self.rank, self.world = int(shard_as[0]), int(shard_as[1])
        # ``always_reduce``: run the step's all-reduce even in a process group of one rank -- the collective branch (RCCL under the
        # "nccl" backend) then executes on a single GPU exactly as it does on eight, which is how the tests prove that path here
        self._reduce = (self.world > 1 and shard_as is None) or (bool(always_reduce) and torch.distributed.is_available() and torch.distributed.is_initialized())

        # ---- engine (GPU kernels); tests may inject a stand-in with the same methods
        if engine is None:
            from .hip_engine import HipEngine
            n_max = max(int(np.asarray(Collo).shape[0]) // self.world + 1, 1 << 14)
            engine = HipEngine(self.uv_layers, precision=precision, max_points=n_max)
        self.engine = engine
        self.device = engine.device
        if hasattr(engine, "warn_if_slow_path"):
            engine.warn_if_slow_path("wave")      # no silent slow path: a depth the fused kernel is not compiled for says so once

        # ---- weights (INF:64-68)
        self._init_rng = np.random.default_rng(seed)
        if ExistModel == 0:
            W, b = self.initialize_NN(self.uv_layers)
        else:
            W, b = self.load_NN(modelDir, self.uv_layers)
        self.n_params = sum(w.size for w in W) + sum(x.size for x in b)
        self.theta = torch.from_numpy(pack_params(W, b)).to(self.device)
        self._shift_state["theta"] = self.theta        # (updated in place: the weight-range check of evaluate_with_finite_gradient reads it)
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_t = 0

        # ---- point sets, column split like INF:40-59, stored SoA on the device
        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)

        Collo = np.asarray(Collo, dtype=np.float64)
        self.x_c, self.y_c, self.t_c = Collo[:, 0:1], Collo[:, 1:2], Collo[:, 2:3]      # host copies, as the reference keeps them (INF:40-42)
        # Device residency: one process holds the whole set; a data-parallel rank holds only ITS rows of each collocation block
        # (uploaded when a block is first used, see _rows), so device memory and upload time scale with 1/world.
        self._n_collo = Collo.shape[0]
        self._collo_host = tuple(np.ascontiguousarray(_col(Collo[:, k]), dtype=np.float32) for k in range(3))
        self._collo_full = tuple(dev(a) for a in self._collo_host) if self.world == 1 else None
        self._collo_cache = {}
        self._sides = {}      # name -> (x, y, t, targets[7,n] or None, out columns)

        def side(name, A, cols, target_cols=None):
            if A is None:
                return
            A = np.asarray(A, dtype=np.float64)
            if A.shape[0] == 0:
                return
            n_glob = A.shape[0]
            s, e = self._shard(0, n_glob)            # this rank's contiguous rows
            A = A[s:e]
            tg = None
            if target_cols is not None and e > s:
                T = np.zeros((7, A.shape[0]), dtype=np.float32)
                for o, cidx in zip(cols, target_cols):
                    T[o] = A[:, cidx]
                tg = dev(T)
            self._sides[name] = (dev(_col(A[:, 0])), dev(_col(A[:, 1])), dev(_col(A[:, 2])), tg, tuple(cols), n_glob)

        side("IC", IC, (0, 1, 2, 3))                     # u,v,ut,vt = 0 at t=0          INF:111-114
        side("SRC", SRC, (0, 1), (3, 4))                 # u,v = source displacement     INF:115-116
        side("NB", UP, (5, 6))                           # s22,s12 = 0 on the free edge  INF:117-118
        side("FIX", FIX, (0, 1))                         # u,v = 0 on the fixed edges    CONF:145-146
        self.SRC, self.IC, self.UP, self.FIX = SRC, IC, UP, FIX

        P = self.n_params
        self._buf = torch.zeros(P + 8 * len(_SLOTS), dtype=torch.float32, device=self.device)
        # ``collective``: "rccl" (default) = torch.distributed.all_reduce of the fused buffer; "p2p" = the library's one-shot all-reduce over
        # IPC-mapped peer buffers with the Adam update in the same kernel (p2p.P2PAllReduce, include/pinn_hip.h: pinn_p2p_*)
        if collective not in ("rccl", "p2p"):
            raise ValueError("collective must be 'rccl' or 'p2p'")
        self._p2p = None
        if collective == "p2p" and self._reduce:
            if not hasattr(self.engine, "lib"):
                raise ValueError("collective='p2p' needs the HIP engine (the one-shot all-reduce is a kernel of libpinn_hip.so)")
            from .p2p import P2PAllReduce
            self._p2p = P2PAllReduce(self.engine.lib, self._buf.numel(), self.pg, timeout_s=p2p_timeout_s)

    def _check_collective(self):
        """collective='p2p': raise if a one-shot all-reduce has failed on this rank (p2p.P2PAllReduce.check: reads a host word, no
        synchronisation) -- called behind every host sync point of train / train_bfgs / getloss, so that a rank that missed a call, or whose
        peer did, stops at once instead of training on with diverged parameters (round-5 review)."""
        if getattr(self, "_p2p", None) is not None:
            self._p2p.check()

    def close(self):
        """Release the P2P communicator (collective: every rank calls it).  Nothing to do for collective='rccl'."""
        p2p, self._p2p = getattr(self, "_p2p", None), None
        if p2p is not None:
            p2p.close()

    # ------------------------------------------------------------------------------------------
    # checkpoints: the reference's [W_list, b_list] pickle (INF:159-186); .npz is accepted too
    # ------------------------------------------------------------------------------------------
    def save_NN(self, fileDir, TYPE=''):
        """INF:159-170.  A name ending in ``.npz`` writes plain arrays (the documented default of this package); any other name writes the
        reference's pickle of [W_list, b_list] (net_api.py)."""
        W, b = unpack_params(self.theta.detach().cpu().numpy(), self.uv_layers)
        write_checkpoint(fileDir, W, b, self.uv_layers)
        if self.verbose:
            print("Save NN parameters successfully...")

    def load_NN(self, fileDir, layers):
        """INF:172-186: ``.npz`` or the reference's pickle (read by an arrays-only unpickler, net_api.py)"""
        num_layers = len(layers)
        uv_weights, uv_biases = read_checkpoint(fileDir)
        # Stored model must have the same number of layers (INF:178)
        assert num_layers == (len(uv_weights) + 1)
        weights = [np.asarray(w, dtype=np.float32) for w in uv_weights]
        biases = [np.asarray(b, dtype=np.float32).reshape(1, -1) for b in uv_biases]
        for i, w in enumerate(weights):
            assert w.shape == (layers[i], layers[i + 1]), "stored weight shape does not match uv_layers"
        if self.verbose:
            print("Load NN parameters successfully...")
        return weights, biases

    def set_weights(self, weights, biases):
        self.theta.copy_(torch.from_numpy(pack_params(weights, biases)).to(self.device))

    # ------------------------------------------------------------------------------------------
    # graph pieces as callables on column arrays [N,1] (INF:188-276)
    # ------------------------------------------------------------------------------------------
    def _fields(self, x, y, t):
        xs = [torch.from_numpy(np.ascontiguousarray(_col(a), dtype=np.float32)).to(self.device) for a in (x, y, t)]
        return self.engine.fields(self.theta, xs[0], xs[1], xs[2], self.lb, self.ub, self.normalize)   # [4,7,N]

    @staticmethod
    def _cols(T):
        return tuple(T[i].detach().cpu().numpy().reshape(-1, 1) for i in range(T.shape[0]))

    # neural_net(X, weights, biases): net_api.NetApi, through these two
    def _current_theta(self):
        return self.theta

    def _net_fields(self, eng, theta, X):
        xs = [torch.from_numpy(np.ascontiguousarray(_col(X[:, k]), dtype=np.float32)).to(eng.device) for k in range(3)]
        return eng.fields(theta, xs[0], xs[1], xs[2], self.lb, self.ub, self.normalize)[0]

    def net_uv(self, x, y, t):                       # INF:201-211
        return self._cols(self._fields(x, y, t)[0])

    def net_e(self, x, y, t):                        # INF:213-219
        F = self._fields(x, y, t)
        e11, e22, e12 = F[1, 0], F[2, 1], F[2, 0] + F[1, 1]
        return self._cols(torch.stack([e11, e22, e12]))

    def net_f_sig(self, x, y, t):                    # INF:221-265
        F = self._fields(x, y, t)
        return self._cols(self._residuals(F))

    def _residuals(self, F):
        E, mu, rho = self.E, self.mu, self.rho
        V, X, Y, T = F[0], F[1], F[2], F[3]
        e11, e22, e12 = X[0], Y[1], Y[0] + X[1]
        coef = E / ((1 + mu) * (1 - 2 * mu))                         # plane strain, INF:238
        sp11 = coef * (1 - mu) * e11 + coef * mu * e22
        sp22 = coef * mu * e11 + coef * (1 - mu) * e22
        sp12 = E / (2 * (1 + mu)) * e12
        f_u = X[4] + Y[6] - rho * T[2]
        f_v = Y[5] + X[6] - rho * T[3]
        return torch.stack([f_u, f_v, T[0] - V[2], T[1] - V[3], V[4] - sp11, V[5] - sp22, V[6] - sp12])

    # BASELINE.json's north_star calls these net_uvp / net_f; the reference's names are net_uv / net_f_sig
    net_uvp = net_uv
    net_f = net_f_sig

    def net_surf_var(self, x, y, t, nx, ny):         # INF:267-276
        u, v, ut, vt, s11, s22, s12 = self.net_uv(x, y, t)
        return s11 * nx + s12 * ny, s12 * nx + s22 * ny

    def callback(self, loss):                        # INF:278-280, SEMI:285-288
        self.count = self.count + 1
        self.loss_rec.append(loss)
        if self.verbose and self.rank == 0:
            print('{} th iterations, Loss: {}'.format(self.count, loss))

    # ------------------------------------------------------------------------------------------
    # one evaluation of loss terms + total gradient on one collocation block
    # ------------------------------------------------------------------------------------------
    def _shard(self, lo, hi):
        n = hi - lo
        return lo + n * self.rank // self.world, lo + n * (self.rank + 1) // self.world

    def _rows(self, lo, hi):
        """Device tensors (x, y, t) of THIS rank's rows of the collocation block [lo, hi)."""
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

    @property
    def _collo(self):
        """this rank's rows of the whole collocation set"""
        return self._rows(0, self._n_collo)

    def _loss_and_grad(self, idx_start, idx_end, sums_out=None, adam=None):
        """Fills self._buf = [grad (P) | 8 floats per slot] with this rank's partial sums, then
        all-reduces.  Everything stays on the device.  Single process only: ``sums_out`` (a view of
        8*len(_SLOTS) floats, zero where no set exists) receives the sums directly instead of the tail of the buffer.
        ``adam = (learning_rate, step)``: the caller wants the Adam update behind this gradient; where the whole evaluation is ONE library call
        (engine.wave_step: collocation set + side sets in one launch, one reduction) and no collective stands in between, the update rides in
        that reduction -- returns True then (the caller skips its own adam_step), False otherwise."""
        P, lay, eng, buf = self.n_params, self.layout, self.engine, self._buf
        n_blk = idx_end - idx_start
        s, e = self._shard(idx_start, idx_end)
        # slots this call's kernels (over)write: the collocation slot if this rank has rows of the block, every active non-empty side set
        writes = {0} if e > s else set()
        for k, name in enumerate(_SLOTS[1:], start=1):
            if name in self._sides and lay[name] != 0.0 and self._sides[name][0].numel() > 0:
                writes.add(k)
        if sums_out is None or self._reduce:
            sums = buf[P:]
            # slots of skipped sets must read zero (getloss toggles the NB weight; a rank without rows of a set contributes nothing to the
            # reduced sum).  The kernels overwrite their slots, so only a slot that was written before and is NOT written now needs zeroing
            # -- in a training loop the same slots are written every step and no launch is spent here.
            stale = getattr(self, "_slots_dirty", set(range(len(_SLOTS)))) - writes
            for k in stale:
                sums[8 * k:8 * k + 8].zero_()
            self._slots_dirty = set(writes)
            if self._reduce:
                # the all-reduce below leaves the GLOBAL total in every slot ANY rank wrote -- also in a slot this rank did not write
                # (a set or block with fewer rows than ranks: _shard gives e == s here).  Such a slot must be zeroed again before the next
                # collective, or the old total is added into it.  Which slots are written somewhere is a global fact (a side set is
                # registered on every rank with its global row count): mark them all.
                self._slots_dirty |= ({0} if n_blk > 0 else set()) | {k for k, name in enumerate(_SLOTS[1:], start=1)
                                                                       if name in self._sides and lay[name] != 0.0}
        else:
            sums = sums_out
        grad = buf[:P]
        tw = [lay["f_uv"] / n_blk] * 4 + [lay["f_s"] / n_blk] * 3
        side = []
        for k, name in enumerate(_SLOTS[1:], start=1):
            if name not in self._sides or lay[name] == 0.0:
                continue
            x, y, t, tg, cols, n = self._sides[name]
            if x.numel() == 0:
                continue
            ow = [0.0] * 7
            for o in cols:
                ow[o] = lay[name] / n
            side.append((x, y, t, tg, ow, sums[8 * k:8 * k + 8]))
        stepped = False
        if e > s and 1 <= len(side) <= 4 and hasattr(eng, "wave_step") and self._step_call:
            # the whole evaluation as ONE library call: repack | one persistent launch for all sets | one reduction (+ Adam)
            x, y, t = self._rows(idx_start, idx_end)
            fold = adam is not None and not self._reduce
            eng.wave_step(self.theta, x, y, t, self.lb, self.ub, self.normalize, tw, side, grad, sums[0:8], self.E, self.mu, self.rho, True,
                          adam=(self.adam_m, self.adam_v, adam[0], adam[1]) if fold else None)
            stepped = fold
        else:
            wrote = False
            if e > s:
                x, y, t = self._rows(idx_start, idx_end)
                eng.wave_loss_grad(self.theta, x, y, t, self.lb, self.ub, self.normalize, tw, self.E, self.mu, self.rho, True,
                                   grad_out=grad, accumulate=False, loss_out=sums[0:8])
                wrote = True
            for i in range(0, len(side), 4):           # all value-only sets of the step in one call (up to PINN_MAX_SETS per call)
                eng.data_loss_grad_multi(self.theta, side[i:i + 4], self.lb, self.ub, self.normalize, grad_out=grad, accumulate=wrote, packed=wrote)
                wrote = True
            if not wrote:
                grad.zero_()
        if self._reduce:
            # one fused buffer [gradient | loss sums] (latency-bound message); enqueued behind the kernels in stream order, and Adam is
            # enqueued behind it: the host never waits
            fold = adam is not None and self._p2p is not None
            stepped = all_reduce_sum(buf, self.pg, getattr(self, "collective_events", None), self._p2p,
                                     (self.theta, self.adam_m, self.adam_v, adam[0], adam[1]) if fold else None, P)
        return stepped

    def _terms_from_sums(self, sums, n_blk):
        """sums: [len(_SLOTS), 8] numpy array of sums of squares -> the reference's loss terms."""
        lay = self.layout
        out = {"loss_f_uv": float(sums[0, :4].sum() / n_blk), "loss_f_s": float(sums[0, 4:7].sum() / n_blk)}
        for k, name in enumerate(_SLOTS[1:], start=1):
            if name in self._sides:
                n = self._sides[name][5]
                cols = list(self._sides[name][4])
                out["loss_" + name] = float(sums[k, cols].sum() / n)
            else:
                out["loss_" + name] = 0.0
        out["loss"] = (lay["f_uv"] * out["loss_f_uv"] + lay["f_s"] * out["loss_f_s"] + lay["IC"] * out["loss_IC"]
                       + lay["SRC"] * out["loss_SRC"] + lay["NB"] * out["loss_NB"] + lay["FIX"] * out["loss_FIX"])
        return out

    # ------------------------------------------------------------------------------------------
    # training drivers
    # ------------------------------------------------------------------------------------------
    def train(self, iter, learning_rate, batch_num, record="pre"):
        """Adam loop of INF:282-319: contiguous collocation blocks, ``iter`` steps per block, all
        side sets fed whole to every block.  Returns the per-step lists
        (loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss).  ``record="pre"`` (default): each recorded value is the loss
        the step's gradient was taken at -- free.  ``record="post"``: the reference's own bookkeeping, every term re-evaluated
        AFTER the update (its four to six extra sess.run calls per step, INF:308-317) -- one more loss+gradient evaluation per step."""
        if record not in ("pre", "post"):
            raise ValueError("record must be 'pre' or 'post'")
        loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss = [], [], [], [], []
        P = self.n_params
        col_num = self._n_collo
        for i in range(batch_num):
            idx_start = int(i * col_num / batch_num)
            idx_end = int((i + 1) * col_num / batch_num)
            rec = torch.zeros((iter, 8 * len(_SLOTS)), dtype=torch.float32, device=self.device)

            def probe():
                self._loss_and_grad(idx_start, idx_end)
                return self._buf

            if iter > 0 and getattr(self.engine, "needs_finite_probe", False) and not self._shift_state.get("probed"):
                # once per model: a synchronous evaluation settles the adjoint shift (Adam's steps are small, it rarely moves after)
                evaluate_with_finite_gradient(self.engine, probe, P, self._shift_state, check=self._check_collective)
                self._shift_state["probed"] = True
            for it in range(iter):
                self.adam_t += 1
                updated = self._loss_and_grad(idx_start, idx_end, sums_out=rec[it], adam=(learning_rate, self.adam_t))     # (sums_out: one launch less per step than copying afterwards)
                if self._reduce:
                    rec[it].copy_(self._buf[P:])
                if not updated:
                    self.engine.adam_step(self.theta, self.adam_m, self.adam_v, self._buf[:P], learning_rate, self.adam_t)
                if record == "post":                 # INF:308-317: the terms at the updated weights
                    self._loss_and_grad(idx_start, idx_end)
                    rec[it].copy_(self._buf[P:])
                if self.verbose and it % 10 == 0 and self.rank == 0:
                    tm = self._terms_from_sums(rec[it].detach().cpu().numpy().reshape(len(_SLOTS), 8), idx_end - idx_start)
                    print('It: %d, Loss: %.3e' % (it, tm["loss"]))
            sums = rec.detach().cpu().numpy().reshape(iter, len(_SLOTS), 8)
            self._check_collective()                 # (behind the block's one host synchronisation)
            if iter > 0 and not bool(torch.isfinite(self.theta).all()):
                raise FloatingPointError("parameters became non-finite during train(): residuals outgrew the 16-bit reverse pass; "
                                         "lower the learning rate or raise engine.adjoint_shift")
            for it in range(iter):
                tm = self._terms_from_sums(sums[it], idx_end - idx_start)
                loss_f_uv.append(tm["loss_f_uv"])
                loss_f_s.append(tm["loss_f_s"])
                loss_IC.append(tm["loss_IC"])
                loss_SRC.append(tm["loss_SRC"])
                loss.append(tm["loss"])
        return loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss

    def train_bfgs(self, batch_num, options: Optional[dict] = None, backend: str = "scipy"):
        """L-BFGS-B stage of INF:321-335: scipy on the host over a float64 copy of the flat
        parameter vector, loss and gradient from the device kernels; ``callback`` fires once per
        function evaluation like ScipyOptimizerInterface's loss_callback.  ``backend="torch"``: the same stage with the
        optimizer on the device as well (lbfgs_on_device), for when the host update is the bottleneck."""
        import scipy.optimize
        P = self.n_params
        opts = dict(BFGS_OPTIONS[self.case])
        if options:
            opts.update(options)
        col_num = self._n_collo
        result = None
        for i in range(batch_num):
            idx_start = int(i * col_num / batch_num)
            idx_end = int((i + 1) * col_num / batch_num)

            def evaluate():
                self._loss_and_grad(idx_start, idx_end)
                return self._buf

            def fun(theta64):
                self.theta.copy_(torch.from_numpy(theta64.astype(np.float32)).to(self.device))
                host = evaluate_with_finite_gradient(self.engine, evaluate, P, self._shift_state, check=self._check_collective)
                tm = self._terms_from_sums(host[P:].reshape(len(_SLOTS), 8), idx_end - idx_start)
                relax_adjoint_shift(self.engine, tm["loss"], self._shift_state)
                self.callback(tm["loss"])
                return tm["loss"], host[:P].astype(np.float64)

            if backend == "torch":
                def loss_and_grad():
                    host = evaluate_with_finite_gradient(self.engine, evaluate, 0, self._shift_state, device_check=P, check=self._check_collective)
                    tm = self._terms_from_sums(host.reshape(len(_SLOTS), 8), idx_end - idx_start)
                    relax_adjoint_shift(self.engine, tm["loss"], self._shift_state)
                    return tm["loss"], self._buf[:P]
                result = lbfgs_on_device(self.theta, loss_and_grad, opts, self.callback)
                continue
            x0 = self.theta.detach().cpu().numpy().astype(np.float64)
            result = scipy.optimize.minimize(fun, x0, jac=True, method='L-BFGS-B', options=opts)
            self.theta.copy_(torch.from_numpy(result.x.astype(np.float32)).to(self.device))
        return result

    # ------------------------------------------------------------------------------------------
    # inference / diagnostics
    # ------------------------------------------------------------------------------------------
    def predict(self, x_star, y_star, t_star):
        """INF:337-347: (u, v, s11, s22, s12, e11, e22, e12), each numpy [M,1]."""
        F = self._fields(x_star, y_star, t_star)
        out = torch.stack([F[0, 0], F[0, 1], F[0, 4], F[0, 5], F[0, 6], F[1, 0], F[2, 1], F[2, 0] + F[1, 1]])
        return self._cols(out)

    probe = predict                                  # INF:349-359 is a verbatim copy of predict

    def getloss(self):
        """INF:361-376: (loss, loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss_NB) on the full sets."""
        n = self._n_collo
        self._loss_and_grad(0, n)
        host = self._buf[self.n_params:].detach().cpu().numpy().reshape(len(_SLOTS), 8)
        self._check_collective()
        tm = self._terms_from_sums(host, n)
        if self.layout["NB"] == 0.0 and "NB" in self._sides:
            # loss_NB is reported by getloss even where it is excluded from the total (INF:117-119,374)
            lay = dict(self.layout)
            self.layout = dict(lay, NB=1.0)
            try:
                self._loss_and_grad(0, n)
                h2 = self._buf[self.n_params:].detach().cpu().numpy().reshape(len(_SLOTS), 8)
                tm["loss_NB"] = self._terms_from_sums(h2, n)["loss_NB"]
            finally:
                self.layout = lay
        return tm["loss"], tm["loss_f_uv"], tm["loss_f_s"], tm["loss_IC"], tm["loss_SRC"], tm["loss_NB"]


class DeepHPMConfined(DeepHPM):
    """Signature of the confined-domain script (CONF:23): its distance/particular networks are
    created but never enter the graph there (CONF:282-294 ignores them), so they are accepted,
    initialised (so that save_NN can write them, CONF:197-217) and otherwise ignored here too."""

    def __init__(self, Collo, SRC, IC, FIXED, DIST, uv_layers, dist_layers, part_layers, lb, ub,
                 uvDir='', partDir='', distDir='', **kw):
        super().__init__(Collo, SRC, IC, None, uv_layers, lb, ub, ExistModel=1 if uvDir else 0, modelDir=uvDir,
                         case="confined", FIX=FIXED, **kw)
        self.DIST = DIST
        self.dist_layers = None if dist_layers is None else [int(v) for v in dist_layers]
        self.part_layers = None if part_layers is None else [int(v) for v in part_layers]
        seed = kw.get("seed", 1111)
        self._aux_nets = {}
        for name, layers, path, off in (("DIST", self.dist_layers, distDir, 1), ("PART", self.part_layers, partDir, 2)):
            if layers is not None:
                self._aux_nets[name] = self.load_NN(path, layers) if path else xavier_init(layers, np.random.default_rng(seed + off))

    def train(self, iter, learning_rate, batch_num, record="pre"):
        """CONF:373-408 returns three lists: (loss_f_uv, loss_f_s, loss)."""
        loss_f_uv, loss_f_s, _, _, loss = super().train(iter, learning_rate, batch_num, record)
        return loss_f_uv, loss_f_s, loss

    def save_NN(self, fileDir, TYPE=''):
        """CONF:197-217: TYPE selects the net ('UV', 'DIST', 'PART'); anything else writes nothing."""
        if TYPE in ('UV', ''):
            return super().save_NN(fileDir)
        if TYPE in ('DIST', 'PART') and TYPE in self._aux_nets:
            W, b = self._aux_nets[TYPE]
            write_checkpoint(fileDir, W, b, self.dist_layers if TYPE == 'DIST' else self.part_layers)
            if self.verbose:
                print("Save %s NN parameters successfully..." % TYPE)

    def getloss(self):
        """CONF:453-472: (loss, loss_f_uv, loss_f_s, loss_IC, loss_SRC, loss_FIX) on the full sets."""
        n = self._n_collo
        self._loss_and_grad(0, n)
        tm = self._terms_from_sums(self._buf[self.n_params:].detach().cpu().numpy().reshape(len(_SLOTS), 8), n)
        return tm["loss"], tm["loss_f_uv"], tm["loss_f_s"], tm["loss_IC"], tm["loss_SRC"], tm["loss_FIX"]
