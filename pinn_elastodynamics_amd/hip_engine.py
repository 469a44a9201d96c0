"""Device engine: PyTorch owns HBM buffers and streams, libpinn_hip.so does the work.

This is the only place the product touches the GPU kernels.  There is NO CPU fallback: without a
GPU or without the built shared library, construction raises.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .capi import DEFAULT_LIB, FLAG_STATE_FP16, FLAG_TWO_KERNEL, FLAG_WEIGHTS_PACKED, PREC, PinnLib, PinnLibError, adjoint_shift


def param_count(layers: Sequence[int]) -> int:
    return sum(layers[i] * layers[i + 1] + layers[i + 1] for i in range(len(layers) - 1))


class HipEngine:
    """loss/gradient, fields and Adam on one GPU, on flat fp32 parameter vectors.

    All tensors are fp32 CUDA(HIP) tensors; points are SoA (x, y, t).  Methods enqueue on the
    current torch stream and return device tensors without synchronising.
    """

    def __init__(self, layers: Sequence[int], precision: str = "f16x3", device: Optional[torch.device] = None,
                 max_points: int = 1 << 18, lib_path: str = DEFAULT_LIB, workspace_bytes: Optional[int] = None,
                 workspace_cap_bytes: int = 8 << 30, fast_state: bool = False):
        if not torch.cuda.is_available():
            raise PinnLibError("HipEngine needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = PinnLib(lib_path)
        self.layers = [int(v) for v in layers]
        self.precision = precision
        # fast_state: PINN_FLAG_STATE_FP16 -- the fused 8-layer collocation kernel parks its states as fp16 only: 17 % faster, but the
        # gradient at trained weights loses accuracy by cancellation (DESIGN_HISTORY.md section 6).  Off by default.
        self.fast_state = bool(fast_state)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.n_params = param_count(self.layers)
        self.adjoint_shift = 0
        self.two_kernel = False               # PINN_FLAG_TWO_KERNEL on every call (set by leave_fused_path_if_weights_out_of_range)
        self.needs_finite_probe = True        # 16-bit reverse pass: the model classes check the first gradient (PINN_ADJOINT_SHIFT)
        if self.lib.supported_width(self.layers[1]) == 0:
            raise PinnLibError(f"hidden width {self.layers[1]} is not supported by the compiled kernels")
        want = workspace_bytes if workspace_bytes is not None else self.lib.workspace_bytes(self.layers, max_points, precision)
        if workspace_bytes is None:
            # larger point sets are walked in several passes; the default cap of 8 GiB already amortises the launches
            # (``workspace_cap_bytes`` raises or lowers it, ``workspace_bytes`` fixes the size outright)
            want = min(want, int(workspace_cap_bytes))
        want = max(want, self.lib.min_workspace_bytes(self.layers, precision))
        if want == 0:
            raise PinnLibError(f"no kernel variant for layers={self.layers} precision={precision}")
        self.ws_bytes = int(want)
        self.ws = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.device)
        self._ws_ptr = (self.ws.data_ptr() + 255) // 256 * 256

        self._lib_path = lib_path

    def for_layers(self, layers: Sequence[int]) -> "HipEngine":
        """A sibling engine for a net of other layer sizes (same precision mode, device and library): neural_net(X, weights, biases) of
        the model classes on weights that are not the model's own net (INF:188-199 takes any weight list)."""
        return HipEngine(layers, precision=self.precision, device=self.device, max_points=1 << 16, lib_path=self._lib_path, fast_state=self.fast_state)

    # ------------------------------------------------------------------------------------------
    def _mode(self, packed: bool) -> int:
        """precision_mode argument of the loss / gradient calls: the precision, PINN_FLAG_WEIGHTS_PACKED when the previous call of this
        engine used the same parameter values (skip the repack launch), and the current adjoint shift (``self.adjoint_shift``, see
        PINN_ADJOINT_SHIFT in include/pinn_hip.h; the model classes adapt it when a gradient comes back non-finite)."""
        return (PREC[self.precision] | (FLAG_WEIGHTS_PACKED if packed else 0) | (FLAG_STATE_FP16 if self.fast_state else 0)
                | (FLAG_TWO_KERNEL if self.two_kernel else 0) | adjoint_shift(self.adjoint_shift))

    def leave_fused_path_if_weights_out_of_range(self, params: torch.Tensor) -> bool:
        """The fused kernels' weight format holds |w| <= 2047 (include/pinn_hip.h); beyond it a call returns NaN throughout.  Called by the
        model classes when a result comes back non-finite: if a weight is out of range, this engine's calls switch to the two-kernel path
        (PINN_FLAG_TWO_KERNEL) for good and True is returned -- the caller repeats its evaluation.  Synchronises (one reduction)."""
        if self.two_kernel or self.precision != "f16x3":      # (bf16x3: the fused format has fp32's range, nothing to leave for)
            return False
        if float(params.detach().abs().max()) <= self.lib.fused_weight_limit():
            return False
        self.two_kernel = True
        return True

    def path(self, head: str = "wave") -> str:
        """Which kernel path this engine's loss + gradient calls of family ``head`` take ('fused-registers', 'fused-lds', 'two-kernel',
        'fp32'): pinn_path_for with this engine's layer list, precision mode word and workspace size (include/pinn_hip.h)."""
        return self.lib.path_for(self.layers, self._mode(False), head, self.ws_bytes)

    def warn_if_slow_path(self, head: str = "wave") -> None:
        """One warning per engine and family when the calls land on the two-kernel path (2.3-5x slower than the fused kernel): the fused
        kernel is compiled for the depths the reference's scripts use, any other layer list still WORKS, just not at the published speed."""
        seen = self.__dict__.setdefault("_slow_path_warned", set())
        if head in seen:
            return
        if self.path(head) == "two-kernel" and not self.two_kernel:
            import warnings
            seen.add(head)
            warnings.warn(f"layers {self.layers} ({self.precision}, '{head}' calls) run on the two-kernel path (chain_kernel + wgrad_kernel): "
                          f"2.3-5x slower than the fused persistent kernel, which is compiled for 4 or 8 hidden layers of width <= 64, 8 of width "
                          f"<= 128, 6 of width <= 160 and the 10 x 128 3-D net -- or the workspace is too small for its scratch images "
                          f"(pinn_path_for, include/pinn_hip.h)", RuntimeWarning, stacklevel=3)

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _chk(t: torch.Tensor, n: Optional[int] = None):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expect contiguous fp32 device tensors"
        if n is not None:
            assert t.numel() == n, f"expected {n} elements, got {t.numel()}"

    def wave_loss_grad(self, params, x, y, t, lb, ub, normalize, term_weights, E=2.5, mu=0.25, rho=1.0, plane_strain=True,
                       grad_out: Optional[torch.Tensor] = None, accumulate: bool = False, loss_out: Optional[torch.Tensor] = None, packed: bool = False):
        """Returns (sumsq[7] device tensor, grad_flat).  grad = d/dparams sum_i term_weights[i]*sumsq[i]."""
        n = x.numel()
        for v in (x, y, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        if grad_out is None:
            grad_out = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            accumulate = False
        if loss_out is None:
            loss_out = torch.empty(8, dtype=torch.float32, device=self.device)
        self.lib.wave2d_loss_grad(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub, normalize,
                                  E, mu, rho, plane_strain, term_weights, loss_out.data_ptr(), grad_out.data_ptr(), accumulate,
                                  self._mode(packed), self._ws_ptr, self.ws_bytes, self._stream())
        return loss_out[:7], grad_out

    def wave_loss_grad_profile(self, params, x, y, t, lb, ub, normalize, term_weights, E=2.5, mu=0.25, rho=1.0, plane_strain=True):
        """Synchronous variant returning HIP-event kernel times in ms: dict(repack, chain, wgrad, reduce)."""
        n = x.numel()
        grad = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        loss = torch.empty(8, dtype=torch.float32, device=self.device)
        ms = self.lib.wave2d_loss_grad_profile(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub,
                                               normalize, E, mu, rho, plane_strain, term_weights, loss.data_ptr(), grad.data_ptr(),
                                               False, self.precision, self._ws_ptr, self.ws_bytes, self._stream())
        return dict(zip(("repack", "chain", "wgrad", "reduce"), ms))

    def data_loss_grad(self, params, x, y, t, lb, ub, normalize, targets, out_weights,
                       grad_out: Optional[torch.Tensor] = None, accumulate: bool = False, loss_out: Optional[torch.Tensor] = None, packed: bool = False):
        """Value-only terms.  targets: [n_out, n] device tensor or None.  Returns (sumsq[n_out], grad)."""
        n = x.numel()
        nout = self.layers[-1]
        for v in (x, y, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        tptr = 0
        if targets is not None:
            self._chk(targets, nout * n)
            tptr = targets.data_ptr()
        if grad_out is None:
            grad_out = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            accumulate = False
        if loss_out is None:
            loss_out = torch.empty(8, dtype=torch.float32, device=self.device)
        self.lib.data_loss_grad(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub, normalize,
                                tptr, out_weights, loss_out.data_ptr(), grad_out.data_ptr(), accumulate, self._mode(packed),
                                self._ws_ptr, self.ws_bytes, self._stream())
        return loss_out[:nout], grad_out

    def data_loss_grad_multi(self, params, sets, lb, ub, normalize, grad_out, accumulate=False, packed=False):
        """Several value-only sets in one call.  sets: list of (x, y, t, targets_or_None, out_weights, loss_out[>=n_out]); the sums of
        set k land in its loss_out, the gradients are summed into grad_out."""
        self._chk(params, self.n_params)
        rows = []
        for x, y, t, tg, ow, lo in sets:
            n = x.numel()
            for v in (x, y, t):
                self._chk(v, n)
            if tg is not None:
                self._chk(tg, self.layers[-1] * n)
            rows.append((x.data_ptr(), y.data_ptr(), t.data_ptr(), n, 0 if tg is None else tg.data_ptr(), ow, lo.data_ptr()))
        self.lib.data_loss_grad_multi(params.data_ptr(), self.layers, rows, lb, ub, normalize, grad_out.data_ptr(), accumulate, self._mode(packed),
                                      self._ws_ptr, self.ws_bytes, self._stream())
        return grad_out

    def wave_step(self, params, x, y, t, lb, ub, normalize, term_weights, sets, grad_out, loss_out, E=2.5, mu=0.25, rho=1.0, plane_strain=True,
                  accumulate: bool = False, adam=None):
        """One training step's evaluation in one library call (pinn_wave2d_step): the collocation batch (x, y, t) with its seven term weights, the
        value-only ``sets`` (as data_loss_grad_multi: (x, y, t, targets_or_None, out_weights, loss_out)), the gradient of everything into
        ``grad_out`` and -- with ``adam = (m, v, lr, step[, beta1, beta2, eps])`` -- the TF1 Adam update of ``params`` behind it.  For the nets of
        the register-state fused kernel that is repack | one persistent launch | one reduction (+ Adam); for every other net the library makes
        the separate calls inside: the same bits either way."""
        n = x.numel()
        for v in (x, y, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        self._chk(grad_out, self.n_params)
        rows = []
        for sx, sy, st, tg, ow, lo in sets:
            m_ = sx.numel()
            for v in (sx, sy, st):
                self._chk(v, m_)
            if tg is not None:
                self._chk(tg, self.layers[-1] * m_)
            rows.append((sx.data_ptr(), sy.data_ptr(), st.data_ptr(), m_, 0 if tg is None else tg.data_ptr(), ow, lo.data_ptr()))
        ad = None
        if adam is not None:
            m, v, lr, step = adam[:4]
            b1, b2, eps = (list(adam[4:]) + [0.9, 0.999, 1e-8][len(adam) - 4:])[:3]
            for a_ in (m, v):
                self._chk(a_, self.n_params)
            ad = (m.data_ptr(), v.data_ptr(), lr, b1, b2, eps, step)
        self.lib.wave2d_step(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub, normalize, E, mu, rho, plane_strain,
                             term_weights, loss_out.data_ptr(), rows, grad_out.data_ptr(), accumulate, ad, self._mode(False), self._ws_ptr, self.ws_bytes,
                             self._stream())
        return grad_out

    def fields(self, params, x, y, t, lb, ub, normalize):
        """Returns [4, n_out, n]: Y and its derivatives w.r.t. x, y, t."""
        n = x.numel()
        nout = self.layers[-1]
        for v in (x, y, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        out = torch.empty((4, nout, n), dtype=torch.float32, device=self.device)
        self.lib.wave2d_fields(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub, normalize,
                               out.data_ptr(), self.precision, self._ws_ptr, self.ws_bytes, self._stream())
        return out

    # ---- plate family (5 streams: value, d/dx, d/dy, d/dt, d2/dt2) --------------------------------------------------
    def net_streams(self, params, x, y, t, lb, ub, normalize):
        """Returns [5, n_out, n]: the net's outputs, their first derivatives and the second time derivative."""
        n, nout = x.numel(), self.layers[-1]
        for v in (x, y, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        out = torch.empty((5, nout, n), dtype=torch.float32, device=self.device)
        self.lib.net_streams(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub, normalize,
                             out.data_ptr(), self.precision, self._ws_ptr, self.ws_bytes, self._stream())
        return out

    def _outs(self, grad_out, accumulate, loss_out):
        if grad_out is None:
            grad_out = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            accumulate = False
        if loss_out is None:
            loss_out = torch.empty(8, dtype=torch.float32, device=self.device)
        return grad_out, accumulate, loss_out

    def plate_loss_grad(self, params, x, y, t, lb, ub, normalize, frozen, term_weights, E=20.0, mu=0.25, rho=1.0,
                        grad_out=None, accumulate=False, loss_out=None):
        """frozen: [2, 5, 5, n] streams of the distance and particular nets.  Returns (sumsq[5], grad wrt this net)."""
        n = x.numel()
        self._chk(frozen, 50 * n)
        grad_out, accumulate, loss_out = self._outs(grad_out, accumulate, loss_out)
        self.lib.plate2d_loss_grad(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub, normalize,
                                   frozen.data_ptr(), E, mu, rho, term_weights, loss_out.data_ptr(), grad_out.data_ptr(), accumulate,
                                   self._mode(False), self._ws_ptr, self.ws_bytes, self._stream())
        return loss_out[:5], grad_out

    def plate_step(self, params, x, y, t, lb, ub, normalize, frozen, term_weights, hole, grad_out, loss_out, hole_loss_out, E=20.0, mu=0.25, rho=1.0,
                   accumulate: bool = False, adam=None):
        """The plate's step in one library call (pinn_plate2d_step): collocation set (frozen: [2, 5, 5, n]) + hole-traction set
        ``hole = (x, y, t, aux[12, n], weights[2])`` -> gradient in ``grad_out``, sums in ``loss_out`` / ``hole_loss_out``, and with
        ``adam = (m, v, lr, step[, beta1, beta2, eps])`` the TF1 Adam update of ``params`` in the same final launch."""
        n = x.numel()
        for v in (x, y, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        self._chk(frozen, 50 * n)
        hx, hy, ht, haux, hw = hole
        hn = hx.numel()
        self._chk(haux, 12 * hn)
        ad = None
        if adam is not None:
            m, v, lr, step = adam[:4]
            b1, b2, eps = (list(adam[4:]) + [0.9, 0.999, 1e-8][len(adam) - 4:])[:3]
            ad = (m.data_ptr(), v.data_ptr(), lr, b1, b2, eps, step)
        self.lib.plate2d_step(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub, normalize, frozen.data_ptr(), E, mu, rho,
                              term_weights, loss_out.data_ptr(), hx.data_ptr(), hy.data_ptr(), ht.data_ptr(), hn, haux.data_ptr(), hw, hole_loss_out.data_ptr(),
                              grad_out.data_ptr(), accumulate, ad, self._mode(False), self._ws_ptr, self.ws_bytes, self._stream())
        return grad_out

    def traction_loss_grad(self, params, x, y, t, lb, ub, normalize, aux, weights, grad_out=None, accumulate=False, loss_out=None, packed=False):
        """aux: [12, n] = D values, P values, nx, ny.  Returns ((sum tx^2, sum ty^2), grad)."""
        n = x.numel()
        self._chk(aux, 12 * n)
        grad_out, accumulate, loss_out = self._outs(grad_out, accumulate, loss_out)
        self.lib.plate2d_traction_loss_grad(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub,
                                            normalize, aux.data_ptr(), weights, loss_out.data_ptr(), grad_out.data_ptr(), accumulate,
                                            self._mode(packed), self._ws_ptr, self.ws_bytes, self._stream())
        return loss_out[:2], grad_out

    def stream_loss_grad(self, params, x, y, t, lb, ub, normalize, targets, weights, grad_out=None, accumulate=False, loss_out=None):
        """targets: [5, n_out, n] or None; weights: 5 x n_out nested list.  Returns (per-output normalised sums, grad)."""
        n, nout = x.numel(), self.layers[-1]
        tptr = 0
        if targets is not None:
            self._chk(targets, 5 * nout * n)
            tptr = targets.data_ptr()
        grad_out, accumulate, loss_out = self._outs(grad_out, accumulate, loss_out)
        self.lib.stream_loss_grad(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), t.data_ptr(), n, lb, ub, normalize, tptr,
                                  weights, loss_out.data_ptr(), grad_out.data_ptr(), accumulate, self.precision, self._ws_ptr,
                                  self.ws_bytes, self._stream())
        return loss_out[:nout], grad_out

    # ---- 3-D Navier-Cauchy extension (layers [4, H, ..., H, 12]; inputs x, y, z, t) -------------------------------------
    def nc3d_loss_grad(self, params, x, y, z, t, lb, ub, normalize, term_weights, E=2.5, mu=0.25, rho=1.0,
                       grad_out=None, accumulate=False, loss_out=None, packed=False):
        """Returns (sumsq[12], grad): residuals (f_u,f_v,f_w,f_ut,f_vt,f_wt,f_s11,f_s22,f_s33,f_s12,f_s13,f_s23), see include/pinn_hip.h."""
        n = x.numel()
        for v in (x, y, z, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        if grad_out is None:
            grad_out = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            accumulate = False
        if loss_out is None:
            loss_out = torch.empty(16, dtype=torch.float32, device=self.device)
        self.lib.nc3d_loss_grad(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), z.data_ptr(), t.data_ptr(), n, lb, ub, normalize,
                                E, mu, rho, term_weights, loss_out.data_ptr(), grad_out.data_ptr(), accumulate, self._mode(packed),
                                self._ws_ptr, self.ws_bytes, self._stream())
        return loss_out[:12], grad_out

    def nc3d_data_loss_grad(self, params, x, y, z, t, lb, ub, normalize, targets, out_weights,
                            grad_out=None, accumulate=False, loss_out=None, packed=False):
        """Value-only terms of the 4-input net.  targets: [n_out, n] device tensor or None.  Returns (sumsq[n_out], grad)."""
        n, nout = x.numel(), self.layers[-1]
        for v in (x, y, z, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        tptr = 0
        if targets is not None:
            self._chk(targets, nout * n)
            tptr = targets.data_ptr()
        if grad_out is None:
            grad_out = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
            accumulate = False
        if loss_out is None:
            loss_out = torch.empty(16, dtype=torch.float32, device=self.device)
        self.lib.nc3d_data_loss_grad(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), z.data_ptr(), t.data_ptr(), n, lb, ub,
                                     normalize, tptr, out_weights, loss_out.data_ptr(), grad_out.data_ptr(), accumulate,
                                     self._mode(packed), self._ws_ptr, self.ws_bytes, self._stream())
        return loss_out[:nout], grad_out

    def nc3d_fields(self, params, x, y, z, t, lb, ub, normalize):
        """Returns [5, n_out, n]: the outputs and their derivatives w.r.t. x, y, z, t."""
        n, nout = x.numel(), self.layers[-1]
        for v in (x, y, z, t):
            self._chk(v, n)
        self._chk(params, self.n_params)
        out = torch.empty((5, nout, n), dtype=torch.float32, device=self.device)
        self.lib.nc3d_fields(params.data_ptr(), self.layers, x.data_ptr(), y.data_ptr(), z.data_ptr(), t.data_ptr(), n, lb, ub, normalize,
                             out.data_ptr(), self.precision, self._ws_ptr, self.ws_bytes, self._stream())
        return out

    def adam_step(self, params, m, v, grad, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
        for a in (params, m, v, grad):
            self._chk(a, self.n_params)
        self.lib.adam_step(params.data_ptr(), m.data_ptr(), v.data_ptr(), grad.data_ptr(), self.n_params, lr, step, beta1, beta2, eps,
                           self._stream())
