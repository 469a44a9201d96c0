"""ctypes binding of the C-ABI declared in include/pinn_hip.h (libpinn_hip.so).

Pointer arguments are passed as plain integers (``tensor.data_ptr()`` for device memory), so the
binding itself has no torch dependency; PyTorch is only the owner of device buffers and streams.
"""
from __future__ import annotations

import ctypes as C
import os

PREC = {"bf16": 0, "f16x3": 1, "f16": 2, "bf16x3": 3, "fp32": 4}
HEADS = {"wave": 0, "data": 1, "plate": 2, "nc3d": 3, "nc3d_data": 4, "streams": 5}      # PINN_HEAD_*
PATHS = {1: "fused-registers", 2: "fused-lds", 3: "two-kernel", 4: "fp32"}                  # PINN_PATH_*
FLAG_WEIGHTS_PACKED = 0x100
FLAG_TWO_KERNEL = 0x400            # PINN_FLAG_TWO_KERNEL: keep the call off the fused kernel (weights beyond its |w| <= 2047 format)
FLAG_STATE_FP16 = 0x200            # PINN_FLAG_STATE_FP16: fused 8-layer collocation kernel parks fp16 states only (faster, not parity-grade)
def adjoint_shift(k: int) -> int:
    """PINN_ADJOINT_SHIFT(k) of include/pinn_hip.h"""
    return (int(k) & 0x1f) << 16


def mode_bits(prec) -> int:
    """precision name (optionally "+packed") or ready-made integer -> the precision_mode argument"""
    return PREC[prec] if isinstance(prec, str) else int(prec)


PREC["f16x3+fp16state"] = PREC["f16x3"] | FLAG_STATE_FP16       # the opt-in fp16-state variant of the fused 8-layer collocation kernel
for _k in list(PREC):                      # "<mode>+packed": the workspace still holds this call's packed weights (PINN_FLAG_WEIGHTS_PACKED)
    PREC[_k + "+packed"] = PREC[_k] | FLAG_WEIGHTS_PACKED

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PINN_HIP_LIB: another build of the same library -- the A / B experiments of tools/ run the product's own bench against build/exp/NAME/libpinn_hip.so)
DEFAULT_LIB = os.environ.get("PINN_HIP_LIB") or os.path.join(_HERE, "lib", "libpinn_hip.so")


class PointSet(C.Structure):
    """pinn_point_set of include/pinn_hip.h"""
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("t", C.c_void_p), ("n", C.c_int64), ("targets", C.c_void_p),
                ("out_weights", C.c_float * 8), ("loss_terms_out", C.c_void_p)]


class AdamState(C.Structure):
    """pinn_adam_state of include/pinn_hip.h"""
    _fields_ = [("m", C.c_void_p), ("v", C.c_void_p), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("step", C.c_int64)]


class RangeState(C.Structure):
    """pinn_range_state of include/pinn_hip.h: what the finite-gradient ladder of pinn_wave2d_loss_grad_checked carries between calls"""
    _fields_ = [("adjoint_shift", C.c_int32), ("two_kernel", C.c_int32), ("attempts", C.c_int32), ("reserved", C.c_int32)]


class PinnLibError(RuntimeError):
    pass


# The profiling hook of the library is process-wide and keeps a raw host pointer: the buffer it writes to is owned HERE, at module
# level, for as long as the hook is on -- never by a PinnLib instance that may be collected while another engine still launches.
_PROFILE_BUFFER = None


class PinnLib:
    """Thin, checked wrapper over the shared library's entry points."""

    def __init__(self, path: str = DEFAULT_LIB):
        if not os.path.exists(path):
            raise PinnLibError(
                f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C pinn_elastodynamics_amd/csrc hip`). There is no CPU fallback.")
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        vp, i32, i64, f64, sz = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t
        pi32, pf64, pf32 = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.pinn_abi_version.restype = i32
        L.pinn_supported_width.argtypes = [i32]
        L.pinn_supported_width.restype = i32
        L.pinn_error_string.argtypes = [i32]
        L.pinn_error_string.restype = C.c_char_p
        L.pinn_workspace_bytes.argtypes = [pi32, i32, i64, i32]
        L.pinn_workspace_bytes.restype = sz
        L.pinn_min_workspace_bytes.argtypes = [pi32, i32, i32]
        L.pinn_min_workspace_bytes.restype = sz
        L.pinn_wave2d_loss_grad.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, f64, f64, f64, i32, pf32,
                                            vp, vp, i32, i32, vp, sz, vp]
        L.pinn_wave2d_loss_grad.restype = i32
        L.pinn_wave2d_loss_grad_profile.argtypes = L.pinn_wave2d_loss_grad.argtypes + [pf32]
        L.pinn_wave2d_loss_grad_profile.restype = i32
        # (round 6) the finite-gradient ladder as library calls: no `accumulate` argument, a pinn_range_state at the end
        L.pinn_wave2d_loss_grad_checked.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, f64, f64, f64, i32, pf32,
                                                    vp, vp, i32, vp, sz, vp, C.POINTER(RangeState)]
        L.pinn_wave2d_loss_grad_checked.restype = i32
        L.pinn_probe_ranges.argtypes = [vp, vp, i64, vp, sz, vp, C.POINTER(i32), pf32]
        L.pinn_probe_ranges.restype = i32
        L.pinn_data_loss_grad.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, vp, pf32, vp, vp, i32, i32, vp, sz, vp]
        L.pinn_data_loss_grad.restype = i32
        L.pinn_data_loss_grad_multi.argtypes = [vp, pi32, i32, C.POINTER(PointSet), i32, pf64, pf64, i32, vp, i32, i32, vp, sz, vp]
        L.pinn_data_loss_grad_multi.restype = i32
        L.pinn_wave2d_step.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, f64, f64, f64, i32, pf32, vp, C.POINTER(PointSet), i32, vp, i32,
                                       C.POINTER(AdamState), i32, vp, sz, vp]
        L.pinn_wave2d_step.restype = i32
        L.pinn_plate2d_step.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, vp, f64, f64, f64, pf32, vp, vp, vp, vp, i64, vp, pf32, vp, vp, i32,
                                        C.POINTER(AdamState), i32, vp, sz, vp]
        L.pinn_plate2d_step.restype = i32
        L.pinn_wave2d_fields.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, vp, i32, vp, sz, vp]
        L.pinn_wave2d_fields.restype = i32
        L.pinn_net_streams.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, vp, i32, vp, sz, vp]
        L.pinn_net_streams.restype = i32
        L.pinn_plate2d_loss_grad.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, vp, f64, f64, f64, pf32, vp, vp, i32, i32, vp, sz, vp]
        L.pinn_plate2d_loss_grad.restype = i32
        L.pinn_plate2d_traction_loss_grad.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, vp, pf32, vp, vp, i32, i32, vp, sz, vp]
        L.pinn_plate2d_traction_loss_grad.restype = i32
        L.pinn_stream_loss_grad.argtypes = [vp, pi32, i32, vp, vp, vp, i64, pf64, pf64, i32, vp, pf32, vp, vp, i32, i32, vp, sz, vp]
        L.pinn_stream_loss_grad.restype = i32
        L.pinn_nc3d_loss_grad.argtypes = [vp, pi32, i32, vp, vp, vp, vp, i64, pf64, pf64, i32, f64, f64, f64, pf32, vp, vp, i32, i32, vp, sz, vp]
        L.pinn_nc3d_loss_grad.restype = i32
        L.pinn_nc3d_data_loss_grad.argtypes = [vp, pi32, i32, vp, vp, vp, vp, i64, pf64, pf64, i32, vp, pf32, vp, vp, i32, i32, vp, sz, vp]
        L.pinn_nc3d_data_loss_grad.restype = i32
        L.pinn_nc3d_fields.argtypes = [vp, pi32, i32, vp, vp, vp, vp, i64, pf64, pf64, i32, vp, i32, vp, sz, vp]
        L.pinn_nc3d_fields.restype = i32
        L.pinn_adam_step.argtypes = [vp, vp, vp, vp, i64, f64, f64, f64, f64, i64, vp]
        L.pinn_adam_step.restype = i32
        L.pinn_debug_set_profile_buffer.argtypes = [vp]
        L.pinn_debug_set_profile_buffer.restype = None
        if hasattr(L, "pinn_debug_set_xcd_bonus"):      # (tuning hook; experiment builds of older trees do not have it)
            L.pinn_debug_set_xcd_bonus.argtypes = [i32]
            L.pinn_debug_set_xcd_bonus.restype = i32
        L.pinn_debug_set_fused.argtypes = [i32]
        L.pinn_debug_set_fused.restype = i32
        L.pinn_fused_weight_limit.argtypes = []
        L.pinn_fused_weight_limit.restype = C.c_float
        L.pinn_path_for.argtypes = [pi32, i32, i32, i32, sz]
        L.pinn_path_for.restype = i32
        L.pinn_debug_path_counts.argtypes = [vp, i32]
        L.pinn_debug_path_counts.restype = None
        L.pinn_debug_profile_ring_stride.argtypes = [i32]
        L.pinn_debug_profile_ring_stride.restype = None
        L.pinn_debug_wall_clock_khz.argtypes = []
        L.pinn_debug_wall_clock_khz.restype = i32
        L.pinn_debug_set_stamp_buffer.argtypes = [vp]
        L.pinn_debug_set_stamp_buffer.restype = None
        L.pinn_debug_profile_ring_arm.argtypes = [i32]
        L.pinn_debug_profile_ring_arm.restype = i32
        L.pinn_debug_profile_ring_read.argtypes = [vp, vp, i32]
        L.pinn_debug_profile_ring_read.restype = i32

    # -- helpers -------------------------------------------------------------------------------
    @staticmethod
    def _ints(v):
        return (C.c_int * len(v))(*[int(x) for x in v])

    @staticmethod
    def _d3(v):
        return (C.c_double * 3)(*[float(x) for x in v])

    @staticmethod
    def _d4(v):
        return (C.c_double * 4)(*[float(x) for x in v])

    @staticmethod
    def _floats(v, n):
        v = list(v) + [0.0] * (n - len(v))
        return (C.c_float * n)(*[float(x) for x in v])

    def check(self, rc: int, what: str):
        if rc != 0:
            raise PinnLibError(f"{what} failed: {self.lib.pinn_error_string(rc).decode()} (code {rc})")

    def error_string(self, rc: int) -> str:
        return f"{self.lib.pinn_error_string(rc).decode()} (code {rc})"

    # -- entry points --------------------------------------------------------------------------
    def abi_version(self) -> int:
        return self.lib.pinn_abi_version()

    def set_profile_buffer(self, enable: bool):
        """Process-wide profiling hook: while on, every loss+gradient call writes its kernel milliseconds {repack, chain or whole
        fused kernel, weight gradient, reductions} into the returned float32[4] (and synchronises the stream)."""
        global _PROFILE_BUFFER
        if enable:
            if _PROFILE_BUFFER is None:
                _PROFILE_BUFFER = (C.c_float * 4)()
            self.lib.pinn_debug_set_profile_buffer(C.cast(_PROFILE_BUFFER, C.c_void_p))
            import numpy as _np
            return _np.ctypeslib.as_array(_PROFILE_BUFFER)
        self.lib.pinn_debug_set_profile_buffer(None)      # (the module-level buffer stays allocated: a launch in flight may still write to it)
        return None

    def profile_ring_arm(self, max_launches: int, every: int = 1) -> int:
        """Start recording the next launches of the fused kernel with HIP events in stream order, nothing synchronises (include/pinn_hip.h);
        ``every`` > 1: the launches of every ``every``-th step only"""
        self.lib.pinn_debug_profile_ring_stride(int(every))
        return int(self.lib.pinn_debug_profile_ring_arm(int(max_launches)))

    def set_stamp_buffer(self, device_ptr) -> None:
        """128 x uint64 device buffer for the fused kernel's shader-clock stamps of workgroup 0 (None: off)"""
        self.lib.pinn_debug_set_stamp_buffer(None if device_ptr is None else C.c_void_p(int(device_ptr)))

    def path_for(self, layers, precision, head: str = "wave", ws_bytes: int = 0) -> str:
        """pinn_path_for: 'fused-registers' | 'fused-lds' | 'two-kernel' | 'fp32' for a loss + gradient call of family ``head``
        ('wave', 'data', 'plate', 'nc3d', 'nc3d_data', 'streams'); ``precision`` is a mode name or the full precision_mode word"""
        mode = PREC[precision] if isinstance(precision, str) else int(precision)
        rc = int(self.lib.pinn_path_for(self._ints(layers), len(layers), mode, HEADS[head], int(ws_bytes)))
        if rc <= 0:
            raise PinnLibError(f"pinn_path_for failed: {self.lib.pinn_error_string(rc).decode()} (code {rc})")
        return PATHS[rc]

    def cache_policy(self, layers, head: str = "wave") -> dict:
        """pinn_debug_cache_policy: per-workgroup bytes of the fused collocation kernel's two memory classes and which of them (if any) the
        instantiation marks non-temporal -- {'policy': 'none' | 'sums' | 'images', 'images_bytes', 'sums_bytes', 'grid_bytes' (256 workgroups)}"""
        im, su = C.c_size_t(0), C.c_size_t(0)
        self.lib.pinn_debug_cache_policy.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.lib.pinn_debug_cache_policy.restype = C.c_int
        rc = int(self.lib.pinn_debug_cache_policy(self._ints(layers), len(layers), HEADS[head], C.byref(im), C.byref(su)))
        if rc < 0:
            raise PinnLibError(f"pinn_debug_cache_policy failed: {self.lib.pinn_error_string(rc).decode()} (code {rc})")
        return {"policy": ("none", "sums", "images")[rc], "images_bytes": int(im.value), "sums_bytes": int(su.value), "grid_bytes": 256 * (int(im.value) + int(su.value))}

    def path_counts(self, reset: bool = False) -> dict:
        """calls per path since the last reset (pinn_debug_path_counts)"""
        buf = (C.c_int64 * 5)()
        self.lib.pinn_debug_path_counts(C.cast(buf, C.c_void_p), int(bool(reset)))
        return {PATHS[i]: int(buf[i]) for i in range(1, 5)}

    def profile_ring_read(self):
        """Stop the recording and return (milliseconds, streams) of the recorded launches, in launch order"""
        import numpy as _np
        cap = 4096
        ms = (C.c_float * cap)()
        tags = (C.c_int * cap)()
        m = int(self.lib.pinn_debug_profile_ring_read(C.cast(ms, C.c_void_p), C.cast(tags, C.c_void_p), cap))
        return _np.array(ms[:m], dtype=_np.float64), _np.array(tags[:m], dtype=_np.int64)

    def profiling(self):
        """Context manager around set_profile_buffer: the hook is always reset, whatever the body raises."""
        lib = self

        class _Ctx:
            def __enter__(self_inner):
                return lib.set_profile_buffer(True)

            def __exit__(self_inner, *exc):
                lib.set_profile_buffer(False)
                return False
        return _Ctx()

    def set_fused(self, enable) -> int:
        """0: two-kernel path, 1: fused kernel where it applies (default)"""
        return int(self.lib.pinn_debug_set_fused(int(bool(enable))))

    def fused_weight_limit(self) -> float:
        return float(self.lib.pinn_fused_weight_limit())

    def supported_width(self, h: int) -> int:
        return self.lib.pinn_supported_width(int(h))

    def workspace_bytes(self, layers, n, prec) -> int:
        return int(self.lib.pinn_workspace_bytes(self._ints(layers), len(layers), int(n), mode_bits(prec)))

    def min_workspace_bytes(self, layers, prec) -> int:
        return int(self.lib.pinn_min_workspace_bytes(self._ints(layers), len(layers), mode_bits(prec)))

    def wave2d_loss_grad(self, params, layers, x, y, t, n, lb, ub, normalize, E, mu, rho, plane_strain, term_weights,
                         loss_out, grad_out, accumulate, prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_wave2d_loss_grad(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub),
                                            int(bool(normalize)), float(E), float(mu), float(rho), int(bool(plane_strain)),
                                            self._floats(term_weights, 7), loss_out, grad_out, int(bool(accumulate)), mode_bits(prec),
                                            ws, int(ws_bytes), stream)
        self.check(rc, "pinn_wave2d_loss_grad")

    def wave2d_loss_grad_checked(self, params, layers, x, y, t, n, lb, ub, normalize, E, mu, rho, plane_strain, term_weights,
                                 loss_out, grad_out, prec, ws, ws_bytes, state: "RangeState", stream=0) -> int:
        """pinn_wave2d_loss_grad_checked: the call + the finite-gradient ladder, synchronous; ``state`` is kept by the caller.  Returns the rc
        (PINN_ERR_RANGE = -7 is a result a caller may want to handle, everything else raises)."""
        rc = self.lib.pinn_wave2d_loss_grad_checked(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub),
                                                    int(bool(normalize)), float(E), float(mu), float(rho), int(bool(plane_strain)),
                                                    self._floats(term_weights, 7), loss_out, grad_out, mode_bits(prec), ws, int(ws_bytes), stream,
                                                    C.byref(state))
        if rc != -7:
            self.check(rc, "pinn_wave2d_loss_grad_checked")
        return rc

    def probe_ranges(self, params, grad, n_params, ws, ws_bytes, stream=0):
        """(gradient finite?, max |w|) -- one small reduction and a stream synchronisation"""
        fin, wmax = C.c_int32(0), C.c_float(0.0)
        self.check(self.lib.pinn_probe_ranges(params or None, grad or None, int(n_params), ws, int(ws_bytes), stream, C.byref(fin), C.byref(wmax)), "pinn_probe_ranges")
        return bool(fin.value), float(wmax.value)

    def wave2d_loss_grad_profile(self, params, layers, x, y, t, n, lb, ub, normalize, E, mu, rho, plane_strain, term_weights,
                                 loss_out, grad_out, accumulate, prec, ws, ws_bytes, stream=0):
        """Synchronous; returns [repack, chain, wgrad, reductions] kernel milliseconds (HIP events)."""
        ms = (C.c_float * 4)()
        rc = self.lib.pinn_wave2d_loss_grad_profile(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb),
                                                    self._d3(ub), int(bool(normalize)), float(E), float(mu), float(rho),
                                                    int(bool(plane_strain)), self._floats(term_weights, 7), loss_out, grad_out,
                                                    int(bool(accumulate)), mode_bits(prec), ws, int(ws_bytes), stream, ms)
        self.check(rc, "pinn_wave2d_loss_grad_profile")
        return [float(v) for v in ms]

    def data_loss_grad(self, params, layers, x, y, t, n, lb, ub, normalize, targets, out_weights, loss_out, grad_out,
                       accumulate, prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_data_loss_grad(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub),
                                          int(bool(normalize)), targets, self._floats(out_weights, 8), loss_out, grad_out,
                                          int(bool(accumulate)), mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_data_loss_grad")

    def data_loss_grad_multi(self, params, layers, sets, lb, ub, normalize, grad_out, accumulate, prec, ws, ws_bytes, stream=0):
        """sets: list of (x, y, t, n, targets_or_0, out_weights, loss_out) with device pointers as integers."""
        arr = (PointSet * len(sets))()
        for k, (x, y, t, n, tg, ow, lo) in enumerate(sets):
            arr[k].x, arr[k].y, arr[k].t, arr[k].n, arr[k].targets, arr[k].loss_terms_out = x or None, y or None, t or None, int(n), tg or None, lo
            for i in range(8):
                arr[k].out_weights[i] = float(ow[i]) if i < len(ow) else 0.0
        rc = self.lib.pinn_data_loss_grad_multi(params, self._ints(layers), len(layers), arr, len(sets), self._d3(lb), self._d3(ub),
                                                int(bool(normalize)), grad_out, int(bool(accumulate)), mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_data_loss_grad_multi")

    def wave2d_step(self, params, layers, x, y, t, n, lb, ub, normalize, E, mu, rho, plane_strain, term_weights, loss_out, sets, grad_out,
                    accumulate, adam, prec, ws, ws_bytes, stream=0):
        """pinn_wave2d_step.  sets: as data_loss_grad_multi; adam: None or (m, v, lr, beta1, beta2, eps, step)."""
        arr = (PointSet * max(1, len(sets)))()
        for k, (sx, sy, st, sn, tg, ow, lo) in enumerate(sets):
            arr[k].x, arr[k].y, arr[k].t, arr[k].n, arr[k].targets, arr[k].loss_terms_out = sx or None, sy or None, st or None, int(sn), tg or None, lo
            for i in range(8):
                arr[k].out_weights[i] = float(ow[i]) if i < len(ow) else 0.0
        ad = None
        if adam is not None:
            ad = AdamState(adam[0], adam[1], float(adam[2]), float(adam[3]), float(adam[4]), float(adam[5]), int(adam[6]))
        rc = self.lib.pinn_wave2d_step(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub), int(bool(normalize)),
                                       float(E), float(mu), float(rho), int(bool(plane_strain)), self._floats(term_weights, 7), loss_out, arr, len(sets),
                                       grad_out, int(bool(accumulate)), C.byref(ad) if ad is not None else None, mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_wave2d_step")

    def plate2d_step(self, params, layers, x, y, t, n, lb, ub, normalize, frozen, E, mu, rho, term_weights, loss_out, hx, hy, ht, hn, haux, hweights,
                     hloss_out, grad_out, accumulate, adam, prec, ws, ws_bytes, stream=0):
        """pinn_plate2d_step.  adam: None or (m, v, lr, beta1, beta2, eps, step)."""
        ad = None
        if adam is not None:
            ad = AdamState(adam[0], adam[1], float(adam[2]), float(adam[3]), float(adam[4]), float(adam[5]), int(adam[6]))
        rc = self.lib.pinn_plate2d_step(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub), int(bool(normalize)), frozen,
                                        float(E), float(mu), float(rho), self._floats(term_weights, 5), loss_out, hx, hy, ht, int(hn), haux,
                                        self._floats(hweights, 2), hloss_out, grad_out, int(bool(accumulate)), C.byref(ad) if ad is not None else None,
                                        mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_plate2d_step")

    def wave2d_fields(self, params, layers, x, y, t, n, lb, ub, normalize, fields_out, prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_wave2d_fields(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub),
                                         int(bool(normalize)), fields_out, mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_wave2d_fields")

    def net_streams(self, params, layers, x, y, t, n, lb, ub, normalize, streams_out, prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_net_streams(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub),
                                       int(bool(normalize)), streams_out, mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_net_streams")

    def plate2d_loss_grad(self, params, layers, x, y, t, n, lb, ub, normalize, frozen, E, mu, rho, term_weights, loss_out, grad_out,
                          accumulate, prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_plate2d_loss_grad(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub),
                                             int(bool(normalize)), frozen, float(E), float(mu), float(rho),
                                             self._floats(term_weights, 5), loss_out, grad_out, int(bool(accumulate)), mode_bits(prec),
                                             ws, int(ws_bytes), stream)
        self.check(rc, "pinn_plate2d_loss_grad")

    def plate2d_traction_loss_grad(self, params, layers, x, y, t, n, lb, ub, normalize, aux, weights, loss_out, grad_out, accumulate,
                                   prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_plate2d_traction_loss_grad(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb),
                                                      self._d3(ub), int(bool(normalize)), aux, self._floats(weights, 2), loss_out,
                                                      grad_out, int(bool(accumulate)), mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_plate2d_traction_loss_grad")

    def stream_loss_grad(self, params, layers, x, y, t, n, lb, ub, normalize, targets, weights, loss_out, grad_out, accumulate, prec,
                         ws, ws_bytes, stream=0):
        w = [float(v) for row in weights for v in row]
        rc = self.lib.pinn_stream_loss_grad(params, self._ints(layers), len(layers), x, y, t, int(n), self._d3(lb), self._d3(ub),
                                            int(bool(normalize)), targets, (C.c_float * len(w))(*w), loss_out, grad_out,
                                            int(bool(accumulate)), mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_stream_loss_grad")

    # -- 3-D Navier-Cauchy extension (4 inputs x, y, z, t; 12 outputs) ---------------------------------------------
    def nc3d_loss_grad(self, params, layers, x, y, z, t, n, lb, ub, normalize, E, mu, rho, term_weights, loss_out, grad_out, accumulate,
                       prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_nc3d_loss_grad(params, self._ints(layers), len(layers), x, y, z, t, int(n), self._d4(lb), self._d4(ub),
                                          int(bool(normalize)), float(E), float(mu), float(rho), self._floats(term_weights, 12), loss_out,
                                          grad_out, int(bool(accumulate)), mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_nc3d_loss_grad")

    def nc3d_data_loss_grad(self, params, layers, x, y, z, t, n, lb, ub, normalize, targets, out_weights, loss_out, grad_out, accumulate,
                            prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_nc3d_data_loss_grad(params, self._ints(layers), len(layers), x, y, z, t, int(n), self._d4(lb), self._d4(ub),
                                               int(bool(normalize)), targets, self._floats(out_weights, 16), loss_out, grad_out,
                                               int(bool(accumulate)), mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_nc3d_data_loss_grad")

    def nc3d_fields(self, params, layers, x, y, z, t, n, lb, ub, normalize, fields_out, prec, ws, ws_bytes, stream=0):
        rc = self.lib.pinn_nc3d_fields(params, self._ints(layers), len(layers), x, y, z, t, int(n), self._d4(lb), self._d4(ub),
                                       int(bool(normalize)), fields_out, mode_bits(prec), ws, int(ws_bytes), stream)
        self.check(rc, "pinn_nc3d_fields")

    def adam_step(self, params, m, v, grad, n_params, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, stream=0):
        rc = self.lib.pinn_adam_step(params, m, v, grad, int(n_params), float(lr), float(beta1), float(beta2), float(eps),
                                     int(step), stream)
        self.check(rc, "pinn_adam_step")
