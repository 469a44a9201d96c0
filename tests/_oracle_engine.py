"""CPU stand-in for HipEngine used ONLY by the CPU-side host-logic tests (tests may call the
oracle; the product never does).  Same method surface, float64 numpy oracle underneath."""
import numpy as np
import torch

from oracle import pinn_oracle as po
from oracle import plate_oracle as pl


class OracleEngine:
    def __init__(self, layers):
        self.layers = [int(v) for v in layers]
        self.device = torch.device("cpu")
        self.n_params = po.param_count(self.layers)
        self.calls = []
        self.adjoint_shift = 0

    def for_layers(self, layers):
        return OracleEngine(layers)

    @staticmethod
    def _np(t):
        return t.detach().cpu().numpy().astype(np.float64)

    def wave_loss_grad(self, params, x, y, t, lb, ub, normalize, term_weights, E=2.5, mu=0.25, rho=1.0, plane_strain=True,
                       grad_out=None, accumulate=False, loss_out=None, packed=False):
        self.calls.append(("wave", x.numel()))
        ss, g, _ = po.wave2d_loss_grad(self._np(params), self.layers, self._np(x), self._np(y), self._np(t), lb, ub, normalize,
                                       E, mu, rho, plane_strain, np.asarray(term_weights, dtype=np.float64))
        if grad_out is None:
            grad_out = torch.zeros(self.n_params, dtype=torch.float32)
            accumulate = False
        if loss_out is None:
            loss_out = torch.zeros(8, dtype=torch.float32)
        gt = torch.from_numpy(g.astype(np.float32))
        grad_out.copy_(grad_out + gt if accumulate else gt)
        loss_out[:7].copy_(torch.from_numpy(ss.astype(np.float32)))
        return loss_out[:7], grad_out

    def data_loss_grad(self, params, x, y, t, lb, ub, normalize, targets, out_weights, grad_out=None, accumulate=False, loss_out=None, packed=False):
        self.calls.append(("data", x.numel()))
        nout = self.layers[-1]
        tg = None if targets is None else self._np(targets).T
        ss, g, _ = po.data_loss_grad(self._np(params), self.layers, self._np(x), self._np(y), self._np(t), lb, ub, normalize, tg,
                                     np.asarray(out_weights, dtype=np.float64)[:nout])
        if grad_out is None:
            grad_out = torch.zeros(self.n_params, dtype=torch.float32)
            accumulate = False
        if loss_out is None:
            loss_out = torch.zeros(8, dtype=torch.float32)
        gt = torch.from_numpy(g.astype(np.float32))
        grad_out.copy_(grad_out + gt if accumulate else gt)
        loss_out[:nout].copy_(torch.from_numpy(ss.astype(np.float32)))
        return loss_out[:nout], grad_out

    def fields(self, params, x, y, t, lb, ub, normalize):
        out = po.wave2d_fields(self._np(params), self.layers, self._np(x), self._np(y), self._np(t), lb, ub, normalize)
        F = np.stack([out["Y"].T] + [d.T for d in out["dY"]])
        return torch.from_numpy(F.astype(np.float32))

    def net_streams(self, params, x, y, t, lb, ub, normalize):
        return torch.from_numpy(pl.net_streams(self._np(params), self.layers, self._np(x), self._np(y), self._np(t)).astype(np.float32))

    def _put(self, g, ss, n, grad_out, accumulate, loss_out):
        if grad_out is None:
            grad_out = torch.zeros(self.n_params, dtype=torch.float32)
            accumulate = False
        if loss_out is None:
            loss_out = torch.zeros(8, dtype=torch.float32)
        gt = torch.from_numpy(g.astype(np.float32))
        grad_out.copy_(grad_out + gt if accumulate else gt)
        loss_out[:n].copy_(torch.from_numpy(np.asarray(ss, dtype=np.float32)))
        return loss_out[:n], grad_out

    def plate_loss_grad(self, params, x, y, t, lb, ub, normalize, frozen, term_weights, E=20.0, mu=0.25, rho=1.0,
                        grad_out=None, accumulate=False, loss_out=None, packed=False):
        fr = self._np(frozen)
        ss, g, _ = pl.plate_loss_grad(self._np(params), self.layers, self._np(x), self._np(y), self._np(t), fr[0], fr[1], E, mu, rho,
                                      np.asarray(term_weights, dtype=np.float64))
        return self._put(g, ss, 5, grad_out, accumulate, loss_out)

    def traction_loss_grad(self, params, x, y, t, lb, ub, normalize, aux, weights, grad_out=None, accumulate=False, loss_out=None, packed=False):
        a = self._np(aux)
        ss, g = pl.traction_loss_grad(self._np(params), self.layers, self._np(x), self._np(y), self._np(t), a[0:5], a[5:10], 0.1, float(weights[0]))
        return self._put(g, ss, 2, grad_out, accumulate, loss_out)

    def stream_loss_grad(self, params, x, y, t, lb, ub, normalize, targets, weights, grad_out=None, accumulate=False, loss_out=None, packed=False):
        w = np.asarray(weights, dtype=np.float64)
        ss, g = pl.stream_loss_grad(self._np(params), self.layers, self._np(x), self._np(y), self._np(t),
                                    None if targets is None else self._np(targets), w)
        return self._put(g, ((w / np.abs(w).max()) * ss).sum(0), self.layers[-1], grad_out, accumulate, loss_out)

    # ---- 3-D Navier-Cauchy extension -----------------------------------------------------------------------
    def nc3d_loss_grad(self, params, x, y, z, t, lb, ub, normalize, term_weights, E=2.5, mu=0.25, rho=1.0,
                       grad_out=None, accumulate=False, loss_out=None, packed=False):
        from oracle import nc3d_oracle as n3
        ss, g, _ = n3.nc3d_loss_grad(self._np(params), self.layers, self._np(x), self._np(y), self._np(z), self._np(t), lb, ub, normalize,
                                     E, mu, rho, np.asarray(term_weights, dtype=np.float64))
        if loss_out is None:
            loss_out = torch.zeros(16, dtype=torch.float32)
        return self._put(g, ss, 12, grad_out, accumulate, loss_out)

    def nc3d_data_loss_grad(self, params, x, y, z, t, lb, ub, normalize, targets, out_weights, grad_out=None, accumulate=False,
                            loss_out=None, packed=False):
        from oracle import nc3d_oracle as n3
        tg = None if targets is None else self._np(targets).T
        ss, g, _ = n3.nc3d_data_loss_grad(self._np(params), self.layers, self._np(x), self._np(y), self._np(z), self._np(t), lb, ub,
                                          normalize, tg, np.asarray(out_weights, dtype=np.float64)[:12])
        if loss_out is None:
            loss_out = torch.zeros(16, dtype=torch.float32)
        return self._put(g, ss, 12, grad_out, accumulate, loss_out)

    def nc3d_fields(self, params, x, y, z, t, lb, ub, normalize):
        from oracle import nc3d_oracle as n3
        out = n3.nc3d_fields(self._np(params), self.layers, self._np(x), self._np(y), self._np(z), self._np(t), lb, ub, normalize)
        return torch.from_numpy(np.stack([out["Y"].T] + [d.T for d in out["dY"]]).astype(np.float32))

    def data_loss_grad_multi(self, params, sets, lb, ub, normalize, grad_out, accumulate=False, packed=False):
        for x, y, t, tg, ow, lo in sets:
            self.data_loss_grad(params, x, y, t, lb, ub, normalize, tg, ow, grad_out=grad_out, accumulate=accumulate, loss_out=lo)
            accumulate = True
        return grad_out

    def adam_step(self, params, m, v, grad, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
        th, mm, vv = po.adam_tf1_step(self._np(params), self._np(grad), self._np(m), self._np(v), step, lr, beta1, beta2, eps)
        params.copy_(torch.from_numpy(th.astype(np.float32)))
        m.copy_(torch.from_numpy(mm.astype(np.float32)))
        v.copy_(torch.from_numpy(vv.astype(np.float32)))
