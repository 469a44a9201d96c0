"""CPU stand-in for HipEngine used ONLY by the CPU-side host-logic tests (tests may call the
oracle; the product never does).  Same method surface, float64 numpy oracle underneath."""
import numpy as np
import torch

from oracle import pinn_oracle as po


class OracleEngine:
    def __init__(self, layers):
        self.layers = [int(v) for v in layers]
        self.device = torch.device("cpu")
        self.n_params = po.param_count(self.layers)
        self.calls = []

    @staticmethod
    def _np(t):
        return t.detach().cpu().numpy().astype(np.float64)

    def wave_loss_grad(self, params, x, y, t, lb, ub, normalize, term_weights, E=2.5, mu=0.25, rho=1.0, plane_strain=True,
                       grad_out=None, accumulate=False, loss_out=None):
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

    def data_loss_grad(self, params, x, y, t, lb, ub, normalize, targets, out_weights, grad_out=None, accumulate=False, loss_out=None):
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

    def adam_step(self, params, m, v, grad, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
        th, mm, vv = po.adam_tf1_step(self._np(params), self._np(grad), self._np(m), self._np(v), step, lr, beta1, beta2, eps)
        params.copy_(torch.from_numpy(th.astype(np.float32)))
        m.copy_(torch.from_numpy(mm.astype(np.float32)))
        v.copy_(torch.from_numpy(vv.astype(np.float32)))
