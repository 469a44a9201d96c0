"""torch-CPU second route for the 3-D Navier-Cauchy head: reverse-mode autograd, written the way the reference writes its 2-D
graph (one ``tf.gradients`` per Jacobian entry, INF:216-218,248-259; forward built again inside net_e, INF:214,227).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Independent of oracle/nc3d_oracle.py (no shared arithmetic): it pins the
closed-form forward-tangent route (tests/test_oracle_nc3d.py, agreement ~1e-13) and is the CPU baseline of ``bench.py --config nc3d``.
The 3-D case itself is a build-side extension (parity unpinned, see nc3d_oracle.py).
"""
from __future__ import annotations

import numpy as np
import torch


def _grad(out, inp):
    return torch.autograd.grad(out, inp, grad_outputs=torch.ones_like(out), create_graph=True)[0]


class TF1ShapedNC3D:
    def __init__(self, weights, biases, lb, ub, normalize=True, E=2.5, mu=0.25, rho=1.0, dtype=torch.float64):
        self.dtype = dtype
        self.weights = [torch.as_tensor(W, dtype=dtype).clone().requires_grad_(True) for W in weights]
        self.biases = [torch.as_tensor(b, dtype=dtype).reshape(1, -1).clone().requires_grad_(True) for b in biases]
        self.lb = torch.as_tensor(lb, dtype=dtype)
        self.ub = torch.as_tensor(ub, dtype=dtype)
        self.normalize = normalize
        self.E, self.mu, self.rho = E, mu, rho

    def neural_net(self, X):
        H = 2.0 * (X - self.lb) / (self.ub - self.lb) - 1.0 if self.normalize else X
        for W, b in zip(self.weights[:-1], self.biases[:-1]):
            H = torch.tanh(torch.add(torch.matmul(H, W), b))
        return torch.add(torch.matmul(H, self.weights[-1]), self.biases[-1])

    def net_uv(self, x, y, z, t):
        out = self.neural_net(torch.cat([x, y, z, t], 1))
        return tuple(out[:, i:i + 1] for i in range(12))

    def net_e(self, x, y, z, t):
        u, v, w = self.net_uv(x, y, z, t)[:3]
        e11, e22, e33 = _grad(u, x), _grad(v, y), _grad(w, z)
        e12 = _grad(u, y) + _grad(v, x)
        e13 = _grad(u, z) + _grad(w, x)
        e23 = _grad(v, z) + _grad(w, y)
        return e11, e22, e33, e12, e13, e23

    def net_f_sig(self, x, y, z, t):
        E, mu, rho = self.E, self.mu, self.rho
        u, v, w, ut, vt, wt, s11, s22, s33, s12, s13, s23 = self.net_uv(x, y, z, t)
        e11, e22, e33, e12, e13, e23 = self.net_e(x, y, z, t)
        lam = E * mu / ((1 + mu) * (1 - 2 * mu))
        G = E / (2 * (1 + mu))
        tr = e11 + e22 + e33
        f_s11 = s11 - (lam * tr + 2 * G * e11)
        f_s22 = s22 - (lam * tr + 2 * G * e22)
        f_s33 = s33 - (lam * tr + 2 * G * e33)
        f_s12 = s12 - G * e12
        f_s13 = s13 - G * e13
        f_s23 = s23 - G * e23
        f_ut = _grad(u, t) - ut
        f_vt = _grad(v, t) - vt
        f_wt = _grad(w, t) - wt
        f_u = _grad(s11, x) + _grad(s12, y) + _grad(s13, z) - rho * _grad(ut, t)
        f_v = _grad(s12, x) + _grad(s22, y) + _grad(s23, z) - rho * _grad(vt, t)
        f_w = _grad(s13, x) + _grad(s23, y) + _grad(s33, z) - rho * _grad(wt, t)
        return f_u, f_v, f_w, f_ut, f_vt, f_wt, f_s11, f_s22, f_s33, f_s12, f_s13, f_s23

    def _cols(self, X):
        X = torch.as_tensor(np.asarray(X), dtype=self.dtype)
        return [X[:, i:i + 1].clone().requires_grad_(True) for i in range(4)]

    def residuals(self, X):
        return torch.cat(self.net_f_sig(*self._cols(X)), 1).detach().numpy()

    def flat_grad(self, X, term_weights=None):
        """(sumsq [12], flat gradient of sum_i w_i sumsq_i in the C-ABI's parameter order W0, b0, W1, b1, ...)."""
        f = torch.cat(self.net_f_sig(*self._cols(X)), 1)
        ss = (f * f).sum(0)
        w = torch.ones(12, dtype=self.dtype) if term_weights is None else torch.as_tensor(np.asarray(term_weights), dtype=self.dtype)
        loss = (ss * w).sum()
        gs = torch.autograd.grad(loss, self.weights + self.biases)
        L = len(self.weights)
        parts = []
        for l in range(L):
            parts.append(gs[l].reshape(-1))
            parts.append(gs[L + l].reshape(-1))
        return ss.detach().numpy(), torch.cat(parts).detach().numpy()
