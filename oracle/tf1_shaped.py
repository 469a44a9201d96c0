"""torch-CPU restatement written op-for-op like the reference's TF1 graph.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

This is the second, independent differentiation route used to pin
``oracle/pinn_oracle.py`` (closed-form forward tangents) and it is also the
"TF1-graph-shaped" CPU baseline timed by ``bench.py``:
TensorFlow 1.x cannot be installed in this image, so the reference's CPU path
is timed as this restatement -- forward graph instantiated twice (net_uv inside
net_f_sig and again inside net_e, INF:227,230), twelve separate reverse passes
for the twelve Jacobian entries (INF:216-218,248-259), seven mean-squares
(INF:104-110) and then the gradient w.r.t. every weight/bias (INF:131-133).

INF = /root/reference/ElasticWaveInfinite/ElasticWave.py
"""
from __future__ import annotations

import torch


def _grad(out, inp):
    # tf.gradients(out, inp)[0] with an implicit all-ones seed (INF:216)
    return torch.autograd.grad(out, inp, grad_outputs=torch.ones_like(out), create_graph=True)[0]


class TF1ShapedWave:
    """Mirrors class DeepHPM's graph pieces (INF:188-265) on torch tensors."""

    def __init__(self, weights, biases, lb, ub, normalize=True, E=2.5, mu=0.25, rho=1.0,
                 dtype=torch.float64):
        self.dtype = dtype
        self.weights = [torch.as_tensor(W, dtype=dtype).clone().requires_grad_(True) for W in weights]
        self.biases = [torch.as_tensor(b, dtype=dtype).reshape(1, -1).clone().requires_grad_(True) for b in biases]
        self.lb = torch.as_tensor(lb, dtype=dtype)
        self.ub = torch.as_tensor(ub, dtype=dtype)
        self.normalize = normalize
        self.E, self.mu, self.rho = E, mu, rho

    def neural_net(self, X):                                   # INF:188-199
        H = 2.0 * (X - self.lb) / (self.ub - self.lb) - 1.0 if self.normalize else X
        for W, b in zip(self.weights[:-1], self.biases[:-1]):
            H = torch.tanh(torch.add(torch.matmul(H, W), b))
        return torch.add(torch.matmul(H, self.weights[-1]), self.biases[-1])

    def net_uv(self, x, y, t):                                 # INF:201-211
        uv_sig = self.neural_net(torch.cat([x, y, t], 1))
        return tuple(uv_sig[:, i:i + 1] for i in range(7))

    def net_e(self, x, y, t):                                  # INF:213-219
        u, v, _, _, _, _, _ = self.net_uv(x, y, t)
        e11 = _grad(u, x)
        e22 = _grad(v, y)
        e12 = _grad(u, y) + _grad(v, x)
        return e11, e22, e12

    def net_f_sig(self, x, y, t):                              # INF:221-265
        E, mu, rho = self.E, self.mu, self.rho
        u, v, ut, vt, s11, s22, s12 = self.net_uv(x, y, t)
        e11, e22, e12 = self.net_e(x, y, t)
        coef = E / ((1 + mu) * (1 - 2 * mu))
        sp11 = coef * (1 - mu) * e11 + coef * mu * e22
        sp22 = coef * mu * e11 + coef * (1 - mu) * e22
        sp12 = E / (2 * (1 + mu)) * e12
        f_s11 = s11 - sp11
        f_s12 = s12 - sp12
        f_s22 = s22 - sp22
        f_ut = _grad(u, t) - ut
        f_vt = _grad(v, t) - vt
        s11_1 = _grad(s11, x)
        s12_2 = _grad(s12, y)
        u_tt = _grad(ut, t)
        s22_2 = _grad(s22, y)
        s12_1 = _grad(s12, x)
        v_tt = _grad(vt, t)
        f_u = s11_1 + s12_2 - rho * u_tt
        f_v = s22_2 + s12_1 - rho * v_tt
        return f_u, f_v, f_ut, f_vt, f_s11, f_s22, f_s12

    def residual_losses(self, xyt):
        """(loss_f_uv, loss_f_s) of INF:104-110 on collocation points xyt [N,3]."""
        xyt = torch.as_tensor(xyt, dtype=self.dtype)
        x = xyt[:, 0:1].clone().requires_grad_(True)
        y = xyt[:, 1:2].clone().requires_grad_(True)
        t = xyt[:, 2:3].clone().requires_grad_(True)
        f = self.net_f_sig(x, y, t)
        loss_f_uv = sum(torch.mean(torch.square(r)) for r in f[:4])
        loss_f_s = sum(torch.mean(torch.square(r)) for r in f[4:])
        return loss_f_uv, loss_f_s, f

    def loss_and_grad(self, xyt, w_f_uv=1.0, w_f_s=1.0):
        """loss = w_f_uv*loss_f_uv + w_f_s*loss_f_s and its gradient w.r.t. all
        weights and biases (what optimizer_Adam.minimize differentiates, INF:131-133).
        Returns (loss_f_uv, loss_f_s, [dW...], [db...], residual tuple)."""
        l_uv, l_s, f = self.residual_losses(xyt)
        loss = w_f_uv * l_uv + w_f_s * l_s
        grads = torch.autograd.grad(loss, self.weights + self.biases)
        n = len(self.weights)
        return l_uv.detach(), l_s.detach(), list(grads[:n]), list(grads[n:]), f

    def flat_grad(self, xyt, w_f_uv=1.0, w_f_s=1.0):
        l_uv, l_s, gW, gb, f = self.loss_and_grad(xyt, w_f_uv, w_f_s)
        parts = []
        for W, b in zip(gW, gb):
            parts += [W.reshape(-1), b.reshape(-1)]
        return l_uv, l_s, torch.cat(parts), f


class MinimalWave(TF1ShapedWave):
    """Same numbers by the minimal algorithm (one forward with three
    jvp tangents + one reverse pass), used only as the second CPU timing leg."""

    def residual_losses(self, xyt):
        xyt = torch.as_tensor(xyt, dtype=self.dtype)
        N = xyt.shape[0]
        sc = 2.0 / (self.ub - self.lb) if self.normalize else torch.ones(3, dtype=self.dtype)
        h = 2.0 * (xyt - self.lb) / (self.ub - self.lb) - 1.0 if self.normalize else xyt
        dh = [torch.zeros(N, 3, dtype=self.dtype) for _ in range(3)]
        for k in range(3):
            dh[k][:, k] = sc[k]
        for W, b in zip(self.weights[:-1], self.biases[:-1]):
            h = torch.tanh(h @ W + b)
            s = 1.0 - h * h
            dh = [s * (d @ W) for d in dh]
        Y = h @ self.weights[-1] + self.biases[-1]
        Jx, Jy, Jt = [d @ self.weights[-1] for d in dh]
        E, mu, rho = self.E, self.mu, self.rho
        coef = E / ((1 + mu) * (1 - 2 * mu))
        c1, c2, G = coef * (1 - mu), coef * mu, E / (2 * (1 + mu))
        e11, e22, e12 = Jx[:, 0], Jy[:, 1], Jy[:, 0] + Jx[:, 1]
        f = (Jx[:, 4] + Jy[:, 6] - rho * Jt[:, 2], Jy[:, 5] + Jx[:, 6] - rho * Jt[:, 3],
             Jt[:, 0] - Y[:, 2], Jt[:, 1] - Y[:, 3],
             Y[:, 4] - (c1 * e11 + c2 * e22), Y[:, 5] - (c2 * e11 + c1 * e22), Y[:, 6] - G * e12)
        loss_f_uv = sum(torch.mean(torch.square(r)) for r in f[:4])
        loss_f_s = sum(torch.mean(torch.square(r)) for r in f[4:])
        return loss_f_uv, loss_f_s, f
