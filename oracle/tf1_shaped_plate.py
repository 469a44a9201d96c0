"""torch-CPU restatement of the plate graph written like the reference's TF1 code: three nets, composite P + D*N, nested
tf.gradients for u_tt / v_tt (PLATE:427-433), plane stress, net_t.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Second, independent differentiation route used to pin oracle/plate_oracle.py."""
from __future__ import annotations

import torch


def _grad(out, inp):
    return torch.autograd.grad(out, inp, grad_outputs=torch.ones_like(out), create_graph=True)[0]


class TF1ShapedPlate:
    def __init__(self, uv, dist, part, E=20.0, mu=0.25, rho=1.0, hole_r=0.1, dtype=torch.float64):
        """uv/dist/part: (weights list [in,out], biases list)."""
        def mk(W, b, train):
            Wt = [torch.as_tensor(w, dtype=dtype).clone().requires_grad_(train) for w in W]
            bt = [torch.as_tensor(x, dtype=dtype).reshape(1, -1).clone().requires_grad_(train) for x in b]
            return Wt, bt
        self.uv_weights, self.uv_biases = mk(*uv, True)
        self.dist_weights, self.dist_biases = mk(*dist, False)
        self.part_weights, self.part_biases = mk(*part, False)
        self.E, self.mu, self.rho, self.hole_r, self.dtype = E, mu, rho, hole_r, dtype

    @staticmethod
    def neural_net(X, weights, biases):                         # PLATE:308-320
        H = X
        for W, b in zip(weights[:-1], biases[:-1]):
            H = torch.tanh(torch.add(torch.matmul(H, W), b))
        return torch.add(torch.matmul(H, weights[-1]), biases[-1])

    def net_uv(self, x, y, t):                                  # PLATE:358-388
        X = torch.cat([x, y, t], 1)
        uv_sig = self.neural_net(X, self.uv_weights, self.uv_biases)
        dist = self.neural_net(X, self.dist_weights, self.dist_biases)
        part = self.neural_net(X, self.part_weights, self.part_biases)
        return tuple(part[:, i:i + 1] + dist[:, i:i + 1] * uv_sig[:, i:i + 1] for i in range(5))

    def net_e(self, x, y, t):                                   # PLATE:390-396
        u, v, _, _, _ = self.net_uv(x, y, t)
        return _grad(u, x), _grad(v, y), _grad(u, y) + _grad(v, x)

    def net_f_sig(self, x, y, t):                               # PLATE:404-439
        E, mu, rho = self.E, self.mu, self.rho
        u, v, s11, s22, s12 = self.net_uv(x, y, t)
        e11, e22, e12 = self.net_e(x, y, t)
        sp11 = E / (1 - mu * mu) * e11 + E * mu / (1 - mu * mu) * e22
        sp22 = E * mu / (1 - mu * mu) * e11 + E / (1 - mu * mu) * e22
        sp12 = E / (2 * (1 + mu)) * e12
        f_s11, f_s12, f_s22 = s11 - sp11, s12 - sp12, s22 - sp22
        s11_1, s12_2 = _grad(s11, x), _grad(s12, y)
        u_tt = _grad(_grad(u, t), t)
        s22_2, s12_1 = _grad(s22, y), _grad(s12, x)
        v_tt = _grad(_grad(v, t), t)
        return s11_1 + s12_2 - rho * u_tt, s22_2 + s12_1 - rho * v_tt, f_s11, f_s22, f_s12

    def net_t(self, x, y, t):                                   # PLATE:452-461
        r = self.hole_r
        u, v, s11, s22, s12 = self.net_uv(x, y, t)
        nx, ny = -x / r, -y / r
        return s11 * nx + s12 * ny, s12 * nx + s22 * ny

    def _cols(self, xyt):
        xyt = torch.as_tensor(xyt, dtype=self.dtype)
        return tuple(xyt[:, i:i + 1].clone().requires_grad_(True) for i in range(3))

    def loss_and_grad(self, collo, hole):
        """loss = 10 (loss_f_uv + loss_f_s + loss_HOLE) (PLATE:187-193,217) and its gradient w.r.t. the uv net only."""
        f = self.net_f_sig(*self._cols(collo))
        loss_f_uv = sum(torch.mean(torch.square(r)) for r in f[:2])
        loss_f_s = sum(torch.mean(torch.square(r)) for r in f[2:])
        tx, ty = self.net_t(*self._cols(hole))
        loss_hole = torch.mean(torch.square(tx)) + torch.mean(torch.square(ty))
        loss = 10 * (loss_f_uv + loss_f_s + loss_hole)
        grads = torch.autograd.grad(loss, self.uv_weights + self.uv_biases)
        n = len(self.uv_weights)
        parts = []
        for W, b in zip(grads[:n], grads[n:]):
            parts += [W.reshape(-1), b.reshape(-1)]
        return dict(loss_f_uv=float(loss_f_uv), loss_f_s=float(loss_f_s), loss_HOLE=float(loss_hole), loss=float(loss)), torch.cat(parts), f
