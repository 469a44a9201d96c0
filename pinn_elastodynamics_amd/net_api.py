"""Pieces of the reference's model-class surface that the three host classes share: weight initialisation as METHODS
(initialize_NN / xavier_init, INF:141-156), neural_net(X, weights, biases) on weights the caller hands in (INF:188-199 -- PLATE:322-356
runs three nets through that one function), and the [W_list, b_list] checkpoint files (INF:159-186).

Checkpoints: ``.npz`` (arrays W0, b0, W1, ... + layers) is this package's documented default -- plain data.  Any other file name is the
reference's format, a pickle of ``[W_list, b_list]``; it is READ through an unpickler that resolves nothing but numpy's array
reconstruction (a pickle from an untrusted source cannot run code here: anything else raises), and written with pickle as the
reference does, so that its scripts keep loading what this package saves.
"""
from __future__ import annotations

import io
import pickle
from typing import Sequence

import numpy as np

_NUMPY_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
}


def _latin1_bytes(text, encoding="latin1"):
    """what protocol <= 2 pickles written by Python 3 call to rebuild a bytes object (``_codecs.encode(str, 'latin1')``) -- that one use only"""
    if not isinstance(text, str) or str(encoding).lower().replace("-", "").replace("_", "") != "latin1":
        raise pickle.UnpicklingError("checkpoint: unexpected _codecs.encode call")
    return text.encode("latin1")


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """lists / tuples of numpy arrays and nothing else"""

    def find_class(self, module, name):
        if (module, name) in _NUMPY_GLOBALS:
            return super().find_class(module, name)
        if (module, name) == ("_codecs", "encode"):
            return _latin1_bytes
        raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}: only numpy arrays are accepted in a [W_list, b_list] file")


def read_checkpoint(fileDir):
    """-> (weights, biases) as lists of arrays, from ``.npz`` or from the reference's pickle (arrays-only unpickler)"""
    if str(fileDir).endswith(".npz"):
        z = np.load(fileDir, allow_pickle=False)
        n = sum(1 for k in z.files if k.startswith("W"))
        return [z[f"W{i}"] for i in range(n)], [z[f"b{i}"] for i in range(n)]
    with open(fileDir, "rb") as f:
        weights, biases = _ArraysOnlyUnpickler(io.BytesIO(f.read()), encoding="latin1").load()
    return list(weights), list(biases)


def write_checkpoint(fileDir, weights, biases, layers):
    if str(fileDir).endswith(".npz"):
        np.savez(fileDir, layers=np.array(layers), **{f"W{i}": w for i, w in enumerate(weights)}, **{f"b{i}": x for i, x in enumerate(biases)})
    else:
        with open(fileDir, "wb") as f:
            pickle.dump([list(weights), list(biases)], f)


def truncated_normal(rng: np.random.Generator, shape, stddev: float) -> np.ndarray:
    """tf.truncated_normal (INF:156): normal draws, redrawn while |z| > 2 standard deviations"""
    W = rng.standard_normal(shape)
    bad = np.abs(W) > 2.0
    while bad.any():
        W[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(W) > 2.0
    return (W * stddev).astype(np.float32)


class NetApi:
    """Mixin: expects ``self._init_rng`` (numpy Generator), ``self.uv_layers``, ``self.engine`` and ``self._net_fields(engine, theta, X)``
    (the value-stream outputs [n_out, N] of a net on points X [N, d_in])."""

    def initialize_NN(self, layers: Sequence[int]):
        """INF:141-150: (weights, biases) -- Xavier truncated-normal matrices [in, out], zero biases [1, out] -- from the model's seeded
        stream (TF1's own stream is not reproducible)."""
        weights, biases = [], []
        for l in range(len(layers) - 1):
            weights.append(self.xavier_init(size=[layers[l], layers[l + 1]]))
            biases.append(np.zeros((1, int(layers[l + 1])), dtype=np.float32))
        return weights, biases

    def xavier_init(self, size):
        """INF:152-156: one [in, out] matrix, truncated normal with stddev sqrt(2 / (in + out))"""
        in_dim, out_dim = int(size[0]), int(size[1])
        return truncated_normal(self._init_rng, (in_dim, out_dim), float(np.sqrt(2.0 / (in_dim + out_dim))))

    def _engine_for(self, layers):
        """the engine that evaluates a net of these layer sizes: the model's own when they are its net's, else a sibling built on demand"""
        layers = [int(v) for v in layers]
        if layers == [int(v) for v in self.engine.layers]:
            return self.engine
        cache = self.__dict__.setdefault("_sibling_engines", {})
        key = tuple(layers)
        if key not in cache:
            cache[key] = self.engine.for_layers(layers)
        return cache[key]

    def neural_net(self, X, weights=None, biases=None):
        """INF:188-199 on X [N, d_in] -> Y [N, n_out], with the weights handed in (lists as initialize_NN / load_NN return them) or, when
        none are given, the model's current ones.  The input normalisation is the model's (INF:191 on, SEMI:198 / PLATE:312 off)."""
        import torch
        from .elastic_wave import pack_params
        X = np.asarray(X)
        if weights is None:
            eng, theta = self.engine, self._current_theta()
        else:
            if biases is None or len(biases) != len(weights):
                raise ValueError("neural_net(X, weights, biases): one bias per weight matrix is required when weights are given (INF:188-199)")
            layers = [int(np.asarray(weights[0]).shape[0])] + [int(np.asarray(w).shape[1]) for w in weights]
            d_in = int(self.engine.layers[0])
            if layers[0] != d_in or X.ndim != 2 or X.shape[1] != d_in:
                # (the value-stream kernels behind this class take its own input columns: 3 = (x, y, t), the 3-D class 4)
                raise ValueError(f"neural_net: this model's kernels evaluate nets on {d_in} input columns; got X {tuple(X.shape)} and a first weight "
                                 f"matrix with {layers[0]} rows")
            try:
                eng = self._engine_for(layers)
            except Exception as e:      # unsupported width / output count of the sibling net: say which
                raise ValueError(f"neural_net: no kernel variant for a net of layers {layers} (hidden width <= 160, one width for all hidden layers, "
                                 f"<= 8 outputs -- 16 for the 4-input class): {e}") from e
            theta = torch.from_numpy(pack_params(weights, biases)).to(eng.device)
        return self._net_fields(eng, theta, X).T.detach().cpu().numpy()
