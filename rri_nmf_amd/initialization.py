"""Starting points for W, T (mirror of /root/reference/src/rri_nmf/initialization.py:9-163).

One-off host step (out of the kernel scope, 'next' row f-2): a thin wrapper over
scikit-learn's randomized SVD, exactly the dependency the reference uses.
"""
from math import sqrt

import numpy as np
from sklearn.utils import check_random_state
from sklearn.utils.extmath import randomized_svd, squared_norm

from .matrixops import normalize

_KNOWN = (None, 'random', 'smart_random', 'nndsvd', 'nndsvda', 'nndsvdar')


def _split_pos_neg(v):
    return np.maximum(v, 0), np.abs(np.minimum(v, 0))


def initialize_nmf(X, n_components, init=None, eps=1e-6, random_state=None, row_normalize=False,
                   n_words_beam=20):
    """W (n x k), H (k x d) >= 0.  'random' | 'smart_random' | 'nndsvd' | 'nndsvda' | 'nndsvdar'."""
    n, d = X.shape
    k = n_components
    # X may be a scipy sparse matrix: randomized_svd and .mean() take it as it is
    if init is None:
        init = 'nndsvd' if k < d else 'random'
    if init == 'random':                      # initialization.py:80-87 (T drawn first)
        rng = check_random_state(random_state)
        H = rng.rand(k, d)
        W = rng.rand(n, k)
        return W, (normalize(H) if row_normalize else H)
    if init == 'smart_random':                # initialization.py:90-102
        rng = check_random_state(random_state)
        scale = np.sqrt(X.mean() / k)
        H = np.abs(scale * rng.randn(k, d))
        W = np.abs(scale * rng.randn(n, k))
        return W, (normalize(H) if row_normalize else H)
    if init not in _KNOWN:
        # the reference runs the SVD first and raises afterwards (initialization.py:153-157)
        raise ValueError('Invalid init parameter: got %r instead of one of %r' % (init, _KNOWN[:1] + _KNOWN[3:]))

    U, S, V = randomized_svd(X, k, random_state=random_state)
    W, H = np.zeros(U.shape), np.zeros(V.shape)
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])         # leading pair is sign-definite (:109-111)
    H[0, :] = np.sqrt(S[0]) * np.abs(V[0, :])
    for j in range(1, k):                              # Boutsidis & Gallopoulos NNDSVD (:114-140)
        up, un = _split_pos_neg(U[:, j])
        vp, vn = _split_pos_neg(V[j, :])
        nup, nvp = sqrt(squared_norm(up)), sqrt(squared_norm(vp))
        nun, nvn = sqrt(squared_norm(un)), sqrt(squared_norm(vn))
        if nup * nvp > nun * nvn:
            u, v, sigma = up / nup, vp / nvp, nup * nvp
        else:
            u, v, sigma = un / nun, vn / nvn, nun * nvn
        scale = np.sqrt(S[j] * sigma)
        W[:, j] = scale * u
        H[j, :] = scale * v
    W[W < eps] = 0
    H[H < eps] = 0
    if init == 'nndsvda':
        fill = X.mean()
        W[W == 0] = fill
        H[H == 0] = fill
    elif init == 'nndsvdar':
        rng = check_random_state(random_state)
        fill = X.mean()
        W[W == 0] = abs(fill * rng.randn(len(W[W == 0])) / 100)
        H[H == 0] = abs(fill * rng.randn(len(H[H == 0])) / 100)
    return W, (normalize(H) if row_normalize else H)
