"""Host-side matrix helpers of the RRI path (mirror of the reference's matrixops module:
same function names and argument meaning, /root/reference/src/rri_nmf/matrixops.py).

These run once, before or after the device loop (preprocessing, validation of the starting
point, label helpers).  The projections INSIDE the loop -- every T row per topic step, the W rows per
sweep / at the end -- are device kernels (k_trow_final, k_proj_rows), not these functions.
"""
import numpy as np
import scipy.sparse as sp


def euclidean_proj_simplex(v_in, s=1):
    """Projection of one vector on {w >= 0, sum(w) = s} (matrixops.py:5-69): threshold
    theta = (sum of the rho largest entries - s) / rho from the descending sort."""
    assert s > 0, "Radius s must be strictly positive (%d <= 0)" % s
    sparse_in = sp.issparse(v_in)
    v = (v_in.toarray() if sparse_in else np.asarray(v_in)).reshape(-1)
    if v.sum() == s and np.all(v >= 0):      # already feasible: returned untouched (:53-55)
        return v
    u = np.sort(v)[::-1]
    css = np.cumsum(u)
    rho = np.nonzero(u * np.arange(1, v.size + 1) > (css - s))[0][-1]
    theta = (css[rho] - s) / (rho + 1.0)
    w = np.clip(v - theta, 0, None).reshape(v_in.shape)
    return sp.csr_matrix(w) if sparse_in else w


def proj_mat_to_simplex(W, s=1.0, axis=1):
    """Row-wise (axis=1) / column-wise (axis=0) simplex projection, in place (matrixops.py:72-100)."""
    if axis == 0:
        return proj_mat_to_simplex(W.T, s, axis=1).T
    scalar = np.isscalar(s)
    if not scalar:
        assert s.size == W.shape[0], ('proj_mat_to_simplex: expected s to have size {n} but s has '
                                      'size {s}'.format(n=W.shape[0], s=s.size))
    for i in range(W.shape[0]):
        W[i, :] = euclidean_proj_simplex(W[i, :], s if scalar else s[i])
    return W


def normalize(X, dim=1, zero_sum_fix=True):
    """Rows (dim=1) or columns (dim=0) scaled to sum 1; empty ones become uniform (matrixops.py:124-163)."""
    if dim not in (0, 1):
        raise Exception('Unknown dim=%r' % (dim,))
    tot = np.sum(X, dim) + np.spacing(1)
    if dim == 1:
        Xn = (1.0 / tot.reshape((tot.size, 1))) * X
        if zero_sum_fix:
            for i in np.nonzero(tot < 1e-10)[0]:
                Xn[i, :] = 1.0 / Xn.shape[1]
    else:
        Xn = X * (1.0 / tot)
        if zero_sum_fix:
            for j in np.nonzero(tot < 1e-10)[0]:
                Xn[:, j] = 1.0 / Xn.shape[0]
    return Xn


def normalize_l2(X, dim=1):
    """Unit l2 norm along dim (matrixops.py:103-121)."""
    if dim == 0:
        return normalize_l2(X.T, 1).T
    if dim != 1:
        raise ValueError("dim must be 0 or 1")
    scale = 1 / np.sqrt(np.sum(X ** 2, 1) + 1e-10)
    return X * scale.reshape(scale.size, 1)


def tfidf(X, return_idf=False):
    """term counts -> tf-idf with idf = log(n / df) (matrixops.py:166-179)."""
    n = X.shape[0]
    idf = np.log(n / ((X > 0).sum(0) + np.spacing(1)))
    if sp.issparse(X):
        idf = sp.coo_matrix(idf)
        out = X.multiply(idf)
    else:
        out = X * idf
    return (out, idf) if return_idf else out


def labels_to_mat(y):
    """(n,) integer labels -> (n, k) one-hot rows; distributions pass through (matrixops.py:182-200)."""
    if y.size == y.shape[0]:
        k = len(np.unique(y))
        out = np.zeros((y.size, k))
        out[np.arange(y.size), y.astype(int)] = 1
        return out
    if abs(y.sum() - y.shape[0]) < 1e-5:
        return y
    k = len(np.unique(y))
    if y.shape[1] == k:
        return normalize(y)
    raise Exception('labels_to_mat: number of columns of y = {0} doesnt match number of unique '
                    'elements {1}'.format(y.shape[1], k))


def harden_distributions(W):
    """1 at the arg-max of every row, 0 elsewhere (matrixops.py:203-209)."""
    out = np.zeros_like(W)
    out[np.arange(W.shape[0]), np.argmax(W, 1)] = 1
    return out


def col_vector(x):
    return x.reshape(x.size, 1)


def stack_matrices(L, dict_key=None, transform=None, dim='tall'):
    """vstack ('tall') / hstack ('fat') of a list of arrays or of L[i][dict_key] (matrixops.py:217-267)."""
    assert dim in ('tall', 'fat'), 'dim must be "tall" or "fat".'
    assert isinstance(L[0], np.ndarray) or (isinstance(L[0], dict) and dict_key), \
        'a list of dicts needs dict_key'
    parts = []
    for item in L:
        if dict_key:
            try:
                m = item[dict_key]
            except TypeError:
                m = getattr(item, dict_key)
        else:
            m = item
        m = np.asarray(m)
        parts.append(transform(m) if transform else m)
    return (np.vstack if dim == 'tall' else np.hstack)(parts)
