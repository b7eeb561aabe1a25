"""Starting points for W, T (mirror of /root/reference/src/rri_nmf/initialization.py:9-163).

One-off host step (out of the kernel scope, 'next' row f-2): a thin wrapper over
scikit-learn's randomized SVD, exactly the dependency the reference uses.
"""
from math import sqrt

import numpy as np
from sklearn.utils import check_random_state
from sklearn.utils.extmath import randomized_svd, squared_norm

from .matrixops import normalize, tfidf

_KNOWN = (None, 'random', 'smart_random', 'nndsvd', 'nndsvda', 'nndsvdar')


def randomized_svd_device(engine, n_components, random_state=None, n_oversamples=10, n_iter='auto', resident=True):
    """Truncated SVD of the X resident in `engine`, by the algorithm of sklearn.utils.extmath.randomized_svd with
    its default settings (Halko et al.: Gaussian test matrix, LU-normalised power iterations, one QR, SVD of the
    small projected matrix, deterministic sign flip) -- the two products with X, which are all of the work, run
    on the device (rri_X_times / rri_Xt_times, float64); the (n or d) x (k+10) factorisations stay in scipy.

    Follows sklearn step by step so that it returns the same U, S, V up to rounding: the random test matrix is
    drawn from the same numpy RandomState, the matrix is transposed when n < d, signs are fixed as svd_flip does.
    resident (round 4, the default on dense handles): the panels of the power iteration stay on the device and are
    normalised there (Cholesky-QR, rri_range_finder); False: every product returns to the host and scipy normalises
    (LU, then QR) exactly as scikit-learn does.
    """
    from scipy import linalg
    n, d = engine.n, engine.d
    rng = check_random_state(random_state)
    m = n_components + n_oversamples
    if n_iter == 'auto':
        n_iter = 7 if n_components < 0.1 * min(n, d) else 4
    transpose = n < d                      # work on the matrix with the smaller second dimension
    A = engine.Xt_times if transpose else engine.X_times       # Q -> A @ Q
    At = engine.X_times if transpose else engine.Xt_times      # Q -> A.T @ Q
    Q = rng.normal(size=(n if transpose else d, m))
    if resident and m <= 64 and hasattr(engine, 'range_finder') and not getattr(engine, 'sparse', False):
        # the panels stay on the device and are normalised there by Cholesky-QR (rri_range_finder) instead of travelling to the
        # host for LU / QR after every product: another basis of the same range, the same U, S, V up to rounding -- and
        # 0.7 s less at 100000 x 10000 (profiles/r04_e2e_*)
        Q, B = engine.range_finder(Q, n_iter, transpose=transpose)
    else:
        lu = lambda Y: linalg.lu(Y, permute_l=True, check_finite=False)[0]
        normalize_q = lu if n_iter > 2 else (lambda Y: Y)
        for _ in range(n_iter):
            Q = normalize_q(A(Q))
            Q = normalize_q(At(Q))
        Q, _ = linalg.qr(A(Q), mode='economic', check_finite=False)
        B = At(Q).T                            # Q.T @ A
    Uhat, s, Vt = linalg.svd(B, full_matrices=False, lapack_driver='gesdd')
    # the same product held column-major: svd_flip's argmax down the columns of a tall row-major U walks it with a stride of a row
    # (88 ms of a 1.5 s fit at 100000 x 10000), and what follows reads U by columns as well
    U = (Uhat.T @ Q.T).T
    if not transpose:                      # svd_flip(U, Vt): the largest |entry| of every column of U is positive
        signs = np.sign(U[np.argmax(np.abs(U), axis=0), np.arange(U.shape[1])])
    else:                                  # svd_flip(U, Vt, u_based_decision=False)
        signs = np.sign(Vt[np.arange(Vt.shape[0]), np.argmax(np.abs(Vt), axis=1)])
    U = U * signs[np.newaxis, :]
    Vt = Vt * signs[:, np.newaxis]
    k = n_components
    if transpose:
        return Vt[:k, :].T, s[:k], U[:, :k].T
    return U[:, :k], s[:k], Vt[:k, :]


def _cholesky_floor(G, shift_rel=0.0):
    """Lower Cholesky factor of G + shift_rel trace(G) I with every pivot held at 1e-14 of its diagonal entry or above: a panel
    that is rank-deficient to working precision keeps a finite factor (host_cholesky in csrc/rri_hip.hip, same arithmetic)."""
    m = G.shape[0]
    L = np.array(G, dtype=np.float64)
    d0 = np.diag(L) + shift_rel * np.trace(L)
    L[np.arange(m), np.arange(m)] = d0
    for j in range(m):
        dj = L[j, j] - np.dot(L[j, :j], L[j, :j])
        floor = 1e-14 * d0[j] if d0[j] > 0 else 1e-300
        if not dj > floor:
            dj = floor
        L[j, j] = np.sqrt(dj)
        L[j + 1:, j] = (L[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
        L[:j, j] = 0.0
    if not np.all(np.isfinite(L)):
        raise ValueError('range finder: the panel holds a non-finite value')
    return L


def _cholqr2(Y, comm_sum):
    """Orthonormal basis of the range of a tall matrix whose rows are spread over the ranks: shifted Cholesky-QR, three passes
    (Y^T Y is all-reduced, L = chol, Y <- Y L^-T; the first pass factorises G + 1e-9 trace(G) I, which a panel of any condition
    number survives -- a Gaussian test matrix times a matrix with one dominant direction is conditioned like 1e8 -- and the next
    two repair what it leaves; rri_range_finder does the same on one device).  Returns this rank's rows of Q."""
    from scipy import linalg
    for rnd in range(3):
        G = comm_sum(Y.T @ Y)
        L = _cholesky_floor(0.5 * (G + G.T), 1e-9 if rnd == 0 else 0.0)
        Y = linalg.solve_triangular(L, Y.T, lower=True, check_finite=False).T
    return Y


def randomized_svd_sharded(engine, n_components, random_state=None, n_oversamples=10, n_iter='auto'):
    """randomized_svd_device for an X whose rows are spread over the ranks of engine.group (n_global >= d): the same
    algorithm -- same Gaussian test matrix on every rank, the same power iterations -- with
      X Q      row-local (every rank its rows of the n x m result),
      X^T Y    a sum over the ranks: all-reduced d x m panel (replicated afterwards),
    the d x m factorisations replicated (the LU of scikit-learn), and the TALL n x m ones -- which scikit-learn
    normalises by LU and finally by QR -- replaced by Cholesky-QR on the all-reduced m x m Gram matrix: another basis of
    the same range, so U, S, V are the same up to rounding (tests: 1e-7 against the one-handle start).
    Returns this rank's rows of U, S, V (replicated)."""
    from scipy import linalg
    g = engine.group
    n, d = g.n_global, engine.d
    if n < d:
        raise NotImplementedError('row-sharded start: n_global < d is not built (shard the transpose, or pass W_in / T_in)')
    comm_sum = lambda A: engine.comm_sum(np.ascontiguousarray(A).ravel()).reshape(A.shape)
    rng = check_random_state(random_state)
    m = n_components + n_oversamples
    if n_iter == 'auto':
        n_iter = 7 if n_components < 0.1 * min(n, d) else 4
    Q = rng.normal(size=(d, m))                                   # replicated: every rank draws the same numbers
    lu = lambda Y: linalg.lu(Y, permute_l=True, check_finite=False)[0]
    tall = (lambda Y: _cholqr2(Y, comm_sum)) if n_iter > 2 else (lambda Y: Y)
    wide = lu if n_iter > 2 else (lambda Y: Y)
    for _ in range(n_iter):
        Y = tall(engine.X_times(Q))                               # rows of X Q, normalised across the ranks
        Q = wide(comm_sum(engine.Xt_times(Y)))                    # X^T Y, replicated
    Qt = _cholqr2(engine.X_times(Q), comm_sum)                    # orthonormal basis of range(X Q): this rank's rows
    B = comm_sum(engine.Xt_times(Qt)).T                           # Q^T X, m x d, replicated
    Uhat, s, Vt = linalg.svd(B, full_matrices=False, lapack_driver='gesdd')
    U = Qt @ Uhat
    # svd_flip(U, Vt): the entry of largest magnitude of every column of U -- over ALL rows -- becomes positive
    loc = np.argmax(np.abs(U), axis=0)
    cand = np.zeros((g.world, 2, U.shape[1]))
    cand[g.rank, 0] = np.abs(U[loc, np.arange(U.shape[1])])
    cand[g.rank, 1] = np.sign(U[loc, np.arange(U.shape[1])])
    cand = comm_sum(cand)
    win = np.argmax(cand[:, 0, :], axis=0)                        # first rank on ties = lowest global row, as np.argmax
    signs = cand[win, 1, np.arange(U.shape[1])]
    signs[signs == 0] = 1.0
    U = U * signs[np.newaxis, :]
    Vt = Vt * signs[:, np.newaxis]
    k = n_components
    return U[:, :k], s[:k], Vt[:k, :]


def _split_pos_neg(v):
    return np.maximum(v, 0), np.abs(np.minimum(v, 0))


_VECTORISED_FROM = 1 << 20   # n * k from which the NNDSVD factors are formed for all topics at once


def _nndsvd_factors_vectorised(U, S, V, comm_sum=None):
    """The loop of initialization.py:109-140 for all topics at once: the same quantities, summed by numpy's
    reductions instead of one strided BLAS dot per topic (0.37 s -> 0.05 s at 100000 x 50; the per-topic loop is
    kept for small problems, where its bits are the reference's).  comm_sum: U holds this rank's rows only -- the
    squared norms of its columns are sums over the ranks."""
    Up, Un = np.maximum(U, 0), np.maximum(-U, 0)
    Vp, Vn = np.maximum(V, 0), np.maximum(-V, 0)
    sq = np.stack([np.einsum('ij,ij->j', Up, Up), np.einsum('ij,ij->j', Un, Un)])
    if comm_sum is not None:
        sq = comm_sum(sq)
    nup, nun = np.sqrt(sq[0]), np.sqrt(sq[1])
    nvp, nvn = np.sqrt(np.einsum('ij,ij->i', Vp, Vp)), np.sqrt(np.einsum('ij,ij->i', Vn, Vn))
    pos = nup * nvp > nun * nvn
    with np.errstate(divide='ignore', invalid='ignore'):
        sigma = np.where(pos, nup * nvp, nun * nvn)
        scale = np.sqrt(S * sigma)
        W = np.where(pos, Up / nup, Un / nun) * scale
        H = np.where(pos[:, None], Vp / nvp[:, None], Vn / nvn[:, None]) * scale[:, None]
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])             # leading pair is sign-definite (:109-111)
    H[0, :] = np.sqrt(S[0]) * np.abs(V[0, :])
    return W, H


def initialize_nmf(X, n_components, init=None, eps=1e-6, random_state=None, row_normalize=False,
                   n_words_beam=20, engine=None):
    """W (n x k), H (k x d) >= 0.  'random' | 'smart_random' | 'nndsvd' | 'nndsvda' | 'nndsvdar'.
    engine: an RRIEngine that already holds X -- the SVD behind the NNDSVD variants then runs its products
    with X on the device (randomized_svd_device); None = scikit-learn on the host, as the reference.
    An engine with a row group attached (engine.group): X is this rank's row block; W comes back for these rows, H
    replicated -- the start the one-handle call would give for the whole matrix (randomized_svd_sharded)."""
    group = getattr(engine, 'group', None)
    if group is not None:
        return _initialize_nmf_sharded(X, n_components, init, eps, random_state, row_normalize, engine, group)
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

    if engine is not None:
        U, S, V = randomized_svd_device(engine, k, random_state=random_state)
    else:
        U, S, V = randomized_svd(X, k, random_state=random_state)
    if U.shape[0] * k >= _VECTORISED_FROM:
        W, H = _nndsvd_factors_vectorised(U, S, V)
    else:
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


def _initialize_nmf_sharded(X, k, init, eps, random_state, row_normalize, engine, group):
    """initialize_nmf for a row block of the group's matrix: what every branch above computes, with the sums over rows
    all-reduced and the random draws made for the WHOLE matrix on every rank (same generator, same order), so that the
    result is the one-handle start cut into row blocks."""
    d = engine.d
    lo, hi, n = group.row_lo, group.row_lo + group.n_local, group.n_global
    comm_sum = lambda A: engine.comm_sum(np.ascontiguousarray(np.asarray(A, dtype=np.float64)).ravel()).reshape(np.shape(A))
    if init is None:
        init = 'nndsvd' if k < d else 'random'
    mean = lambda: float(comm_sum(np.array([engine.X_times(np.ones((d, 1))).sum()]))[0]) / (float(n) * d)
    if init == 'random':
        rng = check_random_state(random_state)
        H = rng.rand(k, d)
        W = rng.rand(n, k)[lo:hi]
        return W, (normalize(H) if row_normalize else H)
    if init == 'smart_random':
        rng = check_random_state(random_state)
        scale = np.sqrt(mean() / k)
        H = np.abs(scale * rng.randn(k, d))
        W = np.abs(scale * rng.randn(n, k))[lo:hi]
        return W, (normalize(H) if row_normalize else H)
    if init not in _KNOWN:
        raise ValueError('Invalid init parameter: got %r instead of one of %r' % (init, _KNOWN[:1] + _KNOWN[3:]))
    U, S, V = randomized_svd_sharded(engine, k, random_state=random_state)
    W, H = _nndsvd_factors_vectorised(U, S, V, comm_sum)
    W[W < eps] = 0
    H[H < eps] = 0
    if init == 'nndsvda':
        fill = mean()
        W[W == 0] = fill
        H[H == 0] = fill
    elif init == 'nndsvdar':
        # initialization.py:147-152 draws one number per zero of W in row-major order of the WHOLE matrix, then one per zero
        # of H, from one generator.  Every rank draws the whole W sequence (the counts of zeros per rank are exchanged) and
        # keeps the stretch that belongs to its rows: the one-handle start, cut into row blocks.
        rng = check_random_state(random_state)
        fill = mean()
        zeros_w = W == 0
        counts = np.zeros(group.world)
        counts[group.rank] = float(zeros_w.sum())
        counts = comm_sum(counts).astype(np.int64)
        before, total = int(counts[:group.rank].sum()), int(counts.sum())
        draws = rng.randn(total)
        W[zeros_w] = np.abs(fill * draws[before:before + int(counts[group.rank])] / 100)
        H[H == 0] = np.abs(fill * rng.randn(int((H == 0).sum())) / 100)
    return W, (normalize(H) if row_normalize else H)


def init_coherence_beam_search(X, n_components, n_words_beam=20):
    """Topics seeded by a beam search for word sets of high pointwise mutual information
    (initialization.py:166-208): every topic starts from the heaviest unused word of normalize(tfidf(X)) and takes,
    n_words_beam - 1 times, the unused word with the best summed PMI against the words it already holds; T gives each
    chosen word its global weight (rows normalised), W = normalize(max(X T^T, 0)).  Stand-alone, as in the reference
    (no `init` name of initialize_nmf leads here): pass the result as W_in / T_in.

    The reference scans the words in two Python loops; here one topic step scores all words at once, with the same
    additions in the same order (the scores, and so the choices, are bit-identical) and np.argmax's first maximum in
    the place of the scan's strict `>`."""
    X = normalize(tfidf(np.asarray(X, dtype=np.float64)))
    C = np.dot(X.T, X)
    k = n_components
    n, d = X.shape
    P_i = np.log(C.sum(1) + np.spacing(1))
    P_ij = np.log(C + np.spacing(1))
    xs = X.sum(0)
    chosen = []
    for _ in range(k):
        j = int(np.argmax(xs))
        xs[j] = 0                              # a word serves one topic only
        tpc = [j]
        for _ in range(1, n_words_beam):
            free = xs > 0
            if not free.any():                 # (the reference fails here with a TypeError on xs[None])
                raise ValueError('init_coherence_beam_search: fewer than k * n_words_beam usable words')
            score = np.zeros(d)
            for c in tpc:
                score += (P_ij[:, c] - P_i) - P_i[c]
            score[~free] = -np.inf
            best = int(np.argmax(score))
            tpc.append(best)
            xs[best] = 0
        chosen.append(tpc)
    xs = X.sum(0)
    T = np.zeros((k, d))
    for t, tpc in enumerate(chosen):
        T[t, tpc] = xs[tpc]                    # weight of a word in its topic: its global importance
    T = normalize(T)
    W = normalize(np.maximum(np.dot(X, T.T), 0))
    return W, T
