"""CPU oracle for the RRI (rank-one residue iteration) hot path of maksimt/rri_nmf.

TEST INFRASTRUCTURE ONLY.  This module is a numpy float64 restatement of the
reference algorithm.  It is the *checker* for the HIP path and the `cpu_baseline`
leg of bench.py.  Nothing under rri_nmf_amd/ imports it; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline may.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against vectors produced by the unmodified reference (oracle/make_golden.py,
run in the build container, outputs committed under tests/golden/).  The
reference's own tests pin no W,T numbers for this path (only NNDSVD init bytes,
tests/conftest.py:8-19), so the vectors were captured from the reference itself.

To stay bit-identical with the reference the numpy operation ORDER of each step
is kept (e.g. `w.dot(X) - g.dot(T)`, zero-column GEMM for the weighted
flavour); the code structure is this repo's own.

Reference citations are `file:line` into /root/reference/src/rri_nmf/.
"""
import copy
import time

import numpy as np
import scipy.sparse

EPS = np.spacing(10)  # nmf.py:52, optimization.py:5


# --------------------------------------------------------------------------
# simplex projection                                     matrixops.py:5-100
# --------------------------------------------------------------------------
def proj_simplex(v_in, s=1):
    """Euclidean projection of a vector on {w>=0, sum w = s} (matrixops.py:5-69).

    Sort based (Duchi et al.): theta from the largest prefix of the descending
    sort that stays positive.  Early exit returns the input object unchanged
    when it already sums to s exactly and is non-negative (matrixops.py:53-55).
    """
    assert s > 0, "Radius s must be strictly positive (%d <= 0)" % s
    v = v_in.toarray() if scipy.sparse.issparse(v_in) else v_in
    m = int(np.prod(v_in.shape))
    v = v.reshape((m,))
    if v.sum() == s and np.all(v >= 0):
        return v
    desc = np.sort(v)[::-1]
    csum = np.cumsum(desc)
    keep = np.nonzero(desc * np.arange(1, m + 1) > (csum - s))[0][-1]
    theta = (csum[keep] - s) / (keep + 1.0)
    w = (v - theta).clip(min=0)
    if scipy.sparse.issparse(v_in):
        return scipy.sparse.csr_matrix(w.reshape(v_in.shape))
    return w.reshape(v_in.shape)


def proj_rows_simplex(M, s=1.0, axis=1):
    """Row-wise (axis=1) or column-wise projection, in place (matrixops.py:72-100)."""
    if axis == 0:
        return proj_rows_simplex(M.T, s, axis=1).T
    if np.isscalar(s):
        for i in range(M.shape[0]):
            M[i, :] = proj_simplex(M[i, :], s)
    else:
        assert s.size == M.shape[0]
        for i in range(M.shape[0]):
            M[i, :] = proj_simplex(M[i, :], s[i])
    return M


# --------------------------------------------------------------------------
# closed-form 1-D quadratic minimiser                    optimization.py:12-88
# --------------------------------------------------------------------------
def qf_min(w, c, s=1.0, ub=1.0):
    """argmin_x w.x + 0.5 x.diag(c).x  s.t. x>=0 [, sum x = s] [, x<=ub].

    Returns (x, nx) with nx the 1-norm of the solution BEFORE projection /
    rescaling (optimization.py:55,74,84).  Branches: scalar c>0 (:53-59, `ub`
    is not applied), scalar c<=0 (:60-74), vector c (:75-87).
    """
    m = w.size
    if s:  # optimization.py:43-49
        if ub:
            ub = min(ub, s)
            assert m * ub >= s, 'Impossible to satisfy sum and upper bound constraints.'
        else:
            ub = s
    if np.isscalar(c):
        if c > 0:
            x = np.maximum(-w, 0) / (c + EPS)
            nx = x.sum()
            if s is not None:
                x = proj_simplex(x, s)
        else:
            x = np.zeros_like(w)
            if s is None:
                hit = np.argwhere(w + c < 0)
                if ub:
                    x[hit] = ub
                else:
                    _unbounded(w, c, s, ub)
            elif s == 1.0:
                x[np.argmin(w)] = 1.0
            else:
                raise NotImplementedError('s={} is not yet implemented'.format(s))
            nx = 1.0
    elif np.shape(w) == np.shape(c):
        if np.any(c < 0) and (s is None and ub is None):
            _unbounded(w, c, s, ub)
        pos = np.argwhere(c > 0).ravel()
        x = np.zeros_like(w)
        x[pos] = np.maximum(-w[pos], 0) / (c[pos] + EPS)
        if ub is not None:
            x = np.minimum(x, ub)
        nx = x.sum()
        if s is not None:
            x = s * x / x.sum()
    return x, nx


def _unbounded(w, c, s, ub):  # optimization.py:105-107
    raise ValueError('Minimum objective is unbounded. w={w}, c={c}, s={s}, '
                     'ub={ub}'.format(w=w, c=c, s=s, ub=ub))


def universal_stopping_condition(obj_history, eps_stop=1e-4):
    """optimization.py:284-291."""
    if len(obj_history) < 2:
        return False
    first = abs(obj_history[0] - obj_history[1])
    last = abs(obj_history[-1] - obj_history[-2])
    return last <= eps_stop * first


# --------------------------------------------------------------------------
# host preprocessing used by the estimators              matrixops.py:124-179
# --------------------------------------------------------------------------
def normalize(X, dim=1, zero_sum_fix=True):
    """Scale rows (dim=1) / columns (dim=0) to sum 1 (matrixops.py:124-163)."""
    if dim == 1:
        tot = np.sum(X, 1) + np.spacing(1)
        Xn = (1.0 / tot.reshape((tot.size, 1))) * X
        if zero_sum_fix:
            for i in np.nonzero(tot < 1e-10)[0]:
                Xn[i, :] = np.ones((1, Xn.shape[1])) * (1.0 / Xn.shape[1])
        return Xn
    if dim == 0:
        tot = np.sum(X, 0) + np.spacing(1)
        Xn = X * (1.0 / tot)
        if zero_sum_fix:
            for j in np.nonzero(tot < 1e-10)[0]:
                Xn[:, j] = np.ones((Xn.shape[0], 1)) * (1.0 / Xn.shape[0])
        return Xn
    raise Exception('Unknown dim=' + str(dim))


def tfidf(X, return_idf=False):
    """matrixops.py:166-179 (dense branch)."""
    n = X.shape[0]
    df = (X > 0).sum(0)
    idf = np.log(n / (df + np.spacing(1)))
    out = X * idf
    return (out, idf) if return_idf else out


# --------------------------------------------------------------------------
# NNDSVD initialisation                                  initialization.py:9-163
# --------------------------------------------------------------------------
def initialize_nmf(X, k, init=None, eps=1e-6, random_state=None, row_normalize=False):
    from sklearn.utils import check_random_state
    from sklearn.utils.extmath import randomized_svd, squared_norm
    n, d = X.shape
    if init is None:
        init = 'nndsvd' if k < d else 'random'
    if init == 'random':  # initialization.py:80-87
        rng = check_random_state(random_state)
        T = rng.rand(k, d)
        W = rng.rand(n, k)
        if row_normalize:
            T = normalize(T)
        return W, T
    if init == 'smart_random':  # initialization.py:90-102
        avg = np.sqrt(X.mean() / k)
        rng = check_random_state(random_state)
        H = np.abs(avg * rng.randn(k, d))
        W = np.abs(avg * rng.randn(n, k))
        if row_normalize:
            H = normalize(H)
        return W, H
    U, S, V = randomized_svd(X, k, random_state=random_state)
    W, H = np.zeros(U.shape), np.zeros(V.shape)
    W[:, 0] = np.sqrt(S[0]) * np.abs(U[:, 0])
    H[0, :] = np.sqrt(S[0]) * np.abs(V[0, :])
    nrm = lambda z: np.sqrt(squared_norm(z))
    for j in range(1, k):  # initialization.py:114-140
        x, y = U[:, j], V[j, :]
        xp, yp = np.maximum(x, 0), np.maximum(y, 0)
        xn, yn = np.abs(np.minimum(x, 0)), np.abs(np.minimum(y, 0))
        xpn, ypn, xnn, ynn = nrm(xp), nrm(yp), nrm(xn), nrm(yn)
        mp, mn = xpn * ypn, xnn * ynn
        if mp > mn:
            u, v, sigma = xp / xpn, yp / ypn, mp
        else:
            u, v, sigma = xn / xnn, yn / ynn, mn
        lbd = np.sqrt(S[j] * sigma)
        W[:, j] = lbd * u
        H[j, :] = lbd * v
    W[W < eps] = 0
    H[H < eps] = 0
    if init == 'nndsvd':
        pass
    elif init == 'nndsvda':
        avg = X.mean()
        W[W == 0] = avg
        H[H == 0] = avg
    elif init == 'nndsvdar':
        rng = check_random_state(random_state)
        avg = X.mean()
        W[W == 0] = abs(avg * rng.randn(len(W[W == 0])) / 100)
        H[H == 0] = abs(avg * rng.randn(len(H[H == 0])) / 100)
    else:
        raise ValueError('Invalid init parameter: got %r' % (init,))
    if row_normalize:
        H = normalize(H)
    return W, H


# --------------------------------------------------------------------------
# objective                                              nmf.py:71-94
# --------------------------------------------------------------------------
def true_objective(X, W, T, reg_w_l2=0, reg_t_l2=0, reg_w_l1=0, reg_t_l1=0, Wm=None, wr=None):
    W2 = np.sum(W ** 2)
    T2 = np.sum(T ** 2)
    T1 = np.sum(np.abs(T))
    W1 = np.sum(np.abs(W))
    R = (X - np.dot(W, T)) ** 2
    if Wm is not None:
        R = Wm * R
    if wr is not None:
        R = wr * R
    base = 0.5 * np.sum(R)
    return base + 0.5 * reg_w_l2 * W2 + 0.5 * reg_t_l2 * T2 + reg_t_l1 * T1 + reg_w_l1 * W1


# --------------------------------------------------------------------------
# per-topic residual products                            nmf.py:633-747
# --------------------------------------------------------------------------
def residual_products_T(X, W, T, t, W_mat=None):
    """(w_t^T R_t , ||w_t||^2) without forming R (Gram form, nmf.py:670-676) or,
    weighted, through the masked residual (nmf.py:687-701)."""
    if W_mat is None:
        w = W[:, t]
        wX = w.T.dot(X)
        g = w.T.dot(W)
        g[t] = 0
        return wX - g.dot(T), (W[:, t] ** 2).sum()
    keep = W[:, t].copy()
    W[:, t] = 0
    Rt = X - np.dot(W, T)
    W[:, t] = keep
    Rt = W_mat * Rt
    wR = np.dot(W[:, t].T, Rt).ravel()
    nw = np.dot((W[:, t] ** 2).reshape(-1, 1).T, W_mat).ravel()
    return wR, nw


def residual_products_W(X, W, T, t, W_mat=None):
    """(R_t t_t , ||t_t||^2): nmf.py:728-734 unweighted, :735-746 weighted."""
    if W_mat is None:
        Xt = X.dot(T[t, :].T)
        g = T.dot(T[t, :].T)
        g[t] = 0
        return Xt - W.dot(g), (T[t, :] ** 2).sum()
    keep = W[:, t].copy()
    W[:, t] = 0
    Rt = X - np.dot(W, T)
    W[:, t] = keep
    Rt = W_mat * Rt
    num = np.dot(Rt, T[t, :].T).ravel()
    nt = np.dot(W_mat, (T[t, :] ** 2).reshape(-1, 1)).ravel()
    return num, nt


# --------------------------------------------------------------------------
# solver                                                 nmf.py:98-560
# --------------------------------------------------------------------------
class _Resets(object):
    """Reset budget + the two degeneracy handlers (nmf.py:751-816).  The
    reference keeps the budget in a module global; here it is per call."""

    def __init__(self, budget, method, fix_seed):
        self.left = budget
        self.method = method
        self.fix_seed = fix_seed
        self.count = 0

    def _reseed(self, X, W, T, t):
        n, d = X.shape
        if self.method == 'max_resid_document':  # nmf.py:770-776, 804-810
            Rp = np.maximum(X - W.dot(T), 0)
            mi = np.argmax((Rp ** 2).sum(1))
            T[t, :] = Rp[mi, :]
            W[:, t] = 0
            W[mi, t] = 1.0
        elif self.method == 'random':  # nmf.py:778-783, 811-816
            if self.fix_seed:
                np.random.seed(t + np.argmax(T[t, :]))
            T[t, :] = np.random.rand(1, d)
            T[t, :] /= T[t, :].sum()
            W[:, t] = np.random.rand(n)

    def after_T(self, X, W, T, t, project_T, t_row_sum):
        if np.sum(T[t, :]) > 1e-10 or self.method is None:  # nmf.py:757-761
            if t_row_sum and project_T and np.abs(np.sum(T[t, ]) - t_row_sum) > 1e-15:
                T[t, :] = proj_simplex(T[t, :], s=t_row_sum)
            return
        if self.left == 0:
            return
        self.left -= 1
        self.count += 1
        self._reseed(X, W, T, t)

    def after_W(self, X, W, T, t):
        if np.sum(W[:, t]) > 1e-10 or self.method is None:  # nmf.py:793-795
            return
        if self.left == 0:
            return
        self.left -= 1
        self.count += 1
        self._reseed(X, W, T, t)


def initialize_and_validate(X, k, W_in, T_in, W_mat, init, random_state, project_T_each_iter,
                            project_W_each_iter, w_row_sum, t_row_sum, fix_W, fix_T):
    """nmf.py:819-880."""
    n, d = X.shape
    if np.prod(np.shape(W_in)) == 0 or np.prod(np.shape(T_in)) == 0:
        src = X if W_mat is None else W_mat * X
        W, T = initialize_nmf(src, k, init, random_state=random_state, row_normalize=False)
        if t_row_sum is not None:
            T = normalize(T) * t_row_sum
        if w_row_sum is not None:
            W = normalize(W) * w_row_sum
    if np.prod(np.shape(W_in)) > 0:
        if not np.shape(W_in) == (n, k):
            raise ValueError('W_in has wrong dimensions, must be n*k')
        W = W_in
    if np.prod(np.shape(T_in)) > 0:
        if not np.shape(T_in) == (k, d):
            raise ValueError('T_in has wrong dimensions, must be k*d')
        T = T_in
    if scipy.sparse.issparse(T):
        T = T.toarray()
    if scipy.sparse.issparse(W):
        W = W.toarray()
    W = np.maximum(W, 0)
    T = np.maximum(T, 0)
    if project_W_each_iter and not fix_W and w_row_sum is not None:
        W = proj_rows_simplex(W, w_row_sum)
    if project_T_each_iter and not fix_T and t_row_sum is not None:
        T = proj_rows_simplex(T, t_row_sum)
    return W, T


def nmf(X, k, w_row=None, W_mat=None, fix_W=False, fix_T=False, random_state=None,
        init='nndsvd', T_in=[], W_in=[], max_iter=200, max_time=600, eps_stop=1e-4,
        compute_obj_each_iter=False, project_W_each_iter=False, w_row_sum=None,
        do_final_project_W=True, project_T_each_iter=False, t_row_sum=None,
        early_stop=None, reset_topic_method='max_resid_document', fix_reset_seed=False,
        n_resets=23, reg_w_l2=0, reg_t_l2=0, reg_w_l1=0, reg_t_l1=0, diagnostics=[],
        objective_always=False, on_sweep=None, eps_gauss_t=None, delta_gauss_t=None,
        store_gradients=False, ind_rows_to_store=None):
    """Restatement of nmf.py:98-560.

    store_gradients (nmf.py:325-327, 411-413, 454-456, 677-686, 706-713, 541-549): the reference raises IndexError
    at :543, where the reshape lambda meant as stack_matrices' `transform` is passed as `dict_key`; this restates
    the evident intent -- out['numer_W'][sweep] the k stacked wR_store rows, out['denom_W'][sweep] the k nw_store
    (k x 1, or k x d for weighted problems).  Not pinned by the reference (it cannot produce these).

    `objective_always=True` reproduces the reference AS SHIPPED, whose module
    logger has level NOTSET so `logger.level <= logging.DEBUG` (nmf.py:366) forces
    compute_obj_each_iter and therefore the obj-history stop rule.
    `on_sweep(iter_no, W, T)` is a test hook called after every sweep.
    """
    out = {}
    n, d = X.shape
    if project_T_each_iter and np.any([reg_w_l1, reg_t_l1]):  # nmf.py:280-285
        project_T_each_iter = False
    if (not project_T_each_iter and not t_row_sum) and (reg_t_l1 < 0 or reg_t_l2 < 0):
        return {'W': np.ones((n, k)), 'T': np.ones((k, d)) * 1e6,
                'obj_history': [-np.inf], 'iter_cputime': [0]}  # nmf.py:292-303
    if (not project_W_each_iter and not w_row_sum) and (reg_w_l1 < 0 or reg_w_l2 < 0):
        return {'W': np.ones((n, k)) * 1e6, 'T': np.ones((k, d)),
                'obj_history': [-np.inf], 'iter_cputime': [0]}  # nmf.py:304-315
    if type(diagnostics) is not list:
        diagnostics = [diagnostics]
    if diagnostics:
        out['diagnostics'] = {f.__name__: [] for f in diagnostics}
    if store_gradients:
        out['numer_W'], out['denom_W'] = {}, {}
    if random_state is None:
        random_state = int(time.time()) % 4294967296
    t0_wall = time.time()
    max_time = max_time - 10
    X_orig = None
    if w_row is not None:  # nmf.py:335-338
        X_orig = X.copy()
        X = np.sqrt(w_row) * X
    if w_row_sum is not None and not np.isscalar(w_row_sum):  # nmf.py:340-344
        w_row_sum = w_row_sum.reshape((w_row_sum.size, 1))
        if w_row is not None:
            w_row_sum = np.sqrt(w_row_sum)
    if n <= k:
        init = 'random'
    t0_cpu = time.process_time()
    W, T = initialize_and_validate(X, k, W_in, T_in, W_mat, init, random_state,
                                   project_T_each_iter, project_W_each_iter, w_row_sum,
                                   t_row_sum, fix_W, fix_T)
    resets = _Resets(n_resets, reset_topic_method, fix_reset_seed)
    iter_cputime = []
    if early_stop:
        last_score = np.inf
        W_prev, T_prev = copy.deepcopy(W), copy.deepcopy(T)
    obj_history = []
    if objective_always:
        compute_obj_each_iter = True
    regs = dict(reg_w_l2=reg_w_l2, reg_t_l2=reg_t_l2, reg_w_l1=reg_w_l1, reg_t_l1=reg_t_l1)
    no_regs = abs(reg_w_l1) + abs(reg_w_l2) + abs(reg_t_l1) + abs(reg_t_l2) == 0
    for f in diagnostics:
        out['diagnostics'][f.__name__].append(f(X, W, T))

    for iter_no in range(max_iter):
        if early_stop:  # nmf.py:381-407
            if callable(early_stop):
                score = early_stop(X, W, T)
            elif compute_obj_each_iter:
                score = np.inf if not obj_history else obj_history[-1]
            if score > last_score:
                W, T = W_prev, T_prev
                obj_history = obj_history[:-1]
                iter_cputime = iter_cputime[:-1]
                for f in diagnostics:
                    out['diagnostics'][f.__name__] = out['diagnostics'][f.__name__][:-1]
                break
            last_score = score
            W_prev, T_prev = copy.deepcopy(W), copy.deepcopy(T)

        if store_gradients:
            out['numer_W'][iter_no], out['denom_W'][iter_no] = [], []
        for t in range(k):  # nmf.py:415-476
            if not fix_T:
                wR, nw = residual_products_T(X, W, T, t, W_mat)
                if store_gradients:
                    if ind_rows_to_store is None:
                        wR_store, nw_store = wR, nw
                    else:   # the same sums over the listed rows only (nmf.py:680-686, 709-713)
                        r_ = np.asarray(ind_rows_to_store)
                        wR_store, nw_store = residual_products_T(X[r_, :], W[r_, :], T, t,
                                                                 None if W_mat is None else W_mat[r_, :])
                    out['numer_W'][iter_no].append(np.array(wR_store, dtype=np.float64))
                    out['denom_W'][iter_no].append(np.array(nw_store, dtype=np.float64))
                if eps_gauss_t and delta_gauss_t:   # nmf.py:422-435: Gaussian mechanism on the T-row sums
                    from scipy.stats import norm as gaussian
                    c2 = 2 * np.log(1.25 / float(delta_gauss_t)) + 0.001
                    df2 = 1000.0
                    sigma2 = c2 * df2 ** 2 * (1 / float(eps_gauss_t)) ** 2
                    N = gaussian(0, np.sqrt(sigma2))
                    wR = wR + N.rvs(np.size(wR)).reshape(np.shape(wR))
                    nw = nw + N.rvs(np.size(nw)).reshape(np.shape(nw))
                    nw = np.maximum(nw, 0)
                numer = wR - reg_t_l1
                denom = nw + reg_t_l2
                s = t_row_sum if project_T_each_iter else None
                T[t, :], nt1 = qf_min(-numer, denom, s=s, ub=t_row_sum)
                if no_regs:
                    W[:, t] = W[:, t] * nt1
                resets.after_T(X, W, T, t, project_T_each_iter, t_row_sum)
            if not fix_W:
                Rt, nt = residual_products_W(X, W, T, t, W_mat)
                numer = Rt - reg_w_l1
                denom = nt + reg_w_l2
                W[:, t], _ = qf_min(-numer, denom, s=None, ub=w_row_sum)
                resets.after_W(X, W, T, t)
                assert np.all(W[:, t] >= 0), 'W contains negative entries'
                assert np.sum(W[:, t]) > 0, 'W[:, t] sums to 0'

        if project_W_each_iter and not fix_W and w_row_sum is not None:  # nmf.py:481-484
            W = proj_rows_simplex(W, w_row_sum)
        if compute_obj_each_iter:
            obj_history.append(true_objective(X, W, T, Wm=W_mat, wr=w_row, **regs))
        iter_cputime.append(time.process_time())
        for f in diagnostics:
            out['diagnostics'][f.__name__].append(f(X, W, T))
        if on_sweep is not None:
            on_sweep(iter_no, W, T)
        if time.time() - t0_wall >= max_time:
            break
        if compute_obj_each_iter and universal_stopping_condition(obj_history, eps_stop=eps_stop):
            break

    iter_cputime = [c - t0_cpu for c in iter_cputime]
    if not project_W_each_iter and w_row_sum is not None and not fix_W and do_final_project_W:
        if np.isscalar(w_row_sum):  # nmf.py:519-529
            for i in range(n):
                W[i, :] = proj_simplex(W[i, :], s=w_row_sum)
        else:
            for i in range(n):
                W[i, :] = proj_simplex(W[i, :], s=w_row_sum[i])
    if w_row is not None:  # nmf.py:531-539
        sub = nmf(X_orig, k, T_in=T, fix_T=True, max_iter=10, w_row_sum=w_row_sum,
                  project_W_each_iter=True, compute_obj_each_iter=compute_obj_each_iter,
                  objective_always=objective_always)
        obj_history.extend(sub['obj_history'] if 'obj_history' in sub else [])
        iter_cputime.extend(sub['iter_cputime'])
        W = sub['W']
    if store_gradients:
        as_row = lambda v: np.asarray(v, dtype=np.float64).reshape((1, np.size(v)))
        for key in ('numer_W', 'denom_W'):
            for it, rows in out[key].items():
                out[key][it] = np.vstack([as_row(v) for v in rows]) if rows else np.zeros((0, 0))
    out['W'] = W
    out['T'] = T
    if compute_obj_each_iter:
        out['obj_history'] = obj_history
    out['iter_cputime'] = iter_cputime
    out['random_state'] = random_state
    out['n_resets_used'] = resets.count
    return out


# --------------------------------------------------------------------------
# timed sweeps for bench.py's cpu_baseline (plain flavour, no objective)
# --------------------------------------------------------------------------
def plain_sweeps(X, W, T, n_sweeps):
    """n_sweeps unconstrained RRI sweeps in place (nmf.py:415-476 with all
    options off): the loop bench.py times on the host cores."""
    k = W.shape[1]
    for _ in range(n_sweeps):
        for t in range(k):
            wR, nw = residual_products_T(X, W, T, t)
            T[t, :], nt1 = qf_min(-wR, nw, s=None, ub=None)
            W[:, t] = W[:, t] * nt1
            Rt, nt = residual_products_W(X, W, T, t)
            W[:, t], _ = qf_min(-Rt, nt, s=None, ub=None)
    return W, T
