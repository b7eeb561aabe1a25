"""nmf(): drop-in for the reference's solver entry point (/root/reference/src/rri_nmf/nmf.py:98-560)
with the sweep / topic loop running on an MI355X through librri_hip.so.

Same signature, defaults, return keys and error behaviour as the reference; keyword-only
additions select the device side:
    dtype   storage type of X (and the mask) in HBM: float32 or float64.  None = float32 when X is
            float32, else float64.  The arithmetic is float64 either way (see csrc/rri_kernels.hpp).
    device  HIP device ordinal.
    device_init  True / False: run the products of the randomized SVD behind the NNDSVD initialisations on the
            device (initialization.randomized_svd_device) or in scikit-learn on the host; None = on the device from
            2e7 entries of X on.  Same algorithm either way.
    group   row-sharded run over the GPUs of one node (distributed.RowGroup): X, W_in (and W_mat, w_row) are
            THIS rank's row block, T_in is the replicated k x d factor; every rank makes the same call and gets its
            rows of W, the common T and the global objective history.  Without W_in / T_in the start is computed
            row-sharded too (initialization.randomized_svd_sharded: the one-handle start, cut into row blocks --
            every init incl. 'nndsvdar', and the weighted start on W_mat .* X for dense inputs; n_global >= d).
            `preprocess` (tf-idf with the document frequencies summed over the ranks, row normalisation) and `w_row`
            (with its refit) run sharded; store_gradients, the Gaussian mechanism and host callbacks stay
            single-handle options.  The reference has one call for the whole X (nmf.py:98-108); this is that call,
            made once per rank.
    schedule  'gram' (default): the residual is never formed, X is read once per topic step; 'residual': the explicit
            residual R = X - W T is kept in HBM and updated by rank-one terms (the form north_star names; unweighted,
            both halves free, k >= 2).  Same results to rounding.
    sparse_pattern  weighted flavour with scipy sparse X and 0/1 sparse W_mat: keep the residual on the observed
            entries only (True), densify on the device (False); None = pattern-only below 35 % observed.

What runs where
    host (numpy, once):   argument checks, warnings and sentinel returns (nmf.py:280-315), the
                          starting point (nmf.py:819-880), the stop rules, callbacks
    device (every sweep): residual products, closed-form updates, projections, resets, objective
There is no CPU fallback for the loop: without the library / a GPU the call raises.
"""
import logging
import os
import time

import numpy as np
import scipy.sparse

from .engine import RRIEngine
from .initialization import initialize_nmf
from .matrixops import normalize, proj_mat_to_simplex
from .optimization import universal_stopping_condition

# The reference configures logging the same way (nmf.py:46-47).  A module logger created like this has
# level NOTSET (0), so `logger.level <= logging.DEBUG` below is true unless the caller sets a level --
# in the reference (nmf.py:366) and therefore here: the objective is then evaluated every sweep and the
# obj-history stop rule is active.  Call `logger.setLevel(logging.WARNING)` for the documented behaviour.
logging.basicConfig(level=logging.WARNING)
logger = logging.getLogger(__name__)

eps_div_by_zero = np.spacing(10)  # nmf.py:52


class TrueObjComputer(object):
    """Handle returned as rtv['obj_calculator'] (nmf.py:58-94): remembers the problem and the last
    objective value; true_objective() re-evaluates it on the device."""

    def __init__(self, X, W, T, reg_w_l2, reg_t_l2, reg_w_l1, reg_t_l1, Wm, wr, dtype=None, device=0,
                 sparse_pattern=None, preprocess=None, group=None):
        self.X, self.W, self.T = X, W, T
        self.reg_w_l2, self.reg_t_l2 = reg_w_l2, reg_t_l2
        self.reg_w_l1, self.reg_t_l1 = reg_w_l1, reg_t_l1
        self.Wm, self.wr = Wm, wr
        self.obj = np.inf
        self._dtype, self._device, self._sparse_pattern = dtype, device, sparse_pattern
        self._preprocess = preprocess      # device-side tf-idf / normalisation that nmf() applied to X (idf resolved)
        self._group = group                # row-sharded: true_objective() is then a collective call

    def true_objective(self):
        X = self.X   # row weights, when used, are already folded into X by nmf() (nmf.py:335-338)
        n, d = X.shape
        k = self.W.shape[1]
        sdt = _storage_dtype(X, self._dtype) if self._preprocess is None else np.dtype(self._dtype or np.float64)
        with _engine_with_problem(X, self.Wm, k, sdt, self._device, self._sparse_pattern) as eng:
            if self._preprocess is not None:
                eng.preprocess(**self._preprocess)
            if self._group is not None:
                eng.attach_group(self._group)
            eng.set_W(self.W)
            eng.set_T(self.T)
            eng.set_params(reg_w_l1=self.reg_w_l1, reg_w_l2=self.reg_w_l2, reg_t_l1=self.reg_t_l1,
                           reg_t_l2=self.reg_t_l2)
            self.obj = eng.objective()
        return self.obj


SPARSE_PATTERN_MAX_DENSITY = 0.35   # below this share of observed entries the pattern-only residual moves fewer bytes


def _observed_csr(X, W_mat):
    """X on the observation pattern of W_mat as ONE canonical CSR matrix (stored entries = observed entries,
    explicit zeros kept), or None when that view does not exist: dense inputs, weights other than 0/1, or X with
    non-zeros outside the pattern (they are invisible to every masked sum, but the reset search of nmf.py:770-773
    looks at all of X)."""
    if not (scipy.sparse.issparse(X) and scipy.sparse.issparse(W_mat)):
        return None
    M = W_mat.tocsr()
    if not M.has_canonical_format:
        M = M.copy()
        M.sum_duplicates()
    Xc = X.tocsr()
    if not Xc.has_canonical_format:
        Xc = Xc.copy()
        Xc.sum_duplicates()
    if Xc.nnz == M.nnz and np.array_equal(Xc.indptr, M.indptr) and np.array_equal(Xc.indices, M.indices):
        vals = Xc.data
    else:
        d = M.shape[1]
        rows_of = lambda A: np.repeat(np.arange(A.shape[0], dtype=np.int64), np.diff(A.indptr))
        key_m = rows_of(M) * d + M.indices
        nz = Xc.data != 0
        key_x = (rows_of(Xc) * d + Xc.indices)[nz]
        pos = np.searchsorted(key_m, key_x)
        pos_ok = pos < key_m.size
        if not np.all(pos_ok) or not np.array_equal(key_m[pos], key_x):
            return None
        vals = np.zeros(M.nnz, dtype=Xc.data.dtype if Xc.data.dtype.kind == 'f' else np.float64)
        vals[pos] = Xc.data[nz]
    return scipy.sparse.csr_matrix((vals, M.indices, M.indptr), shape=M.shape)


def _engine_with_problem(X, W_mat, k, sdt, device, sparse_pattern=None, schedule='gram'):
    """Engine with X and the weights on the device.  scipy sparse X / 0-1 sparse W_mat go up as CSR: onto a
    pattern-only handle (no dense n x d array at all) when X lives on the pattern and the pattern is sparse enough
    -- `sparse_pattern` True / False forces the choice --, else densified / bit-packed on the device (SURVEY.md 8f
    rank 3); dense inputs as they are."""
    n, d = X.shape
    A = None
    if W_mat is not None and sparse_pattern is not False:
        A = _observed_csr(X, W_mat)
        if A is not None and sparse_pattern is None and A.nnz > SPARSE_PATTERN_MAX_DENSITY * float(n) * d:
            A = None
        if A is None and sparse_pattern is True:
            raise ValueError('sparse_pattern=True needs scipy sparse X and 0/1 W_mat with X zero outside the pattern')
    if A is not None:
        eng = RRIEngine(n, d, k, dtype=sdt, weighted='sparse', device=device)
        try:
            eng.upload_observed_csr(A)
        except Exception:
            eng.close()
            raise
        return eng
    eng = RRIEngine(n, d, k, dtype=sdt, weighted=W_mat is not None, device=device,
                    schedule=schedule if W_mat is None else 'gram')
    try:
        _upload_problem(eng, X, W_mat)
    except Exception:
        eng.close()
        raise
    return eng


def _upload_problem(eng, X, W_mat):
    if scipy.sparse.issparse(X):
        eng.upload_X_csr(X)
    else:
        eng.upload_X(X)
    if W_mat is not None:
        if scipy.sparse.issparse(W_mat):
            eng.upload_mask_csr_pattern(W_mat)
        else:
            eng.upload_mask(W_mat)


def _sparse_mask_or_dense(W_mat):
    """a scipy sparse W_mat stays sparse only when it is an observation pattern (all stored values 1)"""
    if W_mat is None or not scipy.sparse.issparse(W_mat):
        return W_mat
    W_mat = W_mat.tocsr(copy=True)      # the caller's matrix is never modified
    W_mat.eliminate_zeros()
    return W_mat if np.all(W_mat.data == 1) else W_mat.toarray()


def _storage_dtype(X, dtype):
    if dtype is not None:
        return np.dtype(dtype)
    return np.dtype(np.float32) if getattr(X, 'dtype', None) == np.float32 else np.dtype(np.float64)


def _is_empty(a):
    return int(np.prod(np.shape(a))) == 0


class _ResidentX(object):
    """What the initialisers need of an X that exists only on the device (after device-side preprocessing):
    its shape and its mean; the products with it go through the engine."""

    def __init__(self, engine):
        self.shape = (engine.n, engine.d)
        self._engine = engine
        self._mean = None

    def mean(self):
        if self._mean is None:
            rows = self._engine.X_times(np.ones((self.shape[1], 1)))
            self._mean = float(rows.sum()) / (float(self.shape[0]) * self.shape[1])
        return self._mean


class ResidentProblem(object):
    """Keeps the device handle of an nmf() call -- X uploaded (and, with `preprocess`, rewritten by tf-idf / normalisation), the
    weights, ||X||^2 -- alive for later calls on the SAME problem: `nmf(X, k, ..., resident=holder)` reuses it when the identity
    of X and W_mat (the very objects), their shapes and types, k, the storage type, the device, the schedule and the
    preprocessing asked for are those of the call that made it, and makes (and keeps) a new one otherwise.  What the estimators'
    `one_iter` loop saves at 100000 x 10000 is the upload and the handle per call (sklearn_interface.py:316-318 re-runs nmf() on
    the same X every time).  The caller promises not to modify X or W_mat in place between the calls -- nothing here can notice --
    and closes the holder (`close()`, or its end of life) when done.  Not for row-sharded calls or per-row weights."""

    def __init__(self):
        self.engine = None
        self.key = None
        self.idf = None
        self.reuses = 0

    def close(self):
        if self.engine is not None:
            try:
                self.engine.close()
            finally:
                self.engine, self.key, self.idf = None, None, None

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001 -- interpreter shutdown
            pass


def _preprocess_spec(preprocess):
    """(tfidf, normalize) of the `preprocess` option: None, a dict {'tfidf': True | idf vector | False,
    'normalize': bool}, or a string / sequence naming the steps ('tfidf', 'normalize')"""
    if preprocess is None:
        return None
    if isinstance(preprocess, str):
        preprocess = (preprocess,)
    if isinstance(preprocess, dict):
        unknown = set(preprocess) - {'tfidf', 'normalize'}
        if unknown:
            raise ValueError('unknown preprocessing step(s): %s' % sorted(unknown))
        tfidf_opt, norm_opt = preprocess.get('tfidf', False), bool(preprocess.get('normalize', False))
    else:
        steps = list(preprocess)
        unknown = set(steps) - {'tfidf', 'normalize'}
        if unknown:
            raise ValueError('unknown preprocessing step(s): %s' % sorted(unknown))
        tfidf_opt, norm_opt = 'tfidf' in steps, 'normalize' in steps
    if tfidf_opt is None or isinstance(tfidf_opt, (bool, np.bool_)):
        tfidf_opt = bool(tfidf_opt)
    if tfidf_opt is False and not norm_opt:
        return None
    return tfidf_opt, norm_opt


def _preprocess_on_host(X, tfidf_opt, norm_opt):
    """matrixops.tfidf / normalize on the host: the route when something (callbacks, weights, sparse input)
    needs the preprocessed X in host memory"""
    from .matrixops import tfidf as host_tfidf
    idf = None
    if tfidf_opt is True:
        X, idf = host_tfidf(X, return_idf=True)
        if scipy.sparse.issparse(X):
            X = X.tocsr()
        idf = np.asarray(idf.todense() if scipy.sparse.issparse(idf) else idf, dtype=np.float64).ravel()
    elif tfidf_opt is not False:
        idf = np.asarray(tfidf_opt, dtype=np.float64).ravel()
        X = X.multiply(idf).tocsr() if scipy.sparse.issparse(X) else X * idf
    if norm_opt and scipy.sparse.issparse(X):     # matrixops.normalize is written for dense arrays: same arithmetic on CSR
        tot = np.asarray(X.sum(1)).ravel() + np.spacing(1)
        X = (scipy.sparse.diags(1.0 / tot) @ X).tocsr()
        empty = np.nonzero(tot < 1e-10)[0]
        if empty.size:
            X = X.tolil()
            for i in empty:
                X[i, :] = 1.0 / X.shape[1]
            X = X.tocsr()
    elif norm_opt:
        X = normalize(X)
    return X, idf


DEVICE_INIT_MIN_ELEMS = 2e7   # from this many entries of X on, the SVD behind NNDSVD uses the device products


def _initialize_and_validate(W_in, T_in, W_mat, X, k, init, random_state, project_T_each_iter,
                             project_W_each_iter, w_row_sum, t_row_sum, fix_W, fix_T, n, d, engine=None, **_):
    """Starting W, T (nmf.py:819-880): initialise BOTH when either input is empty, let W_in / T_in
    override, clamp at 0 (copies: the caller's arrays are never written), project when the
    constraints are kept every sweep."""
    if _is_empty(W_in) or _is_empty(T_in):
        if W_mat is None:
            src = X
        elif scipy.sparse.issparse(W_mat) or scipy.sparse.issparse(X):
            src = scipy.sparse.csr_matrix(W_mat).multiply(X).tocsr()
        else:
            src = W_mat * X
        grp = getattr(engine, 'group', None)
        if grp is not None and W_mat is not None and not getattr(engine, 'sparse', False) and init != 'random':
            # row-sharded weighted start (nmf.py:841-843 factorises W_mat .* X): the weighted handle keeps X and the weights
            # apart, so the rows of W_mat .* X go up once more, onto a scratch handle under the same group, for the
            # products of the sharded randomized SVD (initialization.randomized_svd_sharded)
            with RRIEngine(n, d, k, dtype=engine.dtype, device=engine.device) as scratch:
                scratch.upload_X(np.ascontiguousarray(src, dtype=engine.dtype))
                scratch.attach_group(grp)
                W, T = initialize_nmf(src, k, init, random_state=random_state, row_normalize=False, engine=scratch)
        else:
            W, T = initialize_nmf(src, k, init, random_state=random_state, row_normalize=False,
                                  # the device products see X (dense handles) or X on the pattern = W_mat .* X
                                  engine=engine if (W_mat is None or getattr(engine, 'sparse', False) or grp is not None) else None)
        if t_row_sum is not None:
            T = normalize(T) * t_row_sum
        if w_row_sum is not None:
            W = normalize(W) * w_row_sum
    if not _is_empty(W_in):
        if np.shape(W_in) != (n, k):
            raise ValueError('W_in has wrong dimensions, must be n*k')
        W = W_in
    if not _is_empty(T_in):
        if np.shape(T_in) != (k, d):
            raise ValueError('T_in has wrong dimensions, must be k*d')
        T = T_in
    if scipy.sparse.issparse(T):
        T = T.toarray()
    if scipy.sparse.issparse(W):
        W = W.toarray()
    W = np.maximum(W, 0)
    T = np.maximum(T, 0)
    if project_W_each_iter and not fix_W and w_row_sum is not None:
        W = proj_mat_to_simplex(W, w_row_sum)
    if project_T_each_iter and not fix_T and t_row_sum is not None:
        T = proj_mat_to_simplex(T, t_row_sum)
    return W, T


def _gradient_recorder(eng, X, W_mat, rows, numer_out, denom_out):
    """observe(t, wR, nw) for RRIEngine.sweep_stepwise: appends what _compute_update_T returns as wR_store, nw_store
    (nmf.py:677-686, 706-713) -- the sums over all rows as they come from the device, or, with `ind_rows_to_store`,
    the same sums over those rows only, taken on the host from the current factors"""
    if rows is None:
        def observe(t, wR, nw):
            numer_out.append(wR)
            denom_out.append(nw)
        return observe
    rows = np.asarray(rows)
    Xs = X[rows, :]
    Xs = Xs.toarray() if scipy.sparse.issparse(Xs) else np.asarray(Xs, dtype=np.float64)
    Ms = None
    if W_mat is not None:
        Ms = W_mat[rows, :]
        Ms = Ms.toarray() if scipy.sparse.issparse(Ms) else np.asarray(Ms, dtype=np.float64)

    def observe(t, wR, nw):
        Ws, T = eng.get_W()[rows, :], eng.get_T()
        ws = Ws[:, t].copy()
        if Ms is None:                            # nmf.py:680-686
            wWs = ws.dot(Ws)
            wWs[t] = 0
            numer_out.append(ws.dot(Xs) - wWs.dot(T))
            denom_out.append((ws ** 2).sum())
        else:                                     # nmf.py:709-713
            Wz = Ws.copy()
            Wz[:, t] = 0
            numer_out.append(ws.dot(Ms * (Xs - Wz.dot(T))))
            denom_out.append((ws ** 2).dot(Ms))
    return observe


def _compute_update_T(X, W, T, t, store_gradients=False, ind_rows_to_store=None, W_mat=None, **kwargs):
    """(wR, nw, wR_store, nw_store) of nmf.py:633-715 for the given factors: the sums behind the update of row t of T,
    `w_t^T (X - sum_{j != t} w_j t_j)` and `||w_t||^2` (weighted: through the masked residual, both d-vectors), taken on
    the device.  The reference's test file imports this name; nmf() itself steps on the device and does not call it.
    `dtype=` / `device=` as for nmf(); other keywords (the reference passes its `locals()`) are ignored."""
    X = X.tocsr() if scipy.sparse.issparse(X) else np.asarray(X)
    W_mat = _sparse_mask_or_dense(W_mat)
    W, T = np.asarray(W, dtype=np.float64), np.asarray(T, dtype=np.float64)
    k = W.shape[1]
    if not 0 <= int(t) < k:
        raise IndexError('topic index out of range')
    sdt = _storage_dtype(X, kwargs.get('dtype'))
    with _engine_with_problem(X, W_mat, k, sdt, kwargs.get('device', 0), kwargs.get('sparse_pattern')) as eng:
        eng.set_W(W)
        eng.set_T(T)
        eng.set_params(reset_topic_method=None)
        wR, nw = eng.topic_sums(int(t))
        wR_store = nw_store = None
        if store_gradients:
            numer, denom = [], []
            _gradient_recorder(eng, X, W_mat, ind_rows_to_store, numer, denom)(int(t), wR, nw)
            wR_store, nw_store = numer[0], denom[0]
    return wR, nw, wR_store, nw_store


def _sentinel(W, T):
    return {'W': W, 'T': T, 'obj_history': [-np.inf], 'iter_cputime': [0]}


def nmf(X, k, w_row=None, W_mat=None, fix_W=False, fix_T=False,
        random_state=None, init='nndsvd', T_in=[], W_in=[], max_iter=200,
        max_time=600, eps_stop=1e-4, compute_obj_each_iter=False,
        project_W_each_iter=False, w_row_sum=None,
        do_final_project_W=True, project_T_each_iter=False,
        t_row_sum=None, early_stop=None,
        reset_topic_method='max_resid_document', fix_reset_seed=False,
        n_resets=23,
        reg_w_l2=0, reg_t_l2=0, reg_w_l1=0, reg_t_l1=0,
        diagnostics=[], store_gradients=False,
        ind_rows_to_store=None, eps_gauss_t=None, delta_gauss_t=None,
        *, dtype=None, device=0, device_init=None, sparse_pattern=None, preprocess=None, schedule='gram', group=None,
        resident=None):
    """Non-negative factorisation X ~ W T by rank-one residue iteration; see the module docstring and
    the reference's docstring (nmf.py:109-269) for the parameters.  Returns a dict with 'W', 'T',
    'iter_cputime' (wall seconds since the start, per sweep), 'random_state' and, when the objective
    is tracked, 'obj_history' and 'obj_calculator'; 'diagnostics' when callbacks are given.

    preprocess (keyword only, not in the reference's signature): tf-idf and / or row normalisation of X
    (matrixops.py:124-179) before the factorisation -- {'tfidf': True | idf vector | False, 'normalize': bool} or
    the step names.  A dense X without weights or host callbacks is uploaded raw and rewritten in place on the
    device; every other case preprocesses on the host.  The idf used comes back as rtv['idf'].
    resident (keyword only): a ResidentProblem that keeps the device handle -- X uploaded and preprocessed -- from one call to
    the next on the same problem (see there)."""
    if group is not None:
        # host work that would need the other ranks' rows (the SVD behind the NNDSVD start, document frequencies,
        # per-row weights with their refit) or that decides per rank (callbacks) is not part of the sharded call
        if (_is_empty(W_in) or _is_empty(T_in)) and W_mat is not None and init != 'random' \
                and (scipy.sparse.issparse(W_mat) or scipy.sparse.issparse(X)):
            raise ValueError('a row-sharded weighted call on scipy sparse inputs needs W_in (this rank\'s rows) and T_in')
        # an early_stop callback that carries `device_entries` (what NMF_RS_Estimator.fit installs: the clipped RMSE on its
        # held-out entries, sklearn_interface.py:71-93) is scored by the library over ALL ranks' entries, so every rank takes
        # the same decision (nmf.py:381-407); a callback that wants W, T on the host would see one rank's rows
        host_stop = callable(early_stop) and getattr(early_stop, 'device_entries', None) is None
        # (round 4: per-row weights -- row-local, their refit a fold-in under the same group -- and the device-side tf-idf /
        # normalisation -- document frequencies all-reduced -- run sharded)
        if store_gradients or (eps_gauss_t and delta_gauss_t) or host_stop:
            raise NotImplementedError('store_gradients, the Gaussian mechanism and early_stop callbacks '
                                      'that need W, T on the host are single-handle options')
        if preprocess is not None and (scipy.sparse.issparse(X) or W_mat is not None or w_row is not None or diagnostics):
            raise NotImplementedError('row-sharded preprocessing runs on the device: a dense X without weights or callbacks')
    draw_noise = None
    if eps_gauss_t and delta_gauss_t and not fix_T:
        # Gaussian mechanism on the T-row sums (nmf.py:422-435; Dwork & Roth p. 261)
        from scipy.stats import norm as gaussian
        c2 = 2 * np.log(1.25 / float(delta_gauss_t)) + 0.001
        df2 = 1000.0
        sigma2 = c2 * df2 ** 2 * (1 / float(eps_gauss_t)) ** 2
        noise = gaussian(0, np.sqrt(sigma2))            # draws from numpy's global RNG, as the reference's does
        draw_noise = lambda m: np.asarray(noise.rvs(m), dtype=np.float64).ravel()
    X_given, W_mat_given = X, W_mat
    # scipy sparse X / 0-1 sparse W_mat are ingested as CSR (no host densification); row weights need a dense X
    if scipy.sparse.issparse(X):
        X = X.tocsr() if w_row is None else X.toarray()
    else:
        X = np.asarray(X)
    W_mat = _sparse_mask_or_dense(W_mat)
    rtv = {}
    spec = _preprocess_spec(preprocess)
    device_spec = None
    if spec is not None:
        host_callbacks = bool(diagnostics) or bool(store_gradients) or \
            (callable(early_stop) and getattr(early_stop, 'device_entries', None) is None)
        if scipy.sparse.issparse(X) or W_mat is not None or w_row is not None or host_callbacks:
            X, rtv['idf'] = _preprocess_on_host(X, *spec)
        else:
            device_spec = {'tfidf': spec[0], 'normalize': spec[1]}
    n, d = X.shape

    # ---- option sanity, exactly as nmf.py:280-315 ---------------------------------------------
    if project_T_each_iter and np.any([reg_w_l1, reg_t_l1]):
        logger.warning('This implementation can not solve project_T_each_iter=True with '
                       'regularization. Because WT is no longer scale invariant. Setting '
                       'project_T_each_iter to False.')
        project_T_each_iter = False
    if project_W_each_iter and reg_w_l2 < 0:
        logger.warning('project_W_each_iter={} and reg_w_l2={}<0 doesnt converge with the current '
                       'implementation.'.format(project_W_each_iter, reg_w_l2))
    if (not project_T_each_iter and not t_row_sum) and (reg_t_l1 < 0 or reg_t_l2 < 0):
        logger.error('Unbounded objective. reg_t_l1={}, reg_t_l2={} but project_T_each_iter={} and '
                     't_row_sum={}'.format(reg_t_l1, reg_t_l2, project_T_each_iter, t_row_sum))
        return _sentinel(np.ones((n, k)), np.ones((k, d)) * 1e6)
    if (not project_W_each_iter and not w_row_sum) and (reg_w_l1 < 0 or reg_w_l2 < 0):
        logger.error('Unbounded objective. reg_w_l1={}, reg_w_l2={} but project_W_each_iter={} and '
                     'w_row_sum={}'.format(reg_w_l1, reg_w_l2, project_W_each_iter, w_row_sum))
        return _sentinel(np.ones((n, k)) * 1e6, np.ones((k, d)))

    if type(diagnostics) is not list:
        diagnostics = [diagnostics]
    if diagnostics:
        rtv['diagnostics'] = {f.__name__: [] for f in diagnostics}
    if store_gradients:                          # nmf.py:325-327
        rtv['numer_W'], rtv['denom_W'] = {}, {}
    if random_state is None:
        random_state = int(time.time()) % 4294967296

    wall0 = time.time()
    max_time = max_time - 10  # nmf.py:333

    X_orig = None
    if w_row is not None:                       # nmf.py:335-338
        X_orig = X.copy()
        X = np.sqrt(w_row) * X
    if w_row_sum is not None and not np.isscalar(w_row_sum):   # nmf.py:340-344
        w_row_sum = w_row_sum.reshape((w_row_sum.size, 1))
        if w_row is not None:
            w_row_sum = np.sqrt(w_row_sum)
    if (group.n_global if group is not None else n) <= k:
        init = 'random'

    clock0 = time.perf_counter()
    if logger.level <= logging.DEBUG:           # nmf.py:366-367 (see the note at the logger)
        compute_obj_each_iter = True

    # host preprocessing yields float64 whatever X was (X * idf promotes): the device route stores the same by default
    sdt = _storage_dtype(X, dtype) if device_spec is None else np.dtype(dtype or np.float64)
    needs_init = _is_empty(W_in) or _is_empty(T_in)
    if not _is_empty(W_in) and np.shape(W_in) != (n, k):       # shape errors before any device work (nmf.py:853-860)
        raise ValueError('W_in has wrong dimensions, must be n*k')
    if not _is_empty(T_in) and np.shape(T_in) != (k, d):
        raise ValueError('T_in has wrong dimensions, must be k*d')
    if schedule == 'residual' and W_mat is not None:
        raise NotImplementedError("schedule='residual' belongs to the unweighted flavour (the weighted one always keeps its masked residual)")
    # fix_W / fix_T / k = 1 on a handle of the explicit-residual schedule: one half of every step is missing, and the library
    # steps such calls in the Gram form (T fixed: X T^T once, no pass over the matrix per topic) -- fold-in works on either handle
    # a handle kept from an earlier call on the same problem (ResidentProblem): the same objects, shapes, types and options
    res_key = None
    if resident is not None:
        if group is not None or w_row is not None or (spec is not None and device_spec is None):
            raise ValueError('resident= keeps a handle for plain calls: no group, no w_row, no host-side preprocessing')
        tf_opt = None if device_spec is None else (device_spec['tfidf'] if isinstance(device_spec['tfidf'], bool) else 'given')
        res_key = (id(X_given), tuple(X.shape), str(getattr(X, 'dtype', None)), None if W_mat_given is None else id(W_mat_given),
                   int(k), str(sdt), device, sparse_pattern, schedule,
                   None if device_spec is None else (tf_opt, bool(device_spec['normalize'])))
        if tf_opt == 'given':
            res_key = None               # an idf vector from outside: not worth telling apart -- a fresh handle
    reused = resident is not None and resident.engine is not None and res_key is not None and resident.key == res_key
    if reused:
        eng = resident.engine
        eng.begin_run()
        resident.reuses += 1
    else:
        if resident is not None:
            resident.close()
        eng = _engine_with_problem(X, W_mat, k, sdt, device, sparse_pattern, schedule)
    keep_handle = False
    try:
        if group is not None:
            eng.attach_group(group)
        on_device = device_init if device_init is not None else (float(n) * d >= DEVICE_INIT_MIN_ELEMS)
        X_init = X
        if device_spec is not None:
            idf = resident.idf if reused else eng.preprocess(**device_spec)
            rtv['idf'] = idf
            device_spec = {'tfidf': idf if idf is not None else False, 'normalize': device_spec['normalize']}
            on_device = True                 # the preprocessed X exists only on the device
            X_init = _ResidentX(eng)
        if resident is not None and res_key is not None and not reused:
            resident.engine, resident.key, resident.idf = eng, res_key, rtv.get('idf')
        W, T = _initialize_and_validate(W_in=W_in, T_in=T_in, W_mat=W_mat, X=X_init, k=k, init=init,
                                        random_state=random_state, project_T_each_iter=project_T_each_iter,
                                        project_W_each_iter=project_W_each_iter, w_row_sum=w_row_sum,
                                        t_row_sum=t_row_sum, fix_W=fix_W, fix_T=fix_T, n=n, d=d,
                                        engine=eng if (group is not None or (on_device and needs_init and init not in ('random', 'smart_random'))) else None)
        eng.set_W(W)
        eng.set_T(T)
        eng.set_params(fix_W=fix_W, fix_T=fix_T, project_T_each_iter=project_T_each_iter,
                       t_row_sum=t_row_sum, w_row_sum=w_row_sum, reset_topic_method=reset_topic_method,
                       n_resets=n_resets, reg_w_l1=reg_w_l1, reg_w_l2=reg_w_l2, reg_t_l1=reg_t_l1,
                       reg_t_l2=reg_t_l2, fix_reset_seed=fix_reset_seed)

        def current():
            return eng.get_W(), eng.get_T()

        for f in diagnostics:
            rtv['diagnostics'][f.__name__].append(f(X, W, T))

        iter_cputime, obj_history = [], []
        last_score = np.inf
        if early_stop:
            eng.snapshot()

        # Launch-bound sizes: once o_0 and o_1 are known, the sweep / objective / stop-rule loop below runs on the device in
        # chunks (rri_sweep_until: the persistent kernel keeps every sweep's objective and applies the rule of
        # optimization.py:284-291 itself) -- whenever nothing has to see W, T between the sweeps.  RRI_NMF_CHUNK=0: sweep by sweep.
        chunked = (compute_obj_each_iter and not early_stop and not diagnostics and not store_gradients and draw_noise is None
                   and not (project_W_each_iter and not fix_W and w_row_sum is not None) and group is None
                   and os.environ.get('RRI_NMF_CHUNK', '1') != '0')
        # ... and without an objective to watch (the caller raised the logger's level, see the note at the logger) nothing happens
        # between the sweeps at all: they go to the device in chunks on any handle, bounded by max_time alone
        blind = (not compute_obj_each_iter and not early_stop and not diagnostics and not store_gradients and draw_noise is None
                 and not (project_W_each_iter and not fix_W and w_row_sum is not None) and group is None
                 and os.environ.get('RRI_NMF_CHUNK', '1') != '0')
        iter_no = 0
        while iter_no < max_iter:
            if blind and len(iter_cputime) >= 2 and max_iter - iter_no >= 2:
                per_sweep = max(iter_cputime[-1] - iter_cputime[-2], 1e-6)
                budget = max_time - (time.time() - wall0)
                m = int(min(max_iter - iter_no, max(budget, 0.0) / per_sweep, max(0.5 / per_sweep, 1.0)))
                if m >= 2:
                    c0 = iter_cputime[-1]
                    done = eng.sweep(m)
                    c1 = time.perf_counter()
                    iter_cputime.extend(c0 + (j + 1) * (c1 - c0) / max(done, 1) for j in range(done))
                    iter_no += done
                    if time.time() - wall0 >= max_time:
                        logger.info('STOPPING because max_time after iter %d' % (iter_no - 1))
                        break
                    continue
            if chunked and len(obj_history) >= 2 and max_iter - iter_no >= 2:
                per_sweep = max(iter_cputime[-1] - iter_cputime[-2], 1e-6)      # the latest sweep (the first ones carry one-off costs)
                budget = max_time - (time.time() - wall0)
                m = int(min(max_iter - iter_no, 512, max(budget, 0.0) / per_sweep, max(0.5 / per_sweep, 2.0)))
                res = eng.sweep_until(m, obj_history[-1], eps_stop * abs(obj_history[0] - obj_history[1])) if m >= 2 else None
                if res is None or res[0] < 1:
                    chunked = False                  # not (or no longer) on the persistent path
                else:
                    done, hist = res
                    c0, c1 = iter_cputime[-1], time.perf_counter()
                    stop_now = False
                    for j in range(done):
                        # NaN: the sweep an event interrupted, or one that ran launch by launch -- the kernel left no value
                        obj_history.append(float(hist[j]) if not np.isnan(hist[j]) else eng.objective())
                        logger.info('\tObj: {0:3.3e}'.format(obj_history[-1]))
                        iter_cputime.append(c0 + (j + 1) * (c1 - c0) / done)      # one launch: the sweeps share its time evenly
                        iter_no += 1
                        if universal_stopping_condition(obj_history, eps_stop=eps_stop):
                            logger.info('STOPPING because obj_history after iter %d' % (iter_no - 1))
                            stop_now = True
                            break
                    if stop_now:
                        break
                    if time.time() - wall0 >= max_time:
                        logger.info('STOPPING because max_time after iter %d' % (iter_no - 1))
                        break
                    continue
            if early_stop:                       # nmf.py:381-407
                if callable(early_stop):
                    entries = getattr(early_stop, 'device_entries', None)
                    if entries is not None:   # clipped RMSE on listed entries: evaluated on the device
                        this_score = eng.masked_rmse(*entries)
                    else:
                        Wh, Th = current()
                        this_score = early_stop(X, Wh, Th)
                elif compute_obj_each_iter:
                    this_score = np.inf if not obj_history else obj_history[-1]
                logger.info('Iter %d stopping score %.3f' % (iter_no, this_score))
                if this_score > last_score:
                    eng.rollback()
                    obj_history = obj_history[:-1]
                    iter_cputime = iter_cputime[:-1]
                    for f in diagnostics:
                        rtv['diagnostics'][f.__name__] = rtv['diagnostics'][f.__name__][:-1]
                    break
                last_score = this_score
                eng.snapshot()

            sweep_t0 = time.time()
            observe = None
            if store_gradients:                   # nmf.py:411-413, 454-456: the sums behind every T-row update
                numer_it, denom_it = [], []
                rtv['numer_W'][iter_no], rtv['denom_W'][iter_no] = numer_it, denom_it
                if not fix_T:
                    observe = _gradient_recorder(eng, X, W_mat, ind_rows_to_store, numer_it, denom_it)
            if draw_noise is None and observe is None:
                eng.sweep(1)                      # the topic loop, nmf.py:415-476
            else:
                eng.sweep_stepwise(draw=draw_noise, observe=observe)

            if project_W_each_iter and not fix_W and w_row_sum is not None:   # nmf.py:481-484
                eng.project_W_rows(w_row_sum if np.isscalar(w_row_sum) else w_row_sum.ravel())
            if compute_obj_each_iter:
                obj_history.append(eng.objective())
                logger.info('\tObj: {0:3.3e}'.format(obj_history[-1]))
            iter_cputime.append(time.perf_counter())
            if diagnostics:
                Wh, Th = current()
                for f in diagnostics:
                    rtv['diagnostics'][f.__name__].append(f(X, Wh, Th))
            logger.info('\tTime: %.3fsec' % (time.time() - sweep_t0))
            out_of_time = time.time() - wall0 >= max_time
            if group is not None:            # rank 0's clock decides for everybody
                out_of_time = bool(eng.comm_broadcast([float(out_of_time)], 0)[0])
            if out_of_time:
                logger.info('STOPPING because max_time after iter %d' % iter_no)
                break
            if compute_obj_each_iter and universal_stopping_condition(obj_history, eps_stop=eps_stop):
                logger.info('STOPPING because obj_history after iter %d' % iter_no)
                break
            iter_no += 1

        iter_cputime = [c - clock0 for c in iter_cputime]

        # final row-wise projection of W (nmf.py:519-529)
        if not project_W_each_iter and w_row_sum is not None and not fix_W and do_final_project_W:
            eng.project_W_rows(w_row_sum if np.isscalar(w_row_sum) else w_row_sum.ravel())

        W, T = current()
        n_resets_used = eng.n_resets_used
        keep_handle = resident is not None and resident.engine is eng
    finally:
        if keep_handle:
            pass                         # the holder's from here on
        elif resident is not None and resident.engine is eng:
            resident.close()             # the call failed: a handle in an unknown state is not kept
        else:
            eng.close()

    if w_row is not None:                        # nmf.py:531-539: refit W on the unweighted rows
        sub = nmf(X_orig, k, T_in=T, fix_T=True, max_iter=10, w_row_sum=w_row_sum,
                  project_W_each_iter=True, compute_obj_each_iter=compute_obj_each_iter,
                  dtype=dtype, device=device, group=group)
        obj_history.extend(sub.get('obj_history', []))
        iter_cputime.extend(sub['iter_cputime'])
        W = sub['W']

    if store_gradients:
        # nmf.py:541-549 means to stack the k row vectors of a sweep (its reshape lambda is passed in the place of
        # `dict_key`, which raises; as `transform` it gives this): numer_W[sweep] is k x d, denom_W[sweep] k x 1
        # (k x d for weighted problems)
        as_row = lambda v: np.asarray(v, dtype=np.float64).reshape((1, np.size(v)))
        for key in ('numer_W', 'denom_W'):
            for it, rows in rtv[key].items():
                rtv[key][it] = np.vstack([as_row(v) for v in rows]) if rows else np.zeros((0, 0))
    rtv['W'] = W
    rtv['T'] = T
    if compute_obj_each_iter:
        rtv['obj_history'] = obj_history
        calc = TrueObjComputer(X, W, T, reg_w_l2, reg_t_l2, reg_w_l1, reg_t_l1, W_mat, w_row,
                               dtype=dtype, device=device, sparse_pattern=sparse_pattern, preprocess=device_spec, group=group)
        calc.obj = obj_history[-1] if obj_history else np.inf
        rtv['obj_calculator'] = calc
    rtv['iter_cputime'] = iter_cputime
    rtv['random_state'] = random_state
    rtv['n_resets_used'] = n_resets_used
    return rtv
