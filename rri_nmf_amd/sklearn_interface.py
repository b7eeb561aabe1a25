"""sklearn-style estimators over nmf() -- same classes, constructor arguments and methods as the
reference's src/rri_nmf/sklearn_interface.py (NMF_RS_Estimator :14-182, NMF_TM_Estimator :185-345),
so `fit / fit_transform / one_iter / transform / predict / score` keep working while the solver
underneath runs on the MI355X.
"""
import numpy as np
import scipy.sparse as sp
import sklearn.base
from sklearn.model_selection import train_test_split
from sklearn.utils.validation import check_X_y, check_array

from . import nmf as _nmf_module


def _nmf(*a, **kw):
    return _nmf_module.nmf(*a, **kw)


class _FactorPair(object):
    """W / T storage shared by both estimators (sparsify / densify: :42-57, :230-245)."""

    def sparsify(self):
        self.W = self.W.tocsr() if sp.issparse(self.W) else sp.csr_matrix(self.W)
        self.T = self.T.tocsr() if sp.issparse(self.T) else sp.csr_matrix(self.T)

    def densify(self):
        if sp.issparse(self.W):
            self.W = self.W.toarray()
        if sp.issparse(self.T):
            self.T = self.T.toarray()

    def _warm_start(self):
        """the estimator's own factors continue a fit (:105-112, :254-261)"""
        return (self.W if self.W.size > 0 else []), (self.T if self.T.size > 0 else [])

    def _keep(self, soln):
        self.W = soln.pop('W')
        self.T = soln.pop('T')
        self.nmf_outputs = soln


def _observed_mask(X):
    """W_mat of the recommender flavour: 1 where X holds a non-zero rating (sklearn_interface.py:99-102)"""
    if sp.issparse(X):
        M = X.copy()
        M.data = np.ones(M.nnz)
        return M
    M = np.zeros(X.shape)
    M[X.nonzero()] = 1
    return M


def _ratings_matrix(values, ij, shape):
    """the ratings as a CSR matrix: what the reference densifies with .toarray() (sklearn_interface.py:78-97)
    stays sparse here -- nmf() then keeps X, W_mat and the residual on the observed entries only"""
    A = sp.coo_matrix((np.asarray(values, dtype=np.float64), (ij[:, 0], ij[:, 1])), shape=shape).tocsr()
    A.sum_duplicates()
    A.eliminate_zeros()      # a zero is "not observed" (W_mat = [X != 0])
    return A


class NMF_RS_Estimator(_FactorPair, sklearn.base.BaseEstimator):
    """Recommender flavour: elementwise-weighted RRI (WRRI) on the observed entries, T entries
    clipped to [0, 1], optional early stopping on a 5 % hold-out (sklearn_interface.py:14-182)."""

    def __init__(self, n, d, k, wr1=0, tr1=0, random_state=0,
                 W=np.array([]), T=np.array([]), max_iter=30, nmf_kwargs={},
                 use_validation_early_stopping=True):
        self.n, self.d, self.k = n, d, k
        self.max_iter = max_iter
        self.wr1, self.tr1 = wr1, tr1
        self.random_state = random_state
        self.min_rating = None
        self.max_rating = None
        self.Xpred = np.array([])
        self.use_validation_early_stopping = use_validation_early_stopping
        self.W, self.T = W, T
        self.nmf_kwargs = nmf_kwargs

    def fit(self, X, y=None):
        """X: (m, 2) index pairs (i, j); y: the m observed values X[i, j]."""
        X, y = check_X_y(X, y)
        self.min_rating, self.max_rating = np.min(y), np.max(y)
        grp = self.nmf_kwargs.get('group')
        if grp is not None:
            # row-sharded fit (one rank's rows, LOCAL row indices in X): the clip bounds of predictions and of the early-stop
            # score are those of ALL ratings, so that every rank scores and stops alike
            mm = grp.gather([self.min_rating, self.max_rating], device=self.nmf_kwargs.get('device', 0))
            self.min_rating, self.max_rating = mm[:, 0].min(), mm[:, 1].max()
        shape = (self.n, self.d)
        if self.use_validation_early_stopping:
            ij_tr, ij_val, r_tr, r_val = train_test_split(X, y, test_size=0.05, random_state=0,
                                                          stratify=None)
            Xtr = _ratings_matrix(r_tr, ij_tr, shape)
            Xv = _ratings_matrix(r_val, ij_val, shape).tocoo()
            vi, vj, vr = Xv.row, Xv.col, Xv.data
            lo, hi = self.min_rating, self.max_rating

            def RMSE_val(X_ignored, W, T):
                pred = np.clip(np.einsum('ij,ji->i', W[vi, :], T[:, vj]), lo, hi)
                return np.sqrt(np.mean((pred - vr) ** 2))

            # same score without bringing W, T to the host: nmf() evaluates callbacks that carry `device_entries`
            # with rri_masked_rmse on the device
            RMSE_val.device_entries = (vi, vj, vr, lo, hi)
            self.early_stop = RMSE_val
        else:
            self.early_stop = False
            Xtr = _ratings_matrix(y, X, shape)
        W_in, T_in = self._warm_start()
        soln = _nmf(Xtr, self.k, max_iter=self.max_iter, max_time=7200, compute_obj_each_iter=True,
                    reset_topic_method=None, early_stop=self.early_stop, project_T_each_iter=False,
                    t_row_sum=1.0, project_W_each_iter=False, w_row_sum=None,
                    W_mat=_observed_mask(Xtr), W_in=W_in, T_in=T_in, reg_w_l1=self.wr1,
                    reg_t_l1=self.tr1, random_state=self.random_state, **self.nmf_kwargs)
        self._keep(soln)
        return self

    def fit_from_Xtr(self, Xtr):
        """builds the (index pairs, values) lists from a ratings matrix and fits"""
        Xtr = Xtr.tocsr() if sp.issparse(Xtr) else sp.csr_matrix(Xtr)
        rows, cols = Xtr.nonzero()
        return self.fit(np.column_stack((rows, cols)), Xtr.data)

    def transform(self, Xnew):
        """fold new rows in against the fitted T"""
        soln = _nmf(Xnew, self.k, max_iter=4, max_time=7200, project_W_each_iter=False,
                    project_T_each_iter=False, W_mat=_observed_mask(Xnew), T_in=self.T, fix_T=True,
                    reg_w_l1=self.wr1, reg_t_l1=self.tr1, t_row_sum=1.0, w_row_sum=None,
                    reset_topic_method='random', random_state=self.random_state, **self.nmf_kwargs)
        return soln['W']

    def make_Xpred(self):
        """the full clipped reconstruction, n x d dense (sklearn_interface.py:158-161): kept for callers that want it;
        predict / score below evaluate only the entries they are asked for"""
        if self.Xpred.size == 0:
            self.Xpred = np.clip(np.dot(self.W, self.T), a_min=self.min_rating, a_max=self.max_rating)

    def _predict_entries(self, i, j):
        """clip(W T)[i, j] for index arrays i, j without forming the n x d product"""
        if self.Xpred.size > 0:
            return self.Xpred[i, j]
        W = self.W.toarray() if sp.issparse(self.W) else self.W
        T = self.T.toarray() if sp.issparse(self.T) else self.T
        out = np.empty(len(i))
        for lo in range(0, len(i), 1 << 20):          # chunks: the gathered rows are 2^20 x k doubles
            sl = slice(lo, lo + (1 << 20))
            out[sl] = np.einsum('ek,ke->e', W[i[sl], :], T[:, j[sl]])
        return np.clip(out, self.min_rating, self.max_rating)

    def predict(self, X):
        X = check_array(X)
        # the reference indexes with the float array check_array returns, which modern numpy rejects
        idx = np.asarray(X, dtype=np.intp)
        return self._predict_entries(idx[:, 0], idx[:, 1])

    def score(self, X, y=np.array([])):
        """RMSE of the clipped reconstruction on the given entries (sklearn_interface.py:172-182)"""
        if y.size > 0:
            if sp.issparse(X):
                X = X.toarray()
            return np.sqrt(np.mean((y - self.predict(X)) ** 2))
        if sp.issparse(X):
            C = X.tocoo()
            keep = C.data != 0
            i, j, v = C.row[keep], C.col[keep], C.data[keep]
        else:
            i, j = X.nonzero()
            v = X[i, j]
        return np.sqrt(np.mean((v - self._predict_entries(i, j)) ** 2))


def _all_nonnegative(X):
    """np.all(X >= 0) (sklearn_interface.py:251) without the boolean copy of X, and for a large dense array on several threads
    (numpy's reductions release the GIL): 0.2 s of a 1.8 s fit at 100000 x 10000 went into the plain form.  NaN fails, as there."""
    if not isinstance(X, np.ndarray) or X.ndim != 2 or X.size < (1 << 24):
        return bool(np.all(X >= 0))
    from concurrent.futures import ThreadPoolExecutor
    import os
    nt = max(1, min(16, os.cpu_count() or 1))
    cuts = np.linspace(0, X.shape[0], nt + 1).astype(np.int64)
    with ThreadPoolExecutor(nt) as pool:
        mins = list(pool.map(lambda i: X[cuts[i]:cuts[i + 1]].min() if cuts[i + 1] > cuts[i] else 0.0, range(nt)))
    return bool(np.all(np.asarray(mins) >= 0))          # a NaN minimum compares False


class NMF_TM_Estimator(_FactorPair, sklearn.base.BaseEstimator, sklearn.base.TransformerMixin):
    """Topic-model flavour: rows of T on the simplex after every update, rows of W projected at the
    end (sklearn_interface.py:185-345).

    n, d, k: documents, dictionary size, topics; wr1/wr2/tr1/tr2: l1/l2 penalties on W and T;
    handle_tfidf / handle_normalization: preprocess X inside fit/transform; W, T: optional warm
    start; nmf_kwargs: extra keyword arguments for nmf()."""

    def __init__(self, n, d, k, wr1=0, wr2=0, tr1=0, tr2=0, random_state=0,
                 handle_tfidf=False, handle_normalization=False, max_iter=300,
                 W=np.array([]), T=np.array([]), nmf_kwargs={},
                 do_final_project_W=True, keep_resident=False):
        # keep_resident (not in the reference): fit / one_iter on the SAME array X keep its device handle -- X uploaded and
        # preprocessed -- between the calls (nmf.ResidentProblem; the caller must not modify X in place meanwhile); release()
        # frees it
        self.keep_resident = keep_resident
        self._resident = None
        self.n, self.d, self.k = n, d, k
        self.wr1, self.wr2, self.tr1, self.tr2 = wr1, wr2, tr1, tr2
        self.random_state = random_state
        self.handle_tfidf = handle_tfidf
        self.handle_normalization = handle_normalization
        self.max_iter = max_iter
        self.W, self.T = W, T
        self.nmf_kwargs = nmf_kwargs
        self.do_final_project_W = do_final_project_W

    def _preprocess_kwargs(self, idf=None):
        """handle_tfidf / handle_normalization as nmf()'s `preprocess` option (sklearn_interface.py:243-247, 303-306):
        a dense X is then rewritten on the device after the upload instead of on the host before it"""
        if not (self.handle_tfidf or self.handle_normalization):
            return {}
        tf = False if not self.handle_tfidf else (True if idf is None else idf)
        return {'preprocess': {'tfidf': tf, 'normalize': bool(self.handle_normalization)}}

    def _solve(self, X, max_iter, max_time):
        W_in, T_in = self._warm_start()
        kw = dict(self._preprocess_kwargs(), **self.nmf_kwargs)
        if self.keep_resident and not sp.issparse(X):
            if self._resident is None:
                self._resident = _nmf_module.ResidentProblem()
            kw['resident'] = self._resident
        soln = _nmf(X, self.k, max_iter=max_iter, max_time=max_time, project_W_each_iter=False,
                    w_row_sum=1.0, project_T_each_iter=True, t_row_sum=1.0,
                    do_final_project_W=self.do_final_project_W, W_in=W_in, T_in=T_in,
                    reg_w_l1=self.wr1, reg_w_l2=self.wr2, reg_t_l1=self.tr1, reg_t_l2=self.tr2,
                    random_state=self.random_state, **kw)
        if self.handle_tfidf:
            self.idf = soln['idf']
        self._keep(soln)

    def release(self):
        """free the device handle kept by keep_resident"""
        if self._resident is not None:
            self._resident.close()

    def fit_transform(self, X, y=None):
        assert _all_nonnegative(X), 'X must be non-negative'
        self._solve(X, self.max_iter, 7200)
        return self.W

    def fit(self, X, y=None):
        self.fit_transform(X, y)
        return self

    def one_iter(self, X):
        """one more sweep from the estimator's current factors"""
        self._solve(X, 1, 240)
        return self

    def transform(self, Xnew):
        """express Xnew in the fitted topics (4 sweeps over W with T fixed)"""
        soln = _nmf(Xnew, self.k, max_iter=4, max_time=7200, project_W_each_iter=False, w_row_sum=1.0,
                    t_row_sum=1.0, T_in=self.T, do_final_project_W=self.do_final_project_W, fix_T=True,
                    reg_w_l1=self.wr1, reg_w_l2=self.wr2, reg_t_l1=self.tr1, reg_t_l2=self.tr2,
                    random_state=self.random_state,
                    **self._preprocess_kwargs(idf=self.idf if self.handle_tfidf else None))
        return soln['W']

    def constrained_transform(self, X):
        return self.transform(X)

    def score(self, X, y=None):
        """R^2 of the reconstruction of new documents (sklearn_interface.py:339-345); the residual sum of squares
        is taken on the device (rri_objective) instead of through an n x d product on the host"""
        sst = ((X - np.mean(X, axis=0)) ** 2).sum()
        W = self.transform(X)
        calc = _nmf_module.TrueObjComputer(np.asarray(X), W, self.T, 0, 0, 0, 0, None, None)
        sse = 2.0 * calc.true_objective()
        return 1 - sse / sst
