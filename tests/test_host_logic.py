"""CPU-only tests: the C-ABI surface, the host logic of nmf() that runs before any device work, the
host helper modules, and the estimators' plumbing (with the oracle standing in for the solver)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, relfro
from oracle import rri_oracle as orc
from rri_nmf_amd import _capi, matrixops, optimization
from rri_nmf_amd.synthetic import planted_X, scaled_init


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'rri_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rri_[A-Za-z0-9_]+)\s*\(', text)))


def gpu_present():
    lib = _capi.load_library()
    h = ctypes.c_void_p()
    st = lib.rri_create(ctypes.byref(h), 4, 4, 2, 0, 0, 0, None)
    if st == 0:
        lib.rri_destroy(h)
    return st == 0


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 35
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for nm in names:
        assert hasattr(lib, nm), 'librri_hip.so lacks %s declared in include/rri_hip.h' % nm
    assert sorted(_capi.PROTOTYPES) == names, 'ctypes binding and header disagree'
    typed = _capi.load_library()
    assert typed.rri_abi_version() == _capi.ABI_VERSION


def test_header_is_plain_c():
    """the boundary is a C ABI: the header must compile as C99 on its own (no C++ or HIP types leak into it)"""
    import subprocess
    r = subprocess.run(['gcc', '-x', 'c', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-fsyntax-only',
                        os.path.join(ROOT, 'include', 'rri_hip.h')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_abi_struct_layout_matches_header():
    assert ctypes.sizeof(_capi.Params) == 8 * 4 + 7 * 8
    assert ctypes.sizeof(_capi.Event) == 16
    assert _capi.Params.t_row_sum.offset == 32 and _capi.Params.eps_div.offset == 80


def test_null_and_bad_arguments_are_rejected_without_a_gpu():
    lib = _capi.load_library()
    assert lib.rri_create(None, 4, 4, 2, 0, 0, 0, None) == _capi.RRI_ERR_INVALID
    h = ctypes.c_void_p()
    assert lib.rri_create(ctypes.byref(h), 0, 4, 2, 0, 0, 0, None) == _capi.RRI_ERR_INVALID
    assert b'n,d,k' in lib.rri_last_error(None)
    assert lib.rri_create(ctypes.byref(h), 4, 4, 2, 7, 0, 0, None) == _capi.RRI_ERR_INVALID
    assert lib.rri_sweep(None, 1, None) == _capi.RRI_ERR_INVALID
    assert lib.rri_destroy(None) == _capi.RRI_OK


def test_no_cpu_fallback():
    """without a GPU the product path must fail loudly, never compute on the host"""
    if gpu_present():
        pytest.skip('a GPU is visible here')
    from rri_nmf_amd.nmf import nmf
    from rri_nmf_amd.engine import RRIEngine
    X = planted_X(30, 20, 3, dtype=np.float64)
    W0, T0 = scaled_init(X, 3)
    with pytest.raises(_capi.RRIHipUnavailable):
        RRIEngine(30, 20, 3)
    with pytest.raises(_capi.RRIHipUnavailable):
        nmf(X, 3, W_in=W0, T_in=T0, max_iter=2)
    with pytest.raises(_capi.RRIHipUnavailable):
        _capi.load_library('/nonexistent/librri_hip.so')


def test_nmf_host_checks_before_device_work():
    """argument handling that the reference performs before its loop (nmf.py:280-315, 853-860)"""
    from rri_nmf_amd.nmf import nmf
    g = load_golden('g6_rare_branches')
    X = planted_X(30, 20, 3, dtype=np.float64)
    W0, T0 = scaled_init(X, 3)
    s1 = nmf(X, 3, W_in=W0, T_in=T0, reg_t_l2=-1.0)      # unbounded in T -> sentinel, no exception
    assert s1['obj_history'] == [-np.inf] and s1['iter_cputime'] == [0]
    assert np.array_equal(s1['W'][:2, :2], g['sent_T_W']) and np.array_equal(s1['T'][:2, :2], g['sent_T_T'])
    s2 = nmf(X, 3, W_in=W0, T_in=T0, reg_w_l1=-1.0)
    assert np.array_equal(s2['W'][:2, :2], g['sent_W_W']) and np.array_equal(s2['T'][:2, :2], g['sent_W_T'])
    with pytest.raises(ValueError, match='W_in has wrong dimensions'):
        nmf(X, 3, W_in=W0[:, :2], T_in=T0)
    with pytest.raises(ValueError, match='T_in has wrong dimensions'):
        nmf(X, 3, W_in=W0, T_in=T0[:, :5])
    # (store_gradients and the Gaussian mechanism with W fixed or k = 1 used to be refused here; since round 4 they step on the
    # device like every other configuration -- tests/test_nmf_gpu.py::test_store_gradients_matches_the_oracle)
    Wkeep, Tkeep = W0.copy(), T0.copy()
    assert np.array_equal(W0, Wkeep) and np.array_equal(T0, Tkeep)   # caller arrays untouched


def test_signature_is_the_references():
    import inspect
    from rri_nmf_amd.nmf import nmf
    want = ['X', 'k', 'w_row', 'W_mat', 'fix_W', 'fix_T', 'random_state', 'init', 'T_in', 'W_in', 'max_iter',
            'max_time', 'eps_stop', 'compute_obj_each_iter', 'project_W_each_iter', 'w_row_sum',
            'do_final_project_W', 'project_T_each_iter', 't_row_sum', 'early_stop', 'reset_topic_method',
            'fix_reset_seed', 'n_resets', 'reg_w_l2', 'reg_t_l2', 'reg_w_l1', 'reg_t_l1', 'diagnostics',
            'store_gradients', 'ind_rows_to_store', 'eps_gauss_t', 'delta_gauss_t']
    sig = inspect.signature(nmf)
    pos = [p.name for p in sig.parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert pos == want                                        # nmf.py:98-108
    d = {p.name: p.default for p in sig.parameters.values()}
    assert (d['max_iter'], d['max_time'], d['eps_stop'], d['n_resets'], d['init']) == (200, 600, 1e-4, 23, 'nndsvd')
    assert d['reset_topic_method'] == 'max_resid_document' and d['do_final_project_W'] is True


def test_matrixops_match_oracle():
    g = load_golden('g7_functions')
    for nm in ('rand', 'pos', 'zeros', 'onsimplex', 'ties', 'single', 'neg', 'big'):
        v = g['proj_in_' + nm]
        for s in (1.0, 2.5):
            assert np.array_equal(matrixops.euclidean_proj_simplex(v.copy(), s), g['proj_out_%s_s%g' % (nm, s)])
    rs = np.random.RandomState(0)
    A = rs.rand(7, 5)
    A[3] = 0
    assert np.array_equal(matrixops.normalize(A.copy()), orc.normalize(A.copy()))
    assert np.array_equal(matrixops.normalize(A.copy(), 0), orc.normalize(A.copy(), 0))
    assert np.array_equal(matrixops.tfidf(A.copy()), orc.tfidf(A.copy()))
    B = rs.randn(6, 4)
    assert np.array_equal(matrixops.proj_mat_to_simplex(B.copy(), 1.0), orc.proj_rows_simplex(B.copy(), 1.0))
    sv = np.array([1.0, 2.0, 0.5, 1.0, 3.0, 1.0])
    assert np.array_equal(matrixops.proj_mat_to_simplex(B.copy(), sv), orc.proj_rows_simplex(B.copy(), sv))
    assert np.array_equal(matrixops.proj_mat_to_simplex(B.copy(), 1.0, axis=0), orc.proj_rows_simplex(B.copy(), 1.0, axis=0))
    H = matrixops.harden_distributions(A)
    assert H.sum() == 7 and np.array_equal(np.argmax(H, 1), np.argmax(A, 1))
    assert matrixops.labels_to_mat(np.array([0, 2, 1, 2])).tolist() == [[1, 0, 0], [0, 0, 1], [0, 1, 0], [0, 0, 1]]
    assert matrixops.stack_matrices([A, A]).shape == (14, 5) and matrixops.stack_matrices([A, A], dim='fat').shape == (7, 10)
    assert np.allclose((matrixops.normalize_l2(A + 1) ** 2).sum(1), 1, atol=1e-9)


def test_stop_rules_and_init():
    g = load_golden('g7_functions')
    hist = [10.0, 8.0, 7.5, 7.4999]
    usc = optimization.universal_stopping_condition
    assert np.array_equal(np.array([usc(hist[:1]), usc(hist[:2]), usc(hist[:3]), usc(hist), usc(hist, -1)]), g['usc'])
    assert optimization.first_last_stopping_condition([10.0, 1e-4], 1e-4) and not optimization.first_last_stopping_condition([10.0], 1)
    from rri_nmf_amd.initialization import initialize_nmf
    g8 = load_golden('g8_init')                          # the reference's test_init (tests/test_nmf.py:13-19)
    W, T = initialize_nmf(g8['X'], 2, init='nndsvd', random_state=0)
    assert np.allclose(g8['W_expected'], W) and np.allclose(g8['T_expected'], T)
    for init in ('random', 'smart_random', 'nndsvda', 'nndsvdar'):
        X = planted_X(40, 30, 4, dtype=np.float64)
        a = initialize_nmf(X, 4, init=init, random_state=3)
        b = orc.initialize_nmf(X, 4, init=init, random_state=3)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), init
    with pytest.raises(ValueError):
        initialize_nmf(planted_X(10, 8, 2, dtype=np.float64), 2, init='coherence_pmi')


class _HostProducts(object):
    """stands in for RRIEngine.X_times / Xt_times (the only two things randomized_svd_device asks of an engine)"""
    def __init__(self, X):
        self.X, (self.n, self.d) = X, X.shape
        self.calls = 0

    def X_times(self, B):
        self.calls += 1
        return self.X @ B

    def Xt_times(self, Q):
        self.calls += 1
        return self.X.T @ Q


@pytest.mark.parametrize('shape,k', [((300, 120), 5), ((90, 400), 6), ((200, 40), 10)])
def test_device_randomized_svd_follows_sklearn(shape, k):
    """same test matrix, same normaliser, same sign convention as sklearn.utils.extmath.randomized_svd (what the
    reference's initialization.py:105 calls): tall, wide (transposed internally) and k >= 0.1 min(n, d) (4 power
    iterations instead of 7) cases"""
    from sklearn.utils.extmath import randomized_svd
    from rri_nmf_amd.initialization import randomized_svd_device, initialize_nmf
    X = planted_X(shape[0], shape[1], k, dtype=np.float64)
    fake = _HostProducts(X)
    U, S, V = randomized_svd_device(fake, k, random_state=4)
    U0, S0, V0 = randomized_svd(X, k, random_state=4)
    assert U.shape == U0.shape and V.shape == V0.shape
    assert np.allclose(S, S0, rtol=1e-12, atol=0)
    assert np.abs(U - U0).max() < 1e-9 and np.abs(V - V0).max() < 1e-9
    assert fake.calls == 2 * (7 if k < 0.1 * min(shape) else 4) + 2
    for init in ('nndsvd', 'nndsvda', 'nndsvdar'):
        a = initialize_nmf(X, k, init=init, random_state=4, engine=fake)
        b = initialize_nmf(X, k, init=init, random_state=4)
        assert np.abs(a[0] - b[0]).max() < 1e-8 and np.abs(a[1] - b[1]).max() < 1e-8, init


def test_observed_pattern_view_of_sparse_inputs():
    """nmf()'s choice of the pattern-only handle: X must live on the 0/1 pattern of W_mat; explicit zero ratings
    stay observed; inputs are never modified"""
    import scipy.sparse as sp
    from rri_nmf_amd.nmf import _observed_csr, _sparse_mask_or_dense
    rs = np.random.RandomState(0)
    M = (rs.rand(30, 20) < 0.3).astype(float)
    X = rs.rand(30, 20) * M
    Ms, Xs = sp.csr_matrix(M), sp.csr_matrix(X)
    A = _observed_csr(Xs, Ms)                                  # same structure: taken as it is
    assert A is not None and np.array_equal(A.toarray(), X) and A.nnz == int(M.sum())
    X2 = X.copy()
    obs = np.argwhere(M > 0)
    X2[tuple(obs[3])] = 0.0                                    # an observed zero: X has one stored entry fewer
    A2 = _observed_csr(sp.csr_matrix(X2), Ms)
    assert A2 is not None and A2.nnz == Ms.nnz and np.array_equal(A2.toarray(), X2)
    assert np.array_equal(A2.indices, Ms.indices) and np.array_equal(A2.indptr, Ms.indptr)
    X3 = X.copy()
    X3[tuple(np.argwhere(M == 0)[5])] = 1.5                    # a value outside the pattern: no pattern-only view
    assert _observed_csr(sp.csr_matrix(X3), Ms) is None
    assert _observed_csr(X, Ms) is None and _observed_csr(Xs, M) is None        # dense inputs
    coo = sp.coo_matrix(X)                                     # other sparse formats, unsorted
    assert np.array_equal(_observed_csr(coo, sp.csc_matrix(M)).toarray(), X)
    W5 = sp.csr_matrix(M * 5.0)
    keep = W5.data.copy()
    assert isinstance(_sparse_mask_or_dense(W5), np.ndarray)   # weights other than 0/1: dense path
    Wz = sp.csr_matrix((np.array([1.0, 0.0, 1.0]), np.array([0, 1, 2]), np.array([0, 3] + [3] * 29)), shape=(30, 20))
    out = _sparse_mask_or_dense(Wz)
    assert sp.issparse(out) and out.nnz == 2 and Wz.nnz == 3 and np.array_equal(W5.data, keep)   # inputs untouched


def test_initialize_and_validate_matches_oracle():
    from rri_nmf_amd.nmf import _initialize_and_validate
    X = planted_X(50, 40, 4, dtype=np.float64)
    for kw in (dict(project_T_each_iter=True, project_W_each_iter=True, w_row_sum=1.0, t_row_sum=1.0),
               dict(project_T_each_iter=False, project_W_each_iter=False, w_row_sum=None, t_row_sum=1.0),
               dict(project_T_each_iter=True, project_W_each_iter=False, w_row_sum=1.0, t_row_sum=1.0)):
        a = _initialize_and_validate(W_in=[], T_in=[], W_mat=None, X=X, k=4, init='nndsvd', random_state=0,
                                     fix_W=False, fix_T=False, n=50, d=40, **kw)
        b = orc.initialize_and_validate(X, 4, [], [], None, 'nndsvd', 0, kw['project_T_each_iter'],
                                        kw['project_W_each_iter'], kw['w_row_sum'], kw['t_row_sum'], False, False)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_estimator_plumbing_with_oracle_solver(monkeypatch):
    """the estimators' flag sets / warm starts / one_iter resumability (tests/test_nmf.py:90-110),
    with the CPU oracle substituted for the device solver"""
    import scipy.sparse as sp
    from rri_nmf_amd import sklearn_interface as si
    from rri_nmf_amd.matrixops import proj_mat_to_simplex

    def oracle_nmf(X, *a, **kw):
        kw.setdefault('objective_always', True)   # the reference as shipped
        if sp.issparse(X):                         # the RS estimator hands nmf() CSR ratings and a CSR pattern; the
            X = X.toarray()                        # reference (and so the oracle) takes what .toarray() gives
        if sp.issparse(kw.get('W_mat')):
            kw['W_mat'] = kw['W_mat'].toarray()
        return orc.nmf(X, *a, **kw)
    monkeypatch.setattr(si._nmf_module, 'nmf', oracle_nmf)

    class OracleObjective(object):                 # stands in for the device objective behind NMF_TM_Estimator.score
        def __init__(self, X, W, T, *a, **kw):
            self.v = orc.true_objective(X, W, T)

        def true_objective(self):
            return self.v
    monkeypatch.setattr(si._nmf_module, 'TrueObjComputer', OracleObjective)
    g = load_golden('g1_tm_estimator')
    X = g['X']
    n, d = X.shape
    M = si.NMF_TM_Estimator(n, d, 5, random_state=0, max_iter=10).fit(X)
    assert relfro(M.W, g['W_shipped']) < 1e-8 and relfro(M.T, g['T_shipped']) < 1e-8
    assert np.linalg.norm(X - M.W @ M.T) < np.linalg.norm(X)
    M2 = si.NMF_TM_Estimator(n, d, 5, random_state=0, max_iter=2, do_final_project_W=False).fit(X)
    M2.max_iter = 10
    for _ in range(8):
        M2 = M2.one_iter(X)
    M2.W = proj_mat_to_simplex(M2.W)
    assert np.allclose(M2.T, M.T) and np.allclose(M2.W, M.W)
    Wte = M.transform(g['Xte'])
    assert Wte.shape == (g['Xte'].shape[0], 5) and abs(M.score(g['Xte']) - float(g['score_te'])) < 1.0
    # recommender estimator against the reference's vectors
    g4 = load_golden('g4_wrri')
    R = g4['X']
    E = si.NMF_RS_Estimator(R.shape[0], R.shape[1], 5, random_state=0, max_iter=20).fit_from_Xtr(R)
    assert abs(E.score(R) - float(g4['rs_es_score'])) < 1e-6 and E.score(R) < 1.0   # tests/test_nmf.py:88
    assert len(E.nmf_outputs['obj_history']) == len(g4['rs_es_obj'])
    E2 = si.NMF_RS_Estimator(R.shape[0], R.shape[1], 5, random_state=0, max_iter=20,
                             use_validation_early_stopping=False).fit_from_Xtr(sp.csr_matrix(R))
    assert abs(E2.score(R) - float(g4['rs_noes_score'])) < 1e-6
    i, j = R.nonzero()
    pred = E2.predict(np.column_stack((i, j)).astype(float))
    assert pred.shape == i.shape and abs(E2.score(np.column_stack((i, j)), R[i, j]) - E2.score(R)) < 1e-12
    E2.sparsify(); assert sp.issparse(E2.W); E2.densify(); assert isinstance(E2.W, np.ndarray)


def test_preprocess_option_parsing_and_host_route():
    """nmf()'s `preprocess` option: spellings, and the host route against matrixops (reference matrixops.py:124-179)"""
    import scipy.sparse as sp
    from rri_nmf_amd import nmf as nmf_mod
    from rri_nmf_amd.matrixops import tfidf, normalize
    spec = nmf_mod._preprocess_spec
    assert spec(None) is None and spec({}) is None and spec({'tfidf': False, 'normalize': False}) is None
    assert spec('tfidf') == (True, False) and spec(('normalize', 'tfidf')) == (True, True)
    assert spec({'normalize': 1}) == (False, True) and spec({'tfidf': np.True_}) == (True, False)
    idf_in = np.arange(1.0, 7.0)
    got = spec({'tfidf': idf_in})
    assert got[0] is idf_in and got[1] is False
    with pytest.raises(ValueError):
        spec({'tfidf': True, 'scale': True})
    rng = np.random.RandomState(0)
    X = rng.poisson(0.8, size=(30, 6)).astype(float)
    X[4] = 0
    Xt, idf = tfidf(X, return_idf=True)
    out, idf_out = nmf_mod._preprocess_on_host(X, True, True)
    assert np.array_equal(out, normalize(Xt)) and np.array_equal(idf_out, idf)
    outs, idf_s = nmf_mod._preprocess_on_host(sp.csr_matrix(X), True, True)
    assert sp.issparse(outs) and np.allclose(idf_s, idf)
    assert np.allclose(outs.toarray(), out)
    out2, idf2 = nmf_mod._preprocess_on_host(X, idf_in, False)
    assert np.array_equal(out2, X * idf_in) and np.array_equal(idf2, idf_in)
    out3, idf3 = nmf_mod._preprocess_on_host(X, False, True)
    assert idf3 is None and np.array_equal(out3, normalize(X))


def test_coherence_beam_search_initialiser_against_the_reference():
    """init_coherence_beam_search (initialization.py:166-208) against vectors captured from the reference (G10)"""
    from rri_nmf_amd.initialization import init_coherence_beam_search
    g = load_golden('g10_coherence_init')
    Wa, Ta = init_coherence_beam_search(g['Xa'].copy(), 3, n_words_beam=6)
    assert np.array_equal(Ta > 0, g['Ta'] > 0)                     # the same words in the same topics
    assert np.allclose(Ta, g['Ta'], rtol=1e-13, atol=0) and np.allclose(Wa, g['Wa'], rtol=1e-12, atol=1e-300)
    Wb, Tb = init_coherence_beam_search(g['Xb'].copy(), 4, n_words_beam=5)
    assert np.array_equal(Tb > 0, g['Tb'] > 0)
    assert np.allclose(Tb, g['Tb'], rtol=1e-13, atol=0) and np.allclose(Wb, g['Wb'], rtol=1e-12, atol=1e-300)
    assert np.allclose(Ta.sum(1), 1) and np.allclose(Wa.sum(1), 1) and (Ta > 0).sum(1).tolist() == [6, 6, 6]
    with pytest.raises(ValueError):                                # more words asked for than there are
        init_coherence_beam_search(g['Xa'][:, :5].copy(), 3, n_words_beam=6)


def test_dropin_alias_resolves_the_imports_of_the_reference_test_file():
    """the import lines of the reference's tests (tests/test_nmf.py:3-6, tests/conftest.py:5) resolve against this
    package once rri_nmf_amd.dropin is imported"""
    import subprocess
    import sys
    code = ('import rri_nmf_amd.dropin\n'
            'from rri_nmf.initialization import initialize_nmf\n'
            'from rri_nmf.matrixops import proj_mat_to_simplex, normalize, tfidf\n'
            'from rri_nmf.nmf import nmf, eps_div_by_zero, _compute_update_T\n'
            'from rri_nmf.sklearn_interface import NMF_RS_Estimator, NMF_TM_Estimator\n'
            'import rri_nmf, rri_nmf_amd, rri_nmf_amd.nmf as mine\n'
            'assert rri_nmf is rri_nmf_amd and nmf is mine.nmf and eps_div_by_zero == mine.eps_div_by_zero\n'
            'print("ok")\n')
    from conftest import ROOT
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert res.returncode == 0 and res.stdout.strip() == 'ok', res.stderr[-1500:]
    # it does not shadow another rri_nmf that is already there
    code2 = ('import sys, types\nsys.modules["rri_nmf"] = types.ModuleType("rri_nmf")\n'
             'try:\n    import rri_nmf_amd.dropin\nexcept ImportError as e:\n    print("refused")\n')
    res = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert res.returncode == 0 and res.stdout.strip() == 'refused', res.stderr[-1500:]


def test_row_sharded_call_is_checked_before_any_device_work():
    """nmf(..., group=): what a sharded call cannot take is refused on the host, before a handle exists"""
    from rri_nmf_amd import nmf as nmf_mod
    from rri_nmf_amd.distributed import RowGroup, shard_rows
    X = np.random.RandomState(0).rand(12, 7)
    W0, T0 = np.random.RandomState(1).rand(12, 3), np.random.RandomState(2).rand(3, 7)
    grp = RowGroup(None, 1, 3, [12, 12, 10])
    assert (grp.row_lo, grp.n_local, grp.n_global) == (12, 12, 34)
    view = grp.resized([5, 6, 7])
    assert (view.row_lo, view.n_local, view.n_global) == (5, 6, 18)
    view.close(), grp.close()                                     # a view never destroys the communicator; None is nothing to destroy
    assert [shard_rows(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    import scipy.sparse as sp
    with pytest.raises(ValueError, match='W_in'):          # scipy sparse weighted inputs have no row-sharded start (dense ones: round 4)
        nmf_mod.nmf(sp.csr_matrix(X), 3, W_mat=sp.csr_matrix(np.ones_like(X)), group=grp)
    # round 4: w_row and the device-side preprocessing run sharded (tests/pg_cases.py); these still do not
    for kw in (dict(store_gradients=True), dict(eps_gauss_t=1.0, delta_gauss_t=0.1), dict(early_stop=lambda X, W, T: 0.0),
               dict(preprocess='normalize', w_row=np.ones((12, 1))), dict(preprocess='normalize', W_mat=np.ones_like(X))):
        with pytest.raises(NotImplementedError):
            nmf_mod.nmf(X, 3, W_in=W0, T_in=T0, group=grp, **kw)
    with pytest.raises(NotImplementedError):               # the weighted flavour always keeps its masked residual (fixed halves on a
        nmf_mod.nmf(X, 3, W_in=W0, T_in=T0, W_mat=np.ones_like(X), schedule='residual')     # residual handle run since round 3)


def test_the_nonnegativity_check_of_the_estimator_without_a_boolean_copy():
    """NMF_TM_Estimator.fit asserts np.all(X >= 0) (sklearn_interface.py:251); the large-array form (threads, no copy) must decide alike"""
    from rri_nmf_amd.sklearn_interface import _all_nonnegative
    rs = np.random.RandomState(0)
    X = rs.rand(5000, 4000).astype(np.float32)             # above the threshold of the threaded form
    assert _all_nonnegative(X) and _all_nonnegative(X[:10]) and _all_nonnegative(np.zeros((3, 3)))
    for bad in (-1e-30, np.nan, -np.inf):
        Y = X.copy()
        Y[4999, 3999] = bad
        assert not _all_nonnegative(Y) and not _all_nonnegative(Y[4990:])
