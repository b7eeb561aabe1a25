"""GPU tests of the drop-in surface: nmf() and the estimators on the reference's own fixtures, following
the reference's test file (tests/test_nmf.py) case by case, plus comparisons with the vectors captured
from the reference and with the oracle."""
import logging

import numpy as np
import pytest

from conftest import load_golden, relfro
from rri_nmf_amd.synthetic import planted_X, scaled_init

pytestmark = pytest.mark.gpu
TOL = 2e-9   # see tests/test_hip_parity.py for how this bound is derived


def api():
    from rri_nmf_amd import nmf as nmf_mod
    from rri_nmf_amd import sklearn_interface as si
    return nmf_mod, si


def oracle():
    from oracle import rri_oracle
    return rri_oracle


@pytest.mark.parametrize('ci', [0, 1, 2, 3])
def test_convergence_tm_setting(ci):
    """tests/test_nmf.py:22-42: objective non-increasing, W and T rows on the simplex"""
    nmf_mod, _ = api()
    g = load_golden('g3_tm_settings')
    cases = [{'k': 25}, {'k': 15, 'reg_t_l2': 0.1}, {'k': 15, 'reg_t_l2': -0.1}, {'k': 15, 'reg_w_l2': 0.1}]
    p = dict(cases[ci], max_iter=15, w_row_sum=1.0, random_state=0, eps_stop=1e-4, project_T_each_iter=True,
             project_W_each_iter=True, compute_obj_each_iter=True, t_row_sum=1.0, early_stop=False)
    X = g['X']
    soln = nmf_mod.nmf(X, **p)                       # own NNDSVD start, as the reference's test does
    oh = np.array(soln['obj_history'])
    assert np.all(np.diff(oh) <= 1e-12 * abs(oh[0]))
    W, T = soln['W'], soln['T']
    assert W.min() >= -1e-13 and T.min() >= -1e-13
    assert np.sum(np.abs(W.sum(1) - 1)) + np.sum(np.abs(T.sum(1) - 1)) <= 1e-11
    # and the reference's numbers, from the reference's starting point
    soln = nmf_mod.nmf(X, W_in=g['c%d_W0' % ci], T_in=g['c%d_T0' % ci], **p)
    assert relfro(soln['W'], g['c%d_W' % ci]) < 1e-7 and relfro(soln['T'], g['c%d_T' % ci]) < 1e-7
    want = g['c%d_obj' % ci]
    assert len(soln['obj_history']) == len(want)
    assert np.allclose(soln['obj_history'], want, rtol=1e-9)


def test_convergence_TM_Estimator():
    """tests/test_nmf.py:90-110"""
    _, si = api()
    from rri_nmf_amd.matrixops import proj_mat_to_simplex
    g = load_golden('g1_tm_estimator')
    X = g['X']
    n, d = X.shape
    M = si.NMF_TM_Estimator(n, d, 5, random_state=0, max_iter=10).fit(X)
    assert np.linalg.norm(X - np.dot(M.W, M.T), 'fro') < np.linalg.norm(X, 'fro')
    # as shipped the reference tracks the objective and applies its stop rule: same sweeps, same result
    assert len(M.nmf_outputs['obj_history']) == len(g['obj_shipped'])
    assert np.allclose(M.nmf_outputs['obj_history'], g['obj_shipped'], rtol=1e-9)
    assert relfro(M.W, g['W_shipped']) < 1e-7 and relfro(M.T, g['T_shipped']) < 1e-7
    assert np.array_equal(np.argmax(M.W, 1), np.argmax(g['W_shipped'], 1))
    M2 = si.NMF_TM_Estimator(n, d, 5, random_state=0, max_iter=2, do_final_project_W=False).fit(X)
    M2.max_iter = 10
    for _ in range(7):
        M2 = M2.one_iter(X)
    M2 = M2.one_iter(X)
    M2.W = proj_mat_to_simplex(M2.W)
    assert np.allclose(M2.T, M.T) and np.allclose(M2.W, M.W)
    # fold-in and R^2 on the held-out documents
    M3 = si.NMF_TM_Estimator(n, d, 5, random_state=0, max_iter=10, nmf_kwargs={'eps_stop': -1}).fit(X)
    Wte = M3.transform(g['Xte'])
    assert relfro(Wte, g['Wte']) < 1e-7 and np.array_equal(np.argmax(Wte, 1), g['argmax_te'])
    assert abs(M3.score(g['Xte']) - float(g['score_te'])) < 1e-9
    assert Wte is not None and M3.constrained_transform(g['Xte']).shape == Wte.shape


@pytest.mark.parametrize('prep', [False, True])
def test_resident_handle_between_fit_and_one_iter(monkeypatch, prep):
    """keep_resident (round 4; sklearn_interface.py:316-318 re-runs nmf() on the same X for every one_iter): the handle of fit --
    X uploaded and, with handle_tfidf / handle_normalization, rewritten on the device -- serves the one_iter calls that follow on
    the same array: ONE upload, the same bits as the estimator that makes a handle per call; another array, or an array of
    another shape, gets a handle of its own; release() frees it; a call that raises leaves no handle behind."""
    nmf_mod, si = api()
    from rri_nmf_amd.engine import RRIEngine
    g = load_golden('g1_tm_estimator')
    X = np.ascontiguousarray(g['X'])
    n, d = X.shape
    uploads = []
    real_upload = RRIEngine.upload_X
    monkeypatch.setattr(RRIEngine, 'upload_X', lambda self, A: (uploads.append(A.shape), real_upload(self, A))[1])
    kw = dict(random_state=0, max_iter=3, do_final_project_W=False, handle_tfidf=prep, handle_normalization=prep,
              nmf_kwargs={'eps_stop': -1})
    plain = si.NMF_TM_Estimator(n, d, 5, **kw).fit(X)
    for _ in range(3):
        plain = plain.one_iter(X)
    n_plain = len(uploads)
    del uploads[:]
    kept = si.NMF_TM_Estimator(n, d, 5, keep_resident=True, **kw).fit(X)
    for _ in range(3):
        kept = kept.one_iter(X)
    assert n_plain == 4 and len(uploads) == 1 and kept._resident.reuses == 3
    assert np.array_equal(kept.W, plain.W) and np.array_equal(kept.T, plain.T)
    assert kept.nmf_outputs['obj_history'] == plain.nmf_outputs['obj_history']
    if prep:
        assert np.array_equal(kept.idf, plain.idf)
    # another array (same values): a handle of its own, and it becomes the resident one
    X2 = X.copy()
    kept.one_iter(X2)
    assert len(uploads) == 2 and kept._resident.reuses == 3
    kept.one_iter(X2)
    assert len(uploads) == 2 and kept._resident.reuses == 4
    # the holder used directly; a failing call drops the handle
    holder = nmf_mod.ResidentProblem()
    r1 = nmf_mod.nmf(X, 5, max_iter=2, random_state=0, eps_stop=-1, resident=holder)
    r2 = nmf_mod.nmf(X, 5, max_iter=2, random_state=0, eps_stop=-1, resident=holder)
    assert holder.reuses == 1 and np.array_equal(r1['W'], r2['W']) and np.array_equal(r1['T'], r2['T'])
    with pytest.raises(ValueError):
        nmf_mod.nmf(X, 5, max_iter=2, random_state=0, W_in=np.ones((3, 3)), T_in=np.ones((5, d)), resident=holder)
    r3 = nmf_mod.nmf(X, 5, max_iter=2, random_state=0, eps_stop=-1, resident=holder)
    assert np.array_equal(r3['W'], r1['W'])
    with pytest.raises(ValueError):
        nmf_mod.nmf(X, 5, w_row=np.ones((n, 1)), resident=holder)
    holder.close()
    kept.release()
    assert holder.engine is None and kept._resident.engine is None


def test_logger_level_switches_objective_tracking():
    nmf_mod, _ = api()
    X = planted_X(200, 120, 4, dtype=np.float64)
    W0, T0 = scaled_init(X, 4)
    old = nmf_mod.logger.level
    try:
        nmf_mod.logger.setLevel(logging.WARNING)
        r = nmf_mod.nmf(X, 4, W_in=W0, T_in=T0, max_iter=3)
        assert 'obj_history' not in r and len(r['iter_cputime']) == 3
        nmf_mod.logger.setLevel(logging.NOTSET)
        r = nmf_mod.nmf(X, 4, W_in=W0, T_in=T0, max_iter=3, eps_stop=-1)
        assert len(r['obj_history']) == 3 and r['obj_calculator'].obj == r['obj_history'][-1]
        assert abs(r['obj_calculator'].true_objective() - r['obj_history'][-1]) <= 1e-12 * r['obj_history'][-1]
    finally:
        nmf_mod.logger.setLevel(old)


def test_options_against_oracle():
    nmf_mod, _ = api()
    orc = oracle()
    n, d, k = 400, 150, 6
    X = planted_X(n, d, k, seed=3, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=4)
    Xn = orc.normalize(X.copy())

    def both(Xa, **kw):
        a = nmf_mod.nmf(Xa, k, W_in=W0, T_in=T0, **kw)
        b = orc.nmf(Xa, k, W_in=W0.copy(), T_in=T0.copy(), objective_always=True, **kw)
        return a, b

    # each-sweep W projection + stop rule (same number of sweeps)
    a, b = both(Xn, max_iter=40, eps_stop=1e-3, project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0,
                project_W_each_iter=True)
    assert len(a['obj_history']) == len(b['obj_history']) < 40
    assert relfro(a['W'], b['W']) < 1e-7 and relfro(a['T'], b['T']) < 1e-7
    # early stopping on a callable score with rollback, diagnostics recorded per sweep
    hold = (np.random.RandomState(0).rand(n, d) < 0.1)

    def val_score(Xi, W, T):
        return float(np.sqrt(np.mean(((W @ T) - X)[hold] ** 2))) * (1 if W.sum() < 1e9 else 1)

    def recon(Xi, W, T):
        return float(np.linalg.norm(Xi - W @ T))
    noisy = X * (~hold) + hold * X.mean()
    a, b = both(noisy, max_iter=25, eps_stop=-1, early_stop=val_score, diagnostics=[recon])
    assert len(a['iter_cputime']) == len(b['iter_cputime'])
    assert relfro(a['W'], b['W']) < 1e-7 and relfro(a['T'], b['T']) < 1e-7
    assert np.allclose(a['diagnostics']['recon'], b['diagnostics']['recon'], rtol=1e-8)
    # per-row weights with the recursive refit of W (nmf.py:335-344, 531-539)
    wr = np.random.RandomState(1).rand(n, 1) + 0.5
    a, b = both(Xn, max_iter=4, eps_stop=-1, w_row=wr, w_row_sum=1.0, project_T_each_iter=True,
                t_row_sum=1.0)
    assert relfro(a['W'], b['W']) < 1e-7 and relfro(a['T'], b['T']) < 1e-7
    # float32 storage of X, float64 arithmetic: compared with the oracle on the same fp32-valued X
    X32 = X.astype(np.float32)
    a = nmf_mod.nmf(X32, k, W_in=W0, T_in=T0, max_iter=6, eps_stop=-1)
    b = orc.nmf(X32.astype(np.float64), k, W_in=W0.copy(), T_in=T0.copy(), max_iter=6, eps_stop=-1)
    assert relfro(a['W'], b['W']) < TOL and relfro(a['T'], b['T']) < TOL and a['W'].dtype == np.float64


def test_errors_raised_like_the_reference():
    nmf_mod, _ = api()
    g = load_golden('g6_rare_branches')
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=3, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=4)
    Wd = g['dead_W0']
    with pytest.raises(ValueError, match='unbounded'):
        nmf_mod.nmf(X, k, W_in=Wd, T_in=T0, max_iter=2, eps_stop=-1)
    with pytest.raises(AssertionError, match='sums to 0'):
        nmf_mod.nmf(X, k, W_in=Wd, T_in=T0, max_iter=2, eps_stop=-1, t_row_sum=1.0, w_row_sum=1.0,
                    do_final_project_W=False, reset_topic_method=None)
    r = nmf_mod.nmf(X, k, W_in=Wd, T_in=T0, max_iter=2, eps_stop=-1, t_row_sum=1.0)
    assert r['n_resets_used'] >= 1 and relfro(r['T'], g['dead_mrd_T']) < 1e-7
    with pytest.raises(NotImplementedError):
        Xn = oracle().normalize(X.copy())
        nmf_mod.nmf(Xn, k, W_in=W0, T_in=T0, max_iter=2, project_T_each_iter=True, t_row_sum=2.0,
                    w_row_sum=1.0, reg_t_l2=-50.0)


@pytest.mark.parametrize('ci', [0, 1, 2, 3])
def test_convergence_rs_setting(ci):
    """tests/test_nmf.py:57-78: weighted NMF on the recsys fixture, objective non-increasing"""
    nmf_mod, _ = api()
    g = load_golden('g4_wrri')
    X = g['X']
    Wm = np.zeros(X.shape)
    Wm[X.nonzero()] = 1.0
    cases = [{}, {'reg_w_l1': 0.1, 'reg_t_l1': 0.1}, {'reg_w_l1': 0.1}, {'reg_t_l1': 0.1}]
    p = dict(cases[ci], max_iter=15, random_state=0, W_mat=Wm, compute_obj_each_iter=True, reset_topic_method=None,
             early_stop=False, k=7, project_T_each_iter=False, t_row_sum=1.0, project_W_each_iter=False,
             w_row_sum=None)
    soln = nmf_mod.nmf(X, **p)
    oh = np.array(soln['obj_history'])
    assert np.all(np.diff(oh) <= 1e-9 * abs(oh[0]))
    assert len(oh) == len(g['c%d_obj' % ci]) and np.allclose(oh, g['c%d_obj' % ci], rtol=1e-6)


def test_convergence_RS_Estimator():
    """tests/test_nmf.py:81-88 plus the reference's numbers for both early-stopping settings"""
    _, si = api()
    g = load_golden('g4_wrri')
    X = g['X']
    n, d = X.shape
    E = si.NMF_RS_Estimator(n, d, 5, random_state=0, max_iter=20).fit_from_Xtr(X)
    score = E.score(X)
    assert score < 1.0
    assert abs(score - float(g['rs_es_score'])) < 1e-5 and abs(E.score(g['Xte']) - float(g['rs_es_score_te'])) < 1e-5
    assert len(E.nmf_outputs['obj_history']) == len(g['rs_es_obj'])
    E2 = si.NMF_RS_Estimator(n, d, 5, random_state=0, max_iter=20, use_validation_early_stopping=False)
    E2 = E2.fit_from_Xtr(X)
    assert abs(E2.score(X) - float(g['rs_noes_score'])) < 1e-4
    Wnew = E2.transform(g['Xte'].astype(np.float64))
    assert Wnew.shape == (n, 5) and Wnew.min() >= 0


def test_sparse_inputs_are_ingested_as_csr():
    """scipy sparse X and a sparse 0/1 W_mat give what their dense forms give (same start passed in)"""
    import scipy.sparse as sp
    nmf_mod, _ = api()
    from rri_nmf_amd.engine import RRIEngine
    g = load_golden('g4_wrri')
    X = g['X'].astype(np.float64)
    Xs = sp.csr_matrix(X)
    Wm = np.zeros(X.shape)
    Wm[X.nonzero()] = 1.0
    p = dict(max_iter=5, eps_stop=-1, W_in=g['W0'], T_in=g['T0'], reset_topic_method=None, t_row_sum=1.0)
    a = nmf_mod.nmf(X, 7, W_mat=Wm, **p)
    b = nmf_mod.nmf(Xs, 7, W_mat=sp.csr_matrix(Wm), sparse_pattern=False, **p)     # densified on the device
    assert np.array_equal(a['W'], b['W']) and np.array_equal(a['T'], b['T'])
    assert np.allclose(a['obj_history'], b['obj_history'], rtol=1e-13)
    b = nmf_mod.nmf(Xs, 7, W_mat=sp.csr_matrix(Wm), **p)                           # residual kept on the pattern only
    assert relfro(a['W'], b['W']) < 1e-10 and relfro(a['T'], b['T']) < 1e-10
    assert np.allclose(a['obj_history'], b['obj_history'], rtol=1e-10)
    # unweighted, float32 storage, ragged shape; and own NNDSVD start from the sparse matrix
    rs = np.random.RandomState(0)
    D = (rs.rand(333, 129) < 0.07) * rs.rand(333, 129)
    W0, T0 = scaled_init(D + 0.01, 4, seed=3)
    a = nmf_mod.nmf(D.astype(np.float32), 4, W_in=W0, T_in=T0, max_iter=3, eps_stop=-1)
    b = nmf_mod.nmf(sp.csr_matrix(D.astype(np.float32)), 4, W_in=W0, T_in=T0, max_iter=3, eps_stop=-1)
    assert np.array_equal(a['W'], b['W']) and np.array_equal(a['T'], b['T'])
    c = nmf_mod.nmf(sp.csr_matrix(D), 4, max_iter=3, random_state=0)
    assert c['W'].shape == (333, 4) and np.isfinite(c['W']).all() and c['obj_history'][-1] <= c['obj_history'][0]
    # engine level: malformed CSR is rejected before anything is launched
    with RRIEngine(333, 129, 4) as e:
        bad = sp.csr_matrix(D)
        bad.indices = bad.indices.copy()
        bad.indices[0] = 500
        with pytest.raises(ValueError, match='column index'):
            e.upload_X_csr(bad)


@pytest.mark.parametrize('shape,store', [((700, 333), np.float64), ((257, 1030), np.float32)])
def test_device_products_and_device_init(shape, store):
    """rri_X_times / rri_Xt_times against numpy, and nmf(init='nndsvd') started from the device-assisted SVD
    against the same call started from scikit-learn's (same algorithm: the two must agree to rounding)"""
    from rri_nmf_amd.engine import RRIEngine
    from rri_nmf_amd.initialization import randomized_svd_device
    from sklearn.utils.extmath import randomized_svd
    nmf_mod, _ = api()
    n, d = shape
    k = 6
    X = planted_X(n, d, k, dtype=store)
    X64 = X.astype(np.float64)
    rs = np.random.RandomState(1)
    B, Q = rs.randn(d, 9), rs.randn(n, 5)
    eng = RRIEngine(n, d, k, dtype=store)
    eng.upload_X(X)
    assert relfro(eng.X_times(B), X64 @ B) < 1e-13
    assert relfro(eng.Xt_times(Q), X64.T @ Q) < 1e-13
    assert relfro(eng.X_times(B[:, :1]), X64 @ B[:, :1]) < 1e-13       # a single vector
    U0, S0, V0 = randomized_svd(X64, k, random_state=3)
    # the products on the device, the panels normalised on the host by LU / QR exactly as scikit-learn does ...
    U, S, V = randomized_svd_device(eng, k, random_state=3, resident=False)
    assert np.allclose(S, S0, rtol=1e-10) and np.abs(U - U0).max() < 1e-8 and np.abs(V - V0).max() < 1e-8
    # ... and (the default) the whole range finder resident on the device, its panels normalised by Cholesky-QR
    # (rri_range_finder): another basis of the same range, the same SVD to rounding
    U, S, V = randomized_svd_device(eng, k, random_state=3)
    print('resident range finder %r %s: S %.2e U %.2e V %.2e' % (shape, np.dtype(store).name, np.abs(S / S0 - 1).max(), np.abs(U - U0).max(), np.abs(V - V0).max()))
    assert np.allclose(S, S0, rtol=1e-10) and np.abs(U - U0).max() < 1e-8 and np.abs(V - V0).max() < 1e-8
    # the call itself: Q orthonormal, B = Q^T A, for A = X and A = X^T, with and without power iterations
    for transpose in (False, True):
        A = X64.T if transpose else X64
        for n_iter in (0, 3):
            Q0 = np.random.RandomState(5).randn(A.shape[1], 16)
            Qr, Br = eng.range_finder(Q0, n_iter, transpose=transpose)
            assert np.abs(Qr.T @ Qr - np.eye(16)).max() < 1e-12
            assert relfro(Br, Qr.T @ A) < 1e-12
            Y = A @ Q0
            for _ in range(n_iter):
                Y = A @ np.linalg.qr(A.T @ np.linalg.qr(Y)[0])[0]
            Qh = np.linalg.qr(Y)[0]
            # the same subspace: the projectors agree (on what the panel resolves: its smallest directions are noise at n_iter = 3)
            assert relfro(Qr @ (Qr.T @ Y), Y) < 1e-10 and relfro(Qh @ (Qh.T @ (Qr @ Br)), Qr @ Br) < 1e-6
    # the products do not disturb a factorisation in progress
    W0, T0 = scaled_init(X64, k, seed=2)
    eng.set_W(W0), eng.set_T(T0)
    eng.set_params()
    eng.sweep(1)
    eng.Xt_times(Q), eng.X_times(B)
    eng.sweep(1)
    Wa, Ta = eng.get_W(), eng.get_T()
    eng.set_W(W0), eng.set_T(T0)
    eng.sweep(2)
    assert relfro(Wa, eng.get_W()) < 1e-13 and relfro(Ta, eng.get_T()) < 1e-13
    del eng
    kw = dict(max_iter=5, random_state=0, project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0, eps_stop=-1)
    a = nmf_mod.nmf(X, k, device_init=True, **kw)
    b = nmf_mod.nmf(X, k, device_init=False, **kw)
    # scikit-learn factorises a float32 X in float32 arithmetic; the device products are float64 either way
    tol = 1e-6 if store == np.float64 else 1e-3
    assert relfro(a['W'], b['W']) < tol and relfro(a['T'], b['T']) < tol


def test_gaussian_mechanism_against_the_reference():
    """eps_gauss_t / delta_gauss_t (nmf.py:422-435): the T-row sums leave the device, get the reference's noise
    (same scipy call on numpy's global RNG, same order) and the step finishes on the device"""
    from test_oracle_golden import _g9_cases
    nmf_mod, _ = api()
    g, k, W0, T0, cases = _g9_cases()
    for name, X, M, kw in cases:
        np.random.seed(int(g['seed'][0]))
        r = nmf_mod.nmf(X, k, W_in=W0, T_in=T0, W_mat=M, **kw)
        assert relfro(r['W'], g[name + '_W']) < 1e-8 and relfro(r['T'], g[name + '_T']) < 1e-8, \
            (name, relfro(r['W'], g[name + '_W']), relfro(r['T'], g[name + '_T']))
    # W fixed (round 4: stepped like any other configuration; the oracle comparison is in test_store_gradients_matches_the_oracle).
    # Noise far above the signal drives a denominator to zero: the reference's error, at the same topic as the oracle
    from oracle import rri_oracle as orc
    errs = []
    for f in (lambda **kw: orc.nmf(np.array(cases[0][1], dtype=np.float64), k, W_in=np.array(W0, dtype=np.float64), T_in=np.array(T0, dtype=np.float64), **kw),
              lambda **kw: nmf_mod.nmf(cases[0][1], k, W_in=W0, T_in=T0, **kw)):
        np.random.seed(5)
        with pytest.raises(ValueError, match='unbounded') as ei:
            f(fix_W=True, max_iter=1, eps_gauss_t=1.0, delta_gauss_t=0.1)
        errs.append(str(ei.value))
    assert 'unbounded' in errs[0] and 'unbounded' in errs[1]


def test_store_gradients_matches_the_oracle():
    """store_gradients / ind_rows_to_store (nmf.py:411-413, 454-456, 677-686, 706-713): the sums behind every T-row
    update, read from the device between the two stages of the step.  The reference itself raises at nmf.py:543 (its
    reshape lambda is passed as `dict_key`); oracle and device restate the evident intent."""
    import scipy.sparse as sp
    nmf_mod, _ = api()
    orc = oracle()
    n, d, k = 90, 41, 4
    X = planted_X(n, d, k, seed=5, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=6)
    M = (np.random.RandomState(7).rand(n, d) < 0.4).astype(np.float64)
    rows = [3, 17, 18, 60]
    tm = dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
    cases = [('plain', None, {}, None), ('plain, rows', None, {}, rows), ('topic model', None, tm, None),
             ('weighted', M, dict(t_row_sum=1.0, reset_topic_method=None), None),
             ('weighted, rows', M, dict(t_row_sum=1.0, reset_topic_method=None), rows)]
    for name, Wm, kw, r in cases:
        Xc = X * Wm if Wm is not None else X
        a = orc.nmf(Xc.copy(), k, W_in=W0.copy(), T_in=T0.copy(), W_mat=Wm, max_iter=3, store_gradients=True,
                    ind_rows_to_store=r, **kw)
        b = nmf_mod.nmf(Xc, k, W_in=W0, T_in=T0, W_mat=Wm, max_iter=3, store_gradients=True, ind_rows_to_store=r, **kw)
        assert sorted(b['numer_W']) == sorted(a['numer_W']) == [0, 1, 2], name
        for it in a['numer_W']:
            assert b['numer_W'][it].shape == a['numer_W'][it].shape == (k, d), name
            assert b['denom_W'][it].shape == a['denom_W'][it].shape == ((k, d) if Wm is not None else (k, 1)), name
            assert relfro(b['numer_W'][it], a['numer_W'][it]) < 1e-9, (name, it)
            assert relfro(b['denom_W'][it], a['denom_W'][it]) < 1e-9, (name, it)
        assert relfro(b['W'], a['W']) < 1e-8 and relfro(b['T'], a['T']) < 1e-8, name
    # the pattern-only weighted handle records the same sums as the dense one
    kw = dict(t_row_sum=1.0, reset_topic_method=None, max_iter=2, store_gradients=True, W_in=W0, T_in=T0)
    dense = nmf_mod.nmf(X * M, k, W_mat=M, **kw)
    sparse = nmf_mod.nmf(sp.csr_matrix(X * M), k, W_mat=sp.csr_matrix(M), sparse_pattern=True, **kw)
    for it in dense['numer_W']:
        assert relfro(sparse['numer_W'][it], dense['numer_W'][it]) < 1e-9
        assert relfro(sparse['denom_W'][it], dense['denom_W'][it]) < 1e-12
    # stored before the Gaussian mechanism perturbs them (nmf.py:419-435 come after _compute_update_T)
    np.random.seed(3)
    noisy = nmf_mod.nmf(X, k, W_in=W0, T_in=T0, max_iter=1, store_gradients=True, eps_gauss_t=1e6, delta_gauss_t=0.5)
    quiet = nmf_mod.nmf(X, k, W_in=W0, T_in=T0, max_iter=1, store_gradients=True)
    assert np.allclose(noisy['numer_W'][0][0], quiet['numer_W'][0][0], rtol=1e-12)
    # ind_rows_to_store alone stores nothing (nmf.py:325: only store_gradients switches it on); T fixed: nothing to store
    assert 'numer_W' not in nmf_mod.nmf(X, k, W_in=W0, T_in=T0, max_iter=1, ind_rows_to_store=rows)
    fixed = nmf_mod.nmf(X, k, W_in=W0, T_in=T0, max_iter=1, fix_T=True, store_gradients=True)
    assert fixed['numer_W'][0].size == 0
    # round 4: W fixed (only the T rows step, the kept columns take their scale, nmf.py:450-452) and k = 1 -- the reference's loop
    # has no limit there (nmf.py:417-456); plain and weighted, with the sums stored and with the Gaussian mechanism on them
    for name, Wm, kk, kw in (('W fixed', None, k, dict(fix_W=True)), ('W fixed, topic model', None, k, dict(fix_W=True, **tm)),
                             ('W fixed, weighted', M, k, dict(fix_W=True, t_row_sum=1.0, reset_topic_method=None)),
                             ('k = 1', None, 1, {}), ('k = 1, weighted', M, 1, dict(t_row_sum=1.0, reset_topic_method=None))):
        Xc = X * Wm if Wm is not None else X
        Wk, Tk = W0[:, :kk].copy(), T0[:kk].copy()
        a = orc.nmf(Xc.copy(), kk, W_in=Wk.copy(), T_in=Tk.copy(), W_mat=Wm, max_iter=3, store_gradients=True, **kw)
        b = nmf_mod.nmf(Xc, kk, W_in=Wk, T_in=Tk, W_mat=Wm, max_iter=3, store_gradients=True, **kw)
        for it in a['numer_W']:
            assert relfro(b['numer_W'][it], a['numer_W'][it]) < 1e-9 and relfro(b['denom_W'][it], a['denom_W'][it]) < 1e-9, (name, it)
        assert relfro(b['W'], a['W']) < 1e-8 and relfro(b['T'], a['T']) < 1e-8, name
        np.random.seed(11)
        a = orc.nmf(Xc.copy(), kk, W_in=Wk.copy(), T_in=Tk.copy(), W_mat=Wm, max_iter=2, eps_gauss_t=1e7, delta_gauss_t=0.5, **kw)
        np.random.seed(11)
        b = nmf_mod.nmf(Xc, kk, W_in=Wk, T_in=Tk, W_mat=Wm, max_iter=2, eps_gauss_t=1e7, delta_gauss_t=0.5, **kw)
        assert relfro(b['W'], a['W']) < 1e-7 and relfro(b['T'], a['T']) < 1e-7, (name, 'Gaussian mechanism', relfro(b['W'], a['W']), relfro(b['T'], a['T']))
    # the single-step helper the reference's test file imports (nmf.py:633-715)
    for Wm in (None, M):
        Xc = X * Wm if Wm is not None else X
        for t in (0, k - 1):
            want = orc.residual_products_T(Xc, W0.copy(), T0, t, Wm)
            wR, nw, wRs, nws = nmf_mod._compute_update_T(Xc, W0, T0, t, True, rows, W_mat=Wm, iter_no=0, unused='x')
            assert np.allclose(wR, want[0], rtol=1e-11, atol=1e-12) and np.allclose(nw, want[1], rtol=1e-12)
            sub = orc.residual_products_T(Xc[rows], W0[rows].copy(), T0, t, None if Wm is None else Wm[rows])
            assert np.allclose(wRs, sub[0], rtol=1e-11, atol=1e-12) and np.allclose(nws, sub[1], rtol=1e-12)
    assert nmf_mod._compute_update_T(X, W0, T0, 1, False, None)[2:] == (None, None)
