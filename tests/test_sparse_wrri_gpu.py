"""The weighted flavour on a sparse observation pattern (RRI_WEIGHTED_SPARSE handles: residual kept on the pattern
as a CSR and a CSC copy) against the dense weighted engine and the oracle, on the same inputs."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden, relfro
from rri_nmf_amd.synthetic import planted_X, scaled_init

pytestmark = pytest.mark.gpu


def _problem(n, d, k, frac, seed=0, store=np.float64, empty=False):
    rs = np.random.RandomState(seed + 10)
    M = (rs.rand(n, d) < frac).astype(np.float64)
    if empty:                      # rows / columns without any observation
        M[3:7, :] = 0
        M[:, 5:9] = 0
    X = planted_X(n, d, k, seed=seed, dtype=np.float64) * M
    X[M > 0] += 0.05               # observed entries are non-zero here; explicit zeros are covered separately
    W0, T0 = scaled_init(X, k, seed=seed + 1)
    return X.astype(store), M.astype(store), W0, T0


def _run(X, M, W0, T0, k, sweeps, sparse, store, flags):
    from rri_nmf_amd.engine import RRIEngine
    n, d = X.shape
    with RRIEngine(n, d, k, dtype=store, weighted='sparse' if sparse else True) as e:
        if sparse:
            A = sp.csr_matrix(M)                 # pattern
            A.data = np.asarray(X[M > 0], dtype=store)     # row-major order = CSR order
            e.upload_observed_csr(A)
        else:
            e.upload_X(X)
            e.upload_mask(M)
        e.set_W(W0), e.set_T(T0)
        e.set_params(**flags)
        e.sweep(sweeps)
        return e.get_W(), e.get_T(), e.objective(), e.n_resets_used


FLAGS = {
    'rs_fit': dict(t_row_sum=1.0, reset_topic_method=None),
    'rs_fit_regs': dict(t_row_sum=1.0, reset_topic_method=None, reg_w_l1=0.01, reg_t_l1=0.02, reg_w_l2=0.1, reg_t_l2=0.05),
    'projected': dict(t_row_sum=1.0, project_T_each_iter=True, w_row_sum=2.0, reset_topic_method=None),
    'fix_T': dict(t_row_sum=1.0, fix_T=True, reset_topic_method='random'),
    'fix_W': dict(t_row_sum=1.0, fix_W=True, reset_topic_method=None),
    'resets_T': dict(t_row_sum=1.0, reg_t_l1=1e6),
    'resets_W': dict(t_row_sum=1.0, reg_w_l1=1e6),
}


@pytest.mark.parametrize('name', sorted(FLAGS))
@pytest.mark.parametrize('store', [np.float64, np.float32])
def test_sparse_pattern_matches_dense_weighted_engine(name, store):
    n, d, k = 413, 187, 5
    X, M, W0, T0 = _problem(n, d, k, 0.2, store=store, empty=True)
    flags = FLAGS[name]
    Wd, Td, od, nd = _run(X, M, W0, T0, k, 3, False, store, flags)
    Ws, Ts, os_, ns = _run(X, M, W0, T0, k, 3, True, store, flags)
    tol = 1e-10 if store == np.float64 else 2e-5    # fp32: the two schedules round the stored residual differently
    assert ns == nd
    assert relfro(Ws, Wd) < tol and relfro(Ts, Td) < tol, (relfro(Ws, Wd), relfro(Ts, Td))
    assert abs(os_ - od) <= max(tol, 1e-10) * abs(od)


@pytest.mark.parametrize('frac,n,d', [(0.02, 900, 300), (0.6, 120, 90), (0.08, 64, 2100)])
def test_sparse_pattern_matches_oracle(frac, n, d):
    """short and long segments (8 ... 64 lanes per row / column), against the numpy restatement of nmf.py:687-746"""
    from oracle import rri_oracle as orc
    k = 4
    X, M, W0, T0 = _problem(n, d, k, frac, seed=3)
    kw = dict(t_row_sum=1.0, reset_topic_method=None)
    Ws, Ts, _, _ = _run(X, M, W0, T0, k, 4, True, np.float64, kw)
    ref = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), W_mat=M, max_iter=4, eps_stop=-1, **kw)
    assert relfro(Ws, ref['W']) < 2e-9 and relfro(Ts, ref['T']) < 2e-9, (relfro(Ws, ref['W']), relfro(Ts, ref['T']))


def test_explicit_zero_ratings_and_empty_pattern():
    from rri_nmf_amd.engine import RRIEngine
    from oracle import rri_oracle as orc
    n, d, k = 60, 40, 3
    X, M, W0, T0 = _problem(n, d, k, 0.3, seed=5)
    obs = np.argwhere(M > 0)
    for i, j in obs[::7]:
        X[i, j] = 0.0                                   # observed, and the observed value is 0
    A = sp.csr_matrix((X[M > 0], (obs[:, 0], obs[:, 1])), shape=(n, d))     # keeps the explicit zeros
    assert A.nnz == int(M.sum())
    kw = dict(t_row_sum=1.0, reset_topic_method=None)
    with RRIEngine(n, d, k, dtype=np.float64, weighted='sparse') as e:
        e.upload_observed_csr(A)
        e.set_W(W0), e.set_T(T0)
        e.set_params(**kw)
        e.sweep(3)
        Ws, Ts = e.get_W(), e.get_T()
    ref = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), W_mat=M, max_iter=3, eps_stop=-1, **kw)
    assert relfro(Ws, ref['W']) < 2e-9 and relfro(Ts, ref['T']) < 2e-9
    # nothing observed at all: every column dies -> the reference's assertion (nmf.py:476)
    with RRIEngine(n, d, k, dtype=np.float64, weighted='sparse') as e:
        e.upload_observed_csr(sp.csr_matrix((n, d)))
        e.set_W(W0), e.set_T(T0)
        e.set_params(**kw)
        with pytest.raises(AssertionError):
            e.sweep(1)


def test_nmf_and_estimator_take_the_pattern_only_path():
    """nmf() with scipy sparse X and W_mat: pattern-only handle (default below 35 % observed) = densified handle =
    dense inputs, to rounding; the RS estimator (which now hands nmf() CSR ratings) against the reference's vectors"""
    from rri_nmf_amd import nmf as nmf_mod
    from rri_nmf_amd import sklearn_interface as si
    n, d, k = 300, 120, 4
    X, M, W0, T0 = _problem(n, d, k, 0.15, seed=7)
    kw = dict(W_in=W0, T_in=T0, max_iter=6, eps_stop=-1, t_row_sum=1.0, reset_topic_method=None,
              compute_obj_each_iter=True)
    a = nmf_mod.nmf(sp.csr_matrix(X), k, W_mat=sp.csr_matrix(M), **kw)                          # auto: pattern-only
    b = nmf_mod.nmf(sp.csr_matrix(X), k, W_mat=sp.csr_matrix(M), sparse_pattern=False, **kw)    # densified on device
    c = nmf_mod.nmf(X, k, W_mat=M, **kw)                                                        # dense inputs
    for r in (b, c):
        assert relfro(a['W'], r['W']) < 1e-10 and relfro(a['T'], r['T']) < 1e-10
        assert np.allclose(a['obj_history'], r['obj_history'], rtol=1e-10)
    assert abs(a['obj_calculator'].true_objective() - a['obj_history'][-1]) <= 1e-10 * a['obj_history'][-1]
    with pytest.raises(ValueError):
        Xo = X.copy()
        Xo[M == 0] = 0.3
        nmf_mod.nmf(sp.csr_matrix(Xo), k, W_mat=sp.csr_matrix(M), sparse_pattern=True, **kw)
    # initialisation from the sparse ratings (NNDSVD of W_mat .* X through scikit-learn's sparse path)
    e1 = nmf_mod.nmf(sp.csr_matrix(X), k, W_mat=sp.csr_matrix(M), max_iter=4, eps_stop=-1, t_row_sum=1.0,
                     reset_topic_method=None, random_state=0)
    e2 = nmf_mod.nmf(X, k, W_mat=M, max_iter=4, eps_stop=-1, t_row_sum=1.0, reset_topic_method=None, random_state=0)
    assert relfro(e1['W'], e2['W']) < 1e-6 and relfro(e1['T'], e2['T']) < 1e-6
    g4 = load_golden('g4_wrri')
    R = g4['X']
    E = si.NMF_RS_Estimator(R.shape[0], R.shape[1], 5, random_state=0, max_iter=20).fit_from_Xtr(R)
    assert abs(E.score(R) - float(g4['rs_es_score'])) < 1e-6
    E2 = si.NMF_RS_Estimator(R.shape[0], R.shape[1], 5, random_state=0, max_iter=20,
                             use_validation_early_stopping=False).fit_from_Xtr(sp.csr_matrix(R))
    assert abs(E2.score(R) - float(g4['rs_noes_score'])) < 1e-6


@pytest.mark.parametrize('n,d,store', [(12000, 150, np.float64), (300, 11000, np.float64), (23000, 120, np.float32),
                                       (200, 21000, np.float32)])
def test_more_than_one_block_per_copy(n, d, store):
    """shapes wider than an LDS block (10240 fp32 / 5120 fp64 factors): the row copy is cut into column blocks
    (several Ypart panels), the column copy into row blocks (several Zpart rows)"""
    k = 4
    X, M, W0, T0 = _problem(n, d, k, 0.05, seed=11, store=store)
    flags = dict(t_row_sum=1.0, reset_topic_method=None)
    Wd, Td, od, _ = _run(X, M, W0, T0, k, 3, False, store, flags)
    Ws, Ts, os_, _ = _run(X, M, W0, T0, k, 3, True, store, flags)
    # fp32: with ~6 observations per row the W update is poorly determined, and the two schedules round the stored
    # residual (and, pattern-only, the LDS factor tables) differently: same objective, W to 1e-3
    tol = 1e-10 if store == np.float64 else 1e-3
    assert relfro(Ws, Wd) < tol and relfro(Ts, Td) < tol, (relfro(Ws, Wd), relfro(Ts, Td))
    assert abs(os_ - od) <= (1e-10 if store == np.float64 else 1e-5) * abs(od)


def test_products_and_device_init_on_a_pattern_only_handle():
    """rri_X_times / rri_Xt_times on the observed values (blocked copies, more than one block, more than 64 columns)
    against scipy; nmf(init='nndsvd') of the weighted flavour started from them = started from scikit-learn's"""
    from rri_nmf_amd import nmf as nmf_mod
    from rri_nmf_amd.engine import RRIEngine
    n, d, k = 12500, 310, 5
    X, M, _, _ = _problem(n, d, k, 0.1, seed=2)
    A = sp.csr_matrix(M)
    A.data = X[M > 0]
    rs = np.random.RandomState(0)
    for store in (np.float64, np.float32):
        with RRIEngine(n, d, k, dtype=store, weighted='sparse') as e:
            e.upload_observed_csr(A)
            for m in (1, 15, 64, 70):
                B, Q = rs.randn(d, m), rs.randn(n, m)
                tol = 1e-13 if store == np.float64 else 1e-6
                assert relfro(e.X_times(B), A.astype(store).astype(np.float64) @ B) < tol
                assert relfro(e.Xt_times(Q), A.astype(store).astype(np.float64).T @ Q) < tol
    kw = dict(max_iter=4, eps_stop=-1, t_row_sum=1.0, reset_topic_method=None, random_state=0)
    a = nmf_mod.nmf(A, k, W_mat=sp.csr_matrix(M), device_init=True, **kw)
    b = nmf_mod.nmf(A, k, W_mat=sp.csr_matrix(M), device_init=False, **kw)
    assert relfro(a['W'], b['W']) < 1e-6 and relfro(a['T'], b['T']) < 1e-6, (relfro(a['W'], b['W']), relfro(a['T'], b['T']))
