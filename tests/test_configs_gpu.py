"""Asserting GPU tests for the BASELINE.json configurations that round 1 left to the bench alone, and for the
per-function vectors of the reference (G7) fed through the device.

  C2  10000 x 1000, k = 20: the CPU oracle finishes a sweep there in tens of milliseconds -- full-size comparison,
      plain and topic-model flags, both storage types.
  C4  1000000 x 10000, k = 50 on ONE GPU (40 GB of X): the oracle cannot run; size-independent properties as for C3
      (tests/test_full_size_gpu.py): closed form of a topic step in torch float64, monotone objective, resumability,
      and the row-sharded step protocol on two unequal shards of that matrix against the unsharded handle.
  G4  exact argmax topic assignments on the recommender fixture (13 all-zero rows of W stay exact zeros).
  G7  qf_min (optimization.py:12-88, every branch nmf() can reach) and euclidean_proj_simplex (matrixops.py:5-69)
      vectors captured from the reference, through the T-row update / the row projection of the device.
"""
import numpy as np
import pytest

from conftest import load_golden, relfro
from rri_nmf_amd.synthetic import planted_X, scaled_init

pytestmark = pytest.mark.gpu
EPS = float(np.spacing(10))


def engine(*a, **kw):
    from rri_nmf_amd.engine import RRIEngine
    return RRIEngine(*a, **kw)


def oracle():
    from oracle import rri_oracle
    return rri_oracle


# ---------------------------------------------------------------------------------------------------------------
# C2
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_c2_full_size_against_the_oracle(dtype):
    """BASELINE config 2 at its full size: 2e-9 (summation order only; the same bound as tests/test_hip_parity.py)"""
    n, d, k = 10000, 1000, 20
    X = planted_X(n, d, k, seed=0, dtype=np.float32)          # the bench's generator: fp32-valued X
    X64 = X.astype(np.float64)
    W0, T0 = scaled_init(X64, k, seed=1)
    orc = oracle()
    for sweeps in (1, 5, 10):
        with engine(n, d, k, dtype=dtype) as e:
            e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
            e.sweep(sweeps)
            W, T, nres = e.get_W(), e.get_T(), e.n_resets_used
        ref = orc.nmf(X64, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=sweeps, eps_stop=-1)
        assert nres == 0
        assert relfro(W, ref['W']) < 2e-9 and relfro(T, ref['T']) < 2e-9, (sweeps, relfro(W, ref['W']), relfro(T, ref['T']))
    # topic-model flags (rows of T on the simplex every step, final projection of W), 5 sweeps
    Xn = orc.normalize(X64.copy())
    tm = dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
    T0p = orc.proj_rows_simplex(np.maximum(T0, 0).copy(), 1.0)
    Xs = np.ascontiguousarray(Xn.astype(dtype).astype(np.float64))
    with engine(n, d, k, dtype=dtype) as e:
        e.upload_X(Xn), e.set_W(W0), e.set_T(T0p), e.set_params(**tm)
        e.sweep(5)
        e.project_W_rows(1.0)
        W, T = e.get_W(), e.get_T()
    ref = orc.nmf(Xs, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=5, eps_stop=-1, **tm)
    assert relfro(W, ref['W']) < 2e-9 and relfro(T, ref['T']) < 2e-9, (relfro(W, ref['W']), relfro(T, ref['T']))
    assert np.abs(T.sum(1) - 1).max() < 1e-12 and np.abs(W.sum(1) - 1).max() < 1e-12
    # explicit-residual schedule at the same size (fp32 residual: BASELINE's 1e-4 bar; float64: 2e-9)
    with engine(n, d, k, dtype=dtype, schedule='residual') as e:
        e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
        e.sweep(5)
        W, T = e.get_W(), e.get_T()
    ref = orc.nmf(X64, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=5, eps_stop=-1)
    tol = 2e-9 if dtype == np.float64 else 1e-4
    assert relfro(W, ref['W']) < tol and relfro(T, ref['T']) < tol, (relfro(W, ref['W']), relfro(T, ref['T']))


# ---------------------------------------------------------------------------------------------------------------
# C4 on one GPU
# ---------------------------------------------------------------------------------------------------------------
N4, D4, K4 = 1000000, 10000, 50


@pytest.fixture(scope='module')
def c4_problem():
    import torch
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    Ts = torch.rand(K4, D4, device=dev, generator=g) * (torch.rand(K4, D4, device=dev, generator=g) < 0.3)
    X = torch.empty(N4, D4, device=dev, dtype=torch.float32)
    for lo in range(0, N4, 25000):
        Ws = torch.rand(25000, K4, device=dev, generator=g) * (torch.rand(25000, K4, device=dev, generator=g) < 0.3)
        torch.matmul(Ws, Ts, out=X[lo:lo + 25000])
        X[lo:lo + 25000].add_(torch.rand(25000, D4, device=dev, generator=g), alpha=0.01)
    mean = 0.0
    for lo in range(0, N4, 100000):
        mean += float(X[lo:lo + 100000].sum(dtype=torch.float64))
    a = (mean / (float(N4) * D4) / K4) ** 0.5
    W0 = a * torch.rand(N4, K4, device=dev, generator=g, dtype=torch.float64)
    T0 = a * torch.rand(K4, D4, device=dev, generator=g, dtype=torch.float64)
    torch.cuda.synchronize()
    yield X, W0, T0
    del X, W0, T0
    torch.cuda.empty_cache()


def _f64_matvec(X, v, transpose):
    import torch
    out = torch.zeros(X.shape[1] if transpose else X.shape[0], dtype=torch.float64, device=X.device)
    for lo in range(0, X.shape[0], 10000):
        blk = X[lo:lo + 10000].to(torch.float64)
        if transpose:
            out += blk.t() @ v[lo:lo + 10000]
        else:
            out[lo:lo + 10000] = blk @ v
    return out


def test_c4_topic_step_closed_form_monotone_objective_resumable(c4_problem):
    """more than 2^31 elements of X in one handle: row offsets are 64-bit throughout"""
    import torch
    X, W0, T0 = c4_problem
    W0h, T0h = W0.cpu().numpy(), T0.cpu().numpy()
    t = 7
    with engine(N4, D4, K4, dtype=np.float32) as e:
        e.bind_X_device(X.data_ptr(), X.stride(0))
        e.set_W(W0h), e.set_T(T0h), e.set_params()
        e.update_T_row(t)
        T1 = torch.from_numpy(e.get_T()).to(X.device)
        e.update_W_col(t)
        W1t = torch.from_numpy(np.ascontiguousarray(e.get_W()[:, t])).to(X.device)
        # T row (nmf.py:670-676): wR = w^T X - (w^T W with entry t zeroed) T ; x = max(wR, 0) / (||w||^2 + eps)
        w = W0[:, t]
        gram = w @ W0
        gram[t] = 0
        wR = _f64_matvec(X, w, True) - gram @ T0
        want_T = torch.clamp(wR, min=0) / (w @ w + EPS)
        err_T = float(torch.linalg.norm(T1[t] - want_T) / torch.linalg.norm(want_T))
        assert err_T < 1e-12, err_T
        # W column (nmf.py:728-734) with the new row
        tt = T1[t]
        h = T1 @ tt
        nt = float(h[t])
        h[t] = 0
        Rt = _f64_matvec(X, tt, False) - W0 @ h
        want_W = torch.clamp(Rt, min=0) / (nt + EPS)
        err_W = float(torch.linalg.norm(W1t - want_W) / torch.linalg.norm(want_W))
        assert err_W < 1e-12, err_W
        # sweeps from the start: the objective never increases (the reference's own property, tests/test_nmf.py:40)
        e.set_W(W0h), e.set_T(T0h)
        objs = [e.objective()]
        for _ in range(2):
            e.sweep(1)
            objs.append(e.objective())
        Wa, Ta = e.get_W(), e.get_T()
        assert all(b <= a for a, b in zip(objs, objs[1:])), objs
        assert Wa.min() >= 0 and Ta.min() >= 0 and np.isfinite(Wa).all() and np.isfinite(Ta).all() and e.n_resets_used == 0
        # two calls of one sweep == one call of two sweeps (test_nmf.py:97-110), bit for bit
        e.set_W(W0h), e.set_T(T0h)
        e.sweep(2)
        assert np.array_equal(e.get_W(), Wa) and np.array_equal(e.get_T(), Ta)


def test_c4_two_unequal_row_shards_equal_the_unsharded_handle(c4_problem):
    """the row-sharded step protocol (rri_topic_reduce_local -> sum of the reduce buffers -> rri_topic_finish) on two
    unequal shards (600000 + 400000 rows) of the C4 matrix, in one process on the one GPU of the test box: the sum of
    the two buffers stands in for the all-reduce.  Same factors as one handle holding all rows."""
    import torch
    X, W0, T0 = c4_problem
    W0h, T0h = W0.cpu().numpy(), T0.cpu().numpy()
    cut = 600000
    with engine(N4, D4, K4, dtype=np.float32) as e:
        e.bind_X_device(X.data_ptr(), X.stride(0))
        e.set_W(W0h), e.set_T(T0h), e.set_params()
        e.sweep(1)
        Wa, Ta = e.get_W(), e.get_T()
    from rri_nmf_amd.distributed import make_device_shard
    shards = []
    for lo, hi in ((0, cut), (cut, N4)):
        eng, red, stream = make_device_shard(hi - lo, D4, K4, dtype=np.float32, device_index=0)
        Xs = X[lo:hi]
        eng.bind_X_device(Xs.data_ptr(), Xs.stride(0))
        eng.set_W(W0h[lo:hi]), eng.set_T(T0h), eng.set_params()
        shards.append((eng, red, stream))
    try:
        def allreduce():
            for eng, _, _ in shards:
                eng.synchronize()
            tot = shards[0][1] + shards[1][1]
            torch.cuda.synchronize()
            for _, red, _ in shards:
                red.copy_(tot)
            torch.cuda.synchronize()
        for t in range(K4):
            for eng, _, _ in shards:
                eng.topic_reduce_local(t)
            allreduce()
            for eng, _, _ in shards:
                eng.topic_finish(t)
        for eng, _, _ in shards:                    # the last column's check rides on topic 0's sums
            eng.topic_reduce_local(0)
        allreduce()
        for eng, _, _ in shards:
            eng.topic_finish(-1)
            assert eng.poll() == 0
        W = np.vstack([shards[0][0].get_W(), shards[1][0].get_W()])
        T1, T2 = shards[0][0].get_T(), shards[1][0].get_T()
    finally:
        for eng, _, _ in shards:
            eng.close()
    assert np.array_equal(T1, T2)
    assert relfro(W, Wa) < 1e-10 and relfro(T1, Ta) < 1e-10, (relfro(W, Wa), relfro(T1, Ta))


# ---------------------------------------------------------------------------------------------------------------
# G4: exact topic assignments on the recommender fixture
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_recsys_fixture_topic_assignments_are_exact(dtype):
    """north_star: bit-exact argmax topic assignments on the test fixtures.  The reference's WRRI settings on its
    recsys fixture (tests/test_nmf.py:57-78): 13 rows of W are all zero (users without ratings) -- their argmax is 0 by
    tie only if the device produces exact zeros; the other rows have top-1 / top-2 gaps >= 1e-3 relative."""
    g = load_golden('g4_wrri')
    X = g['X'].astype(np.float64)
    M = np.zeros(X.shape)
    M[X.nonzero()] = 1.0
    cases = [{}, {'reg_w_l1': 0.1, 'reg_t_l1': 0.1}, {'reg_w_l1': 0.1}, {'reg_t_l1': 0.1}]
    for ci, c in enumerate(cases):
        flags = dict(c, reset_topic_method=None, t_row_sum=1.0)
        for S, key in ((1, 'c%d_W_s1'), (2, 'c%d_W_s2'), (6, 'c%d_W_s6'), (15, 'c%d_W')):
            Wr = g[key % ci]
            with engine(X.shape[0], X.shape[1], 7, dtype=dtype, weighted=True) as e:
                e.upload_X(X), e.upload_mask(M), e.set_W(g['W0']), e.set_T(g['T0']), e.set_params(**flags)
                e.sweep(S)
                W = e.get_W()
                am = e.argmax_rows()
            assert np.array_equal(np.argmax(W, 1), np.argmax(Wr, 1)), (ci, S)
            assert np.array_equal(am, np.argmax(Wr, 1)), (ci, S)
            assert np.array_equal(W.max(1) == 0, Wr.max(1) == 0)        # the all-zero rows are exactly zero
    # the estimator's own fit (NMF_RS_Estimator, k = 5), with and without early stopping
    from rri_nmf_amd import sklearn_interface as si
    n, d = X.shape
    for kw, key in ((dict(), 'rs_es_W'), (dict(use_validation_early_stopping=False), 'rs_noes_W')):
        E = si.NMF_RS_Estimator(n, d, 5, random_state=0, max_iter=20, nmf_kwargs={'dtype': dtype}, **kw).fit_from_Xtr(g['X'])
        assert np.array_equal(np.argmax(E.W, 1), np.argmax(g[key], 1)), key


# ---------------------------------------------------------------------------------------------------------------
# G7: the reference's per-function vectors through the device
# ---------------------------------------------------------------------------------------------------------------
def test_g7_simplex_projection_vectors_through_the_device():
    """euclidean_proj_simplex (matrixops.py:5-69) on the reference's vectors: every vector as the single row of a W
    handed to rri_project_W_rows (k_proj_rows), radii 1 and 2.5, and as a per-row radius vector"""
    g = load_golden('g7_functions')
    names = ['rand', 'pos', 'zeros', 'onsimplex', 'ties', 'single', 'neg', 'big']
    for nm in names:
        v = g['proj_in_' + nm]
        for s in (1.0, 2.5):
            want = g['proj_out_%s_s%g' % (nm, s)]
            with engine(1, 2, v.size, dtype=np.float64) as e:
                e.set_W(v.reshape(1, -1))
                e.project_W_rows(s)
                got = e.get_W().ravel()
            assert np.abs(got - want).max() <= 1e-14 * max(1.0, np.abs(want).max()), (nm, s, np.abs(got - want).max())
            assert abs(got.sum() - s) < 1e-13 and got.min() >= 0
    # same vectors as rows of ONE matrix with per-row radii (proj_mat_to_simplex with a vector s, matrixops.py:88-98)
    same = [nm for nm in names if g['proj_in_' + nm].size == 50] + ['pos']
    v = g['proj_in_pos']
    with engine(2, 2, v.size, dtype=np.float64) as e:
        e.set_W(np.vstack([v, v]))
        e.project_W_rows(np.array([1.0, 2.5]))
        got = e.get_W()
    assert np.abs(got[0] - g['proj_out_pos_s1']).max() <= 1e-14 and np.abs(got[1] - g['proj_out_pos_s2.5']).max() <= 1e-14
    assert same


def _trow_through_device(w, c_scalar=None, c_vec=None, s=None, ub=None):
    """qf_min(w, c, s, ub) as the T-row update of a one-document problem evaluates it (nmf.py:437-447):
    numer = w_t^T X = -w with W = [[1]], denom = ||w_t||^2 + reg_t_l2 = c (scalar), or the weighted sums with the
    weights c_vec (vector c); s = t_row_sum when project_T_each_iter, ub = t_row_sum"""
    d = w.size
    if c_vec is None:
        X = (-w).reshape(1, d)
        flags = dict(reg_t_l2=c_scalar - 1.0, project_T_each_iter=s is not None,
                     t_row_sum=(s if s is not None else ub), reset_topic_method=None)
        with engine(1, d, 1, dtype=np.float64) as e:
            e.upload_X(X), e.set_W(np.ones((1, 1))), e.set_T(np.full((1, d), 0.5)), e.set_params(**flags)
            e.update_T_row(0)
            return e.get_T().ravel(), None
    X = (-w / c_vec).reshape(1, d)                       # numer_j = M_j X_j = -w_j, denominator (w_t^2)^T M = c_vec
    flags = dict(project_T_each_iter=s is not None, t_row_sum=(s if s is not None else ub), reset_topic_method=None,
                 fix_W=True)
    with engine(1, d, 1, dtype=np.float64, weighted=True) as e:
        e.upload_X(X), e.upload_mask(c_vec.reshape(1, d)), e.set_W(np.ones((1, 1))), e.set_T(np.full((1, d), 0.5))
        e.set_params(**flags)
        e.sweep(1)
        return e.get_T().ravel(), float(e.get_W()[0, 0])     # fix_W without penalties: W[:,t] *= nx (nmf.py:450-452)


def test_g7_qf_min_vectors_through_the_device():
    g = load_golden('g7_functions')
    w, cvec, cpos = g['qf_w'], g['qf_cvec'], g['qf_cpos']
    close = lambda got, want: np.abs(got - want).max() <= 1e-13 * max(1.0, np.abs(want).max())
    scalar = {'scalar_pos_s1': (0.7, 1.0, 1.0), 'scalar_pos_sNone': (0.7, None, 1.0),
              'scalar_pos_sNone_ubNone': (0.7, None, None), 'scalar_neg_sNone_ub': (-0.3, None, 0.8),
              'scalar_zero_sNone_ub': (0.0, None, 0.8), 'scalar_neg_s1': (-0.3, 1.0, 1.0)}
    for nm, (c, s, ub) in scalar.items():
        x, _ = _trow_through_device(w, c_scalar=c, s=s, ub=ub)
        assert close(x, g['qf_x_' + nm]), (nm, np.abs(x - g['qf_x_' + nm]).max())
    vec = {'vec_pos_ub1': (cpos, None, 1.0), 'vec_pos_ubNone': (cpos, None, None), 'vec_mixed_ub1': (cvec, None, 1.0),
           'vec_pos_s1': (cpos, 1.0, 1.0)}
    for nm, (c, s, ub) in vec.items():
        x, nx = _trow_through_device(w, c_vec=c, s=s, ub=ub)
        assert close(x, g['qf_x_' + nm]), (nm, np.abs(x - g['qf_x_' + nm]).max())
        assert abs(nx - float(g['qf_nx_' + nm])) <= 1e-12 * max(1.0, abs(float(g['qf_nx_' + nm]))), (nm, nx)
    # the reference's exceptions for the unreachable minima (optimization.py:66-67, 72-73, 76-77)
    errs = dict((a, b) for a, b in g['qf_errors'])
    assert errs == {'scalar_neg_unb': 'ValueError', 'vec_neg_unb': 'ValueError', 'scalar_neg_s2': 'NotImplementedError'}
    with pytest.raises(ValueError, match='unbounded'):
        _trow_through_device(w, c_scalar=-0.3, s=None, ub=None)
    with pytest.raises(ValueError, match='unbounded'):
        _trow_through_device(w, c_vec=cvec, s=None, ub=None)
    with pytest.raises(NotImplementedError):
        _trow_through_device(w, c_scalar=-0.3, s=2.0, ub=2.0)
