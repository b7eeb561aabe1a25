"""Pins oracle/rri_oracle.py (the CPU restatement) to vectors captured from the
unmodified reference by oracle/make_golden.py.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import os

from conftest import load_golden, GOLDEN, relfro
from oracle import rri_oracle as orc
from rri_nmf_amd.synthetic import planted_X, scaled_init

def same(a, b, tol=1e-8):
    """Bit-identical in the container the vectors were captured in (same numpy/OpenBLAS build and CPU).
    On another host OpenBLAS may pick other kernels / thread splits and sum in another order; the
    iteration amplifies that (see tests/test_hip_parity.py), so elsewhere: relative Frobenius <= tol."""
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if np.array_equal(a, b):
        return True
    if a.dtype.kind in 'iub' or b.dtype.kind in 'iub':
        return False
    return relfro(a, b) <= tol


def ref_fixture(name):
    return sp.load_npz(os.path.join(GOLDEN, 'ref_data', name + '.npz')).toarray()


# ---------------------------------------------------------------- G7 functions
def test_simplex_projection_vectors():
    g = load_golden('g7_functions')
    for nm in ('rand', 'pos', 'zeros', 'onsimplex', 'ties', 'single', 'neg', 'big'):
        v = g['proj_in_' + nm]
        for s in (1.0, 2.5):
            out = orc.proj_simplex(v.copy(), s)
            assert same(out, g['proj_out_%s_s%g' % (nm, s)]), (nm, s)
            assert abs(out.sum() - s) < 1e-12 and out.min() >= 0


def test_qf_min_all_branches():
    g = load_golden('g7_functions')
    w, cvec, cpos = g['qf_w'], g['qf_cvec'], g['qf_cpos']
    calls = {
        'scalar_pos_s1': (w, 0.7, 1.0, 1.0), 'scalar_pos_sNone': (w, 0.7, None, 1.0),
        'scalar_pos_sNone_ubNone': (w, 0.7, None, None), 'scalar_pos_s2': (w, 0.7, 2.0, 1.0),
        'scalar_neg_sNone_ub': (w, -0.3, None, 0.8), 'scalar_zero_sNone_ub': (w, 0.0, None, 0.8),
        'scalar_neg_s1': (w, -0.3, 1.0, 1.0),
        'vec_pos_ub1': (w, cpos, None, 1.0), 'vec_pos_ubNone': (w, cpos, None, None),
        'vec_mixed_ub1': (w, cvec, None, 1.0), 'vec_pos_s1': (w, cpos, 1.0, 1.0),
    }
    for nm, (ww, c, s, ub) in calls.items():
        x, nx = orc.qf_min(ww.copy(), c, s=s, ub=ub)
        assert same(x, g['qf_x_' + nm]), nm
        assert same(nx, g['qf_nx_' + nm]), nm
    errs = dict(g['qf_errors'])
    assert errs == {'scalar_neg_unb': 'ValueError', 'vec_neg_unb': 'ValueError',
                    'scalar_neg_s2': 'NotImplementedError'}
    with pytest.raises(ValueError):
        orc.qf_min(w.copy(), -0.3, s=None, ub=None)
    with pytest.raises(ValueError):
        orc.qf_min(w.copy(), cvec, s=None, ub=None)
    with pytest.raises(NotImplementedError):
        orc.qf_min(w.copy(), -0.3, s=2.0, ub=3.0)


def test_stop_rule_objective_preprocessing():
    g = load_golden('g7_functions')
    hist = [10.0, 8.0, 7.5, 7.4999]
    got = [orc.universal_stopping_condition(hist[:1]), orc.universal_stopping_condition(hist[:2]),
           orc.universal_stopping_condition(hist[:3]), orc.universal_stopping_condition(hist),
           orc.universal_stopping_condition(hist, -1)]
    assert same(np.array(got), g['usc'])
    X = planted_X(60, 30, 4, seed=5, dtype=np.float64)
    W0, T0 = scaled_init(X, 4, seed=6)
    assert orc.true_objective(X, W0, T0, 0.1, 0.2, 0.3, 0.4) == g['obj_plain']
    assert orc.true_objective(X, W0, T0, 0.1, 0.2, 0.3, 0.4, Wm=g['obj_M']) == g['obj_masked']
    Xt = orc.normalize(orc.tfidf(ref_fixture('text_data_train')))
    assert same(np.array([Xt.sum(), (Xt ** 2).sum()]), g['tfidf_norm_checksum'])


def test_init_known_answer():
    """the one numeric known-answer test the reference holds (tests/test_nmf.py:13-19)"""
    g = load_golden('g8_init')
    W, T = orc.initialize_nmf(g['X'], 2, init='nndsvd', random_state=0)
    assert np.allclose(g['W_expected'], W) and np.allclose(g['T_expected'], T)


# ---------------------------------------------------------------- G5 plain / TM / WRRI on synthetic X
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_plain_synthetic(tag):
    g = load_golden('g5_plain_' + tag)
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    assert same(np.array([X.sum(), (X ** 2).sum()]), g['x_checksum'])
    W0, T0 = scaled_init(X, k, seed=1)
    for S in (1, 5, 30):
        r = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=S, eps_stop=-1)
        assert same(r['W'], g['W_s%d' % S]) and same(r['T'], g['T_s%d' % S]), S
    Xn = orc.normalize(X.copy())
    for S in (1, 5):
        r = orc.nmf(Xn, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=S, eps_stop=-1,
                    project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
        assert same(r['W'], g['tm_W_s%d' % S]) and same(r['T'], g['tm_T_s%d' % S])
    r = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=5, eps_stop=-1,
                reg_w_l1=0.01, reg_t_l1=0.02, reg_w_l2=0.05, reg_t_l2=0.03)
    assert same(r['W'], g['reg_W_s5']) and same(r['T'], g['reg_T_s5'])
    r = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=3, eps_stop=-1, fix_T=True)
    assert same(r['W'], g['fixT_W_s3'])
    r = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=3, eps_stop=-1, fix_W=True)
    assert same(r['W'], g['fixW_W_s3']) and same(r['T'], g['fixW_T_s3'])
    M = (np.random.RandomState(2).rand(n, d) < 0.3).astype(np.float64)
    r = orc.nmf(M * X, k, W_in=W0.copy(), T_in=T0.copy(), W_mat=M, max_iter=4, eps_stop=-1,
                t_row_sum=1.0, reset_topic_method=None, compute_obj_each_iter=True)
    assert same(r['W'], g['wr_W_s4']) and same(r['T'], g['wr_T_s4'])
    assert same(np.array(r['obj_history']), g['wr_obj'])


# ---------------------------------------------------------------- G1/G2 TM estimator flags on the text fixture
def test_tm_fixture():
    g = load_golden('g1_tm_estimator')
    X = orc.normalize(orc.tfidf(ref_fixture('text_data_train')))
    assert same(X, g['X'])
    flags = dict(project_W_each_iter=False, w_row_sum=1.0, project_T_each_iter=True,
                 t_row_sum=1.0, random_state=0, max_time=7200)
    for S in (1, 2, 10):
        r = orc.nmf(X, 5, max_iter=S, eps_stop=-1, W_in=g['W0'].copy(), T_in=g['T0'].copy(), **flags)
        assert same(r['W'], g['W_s%d' % S]) and same(r['T'], g['T_s%d' % S])
        r = orc.nmf(X, 5, max_iter=S, eps_stop=-1, W_in=g['W0'].copy(), T_in=g['T0'].copy(),
                    do_final_project_W=False, **flags)
        assert same(r['W'], g['Wraw_s%d' % S])
    # as shipped: objective each sweep + stop rule
    r = orc.nmf(X, 5, max_iter=10, W_in=g['W0'].copy(), T_in=g['T0'].copy(),
                objective_always=True, **flags)
    assert same(r['W'], g['W_shipped']) and same(np.array(r['obj_history']), g['obj_shipped'])
    assert same(np.argmax(r['W'], 1), np.argmax(g['W_shipped'], 1))
    # own init (sklearn randomized_svd) reproduces the reference's starting point here
    r = orc.nmf(X, 5, max_iter=10, eps_stop=-1, **flags)
    assert np.allclose(r['W'], g['W_s10'], atol=1e-12) and same(np.argmax(r['W'], 1), g['argmax_s10'])
    # fold-in (G2)
    r = orc.nmf(g['Xte'], 5, max_iter=4, max_time=7200, project_W_each_iter=False, w_row_sum=1.0,
                t_row_sum=1.0, T_in=g['T_s10'].copy(), W_in=g['Wte0'].copy(), fix_T=True, random_state=0)
    assert same(r['W'], g['Wte']) and same(np.argmax(r['W'], 1), g['argmax_te'])


# ---------------------------------------------------------------- G3 the reference's own TM convergence settings
def test_tm_settings_properties_and_vectors():
    g = load_golden('g3_tm_settings')
    X = g['X']
    cases = [{'k': 25}, {'k': 15, 'reg_t_l2': 0.1}, {'k': 15, 'reg_t_l2': -0.1}, {'k': 15, 'reg_w_l2': 0.1}]
    for ci, c in enumerate(cases):
        p = dict(c, max_iter=15, w_row_sum=1.0, random_state=0, eps_stop=1e-4,
                 project_T_each_iter=True, project_W_each_iter=True, compute_obj_each_iter=True,
                 t_row_sum=1.0, early_stop=False)
        r = orc.nmf(X, W_in=g['c%d_W0' % ci].copy(), T_in=g['c%d_T0' % ci].copy(), **p)
        assert same(r['W'], g['c%d_W' % ci]) and same(r['T'], g['c%d_T' % ci])
        oh = np.array(r['obj_history'])
        assert same(oh, g['c%d_obj' % ci])
        assert np.all(np.diff(oh) <= 0)                       # tests/test_nmf.py:40
        cv = np.sum(np.abs(r['W'].sum(1) - 1)) + np.sum(np.abs(r['T'].sum(1) - 1))
        assert cv <= 1e-13 and r['W'].min() >= -1e-13 and r['T'].min() >= -1e-13  # :41-55


# ---------------------------------------------------------------- G4 WRRI on the recsys fixture
def test_wrri_fixture():
    g = load_golden('g4_wrri')
    X = g['X']
    Wm = np.zeros(X.shape)
    Wm[X.nonzero()] = 1.0
    cases = [{}, {'reg_w_l1': 0.1, 'reg_t_l1': 0.1}, {'reg_w_l1': 0.1}, {'reg_t_l1': 0.1}]
    for ci, c in enumerate(cases):
        p = dict(c, max_iter=15, random_state=0, W_mat=Wm, compute_obj_each_iter=True,
                 reset_topic_method=None, early_stop=False, k=7, project_T_each_iter=False,
                 t_row_sum=1.0, project_W_each_iter=False, w_row_sum=None)
        r = orc.nmf(X, W_in=g['W0'].copy(), T_in=g['T0'].copy(), **p)
        # denominators down to 1e-27 on this fixture: off-container only M.(WT) is stable
        assert same(Wm * (r['W'] @ r['T']), Wm * (g['c%d_W' % ci] @ g['c%d_T' % ci]), tol=1e-6)
        oh = np.array(r['obj_history'])
        assert same(oh, g['c%d_obj' % ci], tol=1e-6) and np.all(np.diff(oh) <= 0)   # tests/test_nmf.py:78
        for S in (1, 2, 6):
            q = dict(p, max_iter=S, eps_stop=-1)
            r = orc.nmf(X, W_in=g['W0'].copy(), T_in=g['T0'].copy(), **q)
            tolS = 1e-8 if S <= 2 else 1e-3
            assert same(r['W'], g['c%d_W_s%d' % (ci, S)], tolS) and same(r['T'], g['c%d_T_s%d' % (ci, S)], tolS)


# ---------------------------------------------------------------- G6 rare branches
def test_rare_branches():
    g = load_golden('g6_rare_branches')
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=3, dtype=np.float64)
    assert same(np.array([X.sum(), (X ** 2).sum()]), g['x_checksum'])
    W0, T0 = scaled_init(X, k, seed=4)
    Xn = orc.normalize(X.copy())
    r = orc.nmf(Xn, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=3, eps_stop=-1,
                project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0, reg_t_l2=-50.0)
    assert same(r['W'], g['negT_W']) and same(r['T'], g['negT_T'])
    r = orc.nmf(Xn, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=2, eps_stop=-1,
                project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0, reg_w_l2=-5.0,
                do_final_project_W=False)
    assert same(r['W'], g['negW_W']) and same(r['T'], g['negW_T'])
    Wd = g['dead_W0']
    common = dict(W_in=None, T_in=None, max_iter=2, eps_stop=-1)

    def run(**kw):
        return orc.nmf(X, k, **dict(common, W_in=Wd.copy(), T_in=T0.copy(), **kw))
    with pytest.raises(ValueError):
        run(reset_topic_method='max_resid_document')
    assert str(g['dead_unbounded_error']).startswith('ValueError')
    r = run(t_row_sum=1.0, reset_topic_method='max_resid_document')
    assert same(r['W'], g['dead_mrd_W']) and same(r['T'], g['dead_mrd_T']) and r['n_resets_used'] >= 1
    with pytest.raises(AssertionError, match='sums to 0'):
        run(t_row_sum=1.0, w_row_sum=1.0, do_final_project_W=False, reset_topic_method=None)
    with pytest.raises(AssertionError, match='sums to 0'):
        run(t_row_sum=1.0, w_row_sum=1.0, do_final_project_W=False, n_resets=0)
    with pytest.raises(ValueError):
        run(t_row_sum=1.0, reset_topic_method=None)
    r = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=1, eps_stop=-1, t_row_sum=1.0, reg_t_l1=1e6)
    assert same(r['W'], g['l1kill_W']) and same(r['T'], g['l1kill_T'])
    r = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=1, eps_stop=-1, t_row_sum=1.0,
                reg_w_l1=1e6, reset_topic_method='random', fix_reset_seed=True)
    assert same(r['W'], g['l1killW_rnd_W']) and same(r['T'], g['l1killW_rnd_T'])
    r = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=1, eps_stop=-1, t_row_sum=1.0,
                reg_w_l1=1e6, reset_topic_method='max_resid_document')
    assert same(r['W'], g['l1killW_mrd_W']) and same(r['T'], g['l1killW_mrd_T'])
    # sentinels (nmf.py:292-315)
    s1 = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), reg_t_l2=-1.0)
    s2 = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), reg_w_l1=-1.0)
    assert same(s1['W'][:2, :2], g['sent_T_W']) and same(s1['T'][:2, :2], g['sent_T_T'])
    assert same(s2['W'][:2, :2], g['sent_W_W']) and same(s2['T'][:2, :2], g['sent_W_T'])
    assert s1['obj_history'] == [-np.inf]
    # stop rule on synthetic data
    r = orc.nmf(Xn, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=40, eps_stop=1e-3,
                project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0,
                project_W_each_iter=True, compute_obj_each_iter=True)
    assert same(np.array(r['obj_history']), g['stop_obj']) and same(r['W'], g['stop_W'])
    assert len(r['obj_history']) < 40


def _g9_cases():
    g = load_golden('g9_gaussian_mechanism')
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=5, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=6)
    M = (np.random.RandomState(7).rand(n, d) < 0.4).astype(np.float64)
    base = dict(max_iter=3, eps_stop=-1, t_row_sum=1.0, delta_gauss_t=0.1)
    cases = []
    for tag, eps_g in (('small', 1e5), ('large', 1e2)):
        cases.append(('plain_' + tag, X, None, dict(base, eps_gauss_t=eps_g)))
        cases.append(('weighted_' + tag, M * X, M, dict(base, eps_gauss_t=eps_g, reset_topic_method=None)))
    cases.append(('tm_small', X, None, dict(base, eps_gauss_t=1e5, project_T_each_iter=True, w_row_sum=1.0)))
    return g, k, W0, T0, cases


def test_gaussian_mechanism():
    """nmf.py:422-435 with numpy's global RNG seeded as the capture was"""
    g, k, W0, T0, cases = _g9_cases()
    for name, X, M, kw in cases:
        np.random.seed(int(g['seed'][0]))
        r = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), W_mat=M, **kw)
        assert same(r['W'], g[name + '_W']) and same(r['T'], g[name + '_T']), name


def test_oracle_store_gradients_restates_the_intended_stacks():
    """store_gradients has no vectors from the reference (it raises at nmf.py:543): the oracle's restatement is checked
    against the definitions of wR_store / nw_store (nmf.py:670-686) on the first topic step, where W and T are the
    starting factors"""
    from rri_nmf_amd.synthetic import planted_X, scaled_init
    n, d, k = 40, 23, 3
    X = planted_X(n, d, k, seed=1, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=2)
    rows = [1, 5, 30]
    full = orc.nmf(X.copy(), k, W_in=W0.copy(), T_in=T0.copy(), max_iter=2, store_gradients=True)
    part = orc.nmf(X.copy(), k, W_in=W0.copy(), T_in=T0.copy(), max_iter=2, store_gradients=True, ind_rows_to_store=rows)
    W0c, T0c = np.maximum(W0, 0), np.maximum(T0, 0)
    g = W0c[:, 0] @ W0c
    g[0] = 0
    assert np.allclose(full['numer_W'][0][0], W0c[:, 0] @ X - g @ T0c, rtol=1e-13)
    assert np.allclose(full['denom_W'][0][0], (W0c[:, 0] ** 2).sum(), rtol=1e-13)
    gs = W0c[rows, 0] @ W0c[rows, :]
    gs[0] = 0
    assert np.allclose(part['numer_W'][0][0], W0c[rows, 0] @ X[rows, :] - gs @ T0c, rtol=1e-13)
    assert full['numer_W'][1].shape == (k, d) and full['denom_W'][1].shape == (k, 1)
    assert np.array_equal(full['W'], part['W'])          # recording changes nothing
    M = (np.random.RandomState(3).rand(n, d) < 0.5).astype(float)
    w = orc.nmf(X * M, k, W_in=W0.copy(), T_in=T0.copy(), W_mat=M, max_iter=1, t_row_sum=1.0,
                reset_topic_method=None, store_gradients=True)
    assert w['numer_W'][0].shape == (k, d) and w['denom_W'][0].shape == (k, d)
    assert np.allclose(w['denom_W'][0][0], (W0c[:, 0] ** 2) @ M, rtol=1e-13)
