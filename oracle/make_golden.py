"""Captures golden vectors for the RRI path from the UNMODIFIED reference.

Run in the build container only:   python oracle/make_golden.py
Writes tests/golden/*.npz (committed).  The reference pins no W,T numbers for
this path in its own tests, so the vectors are outputs of
the reference itself, run here through oracle/ref_loader.py.

Groups: G1 TM estimator fit, G2 TM transform, G3 the four
TM convergence settings, G4 the four WRRI settings + the RS estimator, G5 plain
RRI on seeded synthetic X, G6 rare branches (c<=0, resets), G7 per-function
vectors (qf_min, simplex projection), G8 init known-answer (tests/conftest.py), G9 Gaussian mechanism,
G10 the coherence beam-search initialiser.
"""
import os
import sys
import logging

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from rri_nmf_amd.synthetic import planted_X, scaled_init  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
ref = ref_loader.load(quiet_objective=True)
rnmf = ref.nmf.nmf


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote', path, os.path.getsize(path), 'bytes')


def fixture(name):
    return sp.load_npz(os.path.join(ref_loader.REF_DATA, name + '.npz')).toarray()


def tm_xform(X):  # tests/conftest.py:31-32
    return ref.matrixops.normalize(ref.matrixops.tfidf(X))


def copy_in(a):
    return np.array(a, dtype=np.float64, copy=True)


# ---------------------------------------------------------------- G1 / G2
def g1_g2():
    TM = ref.sklearn_interface.NMF_TM_Estimator
    X = tm_xform(fixture('text_data_train'))
    Xte = tm_xform(fixture('text_data_test'))
    n, d = X.shape
    k = 5
    # the initial state nmf() starts from (nmf.py:840-878 with the TM flags)
    W0, T0 = ref.nmf._initialize_and_validate(
        W_in=[], T_in=[], W_mat=None, X=X, k=k, init='nndsvd', random_state=0,
        project_T_each_iter=True, project_W_each_iter=False, w_row_sum=1.0, t_row_sum=1.0,
        fix_W=False, fix_T=False, n=n, d=d)
    out = dict(X=X, Xte=Xte, W0=W0, T0=T0)
    for S in (1, 2, 10):
        M = TM(n, d, k, random_state=0, max_iter=S, nmf_kwargs={'eps_stop': -1})
        M.fit(X)
        out['W_s%d' % S], out['T_s%d' % S] = M.W, M.T
        M = TM(n, d, k, random_state=0, max_iter=S, do_final_project_W=False,
               nmf_kwargs={'eps_stop': -1})
        M.fit(X)
        out['Wraw_s%d' % S] = M.W
    # as shipped: logger level NOTSET => objective every sweep + stop rule (nmf.py:366)
    ref.nmf.logger.setLevel(logging.NOTSET)
    M = TM(n, d, k, random_state=0, max_iter=10)
    M.fit(X)
    ref.nmf.logger.setLevel(logging.WARNING)
    out['W_shipped'], out['T_shipped'] = M.W, M.T
    out['obj_shipped'] = np.array(M.nmf_outputs['obj_history'])
    out['argmax_s10'] = np.argmax(out['W_s10'], 1)
    # G2: fold-in of the held-out documents with the 10-sweep topics
    M = TM(n, d, k, random_state=0, max_iter=10, nmf_kwargs={'eps_stop': -1})
    M.fit(X)
    Wte = M.transform(Xte)
    out['Wte'] = Wte
    out['argmax_te'] = np.argmax(Wte, 1)
    # the state transform() starts from: NNDSVD of Xte, T replaced (nmf.py:840-860)
    Wte0, _ = ref.nmf._initialize_and_validate(
        W_in=[], T_in=M.T, W_mat=None, X=Xte, k=k, init='nndsvd', random_state=0,
        project_T_each_iter=False, project_W_each_iter=False, w_row_sum=1.0, t_row_sum=1.0,
        fix_W=False, fix_T=True, n=Xte.shape[0], d=d)
    out['Wte0'] = Wte0
    out['score_te'] = M.score(Xte)
    save('g1_tm_estimator', **out)


# ---------------------------------------------------------------- G3
def g3():
    X = tm_xform(fixture('text_data_train'))
    n, d = X.shape
    out = dict(X=X)
    cases = [{'k': 25}, {'k': 15, 'reg_t_l2': 0.1}, {'k': 15, 'reg_t_l2': -0.1},
             {'k': 15, 'reg_w_l2': 0.1}]
    base = dict(max_iter=15, w_row_sum=1.0, random_state=0, eps_stop=1e-4,
                project_T_each_iter=True, project_W_each_iter=True,
                compute_obj_each_iter=True, t_row_sum=1.0, early_stop=False)
    for ci, c in enumerate(cases):
        p = dict(c)
        p.update(base)
        # raw start, before nmf() clamps and projects it (nmf.py:845-850)
        W0, T0 = ref.initialization.initialize_nmf(X, p['k'], 'nndsvd', random_state=0,
                                                   row_normalize=False)
        T0 = ref.matrixops.normalize(T0) * 1.0
        W0 = ref.matrixops.normalize(W0) * 1.0
        soln = rnmf(X, **p)
        out['c%d_W0' % ci], out['c%d_T0' % ci] = W0, T0
        out['c%d_W' % ci], out['c%d_T' % ci] = soln['W'], soln['T']
        out['c%d_obj' % ci] = np.array(soln['obj_history'])
    save('g3_tm_settings', **out)


# ---------------------------------------------------------------- G4
def g4():
    X = fixture('recsys_data_train')
    Xte = fixture('recsys_data_test')
    n, d = X.shape
    Wm = np.zeros(X.shape)
    Wm[X.nonzero()] = 1.0
    out = dict(X=X, Xte=Xte)
    cases = [{}, {'reg_w_l1': 0.1, 'reg_t_l1': 0.1}, {'reg_w_l1': 0.1}, {'reg_t_l1': 0.1}]
    base = dict(max_iter=15, random_state=0, W_mat=Wm, compute_obj_each_iter=True,
                reset_topic_method=None, early_stop=False, k=7, project_T_each_iter=False,
                t_row_sum=1.0, project_W_each_iter=False, w_row_sum=None)
    W0, T0 = ref.nmf._initialize_and_validate(
        W_in=[], T_in=[], W_mat=Wm, X=X, k=7, init='nndsvd', random_state=0,
        project_T_each_iter=False, project_W_each_iter=False, w_row_sum=None,
        t_row_sum=1.0, fix_W=False, fix_T=False, n=n, d=d)
    out['W0'], out['T0'] = W0, T0
    for ci, c in enumerate(cases):
        p = dict(c)
        p.update(base)
        soln = rnmf(X, **p)
        out['c%d_W' % ci], out['c%d_T' % ci] = soln['W'], soln['T']
        out['c%d_obj' % ci] = np.array(soln['obj_history'])
        for S in (1, 2, 6):  # early sweeps, before ill-conditioning amplifies
            q = dict(p)
            q.update(max_iter=S, eps_stop=-1)
            s2 = rnmf(X, **q)
            out['c%d_W_s%d' % (ci, S)], out['c%d_T_s%d' % (ci, S)] = s2['W'], s2['T']
    RS = ref.sklearn_interface.NMF_RS_Estimator
    for tag, es in (('es', True), ('noes', False)):
        E = RS(n, d, 5, random_state=0, max_iter=20, use_validation_early_stopping=es)
        E = E.fit_from_Xtr(X)
        out['rs_%s_W' % tag], out['rs_%s_T' % tag] = E.W, E.T
        out['rs_%s_obj' % tag] = np.array(E.nmf_outputs['obj_history'])
        out['rs_%s_score' % tag] = E.score(X)
        out['rs_%s_score_te' % tag] = E.score(Xte)
    save('g4_wrri', **out)


# ---------------------------------------------------------------- G5
def g5():
    for tag, (n, d, k) in (('a', (500, 100, 5)), ('b', (2000, 300, 20))):
        X = planted_X(n, d, k, seed=0, dtype=np.float64)
        W0, T0 = scaled_init(X, k, seed=1)
        out = dict(shape=np.array([n, d, k]), x_checksum=np.array([X.sum(), (X ** 2).sum()]),
                   W0_checksum=np.array([W0.sum(), T0.sum()]))
        for S in (1, 5, 30):
            soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=S, eps_stop=-1)
            out['W_s%d' % S], out['T_s%d' % S] = soln['W'], soln['T']
        # topic-model flags on the same X (row sums 1)
        Xn = ref.matrixops.normalize(X.copy())
        for S in (1, 5):
            soln = rnmf(Xn, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=S, eps_stop=-1,
                        project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
            out['tm_W_s%d' % S], out['tm_T_s%d' % S] = soln['W'], soln['T']
        # regularised plain flavour
        soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=5, eps_stop=-1,
                    reg_w_l1=0.01, reg_t_l1=0.02, reg_w_l2=0.05, reg_t_l2=0.03)
        out['reg_W_s5'], out['reg_T_s5'] = soln['W'], soln['T']
        # halves: fix_T (fold-in) and fix_W
        soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=3, eps_stop=-1, fix_T=True)
        out['fixT_W_s3'] = soln['W']
        soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=3, eps_stop=-1, fix_W=True)
        out['fixW_W_s3'], out['fixW_T_s3'] = soln['W'], soln['T']
        # weighted flavour, well-conditioned 30 % mask
        rs = np.random.RandomState(2)
        M = (rs.rand(n, d) < 0.3).astype(np.float64)
        soln = rnmf(M * X, k, W_in=copy_in(W0), T_in=copy_in(T0), W_mat=M, max_iter=4,
                    eps_stop=-1, t_row_sum=1.0, reset_topic_method=None,
                    compute_obj_each_iter=True)
        out['wr_W_s4'], out['wr_T_s4'] = soln['W'], soln['T']
        out['wr_obj'] = np.array(soln['obj_history'])
        save('g5_plain_%s' % tag, **out)


# ---------------------------------------------------------------- G6
def g6():
    n, d, k = 300, 80, 6
    X = planted_X(n, d, k, seed=3, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=4)
    out = dict(shape=np.array([n, d, k]), x_checksum=np.array([X.sum(), (X ** 2).sum()]))
    Xn = ref.matrixops.normalize(X.copy())
    # (a) T side c<=0 with s==1.0 -> one-hot at argmin (optimization.py:68-70)
    soln = rnmf(Xn, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=3, eps_stop=-1,
                project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0, reg_t_l2=-50.0)
    out['negT_W'], out['negT_T'] = soln['W'], soln['T']
    # (b) W side c<=0 with s None, ub=w_row_sum -> entries set to ub (optimization.py:62-65)
    soln = rnmf(Xn, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=2, eps_stop=-1,
                project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0, reg_w_l2=-5.0,
                do_final_project_W=False)
    out['negW_W'], out['negW_T'] = soln['W'], soln['T']
    # (c) dead column.  With no upper bound the reference does NOT reset: c = ||w||^2 = 0 takes
    #     the scalar c<=0 branch and raises (optimization.py:60-67).
    Wd = copy_in(W0)
    Wd[:, 2] = 0.0
    out['dead_W0'] = Wd

    def err_of(**kw):
        try:
            rnmf(X, k, W_in=copy_in(Wd), T_in=copy_in(T0), max_iter=2, eps_stop=-1, **kw)
            return np.array('none')
        except (AssertionError, ValueError, NotImplementedError, NameError) as e:
            return np.array(type(e).__name__ + ':' + str(e)[:40])
    out['dead_unbounded_error'] = err_of(reset_topic_method='max_resid_document')
    #     With ub = t_row_sum the T row comes out all-zero and the reset handlers run
    #     (nmf.py:762-783).
    soln = rnmf(X, k, W_in=copy_in(Wd), T_in=copy_in(T0), max_iter=2, eps_stop=-1,
                t_row_sum=1.0, reset_topic_method='max_resid_document')
    out['dead_mrd_W'], out['dead_mrd_T'] = soln['W'], soln['T']
    #     'random' from the T side is broken in the reference itself (nmf.py:783 uses `n`,
    #     which is not in that function's scope): recorded as an error vector.
    out['dead_rnd_error'] = err_of(t_row_sum=1.0, reset_topic_method='random',
                                   fix_reset_seed=True)
    #     'random' from the W side (nmf.py:811-816) works: kill every W column with a huge l1.
    soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=1, eps_stop=-1,
                t_row_sum=1.0, reg_w_l1=1e6, reset_topic_method='random', fix_reset_seed=True)
    out['l1killW_rnd_W'], out['l1killW_rnd_T'] = soln['W'], soln['T']
    soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=1, eps_stop=-1,
                t_row_sum=1.0, reg_w_l1=1e6, reset_topic_method='max_resid_document')
    out['l1killW_mrd_W'], out['l1killW_mrd_T'] = soln['W'], soln['T']
    #     resets off / exhausted: W column stays zero -> assert (nmf.py:476) when W is bounded,
    #     ValueError when it is not
    out['dead_none_error'] = err_of(t_row_sum=1.0, w_row_sum=1.0, do_final_project_W=False,
                                    reset_topic_method=None)
    out['dead_budget0_error'] = err_of(t_row_sum=1.0, w_row_sum=1.0, do_final_project_W=False,
                                       n_resets=0)
    out['dead_none_unb_error'] = err_of(t_row_sum=1.0, reset_topic_method=None)
    # (d) every T row killed by a huge l1 penalty: resets until the budget (23) runs out
    soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=1, eps_stop=-1,
                t_row_sum=1.0, reg_t_l1=1e6)
    out['l1kill_W'], out['l1kill_T'] = soln['W'], soln['T']
    # (e) unbounded-objective sentinels (nmf.py:292-315)
    s1 = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), reg_t_l2=-1.0)
    s2 = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), reg_w_l1=-1.0)
    out['sent_T_W'], out['sent_T_T'] = s1['W'][:2, :2], s1['T'][:2, :2]
    out['sent_W_W'], out['sent_W_T'] = s2['W'][:2, :2], s2['T'][:2, :2]
    # (f) each-iter W projection + stop rule on synthetic data
    soln = rnmf(Xn, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=40, eps_stop=1e-3,
                project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0,
                project_W_each_iter=True, compute_obj_each_iter=True)
    out['stop_W'], out['stop_T'] = soln['W'], soln['T']
    out['stop_obj'] = np.array(soln['obj_history'])
    save('g6_rare_branches', **out)


# ---------------------------------------------------------------- G7
def g7():
    qf = ref.optimization.qf_min
    proj = ref.matrixops.euclidean_proj_simplex
    rs = np.random.RandomState(7)
    out = {}
    vecs = {
        'rand': rs.randn(37), 'pos': rs.rand(50) * 3, 'zeros': np.zeros(9),
        'onsimplex': np.array([0.25, 0.25, 0.5, 0.0]), 'ties': np.array([0.5, 0.5, 0.5, 0.2, 0.2]),
        'single': np.array([3.0]), 'neg': -rs.rand(11), 'big': rs.rand(1000) * 10,
    }
    for nm, v in vecs.items():
        out['proj_in_' + nm] = v
        for s in (1.0, 2.5):
            out['proj_out_%s_s%g' % (nm, s)] = proj(v.copy(), s)
    w = rs.randn(40)
    cvec = rs.randn(40)
    cpos = np.abs(cvec) + 0.1
    out['qf_w'], out['qf_cvec'], out['qf_cpos'] = w, cvec, cpos
    calls = {
        'scalar_pos_s1': (w, 0.7, 1.0, 1.0), 'scalar_pos_sNone': (w, 0.7, None, 1.0),
        'scalar_pos_sNone_ubNone': (w, 0.7, None, None), 'scalar_pos_s2': (w, 0.7, 2.0, 1.0),
        'scalar_neg_sNone_ub': (w, -0.3, None, 0.8), 'scalar_zero_sNone_ub': (w, 0.0, None, 0.8),
        'scalar_neg_s1': (w, -0.3, 1.0, 1.0),
        'vec_pos_ub1': (w, cpos, None, 1.0), 'vec_pos_ubNone': (w, cpos, None, None),
        'vec_mixed_ub1': (w, cvec, None, 1.0), 'vec_pos_s1': (w, cpos, 1.0, 1.0),
    }
    for nm, (ww, c, s, ub) in calls.items():
        x, nx = qf(ww.copy(), c.copy() if hasattr(c, 'copy') else c, s=s, ub=ub)
        out['qf_x_' + nm], out['qf_nx_' + nm] = x, np.array(nx)
    errs = {}
    for nm, (ww, c, s, ub) in {'scalar_neg_unb': (w, -0.3, None, None),
                               'vec_neg_unb': (w, cvec, None, None),
                               'scalar_neg_s2': (w, -0.3, 2.0, 3.0)}.items():
        try:
            qf(ww.copy(), c, s=s, ub=ub)
            errs[nm] = 'none'
        except Exception as e:  # noqa: BLE001
            errs[nm] = type(e).__name__
    out['qf_errors'] = np.array(sorted(errs.items()))
    # stop rule
    usc = ref.optimization.universal_stopping_condition
    hist = [10.0, 8.0, 7.5, 7.4999]
    out['usc'] = np.array([usc(hist[:1]), usc(hist[:2]), usc(hist[:3]), usc(hist), usc(hist, -1)])
    # objective
    X = planted_X(60, 30, 4, seed=5, dtype=np.float64)
    W0, T0 = scaled_init(X, 4, seed=6)
    M = (rs.rand(60, 30) < 0.4).astype(np.float64)
    O = ref.nmf.TrueObjComputer(X, W0, T0, 0.1, 0.2, 0.3, 0.4, None, None)
    O2 = ref.nmf.TrueObjComputer(X, W0, T0, 0.1, 0.2, 0.3, 0.4, M, None)
    out['obj_plain'], out['obj_masked'] = np.array(O.true_objective()), np.array(O2.true_objective())
    out['obj_M'] = M
    # preprocessing
    Xt = fixture('text_data_train')
    out['tfidf_norm_checksum'] = np.array([tm_xform(Xt).sum(), (tm_xform(Xt) ** 2).sum()])
    save('g7_functions', **out)


# ---------------------------------------------------------------- G8
def g8():
    # tests/conftest.py:8-19 + tests/test_nmf.py:13-19 (byte literals decoded latin-1)
    X = np.array([[1, 0], [0.5, 0.5], [0.25, 0.75]])
    W, T = ref.initialization.initialize_nmf(X, 2, init='nndsvd', random_state=0)
    Wt = np.frombuffer(
        ('\xb9X\x18pb\xbd\xe8?\x00\x00\x00\x00\x00\x00\x00\x00\x114#('
         'e\x8c\xe3?%\x86\x8c"D\x08\xcd?\xbd\xa1('
         '\x84\xe6\xf3\xe0?\xbc\xad\x84\xb3f\xec\xe4?').encode('latin-1')).reshape(3, 2)
    Tt = np.frombuffer(
        ('\x04\x89=\x03\x95\xf6\xee?v)\xdfe\xf9\xf7\xe1?\x00\x00\x00\x00'
         '\x00\x00\x00\x00l\x8d.\xd8\x84%\xe6?').encode('latin-1')).reshape(2, 2)
    assert np.allclose(Wt, W) and np.allclose(Tt, T), 'reference test_init does not reproduce'
    save('g8_init', X=X, W_expected=Wt, T_expected=Tt, W_ref_here=W, T_ref_here=T)


# ---------------------------------------------------------------- G9
def g9():
    """Gaussian mechanism on the T-row sums (nmf.py:422-435), numpy's global RNG seeded before every call.
    `small` = noise far below the signal (sigma 0.016), `large` = sigma 16: denominators clipped at 0 and the
    c <= 0 branches of qf_min on most steps."""
    n, d, k = 240, 90, 4
    X = planted_X(n, d, k, seed=5, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=6)
    rs = np.random.RandomState(7)
    M = (rs.rand(n, d) < 0.4).astype(np.float64)
    out = dict(shape=np.array([n, d, k]), seed=np.array([11]))
    for tag, eps_g in (('small', 1e5), ('large', 1e2)):
        np.random.seed(11)
        soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=3, eps_stop=-1, t_row_sum=1.0,
                    eps_gauss_t=eps_g, delta_gauss_t=0.1)
        out['plain_%s_W' % tag], out['plain_%s_T' % tag] = soln['W'], soln['T']
        np.random.seed(11)
        soln = rnmf(M * X, k, W_in=copy_in(W0), T_in=copy_in(T0), W_mat=M, max_iter=3, eps_stop=-1, t_row_sum=1.0,
                    reset_topic_method=None, eps_gauss_t=eps_g, delta_gauss_t=0.1)
        out['weighted_%s_W' % tag], out['weighted_%s_T' % tag] = soln['W'], soln['T']
    np.random.seed(11)
    soln = rnmf(X, k, W_in=copy_in(W0), T_in=copy_in(T0), max_iter=3, eps_stop=-1, t_row_sum=1.0,
                project_T_each_iter=True, w_row_sum=1.0, eps_gauss_t=1e5, delta_gauss_t=0.1)
    out['tm_small_W'], out['tm_small_T'] = soln['W'], soln['T']
    save('g9_gaussian_mechanism', **out)


# ---------------------------------------------------------------- G10
def g10():
    """init_coherence_beam_search (initialization.py:166-208): the stand-alone topic initialiser (not reachable
    through initialize_nmf's `init` names), on seeded term counts and on the first rows of the text fixture"""
    rs = np.random.RandomState(21)
    Xa = rs.poisson(0.6, size=(120, 40)).astype(np.float64)
    Xa[:, 7] *= 3
    Wa, Ta = ref.initialization.init_coherence_beam_search(Xa.copy(), 3, n_words_beam=6)
    Xb = fixture('text_data_train')[:60, :80].astype(np.float64)
    Xb = Xb[:, Xb.sum(0) > 0]
    Wb, Tb = ref.initialization.init_coherence_beam_search(Xb.copy(), 4, n_words_beam=5)
    save('g10_coherence_init', Xa=Xa, Wa=Wa, Ta=Ta, Xb=Xb, Wb=Wb, Tb=Tb)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g3', 'g4', 'g5', 'g6', 'g7', 'g8', 'g9', 'g10']
    for g in which:
        {'g1': g1_g2, 'g3': g3, 'g4': g4, 'g5': g5, 'g6': g6, 'g7': g7, 'g8': g8, 'g9': g9, 'g10': g10}[g]()
