"""Seeded random differential test: nmf() on the device against the CPU oracle over random shapes, flag sets, storage
types and weightings.  Same outcome = the same exception type, or W, T within tolerance (float64 storage: the
iteration's own rounding sensitivity; fp32 storage: the stored residual's rounding)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import relfro

pytestmark = pytest.mark.gpu


def _case(seed):
    rs = np.random.RandomState(seed)
    hi = int(__import__('os').environ.get('RRI_FUZZ_MAX_DIM', '260'))     # 260 in the suite; larger once in a while
    n, d = int(rs.randint(2, hi)), int(rs.randint(2, hi))
    k = int(rs.choice([1, 2, 3, 5, 8, 17, 33, 70]))
    k = max(1, min(k, 256))
    X = rs.rand(n, d) * (rs.rand(n, d) < rs.choice([0.2, 0.7, 1.0]))
    weighted = rs.choice(['no', 'dense', 'sparse'])
    M = None
    if weighted != 'no':
        M = (rs.rand(n, d) < rs.choice([0.15, 0.5])).astype(np.float64)
        X = X * M
    kw = dict(max_iter=int(rs.randint(1, 4)), eps_stop=-1)
    # the weighted flavour always with the clip the recommender estimator sets: without it a reset can leave T rows of
    # 1e-15 and W entries that are rounding noise divided by eps (seed 32 once: the oracle and two device schedules,
    # each right, 1e-4 apart)
    if rs.rand() < 0.5 or weighted != 'no':
        kw['t_row_sum'] = float(rs.choice([1.0, 2.0]))
        if rs.rand() < 0.5 and weighted == 'no':
            kw['project_T_each_iter'] = True
    if rs.rand() < 0.4:
        kw['w_row_sum'] = float(rs.choice([1.0, 3.0]))
        if rs.rand() < 0.3:
            kw['project_W_each_iter'] = True
    if rs.rand() < 0.3:
        kw.update(reg_w_l1=float(rs.choice([0.0, 0.01])), reg_t_l1=float(rs.choice([0.0, 0.02])),
                  reg_w_l2=float(rs.choice([0.0, 0.1])), reg_t_l2=float(rs.choice([0.0, 0.05, -0.01])))
    r = rs.rand()
    if r < 0.12:
        kw['fix_T'] = True
    elif r < 0.24:
        kw['fix_W'] = True
    # resets in the weighted flavour only as the recommender estimator has them (none): a reset leaves a topic that
    # is barely determined, and runs with 20+ resets (seeds 0, 32 once) end 1e-5 apart on rounding noise alone --
    # enough for a column to die in one run and not in the other
    kw['reset_topic_method'] = None if weighted != 'no' else [None, 'max_resid_document'][int(rs.rand() < 0.6)]
    W0, T0 = rs.rand(n, k) + 0.01, rs.rand(k, d) + 0.01
    if rs.rand() < 0.15:
        W0[:, int(rs.randint(k))] = 0.0               # a dead column at the start
    store = rs.choice(['f64', 'f64', 'f32'])
    return X, M, weighted, k, W0, T0, kw, store


def _outcome(fn):
    try:
        return 'ok', fn()
    except (ValueError, AssertionError, NotImplementedError) as e:
        return type(e).__name__, str(e)


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('RRI_FUZZ_CASES', '80'))))
def test_random_case_matches_the_oracle(seed):
    from rri_nmf_amd import nmf as nmf_mod
    from oracle import rri_oracle as orc
    X, M, weighted, k, W0, T0, kw, store = _case(seed)
    dt = np.float64 if store == 'f64' else np.float32
    Xs = X.astype(dt)
    if weighted == 'sparse':
        args = dict(X=sp.csr_matrix(Xs), W_mat=sp.csr_matrix(M))
    elif weighted == 'dense':
        args = dict(X=Xs, W_mat=M.astype(dt))
    else:
        args = dict(X=Xs, W_mat=None)
    a = _outcome(lambda: nmf_mod.nmf(args['X'], k, W_mat=args['W_mat'], W_in=W0, T_in=T0, dtype=dt, **kw))
    b = _outcome(lambda: orc.nmf(Xs.astype(np.float64), k, W_mat=M, W_in=W0.copy(), T_in=T0.copy(), **kw))
    assert a[0] == b[0], (a[0], b[0], a[1] if a[0] != 'ok' else '', b[1] if b[0] != 'ok' else '', kw, weighted, store)
    if a[0] == 'ok':
        # fp32 storage of a maintained residual (weighted) rounds it at every update; everything else is float64
        tol = 1e-7 if store == 'f64' else (5e-3 if weighted != 'no' else 1e-7)
        if X.shape[0] * X.shape[1] > 260 * 260:
            # larger problems amplify rounding more: the yardstick is the oracle against itself with
            # its start perturbed by one ulp
            Wp = W0 * (1.0 + 2.0 ** -52 * np.sign(np.random.RandomState(7).randn(*W0.shape)))
            c = orc.nmf(Xs.astype(np.float64), k, W_mat=M, W_in=Wp, T_in=T0.copy(), **kw)
            tol = max(tol, 30.0 * max(relfro(c['W'], b[1]['W']), relfro(c['T'], b[1]['T'])))
        ew, et = relfro(a[1]['W'], b[1]['W']), relfro(a[1]['T'], b[1]['T'])
        assert ew < tol and et < tol, (ew, et, kw, weighted, store, X.shape, k)


@pytest.mark.parametrize('seed', range(1000, 1000 + int(__import__('os').environ.get('RRI_FUZZ_RESIDUAL_CASES', '60'))))
def test_random_case_on_the_explicit_residual_schedule(seed):
    """the same differential test for nmf(..., schedule='residual') on the unweighted cases: same exception or same factors as
    the oracle; an fp32 residual is rounded at every update (and rebuilt every sweep): the weighted flavour's bound.  Fixed
    halves and k = 1 run on such a handle too (round 3: stepped in the Gram form, the residual is not touched)"""
    from rri_nmf_amd import nmf as nmf_mod
    from oracle import rri_oracle as orc
    X, M, weighted, k, W0, T0, kw, store = _case(seed)
    if weighted != 'no':
        # refused -- unless the reference's own "unbounded objective" sentinel (nmf.py:292-315) comes first
        try:
            r = nmf_mod.nmf(X if weighted == 'no' else X * M, k, W_mat=M, W_in=W0, T_in=T0, schedule='residual', **kw)
        except NotImplementedError:
            return
        assert r['obj_history'] == [-np.inf], 'neither refused nor a sentinel return'
        return
    dt = np.float64 if store == 'f64' else np.float32
    Xs = X.astype(dt)
    a = _outcome(lambda: nmf_mod.nmf(Xs, k, W_in=W0, T_in=T0, dtype=dt, schedule='residual', **kw))
    b = _outcome(lambda: orc.nmf(Xs.astype(np.float64), k, W_in=W0.copy(), T_in=T0.copy(), **kw))
    assert a[0] == b[0], (a[0], b[0], a[1] if a[0] != 'ok' else '', b[1] if b[0] != 'ok' else '', kw, store)
    if a[0] == 'ok':
        gram_form = bool(kw.get('fix_W') or kw.get('fix_T') or k < 2)       # no stored residual in play: float64 throughout
        tol = 1e-7 if (store == 'f64' or gram_form) else 5e-3
        ew, et = relfro(a[1]['W'], b[1]['W']), relfro(a[1]['T'], b[1]['T'])
        assert ew < tol and et < tol, (ew, et, kw, store, X.shape, k)
