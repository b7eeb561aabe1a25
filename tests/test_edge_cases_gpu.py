"""Differential test on awkward inputs: nmf() on the device and the CPU oracle must agree on the OUTCOME --
either the same exception type (the reference's error conventions) or W, T within tolerance."""
import numpy as np
import pytest

from conftest import relfro

pytestmark = pytest.mark.gpu


def outcome(fn):
    try:
        r = fn()
        return 'ok', r
    except (ValueError, AssertionError, NotImplementedError) as e:
        return type(e).__name__, str(e)


def both(X, k, W0, T0, **kw):
    from rri_nmf_amd import nmf as nmf_mod
    from oracle import rri_oracle as orc
    a = outcome(lambda: nmf_mod.nmf(X, k, W_in=W0, T_in=T0, eps_stop=-1, **kw))
    b = outcome(lambda: orc.nmf(np.asarray(X, dtype=np.float64), k, W_in=np.array(W0, dtype=np.float64),
                                T_in=np.array(T0, dtype=np.float64), eps_stop=-1, **kw))
    return a, b


def agree(a, b, tol=1e-8):
    assert a[0] == b[0], (a[0], b[0], a[1] if a[0] != 'ok' else '', b[1] if b[0] != 'ok' else '')
    if a[0] == 'ok':
        assert relfro(a[1]['W'], b[1]['W']) < tol and relfro(a[1]['T'], b[1]['T']) < tol, \
            (relfro(a[1]['W'], b[1]['W']), relfro(a[1]['T'], b[1]['T']))
        assert a[1]['W'].shape == b[1]['W'].shape and a[1]['T'].shape == b[1]['T'].shape


def rnd(seed, *shape):
    return np.random.RandomState(seed).rand(*shape)


CASES = {
    'one_by_one': lambda: (rnd(0, 1, 1) + 0.5, 1, rnd(1, 1, 1) + 0.1, rnd(2, 1, 1) + 0.1, dict(max_iter=3)),
    'k_ge_d': lambda: (rnd(0, 7, 2) + 0.1, 3, rnd(1, 7, 3), rnd(2, 3, 2), dict(max_iter=4)),
    'n_le_k': lambda: (rnd(0, 3, 9) + 0.1, 3, rnd(1, 3, 3), rnd(2, 3, 9), dict(max_iter=4)),
    'all_zero_X': lambda: (np.zeros((20, 12)), 3, rnd(1, 20, 3), rnd(2, 3, 12), dict(max_iter=2)),
    'all_zero_X_bounded': lambda: (np.zeros((20, 12)), 3, rnd(1, 20, 3), rnd(2, 3, 12),
                                   dict(max_iter=2, t_row_sum=1.0, w_row_sum=1.0, do_final_project_W=False)),
    'empty_rows_and_cols': lambda: ((lambda X: (X.__setitem__((slice(3, 9), slice(None)), 0), X.__setitem__((slice(None), slice(10, 20)), 0), X)[2])(rnd(0, 40, 30)),
                                    4, rnd(1, 40, 4), rnd(2, 4, 30), dict(max_iter=5)),
    'k1_topic_model': lambda: (rnd(0, 50, 33), 1, rnd(1, 50, 1), rnd(2, 1, 33),
                               dict(max_iter=4, project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
    'k64_small_n': lambda: (rnd(0, 70, 90), 64, rnd(1, 70, 64) * 0.2, rnd(2, 64, 90) * 0.2, dict(max_iter=3)),
    'per_row_w_sums': lambda: (rnd(0, 60, 25), 3, rnd(1, 60, 3), rnd(2, 3, 25),
                               dict(max_iter=3, w_row_sum=rnd(3, 60) + 0.5, project_W_each_iter=True,
                                    project_T_each_iter=True, t_row_sum=1.0)),
    'simplex_radius_2': lambda: (rnd(0, 60, 25), 3, rnd(1, 60, 3), rnd(2, 3, 25),
                                 dict(max_iter=4, project_T_each_iter=True, t_row_sum=2.0, w_row_sum=3.0)),
    'negative_l2_with_bounds': lambda: (rnd(0, 60, 25), 3, rnd(1, 60, 3), rnd(2, 3, 25),
                                        dict(max_iter=3, project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0,
                                             reg_t_l2=-0.05, reg_w_l2=-0.01, project_W_each_iter=True)),
    'c_le_0_radius_not_1': lambda: (rnd(0, 60, 25), 3, rnd(1, 60, 3), rnd(2, 3, 25),
                                    dict(max_iter=2, project_T_each_iter=True, t_row_sum=2.0, w_row_sum=1.0,
                                         reg_t_l2=-500.0)),
    'weighted_tiny': lambda: (rnd(0, 9, 7) * (rnd(5, 9, 7) < 0.5), 2, rnd(1, 9, 2), rnd(2, 2, 7),
                              dict(max_iter=3, W_mat=(rnd(5, 9, 7) < 0.5).astype(float), t_row_sum=1.0,
                                   reset_topic_method=None)),
    'weighted_unbounded': lambda: (rnd(0, 30, 20), 3, rnd(1, 30, 3), rnd(2, 3, 20),
                                   dict(max_iter=2, W_mat=(rnd(5, 30, 20) < 0.5).astype(float), reg_t_l2=-5.0,
                                        t_row_sum=None, project_T_each_iter=True)),
    'weighted_T_row_resets': lambda: (rnd(0, 60, 40) * (rnd(5, 60, 40) < 0.5), 3, rnd(1, 60, 3), rnd(2, 3, 40),
                                      dict(max_iter=2, W_mat=(rnd(5, 60, 40) < 0.5).astype(float), t_row_sum=1.0,
                                           reg_t_l1=1e6, reset_topic_method='max_resid_document')),
    'weighted_W_col_resets': lambda: (rnd(0, 60, 40) * (rnd(5, 60, 40) < 0.5), 3, rnd(1, 60, 3), rnd(2, 3, 40),
                                      dict(max_iter=2, W_mat=(rnd(5, 60, 40) < 0.5).astype(float), t_row_sum=1.0,
                                           reg_w_l1=1e6, reset_topic_method='max_resid_document')),
    'k300_past_the_resident_W_tile': lambda: (rnd(0, 420, 350), 300, rnd(1, 420, 300) * 0.1, rnd(2, 300, 350) * 0.1,
                                              dict(max_iter=2, compute_obj_each_iter=True)),
    'k1000_topic_model': lambda: (rnd(0, 90, 1100), 1000, rnd(1, 90, 1000) * 0.05, rnd(2, 1000, 1100) * 0.05,
                                  dict(max_iter=1, project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
    'k300_weighted': lambda: (rnd(0, 350, 200) * (rnd(5, 350, 200) < 0.5), 300, rnd(1, 350, 300) * 0.1, rnd(2, 300, 200) * 0.1,
                              dict(max_iter=1, W_mat=(rnd(5, 350, 200) < 0.5).astype(float), t_row_sum=1.0,
                                   reset_topic_method=None)),
    'fp32_input_ragged_d': lambda: (rnd(0, 130, 1027).astype(np.float32), 5, rnd(1, 130, 5), rnd(2, 5, 1027),
                                    dict(max_iter=3)),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_same_outcome_as_oracle(name):
    X, k, W0, T0, kw = CASES[name]()
    a, b = both(X, k, W0, T0, **kw)
    agree(a, b)


def test_ragged_d_weighted_on_recycled_memory():
    """d not a multiple of the 16-byte vector: the pad column of the maintained residual is streamed by the passes
    and must hold zeros whatever the allocator hands out (a NaN pattern there once zeroed rows of W)"""
    import torch
    from rri_nmf_amd.engine import RRIEngine
    from oracle import rri_oracle as orc
    n, d, k = 412, 187, 5
    rs = np.random.RandomState(0)
    M = (rs.rand(n, d) < 0.2).astype(np.float64)
    X = rs.rand(n, d) * M
    W0, T0 = rnd(1, n, k), rnd(2, k, d)
    kw = dict(t_row_sum=1.0, reset_topic_method=None)
    ref = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), W_mat=M, max_iter=2, eps_stop=-1, **kw)
    for store in (np.float64, np.float32):
        for _ in range(2):
            junk = torch.full((n * (d + 8),), float('nan'), dtype=torch.float64, device='cuda')   # poison, then free
            del junk
            torch.cuda.empty_cache()
            with RRIEngine(n, d, k, dtype=store, weighted=True) as e:
                e.upload_X(X.astype(store)), e.upload_mask(M.astype(store))
                e.set_W(W0), e.set_T(T0)
                e.set_params(**kw)
                e.sweep(2)
                tol = 1e-9 if store == np.float64 else 1e-4
                assert relfro(e.get_W(), ref['W']) < tol and relfro(e.get_T(), ref['T']) < tol
