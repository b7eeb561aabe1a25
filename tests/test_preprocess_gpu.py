"""tf-idf / row normalisation of the resident X on the device against matrixops.tfidf /
normalize (reference matrixops.py:124-179), and the `preprocess` option of nmf() / the TM estimator's handle_* flags
against the same factorisation of a host-preprocessed X."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import relfro

pytestmark = pytest.mark.gpu


def counts(n, d, seed, empty_rows=(), empty_cols=()):
    rng = np.random.RandomState(seed)
    X = rng.poisson(0.7, size=(n, d)).astype(np.float64)
    X[list(empty_rows), :] = 0
    X[:, list(empty_cols)] = 0
    return X


def resident(eng):
    return eng.X_times(np.eye(eng.d))


@pytest.mark.parametrize('store,tol', [(np.float64, 1e-15), (np.float32, 2e-7)])
@pytest.mark.parametrize('n,d', [(57, 33), (300, 257), (1000, 64)])
def test_device_tfidf_and_normalize_match_matrixops(n, d, store, tol):
    from rri_nmf_amd.engine import RRIEngine
    from rri_nmf_amd.matrixops import tfidf, normalize
    X = counts(n, d, n + d, empty_rows=(3, n - 1), empty_cols=(0, d - 2))
    Xt, idf = tfidf(X, return_idf=True)
    want = {'tfidf': Xt, 'normalize': normalize(X), 'both': normalize(Xt)}
    for which, ref in want.items():
        with RRIEngine(n, d, 3, dtype=store) as eng:
            eng.upload_X(X.astype(store))
            if which != 'normalize':
                assert np.array_equal(eng.column_positive_counts(), (X > 0).sum(0))
            got_idf = eng.preprocess(tfidf=which != 'normalize', normalize=which != 'tfidf')
            got = resident(eng)
        if which != 'normalize':
            assert np.array_equal(got_idf, np.asarray(idf).ravel())
        assert np.max(np.abs(got - ref)) <= tol * max(1.0, np.max(np.abs(ref))), which
        if which != 'tfidf':      # empty documents become uniform (normalize's zero-sum fix)
            assert np.allclose(got[3], 1.0 / d, rtol=tol * 10) and np.allclose(got.sum(1), 1.0, atol=1e-5)


def test_given_idf_is_applied_to_new_documents():
    from rri_nmf_amd.engine import RRIEngine
    from rri_nmf_amd.matrixops import normalize
    X = counts(40, 21, 5)
    idf = np.linspace(0.1, 3.0, 21)
    with RRIEngine(40, 21, 2, dtype=np.float64) as eng:
        eng.upload_X(X)
        assert np.array_equal(eng.preprocess(tfidf=idf, normalize=True), idf)
        assert np.max(np.abs(resident(eng) - normalize(X * idf))) < 1e-15


def test_scaling_is_refused_where_it_cannot_apply():
    import torch
    from rri_nmf_amd.engine import RRIEngine
    X = counts(64, 32, 1)
    with RRIEngine(64, 32, 2, dtype=np.float64, weighted=True) as eng:
        eng.upload_X(X)
        eng.upload_mask((X > 0).astype(np.float64))
        with pytest.raises(ValueError):
            eng.scale_X(np.ones(32))
    with RRIEngine(64, 32, 2, dtype=np.float64) as eng:
        xt = torch.as_tensor(X, device='cuda')
        torch.cuda.synchronize()
        eng.bind_X_device(xt.data_ptr(), xt.stride(0))
        with pytest.raises(ValueError):          # caller-owned memory is never rewritten
            eng.scale_X(np.ones(32))
        with pytest.raises(ValueError):
            eng.scale_X(np.ones(31))


@pytest.mark.parametrize('init', ['nndsvd', 'random', 'smart_random'])
def test_nmf_preprocess_option_matches_host_preprocessing(init):
    from rri_nmf_amd import nmf as nmf_mod
    from rri_nmf_amd.matrixops import tfidf, normalize
    X = counts(240, 150, 9, empty_rows=(7,))
    Xt, idf = tfidf(X, return_idf=True)
    Xh = normalize(Xt)
    kw = dict(max_iter=6, random_state=0, init=init, project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0,
              compute_obj_each_iter=True)
    a = nmf_mod.nmf(Xh, 4, **kw)
    b = nmf_mod.nmf(X, 4, preprocess={'tfidf': True, 'normalize': True}, **kw)
    assert np.array_equal(b['idf'], np.asarray(idf).ravel())
    assert relfro(b['W'], a['W']) < 1e-8 and relfro(b['T'], a['T']) < 1e-8
    assert np.allclose(b['obj_history'], a['obj_history'], rtol=1e-10)
    # the re-evaluation (after the final projection of W) applies the same preprocessing to the raw X again
    want = a['obj_calculator'].true_objective()
    assert abs(b['obj_calculator'].true_objective() - want) <= 1e-9 * abs(want)
    # step names instead of the dict; sparse input and host callbacks take the host route to the same result
    c = nmf_mod.nmf(sp.csr_matrix(X), 4, preprocess=('tfidf', 'normalize'), **kw)
    assert relfro(c['W'], a['W']) < 1e-8 and np.allclose(c['idf'], b['idf'])
    seen = []
    e = nmf_mod.nmf(X, 4, preprocess='normalize', diagnostics=[lambda Xd, W, T: seen.append(Xd.sum())],
                    **dict(kw, max_iter=2))
    assert e['idf'] is None and abs(seen[0] - X.shape[0]) < 1e-9      # the callback saw the normalised X
    with pytest.raises(ValueError):
        nmf_mod.nmf(X, 4, preprocess=('tfidf', 'whiten'), **kw)


def test_tm_estimator_flags_use_the_option():
    from rri_nmf_amd import sklearn_interface as si
    from rri_nmf_amd.matrixops import tfidf, normalize
    X = counts(200, 120, 11)
    Xte = counts(50, 120, 12)
    Xt, idf = tfidf(X, return_idf=True)
    plain = si.NMF_TM_Estimator(200, 120, 4, random_state=0, max_iter=8).fit(normalize(Xt))
    flagged = si.NMF_TM_Estimator(200, 120, 4, random_state=0, max_iter=8, handle_tfidf=True,
                                  handle_normalization=True).fit(X)
    assert np.array_equal(flagged.idf, np.asarray(idf).ravel())
    assert relfro(flagged.W, plain.W) < 1e-8 and relfro(flagged.T, plain.T) < 1e-8
    assert relfro(flagged.transform(Xte), plain.transform(normalize(Xte * idf))) < 1e-8
    only_norm = si.NMF_TM_Estimator(200, 120, 4, random_state=0, max_iter=3, handle_normalization=True).fit(X)
    ref = si.NMF_TM_Estimator(200, 120, 4, random_state=0, max_iter=3).fit(normalize(X))
    assert relfro(only_norm.W, ref.W) < 1e-8 and not hasattr(only_norm, 'idf')
