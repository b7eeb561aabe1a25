"""GPU parity tests of the HIP RRI path, called through the C ABI (RRIEngine -> librri_hip.so),
against (i) golden vectors captured from the unmodified reference and (ii) the CPU oracle on the
same seeded inputs.

`dtype` is the STORAGE type of X in HBM (fp32 or fp64).  The arithmetic of the device path is float64
in both cases (X elements are converted as they stream through the pass kernel), so for an fp32 X the
reference is the oracle run on the same fp32-valued X upcast to float64, exactly as BASELINE.md's CPU
baseline does.

Tolerance (relative Frobenius distance to the float64 reference result), both storage types:
    2e-9                   same algorithm, different summation order.  The iteration itself amplifies
                           rounding: perturbing the oracle's start by 8e-16 (relative) moves its own
                           W by 2.7e-11 after 5 sweeps on the 2000x300 k=20 case (max(.,0) switching of
                           near-zero entries), so agreement much below 1e-10 is not a property even of
                           two runs of the reference with different BLAS threading.
BASELINE.json's bar for the fp32 configuration is 1e-4; an all-fp32 arithmetic (numpy sgemv) misses it
on these inputs (4e-3 at 10k x 1k k=20 after 20 sweeps), which is why the kernels accumulate in float64.
"""
import numpy as np
import pytest

from conftest import load_golden, relfro
from rri_nmf_amd.synthetic import planted_X, scaled_init

pytestmark = pytest.mark.gpu

TOL = {np.float64: 2e-9, np.float32: 2e-9}


def engine(*a, **kw):
    from rri_nmf_amd.engine import RRIEngine
    return RRIEngine(*a, **kw)


def oracle():
    from oracle import rri_oracle
    return rri_oracle


def stored(X, dtype):
    """X as the device holds it (rounded to the storage type), in float64 for the oracle"""
    return np.ascontiguousarray(X.astype(dtype).astype(np.float64))


def run_oracle(X, W0, T0, sweeps, **kw):
    k = W0.shape[1]
    return oracle().nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=sweeps, eps_stop=-1, **kw)


def run_engine(X, W0, T0, sweeps, dtype, final_proj=None, **params):
    n, d = X.shape
    k = W0.shape[1]
    with engine(n, d, k, dtype=dtype) as e:
        e.upload_X(X)
        e.set_W(np.maximum(W0, 0))
        e.set_T(np.maximum(T0, 0))
        e.set_params(**params)
        e.sweep(sweeps)
        if final_proj is not None:
            e.project_W_rows(final_proj)
        return e.get_W(), e.get_T(), e.n_resets_used


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_plain_flavour_vs_reference_vectors(tag, dtype):
    g = load_golden('g5_plain_' + tag)
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    Xs = stored(X, dtype)
    for S in (1, 5, 30):
        W, T, _ = run_engine(X, W0, T0, S, dtype)
        ref = run_oracle(Xs, W0, T0, S)
        if dtype == np.float64:   # oracle == reference's vectors (bit-identical in the build container;
            # another host's BLAS sums in another order, so only to the iteration's own sensitivity here)
            assert relfro(ref['W'], g['W_s%d' % S]) < TOL[dtype] and relfro(ref['T'], g['T_s%d' % S]) < TOL[dtype]
            assert relfro(W, g['W_s%d' % S]) < TOL[dtype] and relfro(T, g['T_s%d' % S]) < TOL[dtype]
        assert relfro(W, ref['W']) < TOL[dtype], (S, relfro(W, ref['W']))
        assert relfro(T, ref['T']) < TOL[dtype], (S, relfro(T, ref['T']))


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_sweeps_are_resumable(dtype):
    """30 sweeps in one call == 30 calls of one sweep (carry across calls is exact)."""
    g = load_golden('g5_plain_a')
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    with engine(n, d, k, dtype=dtype) as e:
        e.upload_X(X); e.set_W(W0); e.set_T(T0); e.set_params()
        for _ in range(30):
            e.sweep(1)
        W1, T1 = e.get_W(), e.get_T()
    W2, T2, _ = run_engine(X, W0, T0, 30, dtype)
    assert np.array_equal(W1, W2) and np.array_equal(T1, T2)
    assert relfro(W1, run_oracle(stored(X, dtype), W0, T0, 30)['W']) < TOL[dtype]


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_topic_model_flavour(tag, dtype):
    g = load_golden('g5_plain_' + tag)
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    orc = oracle()
    Xn = orc.normalize(X.copy())
    # nmf.py:875-878: T is projected once before the loop when project_T_each_iter
    T0p = orc.proj_rows_simplex(np.maximum(T0, 0).copy(), 1.0)
    tm = dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
    for S in (1, 5):
        W, T, _ = run_engine(Xn, W0, T0p, S, dtype, final_proj=1.0, **tm)
        ref = run_oracle(stored(Xn, dtype), W0, T0, S, **tm)
        if dtype == np.float64:
            assert relfro(W, g['tm_W_s%d' % S]) < TOL[dtype] and relfro(T, g['tm_T_s%d' % S]) < TOL[dtype]
        assert relfro(W, ref['W']) < TOL[dtype], relfro(W, ref['W'])
        assert relfro(T, ref['T']) < TOL[dtype], relfro(T, ref['T'])
        assert np.abs(T.sum(1) - 1).max() < 1e-12 and np.abs(W.sum(1) - 1).max() < 1e-12
        assert W.min() >= 0 and T.min() >= 0


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_regularised_and_fixed_halves(dtype):
    g = load_golden('g5_plain_a')
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    Xs = stored(X, dtype)
    regs = dict(reg_w_l1=0.01, reg_t_l1=0.02, reg_w_l2=0.05, reg_t_l2=0.03)
    W, T, _ = run_engine(X, W0, T0, 5, dtype, **regs)
    ref = run_oracle(Xs, W0, T0, 5, **regs)
    if dtype == np.float64:
        assert relfro(W, g['reg_W_s5']) < TOL[dtype] and relfro(T, g['reg_T_s5']) < TOL[dtype]
    assert relfro(W, ref['W']) < TOL[dtype] and relfro(T, ref['T']) < TOL[dtype]
    W, T, _ = run_engine(X, W0, T0, 3, dtype, fix_T=True)
    ref = run_oracle(Xs, W0, T0, 3, fix_T=True)
    assert relfro(W, ref['W']) < TOL[dtype] and np.array_equal(T, np.maximum(T0, 0))
    W, T, _ = run_engine(X, W0, T0, 3, dtype, fix_W=True)
    ref = run_oracle(Xs, W0, T0, 3, fix_W=True)
    if dtype == np.float64:
        assert relfro(W, g['fixW_W_s3']) < TOL[dtype] and relfro(T, g['fixW_T_s3']) < TOL[dtype]
    assert relfro(W, ref['W']) < TOL[dtype] and relfro(T, ref['T']) < TOL[dtype]


@pytest.mark.parametrize('shape', [(700, 333, 6), (65, 40, 1), (5003, 257, 64), (1200, 90, 80), (64, 30, 3)])
def test_fixed_T_whole_sweep_launch_matches_the_launch_per_topic_schedule(monkeypatch, shape):
    """k_wsweep_rows (round 4): with T fixed the W half of every topic of a sweep is ONE launch -- row i of W needs only its own
    entries, (X T^T)[i, :] and T T^T -- and the column checks are taken in topic order afterwards (k_wsweep_verdict).  Against
    the launch-per-topic schedule (RRI_WSWEEP=0: k_tgram, k_wcol, k_check_wcol per topic) and the oracle: plain, penalties, the
    bound w_row_sum, the non-positive-denominator branch of qf_min, both storage types, ragged row counts (not a multiple of 64),
    k = 1 and k beyond 64; sweeps in one call and call by call."""
    n, d, k = shape
    X = planted_X(n, d, max(k, 2), seed=n + k, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=5)
    monkeypatch.setenv('RRI_ONCHIP', '0')

    def run(sw, store, sweeps=3, **flags):
        monkeypatch.setenv('RRI_WSWEEP', sw)
        # (nmf()'s final projection of the rows of W when w_row_sum is given, nmf.py:519-529)
        return run_engine(X, W0, T0, sweeps, store, final_proj=flags.get('w_row_sum'), fix_T=True, **flags)

    for store in (np.float64, np.float32):
        for flags in (dict(), dict(reg_w_l1=0.02, reg_w_l2=0.05), dict(t_row_sum=1.0, w_row_sum=1.0),
                      dict(w_row_sum=0.7, reg_w_l2=-1e9)):       # the last: denom <= 0, entries jump to the bound or to 0
            a, b = run('1', store, **flags), run('0', store, **flags)
            # T: untouched -- but for the row a reset rewrites from the residual (k = 80 empties a column of W: nmf.py:804-810)
            assert relfro(a[1], b[1]) < 1e-12 and relfro(a[0], b[0]) < 1e-12, (store, flags, relfro(a[0], b[0]), relfro(a[1], b[1]))
            ref = run_oracle(stored(X, store), W0, T0, 3, fix_T=True, **flags)
            assert relfro(a[0], ref['W']) < TOL[store], (store, flags, relfro(a[0], ref['W']))
    # call by call == one call; the objective after the sweeps comes from the cross terms the launch left (no pass over X)
    with engine(n, d, k, dtype=np.float64) as e:
        e.upload_X(X); e.set_W(W0); e.set_T(T0); e.set_params(fix_T=True)
        for _ in range(3):
            e.sweep(1)
        W1, o1 = e.get_W(), e.objective()
    with engine(n, d, k, dtype=np.float64) as e:
        e.upload_X(X); e.set_W(W0); e.set_T(T0); e.set_params(fix_T=True)
        e.sweep(3)
        W3, T3, o3 = e.get_W(), e.get_T(), e.objective()
    assert np.array_equal(W1, W3) and o1 == o3
    # (T3, not T0: at k = 80 a column of W empties and the default reset rewrites its row of T, fixed or not -- nmf.py:804-810)
    assert abs(o3 - 0.5 * np.linalg.norm(X - W3 @ T3) ** 2) <= 1e-9 * abs(o3)


def test_round4_launches_give_the_same_bits_run_to_run(monkeypatch):
    """Every sum of the round's new launches has a fixed order (no atomics in the data path): the one-pass weighted step on a
    sparse 0/1 mask (k_wmcorr_cols, k_wreduce), the whole-sweep W half with T fixed (k_wsweep_rows) and the rebuild with its
    column ranges (k_resid_mfma) return the same bits from one run to the next."""
    monkeypatch.setenv('RRI_ONCHIP', '0')
    n, d, k = 6000, 1300, 7
    X = planted_X(n, d, k, seed=11, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=12)
    M = (np.random.RandomState(2).rand(n, d) < 0.05).astype(np.float64)
    M[0, :] = 1.0
    runs = []
    for _ in range(3):
        a = run_weighted(stored(M * X, np.float32), M, W0, T0, 4, np.float32, t_row_sum=1.0, reset_topic_method=None)
        b = run_engine(X, W0, T0, 4, np.float32, fix_T=True, reg_w_l1=0.01)
        runs.append((a[0], a[1], a[2], b[0]))
    for r in runs[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(r[:2], runs[0][:2])) and r[2] == runs[0][2] and np.array_equal(r[3], runs[0][3])


def test_fixed_T_whole_sweep_launch_halts_where_the_reference_does(monkeypatch):
    """A column of W that the update empties (nmf.py:471-476, 793-816) in the middle of a whole-sweep launch: with a reset method
    the run pauses at that topic with the later columns as they were before the sweep (k_wsweep_repair), the reset is drawn and
    the sweep goes on from the next topic -- the same W, T and reset count as the launch-per-topic schedule and the oracle;
    without one, the reference's assertion."""
    n, d, k = 900, 120, 5
    X = planted_X(n, d, k, seed=3, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=4)
    X[:, :10] = 0.0                      # topic 2 lives on columns where X is empty: its column of W goes to zero
    T0[2, :] = 0.0
    T0[2, :10] = 1.0
    monkeypatch.setenv('RRI_ONCHIP', '0')
    out = {}
    for sw in ('1', '0'):
        monkeypatch.setenv('RRI_WSWEEP', sw)
        with engine(n, d, k, dtype=np.float64) as e:
            e.upload_X(X); e.set_W(W0); e.set_T(T0)
            e.set_params(fix_T=True, reset_topic_method='random', fix_reset_seed=True, n_resets=2)
            e.sweep(3)
            out[sw] = (e.get_W(), e.get_T(), e.n_resets_used, list(e.reset_log))
        with engine(n, d, k, dtype=np.float64) as e:
            e.upload_X(X); e.set_W(W0); e.set_T(T0)
            e.set_params(fix_T=True, reset_topic_method=None)
            with pytest.raises(AssertionError, match='sums to 0'):
                e.sweep(2)
    a, b = out['1'], out['0']
    assert a[2] == b[2] >= 1 and a[3] == b[3], (a[2], b[2], a[3], b[3])
    assert relfro(a[0], b[0]) < 1e-12 and relfro(a[1], b[1]) < 1e-12, (relfro(a[0], b[0]), relfro(a[1], b[1]))
    ref = run_oracle(X, W0, T0, 3, fix_T=True, reset_topic_method='random', fix_reset_seed=True, n_resets=2)
    assert relfro(a[0], ref['W']) < 2e-9 and relfro(a[1], ref['T']) < 2e-9, (relfro(a[0], ref['W']), relfro(a[1], ref['T']))


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_text_fixture_topic_assignments_are_exact(dtype):
    """BASELINE north_star: bit-exact argmax topic assignments on the reference's fixture."""
    g = load_golden('g1_tm_estimator')
    X, W0, T0 = g['X'], g['W0'], g['T0']
    tm = dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
    for S in (1, 2, 10):
        W, T, _ = run_engine(X, W0, T0, S, dtype, final_proj=1.0, **tm)
        ref = run_oracle(stored(X, dtype), W0, T0, S, **tm)
        assert relfro(W, ref['W']) < TOL[dtype] and relfro(T, ref['T']) < TOL[dtype]
        if dtype == np.float64:
            assert relfro(W, g['W_s%d' % S]) < TOL[dtype] and relfro(T, g['T_s%d' % S]) < TOL[dtype]
    # exact topic assignments against the REFERENCE's float64 run, for both storage types
    assert np.array_equal(np.argmax(W, 1), g['argmax_s10'])
    n, d = X.shape
    with engine(n, d, 5, dtype=dtype) as e:
        e.upload_X(X); e.set_W(W); e.set_T(T); e.set_params()
        assert np.array_equal(e.argmax_rows(), g['argmax_s10'])
    # fold-in of held-out documents (G2): fix_T, 4 sweeps, final projection
    Wte, Tte, nres = run_engine(g['Xte'], g['Wte0'], g['T_s10'], 4, dtype, final_proj=1.0, fix_T=True,
                                t_row_sum=1.0, w_row_sum=1.0)
    assert nres == 0 and np.array_equal(Tte, g['T_s10'])      # T untouched, no reset events
    assert relfro(Wte, g['Wte']) < (TOL[dtype] if dtype == np.float64 else 1e-6)
    assert np.array_equal(np.argmax(Wte, 1), g['argmax_te'])


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_objective_matches_oracle(dtype):
    orc = oracle()
    X = planted_X(700, 333, 6, seed=5, dtype=np.float64)
    W0, T0 = scaled_init(X, 6, seed=6)
    with engine(700, 333, 6, dtype=dtype) as e:
        e.upload_X(X); e.set_W(W0); e.set_T(T0)
        e.set_params(reg_w_l1=0.3, reg_w_l2=0.1, reg_t_l1=0.4, reg_t_l2=0.2)
        got = e.objective()
    want = orc.true_objective(X, W0, T0, 0.1, 0.2, 0.3, 0.4)
    want = orc.true_objective(stored(X, dtype), W0, T0, 0.1, 0.2, 0.3, 0.4)
    assert abs(got - want) <= 1e-11 * abs(want)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_rare_branches(dtype):
    g = load_golden('g6_rare_branches')
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=3, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=4)
    orc = oracle()
    Xn = orc.normalize(X.copy())
    T0p = orc.proj_rows_simplex(np.maximum(T0, 0).copy(), 1.0)
    tol = TOL[dtype] if dtype == np.float64 else 2e-5   # goldens were made with the float64 X
    # T side c<=0 -> one-hot rows (optimization.py:68-70)
    W, T, _ = run_engine(Xn, W0, T0p, 3, dtype, final_proj=1.0, project_T_each_iter=True, t_row_sum=1.0,
                         w_row_sum=1.0, reg_t_l2=-50.0)
    assert relfro(T, g['negT_T']) < tol and relfro(W, g['negT_W']) < tol
    # W side c<=0 -> entries at ub (optimization.py:62-65)
    W, T, _ = run_engine(Xn, W0, T0p, 2, dtype, project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0,
                         reg_w_l2=-5.0)
    assert relfro(T, g['negW_T']) < tol and relfro(W, g['negW_W']) < tol
    Wd = g['dead_W0']
    # dead column, no bound: unbounded (reference raises ValueError)
    with pytest.raises(ValueError, match='unbounded'):
        run_engine(X, Wd, T0, 2, dtype)
    # dead column with ub: reset to the max-residual document (on the device)
    W, T, nres = run_engine(X, Wd, T0, 2, dtype, t_row_sum=1.0)
    assert nres >= 1 and relfro(T, g['dead_mrd_T']) < tol and relfro(W, g['dead_mrd_W']) < tol
    # resets off / exhausted: the reference's assert
    with pytest.raises(AssertionError, match='sums to 0'):
        run_engine(X, Wd, T0, 2, dtype, t_row_sum=1.0, w_row_sum=1.0, reset_topic_method=None)
    with pytest.raises(AssertionError, match='sums to 0'):
        run_engine(X, Wd, T0, 2, dtype, t_row_sum=1.0, w_row_sum=1.0, n_resets=0)
    with pytest.raises(ValueError, match='unbounded'):
        run_engine(X, Wd, T0, 2, dtype, t_row_sum=1.0, reset_topic_method=None)
    # every T row killed by a huge l1 penalty: 6 resets in one sweep
    W, T, nres = run_engine(X, W0, T0, 1, dtype, t_row_sum=1.0, reg_t_l1=1e6)
    assert nres == k and relfro(T, g['l1kill_T']) < tol and relfro(W, g['l1kill_W']) < tol
    # W columns killed: max-residual resets from the W side
    W, T, nres = run_engine(X, W0, T0, 1, dtype, t_row_sum=1.0, reg_w_l1=1e6)
    assert nres == k and relfro(T, g['l1killW_mrd_T']) < tol and relfro(W, g['l1killW_mrd_W']) < tol
    # ... and 'random' resets with the reference's seeding (numpy global RNG on the host)
    W, T, nres = run_engine(X, W0, T0, 1, dtype, t_row_sum=1.0, reg_w_l1=1e6, reset_topic_method='random',
                            fix_reset_seed=True)
    assert nres == k and relfro(T, g['l1killW_rnd_T']) < max(tol, 1e-7) and relfro(W, g['l1killW_rnd_W']) < max(tol, 1e-7)


def test_ragged_shapes():
    """d not a multiple of the 16-byte vector, n not a multiple of any tile, k = 1, 2, 3."""
    orc = oracle()
    for (n, d, k) in [(37, 5, 1), (130, 1027, 2), (1000, 2051, 3), (65, 64, 7)]:
        X = planted_X(n, d, max(k, 2), seed=11, dtype=np.float64)
        W0, T0 = scaled_init(X, k, seed=12)
        for dtype in (np.float64, np.float32):
            ref = run_oracle(stored(X, dtype), W0, T0, 4)
            W, T, _ = run_engine(X, W0, T0, 4, dtype)
            assert relfro(W, ref['W']) < TOL[dtype] and relfro(T, ref['T']) < TOL[dtype], (n, d, k, dtype)


def test_half_steps_match_a_sweep():
    X = planted_X(300, 200, 4, seed=21, dtype=np.float64)
    W0, T0 = scaled_init(X, 4, seed=22)
    Wa, Ta, _ = run_engine(X, W0, T0, 2, np.float64)
    with engine(300, 200, 4, dtype=np.float64) as e:
        e.upload_X(X); e.set_W(W0); e.set_T(T0); e.set_params()
        for _ in range(2):
            for t in range(4):
                e.update_T_row(t)
                e.update_W_col(t)
        Wb, Tb = e.get_W(), e.get_T()
    assert relfro(Wb, Wa) < 1e-12 and relfro(Tb, Ta) < 1e-12


def test_rank1_update_and_copy_bench_run():
    X = planted_X(2048, 1024, 4, seed=1, dtype=np.float32)
    W0, T0 = scaled_init(X, 4, seed=2)
    with engine(2048, 1024, 4, dtype=np.float32) as e:
        e.upload_X(X); e.set_W(W0); e.set_T(T0); e.set_params()
        assert e.bench_rank1_update(3) > 0
        assert e.bench_stream_copy(3) > 0
        e.timing_enable(True)
        e.sweep(2)
        cnt, ms = e.timing_read(0)
        # this shape runs as the register-resident persistent launch (one timed launch per call); the launch-per-phase
        # schedule times a pass per topic step and the prologue's
        onchip = e.onchip_info()[0]
        assert cnt == (1 if onchip else 2 * 4 + 1) and ms > 0
        e.timing_enable(True, every=4)
        e.sweep(2)
        assert e.timing_read(0)[0] == (1 if onchip else 2)
    import os
    old = os.environ.get('RRI_ONCHIP')
    os.environ['RRI_ONCHIP'] = '0'
    try:
        with engine(2048, 1024, 4, dtype=np.float32) as e:
            e.upload_X(X); e.set_W(W0); e.set_T(T0); e.set_params()
            e.timing_enable(True)
            e.sweep(2)
            assert e.timing_read(0)[0] == 2 * 4 + 1
    finally:
        if old is None:
            os.environ.pop('RRI_ONCHIP', None)
        else:
            os.environ['RRI_ONCHIP'] = old


# ---------------------------------------------------------------------------------------------------------
# elementwise-weighted flavour (WRRI; nmf.py:687-701, 735-746)
# ---------------------------------------------------------------------------------------------------------
def run_weighted(X, M, W0, T0, sweeps, dtype, **params):
    n, d = X.shape
    k = W0.shape[1]
    with engine(n, d, k, dtype=dtype, weighted=True) as e:
        e.upload_X(X)
        e.upload_mask(M)
        e.set_W(np.maximum(W0, 0))
        e.set_T(np.maximum(T0, 0))
        e.set_params(**params)
        e.sweep(sweeps)
        if params.get('w_row_sum') is not None and not params.get('fix_W'):
            e.project_W_rows(params['w_row_sum'])      # nmf()'s final projection (nmf.py:519-529)
        return e.get_W(), e.get_T(), e.objective()


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_weighted_flavour_synthetic(tag, dtype):
    """well-conditioned 30 % mask: W and T themselves are comparable"""
    g = load_golden('g5_plain_' + tag)
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    M = (np.random.RandomState(2).rand(n, d) < 0.3).astype(np.float64)
    Xm = stored(M * X, dtype)
    flags = dict(t_row_sum=1.0, reset_topic_method=None)
    W, T, obj = run_weighted(Xm, M, W0, T0, 4, dtype, **flags)
    ref = run_oracle(Xm, W0, T0, 4, W_mat=M, compute_obj_each_iter=True, **flags)
    # float64 residual: summation-order only.  fp32 residual: E is rounded to fp32 at every store (and
    # refreshed each sweep), so the bar is BASELINE's 1e-4
    tol = 5e-9 if dtype == np.float64 else 1e-4
    assert relfro(W, ref['W']) < tol and relfro(T, ref['T']) < tol, (relfro(W, ref['W']), relfro(T, ref['T']))
    assert abs(obj - ref['obj_history'][-1]) < (1e-9 if dtype == np.float64 else 1e-5) * abs(obj)
    if dtype == np.float64:
        assert relfro(W, g['wr_W_s4']) < tol and relfro(T, g['wr_T_s4']) < tol
    # l1 penalties, general (non 0/1) weights, fixed halves
    Mw = M * (0.5 + np.random.RandomState(3).rand(n, d))
    for kw in (dict(reg_w_l1=0.05, reg_t_l1=0.02), dict(fix_T=True), dict(fix_W=True),
               dict(w_row_sum=2.0, reg_w_l2=0.1, reg_t_l2=0.2)):
        W, T, _ = run_weighted(Xm, Mw, W0, T0, 3, dtype, **dict(flags, **kw))
        ref = run_oracle(Xm, W0, T0, 3, W_mat=Mw, **dict(flags, **kw))
        assert relfro(W, ref['W']) < tol and relfro(T, ref['T']) < tol, (kw, relfro(W, ref['W']), relfro(T, ref['T']))


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_weighted_recsys_fixture(dtype):
    """the reference's own WRRI test settings (tests/test_nmf.py:57-78) on its recsys fixture.  Some T entries
    there have denominators ~1e-27: W, T agree only in the early sweeps, the masked
    reconstruction M.(WT) and the objective agree throughout."""
    g = load_golden('g4_wrri')
    X = g['X'].astype(np.float64)
    M = np.zeros(X.shape)
    M[X.nonzero()] = 1.0
    cases = [{}, {'reg_w_l1': 0.1, 'reg_t_l1': 0.1}, {'reg_w_l1': 0.1}, {'reg_t_l1': 0.1}]
    for ci, c in enumerate(cases):
        flags = dict(c, reset_topic_method=None, t_row_sum=1.0)
        Wr, Tr = g['c%d_W_s1' % ci], g['c%d_T_s1' % ci]
        W, T, _ = run_weighted(X, M, g['W0'], g['T0'], 1, dtype, **flags)
        if dtype == np.float64:       # first sweep: every entry is well determined
            assert relfro(W, Wr) < 1e-9 and relfro(T, Tr) < 1e-9, (relfro(W, Wr), relfro(T, Tr))
        assert relfro(M * (W @ T), M * (Wr @ Tr)) < (1e-9 if dtype == np.float64 else 1e-5)
        tol_rec = 1e-6 if dtype == np.float64 else 1e-4
        for S, Wr, Tr in ((2, g['c%d_W_s2' % ci], g['c%d_T_s2' % ci]), (6, g['c%d_W_s6' % ci], g['c%d_T_s6' % ci]),
                          (15, g['c%d_W' % ci], g['c%d_T' % ci])):
            W, T, obj = run_weighted(X, M, g['W0'], g['T0'], S, dtype, **flags)
            assert relfro(M * (W @ T), M * (Wr @ Tr)) < tol_rec, (S, relfro(M * (W @ T), M * (Wr @ Tr)))
        assert abs(obj - g['c%d_obj' % ci][-1]) < tol_rec * abs(obj)
        # monotone objective, as the reference's test asserts
        objs = []
        with engine(X.shape[0], X.shape[1], 7, dtype=dtype, weighted=True) as e:
            e.upload_X(X); e.upload_mask(M); e.set_W(g['W0']); e.set_T(g['T0']); e.set_params(**flags)
            for _ in range(15):
                e.sweep(1)
                objs.append(e.objective())
        assert np.all(np.diff(objs) <= 1e-9 * abs(objs[0]))
        assert np.allclose(objs, g['c%d_obj' % ci], rtol=(1e-6 if dtype == np.float64 else 1e-4))


@pytest.mark.parametrize('shape', [(700, 333, 6), (257, 1030, 5), (4133, 520, 4), (31, 7, 2), (9000, 64, 3)])
def test_sparse_mask_correction_walks_set_bits_only(monkeypatch, shape):
    """k_wmcorr_cols (round 4): below 12 % density the mask-only correction of the one-pass weighted step reads a second packed
    copy of the mask with the rows in the bits and adds u[row] for the SET bits alone.  Same sums as the kernel that walks every
    bit (RRI_WMCORR_COLS=0), other row-block partition: agreement to summation order, on ragged shapes (rows not a multiple of
    32 or of a row block, columns not a multiple of 4 / 64 / 256), and against the oracle.  The same launch then takes the T-row
    step's second column sum nw = (w^2)^T M as well and the read-modify-write pass leaves it out (RRI_WNW_MASK=0: it does not)."""
    n, d, k = shape
    X = planted_X(n, d, k, seed=n, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    M = (np.random.RandomState(2).rand(n, d) < 0.06).astype(np.float64)
    M[:, 0] = 1.0          # a full column and an empty one beside the ragged rest
    M[:, d - 1] = 0.0
    M[0, :] = 1.0          # every topic row stays determined
    Xm = M * X
    flags = dict(t_row_sum=1.0, reset_topic_method=None)
    monkeypatch.setenv('RRI_ONCHIP', '0')
    out = {}
    # (set bits only, nw = (w^2)^T M from the same launch) | (set bits only, nw from the pass) | (every bit, nw from the pass)
    for sw in (('1', '1'), ('1', '0'), ('0', '1')):
        monkeypatch.setenv('RRI_WMCORR_COLS', sw[0])
        monkeypatch.setenv('RRI_WNW_MASK', sw[1])
        out[sw] = run_weighted(Xm, M, W0, T0, 3, np.float64, **flags)
    b = out[('0', '1')]
    for sw in (('1', '1'), ('1', '0')):
        a = out[sw]
        assert relfro(a[0], b[0]) < 1e-11 and relfro(a[1], b[1]) < 1e-11 and abs(a[2] - b[2]) <= 1e-11 * abs(b[2]), \
            (sw, relfro(a[0], b[0]), relfro(a[1], b[1]))
    a = out[('1', '1')]
    ref = run_oracle(Xm, W0, T0, 3, W_mat=M, compute_obj_each_iter=True, **flags)
    assert relfro(M * (a[0] @ a[1]), M * (ref['W'] @ ref['T'])) < 1e-8
    assert abs(a[2] - ref['obj_history'][-1]) < 1e-8 * abs(a[2])
    # fixed halves, penalties and fp32 storage through the same kernels
    for kw, dt in ((dict(fix_W=True), np.float64), (dict(fix_T=True), np.float64), (dict(reg_w_l1=0.05, reg_t_l2=0.02), np.float32)):
        def outcome(sw):
            monkeypatch.setenv('RRI_WMCORR_COLS', sw)
            try:
                return run_weighted(stored(Xm, dt), M, W0, T0, 2, dt, **dict(flags, **kw))
            except (AssertionError, ValueError) as exc:      # the reference's own stops (an emptied column on the 31 x 7 case)
                return str(exc)
        a, b = outcome('1'), outcome('0')
        if isinstance(a, str) or isinstance(b, str):
            assert a == b, (kw, a, b)
            continue
        tol = 1e-11 if dt == np.float64 else 1e-5
        assert relfro(a[0], b[0]) < tol and relfro(a[1], b[1]) < tol, (kw, relfro(a[0], b[0]), relfro(a[1], b[1]))


@pytest.mark.gpu
@pytest.mark.parametrize('store', [np.float32, np.float64])
def test_objective_without_a_pass_over_X(store, monkeypatch):
    """after a complete sweep rri_objective assembles 1/2||X - WT||^2 from ||X||^2, the cross terms the W halves
    left behind and the two Gram matrices; it must agree with the residual-based value (and with numpy) and fall back
    to the residual whenever something changed W or T from outside"""
    from rri_nmf_amd.engine import RRIEngine
    n, d, k = 2111, 517, 7
    X = planted_X(n, d, k, seed=0, dtype=store)
    X64 = X.astype(np.float64)
    W0, T0 = scaled_init(X64, k, seed=1)
    direct = lambda W, T: 0.5 * float(((X64 - W @ T) ** 2).sum())
    for flags in (dict(), dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0), dict(fix_T=True),
                  dict(reg_w_l1=0.01, reg_t_l2=0.1)):
        with RRIEngine(n, d, k, dtype=store) as e:
            e.upload_X(X), e.set_W(W0), e.set_T(T0)
            e.set_params(**flags)
            o0 = e.objective_parts()[0]                      # nothing has run: residual path
            assert abs(o0 - direct(W0, T0)) <= 1e-12 * o0
            for sweeps in (1, 2):
                e.sweep(sweeps)
                fast = e.objective_parts()[0]
                W, T = e.get_W(), e.get_T()
                want = direct(W, T)
                assert abs(fast - want) <= 1e-11 * max(want, 1e-3 * float((X64 ** 2).sum())), (flags, fast, want)
            e.set_W(W)                                       # same values, but set from outside: back to the residual
            slow = e.objective_parts()[0]
            assert abs(slow - want) <= 1e-12 * want and abs(slow - fast) <= 1e-10 * want
            e.update_T_row(1)                                # a lone half step: the cross terms are stale
            W, T = e.get_W(), e.get_T()
            assert abs(e.objective_parts()[0] - direct(W, T)) <= 1e-12 * direct(W, T)


@pytest.mark.gpu
def test_the_abi_from_plain_c(tmp_path):
    """tests/c/abi_smoke.c: the boundary used by a C program with no Python or torch in the process (gcc, -lrri_hip)"""
    import os
    import subprocess
    from conftest import ROOT
    from rri_nmf_amd import _capi
    libdir = os.path.dirname(_capi.LIB_PATH)
    exe = str(tmp_path / 'abi_smoke')
    subprocess.run(['gcc', '-std=c99', '-pthread', '-Wall', '-Wextra', '-Werror', '-I', os.path.join(ROOT, 'include'),
                    os.path.join(ROOT, 'tests', 'c', 'abi_smoke.c'), '-o', exe, '-L', libdir, '-lrri_hip', '-lm',
                    '-Wl,-rpath,' + libdir], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    tag, o0, o1, cw, ct = lines[0].split()
    assert tag == 'ok' and float(o1) < 0.2 * float(o0) and float(cw) > 0 and float(ct) > 0
    # two row blocks on two handles in the one C process, collectives inside rri_sweep: the one-handle factors
    tag, ew, et = lines[1].split()
    assert tag == 'sharded' and float(ew) < 1e-9 and float(et) < 1e-9, lines[1]
    # R <- R - a b^T on an explicit-residual handle: at most one fp32 ulp from the same arithmetic in C, products 1e-12
    tag, ulps, ey = lines[2].split()
    assert tag == 'residual' and float(ulps) <= 1.01 and float(ey) < 1e-12, lines[2]


def test_bench_line_keeps_its_contract():
    """`python bench.py` prints ONE JSON line with the keys the driver reads, the roofline and cpu_baseline objects
    included (smallest BASELINE configuration, a few steps)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--config', 'c2', '--steps', '3', '--warmup', '1'],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key, typ in (('metric', str), ('value', float), ('unit', str), ('n_gpus', int), ('steps', int), ('warmup', int),
                     ('ms_per_step', float), ('higher_is_better', bool), ('scaling', str), ('dtype', str),
                     ('data', str), ('config', dict), ('roofline', dict), ('cpu_baseline', dict)):
        assert isinstance(j[key], typ), key
    assert j['vs_baseline'] is None and j['n_gpus'] == 1 and j['steps'] == 3 and j['warmup'] == 1
    assert j['higher_is_better'] is True and j['scaling'] in ('weak', 'strong') and 'workload' in j['config']
    assert abs(j['ms_per_step'] * j['value'] - 1e3) < 1e-6 * 1e3          # sweeps/s and ms per sweep agree
    r = j['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and r['peak'] > 0 and r['achieved'] > 0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 'traffic' in r
    c = j['cpu_baseline']
    assert c['value'] > 0 and c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['unit'] == j['unit'] and c['sample']
    assert c['one_thread']['value'] > 0 and c['one_thread']['cores'] == 1
    # parity in the same run, at C2's FULL size: the device path against the CPU oracle after 1 and 2 sweeps (2e-9:
    # summation order only, the bound of this file), next to the oracle's own 1-ulp sensitivity
    ps = j['parity_sample']
    assert ps['rows'] == 10000 and ps['sweeps'] == 2
    for key in ('after_1_sweeps', 'after_2_sweeps'):
        assert ps[key]['relfro_W'] < 2e-9 and ps[key]['relfro_T'] < 2e-9, (key, ps[key])
    assert ps['reference_self_sensitivity']['relfro_W'] > 0.0          # the control measures something
    # the explicit-residual schedule ran beside it (fp32 residual: BASELINE's 1e-4 bar against the default schedule)
    sc = j['rank1_update']['schedule']
    assert sc['sweeps_per_s'] > 0 and max(sc['vs_default_schedule_after_4_sweeps'].values()) < 1e-4, sc
    assert j['rank1_update']['achieved'] > 0


@pytest.mark.parametrize('shape', [(700, 333, 6), (257, 1030, 5), (1501, 100, 3), (5003, 1000, 22), (64, 4100, 2), (9, 7, 2), (300, 500, 1)])
@pytest.mark.parametrize('store', [np.float32, np.float64])
def test_the_lds_dma_pass_gives_the_register_pass_its_bits(monkeypatch, shape, store):
    """k_pass_dma (round 4): the read-only pass with its rows staged through an LDS ring that LDS-DMA fills.  Same arithmetic in
    the same order as k_pass, so the SAME BITS -- on ragged shapes (rows not a multiple of the chunk, columns not a multiple of
    a lane's vector or of a panel, fewer columns than one wave covers, more panels than one workgroup), both storage types,
    plain and topic-model flags, fixed halves, with and without interleaved row chunks.  The library takes the ring by itself only
    where X streams from HBM; RRI_PASS_DMA=1 forces it here."""
    from rri_nmf_amd.engine import RRIEngine
    n, d, k = shape
    X = planted_X(n, d, k, seed=n + d, dtype=store)
    W0, T0 = scaled_init(X, k, seed=3)
    T0s = T0 / T0.sum(1, keepdims=True)
    monkeypatch.setenv('RRI_ONCHIP', '0')          # the launch-per-phase schedule: that is where the pass runs

    def run(dma, il, flags, T, sub='0'):
        monkeypatch.setenv('RRI_PASS_DMA', dma)
        monkeypatch.setenv('RRI_PASS_IL', il)
        monkeypatch.setenv('RRI_PASS_DMA_SUB', sub)       # row blocks a workgroup of the ring kernel walks as one stream
        monkeypatch.setenv('RRI_PASS_WGS', '64' if sub != '0' else '2048')   # several row blocks even at these sizes
        with RRIEngine(n, d, k, dtype=store) as e:
            e.upload_X(X), e.set_W(W0), e.set_T(T), e.set_params(**flags)
            e.sweep(3)
            return e.get_W(), e.get_T(), e.objective()

    for flags, T in ((dict(), T0), (dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0), T0s), (dict(fix_W=True), T0),
                     (dict(reg_w_l1=0.01, reg_t_l2=0.02), T0)):
        for il in ('0', '1'):
            a, b = run('1', il, flags, T), run('0', il, flags, T)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (flags, il, relfro(a[0], b[0]), relfro(a[1], b[1]))
        # three row blocks per workgroup (contiguous rows), the last group ragged: still one row of column sums per row block
        a, b = run('1', '0', flags, T, sub='3'), run('0', '0', flags, T, sub='3')
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], (flags, 'sub 3', relfro(a[0], b[0]), relfro(a[1], b[1]))
