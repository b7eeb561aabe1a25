"""The explicit-residual form of the path (BASELINE north_star: "outer-product residual update R <- R -+ v_t u_t^T"),
through the C ABI: RRI_UNWEIGHTED_RESIDUAL handles (RRIEngine(schedule='residual')).

  * the update as an operation of its own (rri_residual_update = k_pass<UPD>): the residual it writes, the row dots
    and the column sums it returns, against numpy float64 -- ragged shapes, both storage types, one and two terms,
    and one launch at BASELINE's full C3 size against torch float64 on the same device data;
  * whole sweeps of that schedule against the CPU oracle (the reference's Gram form, nmf.py:670-676, 728-734) and
    against the vectors captured from the reference, plain / topic-model / regularised, with reset events.

Tolerances: float64 storage 2e-9 (summation order only, as tests/test_hip_parity.py); fp32 storage of R: the stored
residual is rounded to fp32 at every update and rebuilt once per sweep -- 1e-4, BASELINE's bar, stated per assert.
"""
import numpy as np
import pytest

from conftest import load_golden, relfro
from rri_nmf_amd.synthetic import planted_X, scaled_init

pytestmark = pytest.mark.gpu

TOL = {np.float64: 2e-9, np.float32: 1e-4}


def engine(*a, **kw):
    from rri_nmf_amd.engine import RRIEngine
    return RRIEngine(*a, **kw)


def oracle():
    from oracle import rri_oracle
    return rri_oracle


def stored(X, dtype):
    return np.ascontiguousarray(np.asarray(X).astype(dtype).astype(np.float64))


def run_residual(X, W0, T0, sweeps, dtype, final_proj=None, schedule='residual', **params):
    n, d = X.shape
    k = W0.shape[1]
    with engine(n, d, k, dtype=dtype, schedule=schedule) as e:
        e.upload_X(X)
        e.set_W(np.maximum(W0, 0))
        e.set_T(np.maximum(T0, 0))
        e.set_params(**params)
        e.sweep(sweeps)
        if final_proj is not None:
            e.project_W_rows(final_proj)
        return e.get_W(), e.get_T(), e.n_resets_used


def run_oracle(X, W0, T0, sweeps, **kw):
    return oracle().nmf(X, W0.shape[1], W_in=W0.copy(), T_in=T0.copy(), max_iter=sweeps, eps_stop=-1, **kw)


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('shape', [(37, 5, 2), (130, 1027, 3), (1000, 2051, 4), (2111, 517, 7), (65, 64, 2)])
def test_rank_one_update_against_numpy(shape, dtype):
    """R <- R - a b^T (- a2 b2^T) with non-zero factors: the written R, y = R_new t and z = R_new^T w"""
    n, d, k = shape
    rs = np.random.RandomState(n + d)
    X = planted_X(n, d, max(k, 2), seed=3, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=4)
    ulp = {np.float64: 2.0 ** -52, np.float32: 2.0 ** -23}[dtype]
    with engine(n, d, k, dtype=dtype, schedule='residual') as e:
        e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
        e.residual_rebuild()
        R0 = e.get_residual().astype(np.float64)
        want0 = stored(X, dtype) - W0 @ T0
        # X - W T is a k-term sum per entry in another order than numpy's, then rounded to the storage type
        assert np.abs(R0 - want0).max() <= ulp * np.abs(stored(X, dtype)).max() * 1.01 + 1e-15
        for two in (False, True):
            a, b = rs.rand(n) - 0.3, rs.rand(d) - 0.3
            a2, b2 = (rs.rand(n) - 0.5, rs.rand(d) - 0.5) if two else (None, None)
            trow, wcol = rs.rand(d), rs.rand(n)
            Rb = e.get_residual().astype(np.float64)
            y, z = e.residual_update(a, b, trow, wcol, a2=a2, b2=b2)
            R1 = e.get_residual().astype(np.float64)
            want = Rb - np.outer(a, b) - (np.outer(a2, b2) if two else 0.0)
            # one rounding to the storage type (the kernel fuses multiply-add in float64; numpy rounds the product first)
            assert np.abs(R1 - want).max() <= ulp * max(np.abs(want).max(), 1.0) * 1.01, (two, np.abs(R1 - want).max())
            assert relfro(R1, want) < 2 * ulp
            # the fused products are taken on the residual AS STORED
            assert relfro(y, R1 @ trow) < 1e-13 and relfro(z, R1.T @ wcol) < 1e-13, (relfro(y, R1 @ trow), relfro(z, R1.T @ wcol))


def test_rank_one_update_at_full_size():
    """one launch at C3 size (100000 x 10000 fp32): the residual written by the update kernel and its fused products
    against torch float64 on the same device data"""
    import torch
    N, D, K = 100000, 10000, 50
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    X = torch.rand(N, D, device=dev, generator=g, dtype=torch.float32)
    W0 = 0.1 * torch.rand(N, K, device=dev, generator=g, dtype=torch.float64)
    T0 = 0.1 * torch.rand(K, D, device=dev, generator=g, dtype=torch.float64)
    a = torch.rand(N, device=dev, generator=g, dtype=torch.float64) - 0.3
    b = torch.rand(D, device=dev, generator=g, dtype=torch.float64) - 0.3
    trow = torch.rand(D, device=dev, generator=g, dtype=torch.float64)
    wcol = torch.rand(N, device=dev, generator=g, dtype=torch.float64)
    torch.cuda.synchronize()
    with engine(N, D, K, dtype=np.float32, schedule='residual') as e:
        e.bind_X_device(X.data_ptr(), X.stride(0))
        e.set_W(W0.cpu().numpy()), e.set_T(T0.cpu().numpy()), e.set_params()
        e.residual_rebuild()
        y, z = e.residual_update(a.cpu().numpy(), b.cpu().numpy(), trow.cpu().numpy(), wcol.cpu().numpy())
        R1 = e.get_residual()                      # 4 GB, host
    num = den = 0.0
    ymax = 0.0
    zacc = torch.zeros(D, dtype=torch.float64, device=dev)
    yt = torch.from_numpy(y).to(dev)
    for lo in range(0, N, 10000):
        hi = lo + 10000
        r0 = (X[lo:hi].double() - W0[lo:hi] @ T0).float().double()        # as the rebuild stores it
        want = (r0 - a[lo:hi, None] * b[None, :]).float()
        got = torch.from_numpy(R1[lo:hi]).to(dev)
        diff = (got - want).double()
        num += float((diff * diff).sum())
        den += float((want.double() ** 2).sum())
        assert float(diff.abs().max()) <= 2.0 ** -23 * float(want.abs().max()) * 2.01   # at most one fp32 ulp apart
        gd = got.double()
        ymax = max(ymax, float(((gd @ trow) - yt[lo:hi]).abs().max() / (gd @ trow).abs().max()))
        zacc += gd.t() @ wcol[lo:hi]
    assert (num / den) ** 0.5 < 3e-8, (num / den) ** 0.5     # rare 1-ulp differences where the two GEMMs round apart
    assert ymax < 1e-12, ymax
    zt = torch.from_numpy(z).to(dev)
    assert float(torch.linalg.norm(zt - zacc) / torch.linalg.norm(zacc)) < 1e-12


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_residual_schedule_vs_oracle_and_reference_vectors(tag, dtype):
    g = load_golden('g5_plain_' + tag)
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    Xs = stored(X, dtype)
    tol = TOL[dtype]
    for S in (1, 5, 30):
        W, T, _ = run_residual(X, W0, T0, S, dtype)
        ref = run_oracle(Xs, W0, T0, S)
        assert relfro(W, ref['W']) < tol and relfro(T, ref['T']) < tol, (S, relfro(W, ref['W']), relfro(T, ref['T']))
        if dtype == np.float64:       # the reference's own vectors
            assert relfro(W, g['W_s%d' % S]) < tol and relfro(T, g['T_s%d' % S]) < tol
        Wg, Tg, _ = run_residual(X, W0, T0, S, dtype, schedule='gram')     # and the default schedule
        assert relfro(W, Wg) < tol and relfro(T, Tg) < tol
    # regularised
    regs = dict(reg_w_l1=0.01, reg_t_l1=0.02, reg_w_l2=0.05, reg_t_l2=0.03)
    W, T, _ = run_residual(X, W0, T0, 5, dtype, **regs)
    ref = run_oracle(Xs, W0, T0, 5, **regs)
    assert relfro(W, ref['W']) < tol and relfro(T, ref['T']) < tol
    if dtype == np.float64:
        assert relfro(W, g['reg_W_s5']) < tol and relfro(T, g['reg_T_s5']) < tol
    # topic-model flags: T rows on the simplex every step, final projection of W
    orc = oracle()
    Xn = orc.normalize(X.copy())
    T0p = orc.proj_rows_simplex(np.maximum(T0, 0).copy(), 1.0)
    tm = dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)
    for S in (1, 5):
        W, T, _ = run_residual(Xn, W0, T0p, S, dtype, final_proj=1.0, **tm)
        ref = run_oracle(stored(Xn, dtype), W0, T0, S, **tm)
        assert relfro(W, ref['W']) < tol and relfro(T, ref['T']) < tol, (S, relfro(W, ref['W']), relfro(T, ref['T']))
        assert np.abs(T.sum(1) - 1).max() < 1e-12 and np.abs(W.sum(1) - 1).max() < 1e-12


def test_residual_schedule_ragged_shapes_and_resumability():
    for (n, d, k) in [(130, 1027, 2), (1000, 2051, 3), (65, 64, 7)]:
        X = planted_X(n, d, max(k, 2), seed=11, dtype=np.float64)
        W0, T0 = scaled_init(X, k, seed=12)
        for dtype in (np.float64, np.float32):
            ref = run_oracle(stored(X, dtype), W0, T0, 4)
            W, T, _ = run_residual(X, W0, T0, 4, dtype)
            assert relfro(W, ref['W']) < TOL[dtype] and relfro(T, ref['T']) < TOL[dtype], (n, d, k, dtype)
    X = planted_X(700, 333, 6, seed=5, dtype=np.float64)
    W0, T0 = scaled_init(X, 6, seed=6)
    Wa, Ta, _ = run_residual(X, W0, T0, 3, np.float32)
    with engine(700, 333, 6, dtype=np.float32, schedule='residual') as e:
        e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
        for _ in range(3):
            e.sweep(1)
        assert np.array_equal(e.get_W(), Wa) and np.array_equal(e.get_T(), Ta)      # carry across calls is exact
        o1 = e.objective()
        e.sweep(1)                       # the objective stored the residual: the sweep reuses it
        Wb = e.get_W()
    with engine(700, 333, 6, dtype=np.float32, schedule='residual') as e:
        e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
        e.sweep(4)
        assert relfro(e.get_W(), Wb) < 1e-6
        assert e.objective() <= o1
    # half steps: k T-row / W-column calls equal a sweep to rounding
    with engine(700, 333, 6, dtype=np.float64, schedule='residual') as e:
        e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
        for t in range(6):
            e.update_T_row(t)
            e.update_W_col(t)
        Wh, Th = e.get_W(), e.get_T()
    Ws, Ts, _ = run_residual(X, W0, T0, 1, np.float64)
    assert relfro(Wh, Ws) < 1e-10 and relfro(Th, Ts) < 1e-10
    # fixed halves on a handle of this schedule: one half of every step is missing, the library steps such calls in the Gram
    # form (T fixed: X T^T once, no pass per topic) -- bit for bit what a Gram-form handle gives; switching back and forth
    # on ONE handle keeps no sums of the other form
    for fixed in (dict(fix_T=True), dict(fix_W=True)):
        out = []
        for schedule in ('residual', 'gram'):
            with engine(700, 333, 6, dtype=np.float64, schedule=schedule) as e:
                e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
                e.sweep(2)                               # both halves free (the residual handle: its own schedule)
                e.set_params(**fixed)
                e.sweep(2)
                mid = (e.get_W(), e.get_T())
                e.set_params()
                e.sweep(1)
                out.append(mid + (e.get_W(), e.get_T(), e.objective()))
        a, b = out
        assert relfro(a[0], b[0]) < 1e-9 and relfro(a[1], b[1]) < 1e-9, fixed
        assert relfro(a[2], b[2]) < 1e-9 and relfro(a[3], b[3]) < 1e-9 and abs(a[4] - b[4]) <= 1e-9 * abs(b[4])
        if 'fix_T' in fixed:
            assert np.array_equal(a[1], out[0][1])


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_residual_schedule_rare_branches(dtype):
    """reset events and error conventions (nmf.py:751-816, 475-476) on the explicit-residual schedule, against the
    vectors captured from the reference"""
    g = load_golden('g6_rare_branches')
    n, d, k = [int(v) for v in g['shape']]
    X = planted_X(n, d, k, seed=3, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=4)
    tol = 2e-9 if dtype == np.float64 else 1e-4
    Wd = g['dead_W0']
    with pytest.raises(ValueError, match='unbounded'):
        run_residual(X, Wd, T0, 2, dtype)
    W, T, nres = run_residual(X, Wd, T0, 2, dtype, t_row_sum=1.0)
    assert nres >= 1 and relfro(T, g['dead_mrd_T']) < tol and relfro(W, g['dead_mrd_W']) < tol
    with pytest.raises(AssertionError, match='sums to 0'):
        run_residual(X, Wd, T0, 2, dtype, t_row_sum=1.0, w_row_sum=1.0, reset_topic_method=None)
    W, T, nres = run_residual(X, W0, T0, 1, dtype, t_row_sum=1.0, reg_t_l1=1e6)
    assert nres == k and relfro(T, g['l1kill_T']) < tol and relfro(W, g['l1kill_W']) < tol
    W, T, nres = run_residual(X, W0, T0, 1, dtype, t_row_sum=1.0, reg_w_l1=1e6)
    assert nres == k and relfro(T, g['l1killW_mrd_T']) < tol and relfro(W, g['l1killW_mrd_W']) < tol
    orc = oracle()
    Xn = orc.normalize(X.copy())
    T0p = orc.proj_rows_simplex(np.maximum(T0, 0).copy(), 1.0)
    gtol = tol if dtype == np.float64 else 2e-4      # goldens were made with the float64 X
    W, T, _ = run_residual(Xn, W0, T0p, 3, dtype, final_proj=1.0, project_T_each_iter=True, t_row_sum=1.0,
                           w_row_sum=1.0, reg_t_l2=-50.0)
    assert relfro(T, g['negT_T']) < gtol and relfro(W, g['negT_W']) < gtol
    W, T, _ = run_residual(Xn, W0, T0p, 2, dtype, project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0, reg_w_l2=-5.0)
    assert relfro(T, g['negW_T']) < gtol and relfro(W, g['negW_W']) < gtol


def test_nmf_with_the_residual_schedule_on_the_text_fixture():
    """nmf(..., schedule='residual') through the drop-in surface: the reference's topic-model fit (G1), exact topic
    assignments"""
    from rri_nmf_amd import nmf as nmf_mod
    g = load_golden('g1_tm_estimator')
    X, W0, T0 = g['X'], g['W0'], g['T0']
    for dtype in (np.float64, np.float32):
        r = nmf_mod.nmf(X, 5, W_in=W0, T_in=T0, max_iter=10, eps_stop=-1, project_T_each_iter=True, t_row_sum=1.0,
                        w_row_sum=1.0, project_W_each_iter=False, dtype=dtype, schedule='residual')
        assert relfro(r['W'], g['W_s10']) < (2e-9 if dtype == np.float64 else 1e-4)
        assert np.array_equal(np.argmax(r['W'], 1), g['argmax_s10'])
    # fold-in through the estimator (NMF_TM_Estimator.transform: fix_T, max_iter = 4, sklearn_interface.py:327-333) with the
    # explicit-residual schedule requested for every call: the reference's held-out vectors of G1
    from rri_nmf_amd import sklearn_interface as si
    n, d = X.shape
    M = si.NMF_TM_Estimator(n, d, 5, random_state=0, max_iter=10, nmf_kwargs={'eps_stop': -1, 'schedule': 'residual'}).fit(X)
    Wte = M.transform(g['Xte'])
    assert relfro(Wte, g['Wte']) < 1e-7 and np.array_equal(np.argmax(Wte, 1), g['argmax_te'])


def test_tile_rotation_is_calibrated_per_handle_and_changes_no_bit(monkeypatch, capfd):
    """calibrate_rot (round 4): a handle that keeps a residual of 10^8 elements or more times its read-modify-write pass under the
    two ways of dealing its tiles to the XCDs before the first sweep (null updates: the residual rewritten with its own values)
    and keeps the faster.  The rotation decides which workgroup computes a tile and nothing in any sum: the explicit-residual
    schedule and the dense weighted flavour return the same bits with the calibration on (default), off, and with either rotation
    forced; RRI_ROT_DEBUG shows that the calibration ran, once per handle."""
    from rri_nmf_amd.engine import RRIEngine
    n, d, k = 26000, 4000, 6                       # 1.04e8 elements: the smallest size class that calibrates
    X = planted_X(n, d, k, seed=21, dtype=np.float32)
    W0, T0 = scaled_init(X.astype(np.float64), k, seed=22)
    M = (np.random.RandomState(3).rand(n, d) < 0.08).astype(np.float32)
    M[0, :] = 1.0
    monkeypatch.setenv('RRI_ONCHIP', '0')

    def run(kind, **env):
        for key in ('RRI_ROT_CAL', 'RRI_PASS_ROT', 'RRI_ROT_DEBUG'):
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        if kind == 'residual':
            e = RRIEngine(n, d, k, dtype=np.float32, schedule='residual')
            e.upload_X(X)
            flags = {}
        else:
            e = RRIEngine(n, d, k, dtype=np.float32, weighted=True)
            e.upload_X(X * M)
            e.upload_mask(M)
            flags = dict(t_row_sum=1.0, reset_topic_method=None)
        with e:
            e.set_W(W0); e.set_T(T0); e.set_params(**flags)
            e.sweep(1)
            e.sweep(1)
            return e.get_W(), e.get_T(), e.objective()

    for kind in ('residual', 'weighted'):
        capfd.readouterr()
        base = run(kind, RRI_ROT_DEBUG='1')
        err = capfd.readouterr().err
        assert err.count('tile rotation 0:') == 1 and err.count('tile rotation 1:') == 1, err      # once per handle, not per sweep
        for env in (dict(RRI_ROT_CAL='0'), dict(RRI_PASS_ROT='0'), dict(RRI_PASS_ROT='1'), dict(RRI_PASS_ROT='3')):
            other = run(kind, **env)
            assert np.array_equal(other[0], base[0]) and np.array_equal(other[1], base[1]) and other[2] == base[2], (kind, env)
