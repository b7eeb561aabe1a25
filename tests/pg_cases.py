"""Process-group cases of the GPU suite, run as CHILD processes of the tests (tests/test_group_gpu.py):

    python tests/pg_cases.py <case> <out.npz> <log file>          (RANK / WORLD_SIZE / MASTER_PORT from the environment)

Why children: a process blocked inside RCCL or hipStreamSynchronize cannot be interrupted by pytest-timeout's signal
(round 1 lost a run to such a stall without a trace).  Every child arms `faulthandler.dump_traceback_later(..., exit=True)`
on its log file first: a stalled rendezvous, bootstrap or collective leaves the Python stacks of all threads there
(the ctypes / torch.distributed frame it sits in) and a non-zero exit code; the parent joins with a timeout, kills the
exact PIDs it started and puts the log's tail into the assertion message.  Logs live under gpurun_out/pg_logs/.
"""
import faulthandler
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402

HANG_AFTER_S = 150


def _problem(n, d, k, weighted, store, seed=0):
    from rri_nmf_amd.synthetic import planted_X, scaled_init, observed_mask
    X = planted_X(n, d, max(k, 2), seed=seed, dtype=store)
    M = None
    if weighted:
        M = observed_mask(n, d, 0.3, seed=2, dtype=store)
        X = X * M
    W0, T0 = scaled_init(X, k, seed=1)
    return X, M, W0, T0


# name -> (n, d, k, sweeps, weighted, storage, nmf flags); shared with the parent, which runs the one-handle reference
GROUP_CASES = {
    'plain_f32': (1501, 700, 6, 3, False, 'float32', dict()),
    'topic_model_f64': (900, 333, 5, 3, False, 'float64', dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
    'regularised_f64': (800, 300, 4, 3, False, 'float64', dict(reg_w_l1=0.01, reg_t_l1=0.02, reg_w_l2=0.05, reg_t_l2=0.03)),
    'fold_in_fix_T': (1000, 400, 5, 4, False, 'float64', dict(fix_T=True, t_row_sum=1.0, w_row_sum=1.0)),
    'fix_W': (1000, 400, 5, 3, False, 'float64', dict(fix_W=True)),
    'rank_one_k1': (700, 260, 1, 3, False, 'float64', dict()),
    'weighted_f32': (1201, 515, 5, 3, True, 'float32', dict(t_row_sum=1.0, reset_topic_method=None)),
    'weighted_f64_regs': (700, 260, 4, 3, True, 'float64', dict(t_row_sum=1.0, reset_topic_method=None, reg_w_l1=0.01, reg_t_l1=0.02)),
    'weighted_fix_T': (700, 260, 4, 3, True, 'float64', dict(t_row_sum=1.0, reset_topic_method=None, fix_T=True)),
    'pattern_only_f64': (1201, 515, 5, 3, 'sparse', 'float64', dict(t_row_sum=1.0, reset_topic_method=None)),
    'resets_W_max_resid': (600, 200, 4, 2, False, 'float64', dict(t_row_sum=1.0, reg_w_l1=1e6)),
    'resets_T_max_resid': (600, 200, 4, 2, False, 'float64', dict(t_row_sum=1.0, reg_t_l1=1e6)),
    'resets_W_random': (600, 200, 4, 2, False, 'float64', dict(t_row_sum=1.0, reg_w_l1=1e6, reset_topic_method='random', fix_reset_seed=True)),
    'weighted_resets_T': (600, 200, 4, 2, True, 'float64', dict(t_row_sum=1.0, reg_t_l1=1e6)),
    # T fixed: the W half of all topics is one launch per rank (k_wsweep_rows) and the k column sums travel in ONE all-reduce; a
    # column that empties in the middle of a launch pauses every rank at that topic, the reset rewrites T[t,:] and W[:,t], and the
    # sweep resumes at the next topic with X T^T and T T^T taken again
    'fold_in_resets_W_random': (600, 200, 4, 2, False, 'float64', dict(fix_T=True, t_row_sum=1.0, reg_w_l1=1e6, reset_topic_method='random',
                                                                     fix_reset_seed=True)),
    'fold_in_resets_W_max_resid': (600, 200, 4, 2, False, 'float64', dict(fix_T=True, t_row_sum=1.0, reg_w_l1=1e6)),
    # the explicit-residual schedule row-sharded: every rank keeps its rows of R, the reference is the same schedule on one handle
    'residual_schedule_f64': (1501, 700, 6, 3, False, 'float64', dict(schedule='residual')),
    'residual_schedule_f32_tm': (2600, 1200, 7, 3, False, 'float32', dict(schedule='residual', project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
    'residual_schedule_resets_W': (600, 200, 4, 2, False, 'float64', dict(schedule='residual', t_row_sum=1.0, reg_w_l1=1e6)),
    # fixed halves asked of the explicit-residual schedule (the estimators' transform with nmf_kwargs={'schedule': 'residual'}):
    # the library steps them in the Gram form on that handle
    'residual_schedule_fold_in': (1000, 400, 5, 4, False, 'float64', dict(schedule='residual', fix_T=True, t_row_sum=1.0, w_row_sum=1.0)),
    # early stopping on held-out entries scored on the device (what NMF_RS_Estimator.fit installs by default,
    # sklearn_interface.py:71-93, nmf.py:381-407): every rank scores ITS rows' entries, the score is all-reduced
    'weighted_early_stop': (1201, 515, 5, 14, True, 'float64', dict(t_row_sum=1.0, reset_topic_method=None, _early_stop=True)),
    'pattern_only_early_stop': (1201, 515, 5, 14, 'sparse', 'float32', dict(t_row_sum=1.0, reset_topic_method=None, _early_stop=True)),
    'c4_proportions_unequal': (100003, 1000, 50, 1, False, 'float32', dict()),
    # no W_in / T_in: the start itself is computed row-sharded (initialization.randomized_svd_sharded and friends)
    'start_nndsvd': (1501, 700, 6, 3, False, 'float64', dict(_init='nndsvd', project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
    'start_nndsvda_f32': (2000, 333, 8, 2, False, 'float32', dict(_init='nndsvda')),
    'start_smart_random': (900, 333, 5, 2, False, 'float64', dict(_init='smart_random')),
    'start_nndsvd_fold_in': (1000, 400, 5, 4, False, 'float64', dict(_init='nndsvd', _T_in=True, fix_T=True, t_row_sum=1.0, w_row_sum=1.0)),
    # round 4: what nmf(group=) still refused.  nndsvdar: one draw per zero of the WHOLE W (initialization.py:147-152) -- every rank
    # draws the whole sequence and keeps its rows' stretch; the weighted start (nmf.py:841-843 factorises W_mat .* X) through a scratch
    # handle under the same group; tf-idf + normalisation on the device with the document frequencies all-reduced (matrixops.py:166-179);
    # per-row weights (nmf.py:335-344) with their refit (nmf.py:531-539) as a fold-in under the same group
    'start_nndsvdar': (1501, 700, 6, 3, False, 'float64', dict(_init='nndsvdar', project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
    'start_weighted_nndsvd': (1201, 515, 5, 3, True, 'float64', dict(_init='nndsvd', t_row_sum=1.0, reset_topic_method=None)),
    'preprocess_tfidf_normalize': (1501, 700, 6, 3, False, 'float64', dict(_counts=True, preprocess={'tfidf': True, 'normalize': True},
                                                                          project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
    'row_weights_with_refit': (1000, 400, 5, 3, False, 'float64', dict(_w_row=True, w_row_sum=1.0, project_T_each_iter=True, t_row_sum=1.0)),
}


def nmf_inputs(name):
    """(X, W_mat, W0, T0, k, keyword arguments of nmf()) of a case, all rows"""
    import scipy.sparse as sp
    n, d, k, sweeps, weighted, store, flags = GROUP_CASES[name]
    flags = dict(flags)
    X, M, W0, T0 = _problem(n, d, k, weighted, np.dtype(store))
    init = flags.pop('_init', None)
    keep_T = flags.pop('_T_in', False)
    if flags.pop('_counts', False):       # term counts with many zeros (tf-idf is trivial on a dense positive matrix)
        X = np.random.RandomState(5).poisson(0.4 * X / X.mean()).astype(np.dtype(store))
        W0, T0 = _problem(n, d, k, False, np.dtype(store), seed=3)[2:]
        T0 = T0 / T0.sum(1, keepdims=True)
    if flags.pop('_w_row', False):        # per-row weights, a column vector as the reference broadcasts it (nmf.py:337)
        flags['w_row'] = 0.5 + np.random.RandomState(6).rand(n, 1)
    held_out = None
    if flags.pop('_early_stop', False):
        # 5 % of the observed entries are held out of the fit (mask 0 there) and scored after every sweep
        rs = np.random.RandomState(7)
        ii, jj = np.nonzero(M)
        pick = rs.rand(ii.size) < 0.05
        held_out = (ii[pick], jj[pick], np.asarray(X[ii[pick], jj[pick]], dtype=np.float64), 0.0, float(X.max()))
        M = M.copy()
        M[ii[pick], jj[pick]] = 0
        X = X * M
    kw = dict(max_iter=sweeps, eps_stop=-1, compute_obj_each_iter=True, dtype=np.dtype(store), **flags)
    if held_out is not None:
        kw['early_stop'] = held_out_score(held_out, 0, n)
    if init is not None:       # the start comes from initialize_nmf (the fold-in case keeps T_in, as the estimators' transform does)
        kw.update(init=init, random_state=0, device_init=True)
        W0 = []
        if not keep_T:
            T0 = []
    if weighted == 'sparse':
        A = sp.csr_matrix(M)
        A.data = np.asarray(X[M > 0], dtype=np.float64)
        return A, sp.csr_matrix(M), W0, T0, k, kw
    return X, M, W0, T0, k, kw


def held_out_score(entries, lo, hi):
    """an early_stop callback for nmf(): the clipped RMSE on the held-out entries of rows [lo, hi), local row indices, carried
    as `device_entries` so that the library scores them (collectively when the call is row-sharded)"""
    vi, vj, vr, clip_lo, clip_hi = entries
    keep = (vi >= lo) & (vi < hi)

    def score(X_ignored, W, T):
        pred = np.clip(np.einsum('ij,ji->i', W[vi[keep] - lo, :], T[:, vj[keep]]), clip_lo, clip_hi)
        return np.sqrt(np.mean((pred - vr[keep]) ** 2))

    score.device_entries = (vi[keep] - lo, vj[keep], vr[keep], clip_lo, clip_hi)
    score.all_entries = entries
    return score


def _init_pg(backend, **kw):
    import datetime
    import torch.distributed as dist
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    dist.init_process_group(backend, init_method='tcp://127.0.0.1:%s' % os.environ['MASTER_PORT'], rank=rank,
                            world_size=world, timeout=datetime.timedelta(seconds=90), **kw)
    return rank, world


def case_group_host_transport(out, name):
    """nmf(X_rows, ..., group=RowGroup.over_torch(...)): the library's own sharded sweep (collectives inside rri_sweep)
    with gloo as the transport of its host-callback communicator; unequal row blocks"""
    import torch.distributed as dist
    from rri_nmf_amd import nmf as nmf_mod
    from rri_nmf_amd.distributed import RowGroup
    rank, world = _init_pg('gloo')
    try:
        X, M, W0, T0, k, kw = nmf_inputs(name)
        n = X.shape[0]
        cut = [0] + [int(round(n * (0.6 if world == 2 else (r + 1.0) / world))) if r < world - 1 else n for r in range(world)]
        lo, hi = cut[rank], cut[rank + 1]
        if rank == 1 and world == 2 and os.environ.get('RRI_TEST_SEED_OTHER_RANK'):
            np.random.seed(12345)         # only rank 0's generator may matter for 'random' resets
        if callable(kw.get('early_stop')):
            kw['early_stop'] = held_out_score(kw['early_stop'].all_entries, lo, hi)      # this rank's rows, local indices
        if kw.get('w_row') is not None:
            kw['w_row'] = kw['w_row'][lo:hi]                                            # this rank's rows
        with RowGroup.over_torch(hi - lo) as grp:
            assert (grp.row_lo, grp.n_global) == (lo, n)
            r = nmf_mod.nmf(X[lo:hi], k, W_mat=None if M is None else M[lo:hi], W_in=W0[lo:hi] if len(W0) else [], T_in=T0, group=grp, **kw)
            obj2 = r['obj_calculator'].true_objective()       # collective re-evaluation on a fresh handle
        np.savez(out, W=r['W'], T=r['T'], obj=np.array(r['obj_history']), obj2=obj2, resets=r['n_resets_used'], lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


def estimator_problem():
    from rri_nmf_amd.synthetic import planted_X
    X = planted_X(1200, 300, 6, seed=9, dtype=np.float64)
    return X / X.sum(1, keepdims=True)            # documents as distributions, what the topic-model estimator expects


def case_group_estimator(out):
    """NMF_TM_Estimator.fit on a row block with nmf_kwargs={'group': ...}: the estimator's own NNDSVD start computed
    row-sharded, 5 sweeps, final projection; then one_iter from its factors"""
    import torch.distributed as dist
    from rri_nmf_amd import sklearn_interface as si
    from rri_nmf_amd.distributed import RowGroup
    rank, world = _init_pg('gloo')
    try:
        X = estimator_problem()
        n, d = X.shape
        lo, hi = (0, 700) if rank == 0 else (700, n)
        with RowGroup.over_torch(hi - lo) as grp:
            E = si.NMF_TM_Estimator(hi - lo, d, 6, random_state=0, max_iter=5,
                                    nmf_kwargs={'group': grp, 'eps_stop': -1, 'device_init': True}).fit(X[lo:hi])
            W5, T5 = E.W.copy(), E.T.copy()
            E.one_iter(X[lo:hi])
        np.savez(out, W=W5, T=T5, W6=E.W, T6=E.T, obj=np.array(E.nmf_outputs['obj_history']), lo=lo, hi=hi)
    finally:
        dist.destroy_process_group()


def rs_estimator_problem():
    """ratings on a 5-star scale: (index pairs, values, start) of a 900 x 260 problem, 20 % observed"""
    rs = np.random.RandomState(11)
    n, d, k = 900, 260, 4
    Wt, Tt = rs.rand(n, k), rs.rand(k, d)
    R = np.clip(np.round(1 + 4 * (Wt @ Tt) / (Wt @ Tt).max()), 1, 5)
    obs = rs.rand(n, d) < 0.2
    ii, jj = np.nonzero(obs)
    W0, T0 = 0.5 * rs.rand(n, k), rs.rand(k, d)
    return np.column_stack((ii, jj)), R[ii, jj], W0, T0, (n, d, k)


def case_group_rs_estimator(out):
    """NMF_RS_Estimator.fit with its DEFAULTS (early stopping on a 5 % hold-out, sklearn_interface.py:71-123) on a row block:
    nmf_kwargs={'group': ...}, warm start given (the weighted start is not computed row-sharded)"""
    import torch.distributed as dist
    from rri_nmf_amd import sklearn_interface as si
    from rri_nmf_amd.distributed import RowGroup
    rank, world = _init_pg('gloo')
    try:
        ij, y, W0, T0, (n, d, k) = rs_estimator_problem()
        lo, hi = (0, 520) if rank == 0 else (520, n)
        mine = (ij[:, 0] >= lo) & (ij[:, 0] < hi)
        if rank == 1:
            y = y.copy()
            y[mine & (y == 1)] = 2          # the lowest rating occurs on rank 0 only: the clip bounds must still be the global ones
        with RowGroup.over_torch(hi - lo) as grp:
            E = si.NMF_RS_Estimator(hi - lo, d, k, W=W0[lo:hi], T=T0, max_iter=12, nmf_kwargs={'group': grp, 'dtype': np.float64})
            E.fit(np.column_stack((ij[mine, 0] - lo, ij[mine, 1])), y[mine])
            score = E.score(np.column_stack((ij[mine, 0] - lo, ij[mine, 1])), y[mine])
        np.savez(out, W=E.W, T=E.T, obj=np.array(E.nmf_outputs['obj_history']), lo=lo, hi=hi, clip=np.array([E.min_rating, E.max_rating]),
                 score=score)
    finally:
        dist.destroy_process_group()


def case_group_rccl_single_rank(out):
    """the in-library RCCL communicator with one rank (world sizes above one need one GPU per rank: the driver's run):
    rri_comm_unique_id / rri_comm_create / rri_attach_comm, then the collective entry points"""
    from conftest import load_golden, relfro
    from rri_nmf_amd import nmf as nmf_mod
    from rri_nmf_amd.distributed import RowGroup
    from rri_nmf_amd.engine import RRIEngine
    res = {}
    exchange = lambda obj: [obj]                       # one rank: no host channel needed at all (no torch.distributed)
    with RowGroup.rccl(1501, device=0, exchange=exchange, rank=0, world=1) as grp:
        X, M, W0, T0 = _problem(1501, 700, 6, False, np.float32)
        with RRIEngine(1501, 700, 6, dtype=np.float32) as e:
            e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
            e.sweep(3)
            Wa, Ta, oa = e.get_W(), e.get_T(), e.objective()
        with RRIEngine(1501, 700, 6, dtype=np.float32) as e:
            e.attach_group(grp)
            e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
            e.sweep(2)
            e.sweep(1)
            Wb, Tb, ob = e.get_W(), e.get_T(), e.objective()
            res['allreduce_calls'] = e.comm_stats()[2]
        res['plain'] = [relfro(Wb, Wa), relfro(Tb, Ta), abs(ob - oa) / abs(oa)]
    # a second group in the same process (communicators come and go with nmf() calls of a host program)
    g = load_golden('g6_rare_branches')
    n, d, k = [int(v) for v in g['shape']]
    from rri_nmf_amd.synthetic import planted_X, scaled_init
    X = planted_X(n, d, k, seed=3, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=4)
    with RowGroup.rccl(n, device=0, exchange=exchange, rank=0, world=1) as grp:
        for key, flags, gW, gT in (('resets_W', dict(t_row_sum=1.0, reg_w_l1=1e6), 'l1killW_mrd_W', 'l1killW_mrd_T'),
                                   ('resets_T', dict(t_row_sum=1.0, reg_t_l1=1e6), 'l1kill_W', 'l1kill_T')):
            r = nmf_mod.nmf(X, k, W_in=W0, T_in=T0, max_iter=1, eps_stop=-1, group=grp, **flags)
            res[key] = [relfro(r['W'], g[gW]), relfro(r['T'], g[gT]), float(r['n_resets_used'])]
        Xw, Mw, W0w, T0w = _problem(700, 260, 4, True, np.float64)
        flags = dict(t_row_sum=1.0, reset_topic_method=None, max_iter=3, eps_stop=-1)
        a = nmf_mod.nmf(Xw, 4, W_mat=Mw, W_in=W0w, T_in=T0w, **flags)
        b = nmf_mod.nmf(Xw, 4, W_mat=Mw, W_in=W0w, T_in=T0w, group=grp.resized([700]), **flags)
        res['weighted'] = [relfro(b['W'], a['W']), relfro(b['T'], a['T']), 0.0]
    with open(out, 'w') as f:
        json.dump(res, f)


def case_group_bad_score_entry(out):
    """rri_masked_rmse is a collective under a communicator: a rank whose OWN entry list is bad (an index out of range) must
    still enter the all-reduce -- its peer would block in it otherwise -- and both ranks must see the failure; the next score,
    with good lists, must work (the ranks are still in step)"""
    import torch.distributed as dist
    from rri_nmf_amd.distributed import RowGroup
    from rri_nmf_amd.engine import RRIEngine
    rank, world = _init_pg('gloo')
    res = {}
    try:
        n, d, k = 400, 90, 3
        X, M, W0, T0 = _problem(n, d, k, False, np.dtype('float64'))
        lo, hi = (0, 240) if rank == 0 else (240, n)
        with RowGroup.over_torch(hi - lo) as grp, RRIEngine(hi - lo, d, k, dtype=np.float64) as e:
            e.attach_group(grp)
            e.upload_X(X[lo:hi]), e.set_W(W0[lo:hi]), e.set_T(T0), e.set_params()
            I = np.arange(0, hi - lo, 7)
            J = (I * 3) % d
            vals = X[lo:hi][I, J]
            Ibad = I.copy()
            if rank == 1:
                Ibad[2] = hi - lo                  # one row beyond this rank's block
            try:
                e.masked_rmse(Ibad, J, vals, 0.0, 10.0)
                res['first'] = 'no error'
            except ValueError as ex:              # RRI_ERR_INVALID
                res['first'] = str(ex)
            res['second'] = e.masked_rmse(I, J, vals, 0.0, 10.0)
            WT = W0.astype(np.float64).dot(T0.astype(np.float64))
            Iall = np.concatenate([np.arange(0, 240, 7), 240 + np.arange(0, n - 240, 7)])
            Jall = np.concatenate([(np.arange(0, 240, 7) * 3) % d, (np.arange(0, n - 240, 7) * 3) % d])
            res['want'] = float(np.sqrt(np.mean((np.clip(WT[Iall, Jall], 0.0, 10.0) - X[Iall, Jall]) ** 2)))
    finally:
        dist.destroy_process_group()
    with open(out, 'w') as f:
        json.dump(res, f)


def case_group_closed(out):
    """a closed RowGroup (and a view of it) must be refused by attach_group instead of detaching the handle"""
    from rri_nmf_amd.distributed import RowGroup
    from rri_nmf_amd.engine import RRIEngine
    res = {}
    grp = RowGroup.rccl(300, device=0, exchange=lambda obj: [obj], rank=0, world=1)
    view = grp.resized([200])
    res['open_before'] = (not grp.closed) and (not view.closed)
    with RRIEngine(200, 50, 3, dtype=np.float64) as e:
        e.attach_group(view)
        res['attached_world'] = e.comm_stats()[1]
    grp.close()
    res['closed_after'] = grp.closed and view.closed
    res['refused'] = []
    for name, g, rows in (('group', grp, 300), ('view', view, 200)):
        with RRIEngine(rows, 50, 3, dtype=np.float64) as e:
            try:
                e.attach_group(g)
            except ValueError as ex:
                if 'closed' in str(ex):
                    res['refused'].append(name)
    grp.close()
    res['idempotent'] = True
    with open(out, 'w') as f:
        json.dump(res, f)


def case_legacy_protocol_single_rank_nccl(out):
    """ShardedRRI (collective in the caller's hands: torch.distributed on the engine's stream) with a one-rank RCCL
    group of torch's: the split step protocol equals rri_sweep, with and without reset events"""
    import torch
    import torch.distributed as dist
    from conftest import load_golden, relfro
    from rri_nmf_amd.distributed import ShardedRRI, make_device_shard
    from rri_nmf_amd.engine import RRIEngine
    from rri_nmf_amd.synthetic import planted_X, scaled_init
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')      # RCCL's bootstrap socket: loopback (see RowGroup.rccl)
    _init_pg('nccl', device_id=torch.device('cuda', 0))
    res = {}
    try:
        n, d, k = 3000, 1100, 5
        X = planted_X(n, d, k, seed=0, dtype=np.float32)
        W0, T0 = scaled_init(X, k, seed=1)
        with RRIEngine(n, d, k, dtype=np.float32) as e:
            e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params()
            e.sweep(3)
            Wa, Ta, obja = e.get_W(), e.get_T(), e.objective()
        eng, red, stream = make_device_shard(n, d, k, dtype=np.float32, device_index=0)
        eng.upload_X(X), eng.set_W(W0), eng.set_T(T0), eng.set_params()
        drv = ShardedRRI(eng, red, k, stream=stream)
        drv.sweep(2)
        drv.sweep(1)
        res['plain'] = [relfro(eng.get_W(), Wa), relfro(eng.get_T(), Ta), abs(drv.objective() - obja) / abs(obja)]
        res['allreduce_calls'] = drv.allreduce_calls
        eng.close()
        g = load_golden('g6_rare_branches')
        n, d, k = [int(v) for v in g['shape']]
        X = planted_X(n, d, k, seed=3, dtype=np.float64)
        W0, T0 = scaled_init(X, k, seed=4)
        for key, flags, gW, gT in (('resets_W', dict(t_row_sum=1.0, reg_w_l1=1e6), 'l1killW_mrd_W', 'l1killW_mrd_T'),
                                   ('resets_T', dict(t_row_sum=1.0, reg_t_l1=1e6), 'l1kill_W', 'l1kill_T')):
            with RRIEngine(n, d, k, dtype=np.float64) as e:
                e.upload_X(X), e.set_W(W0), e.set_T(T0), e.set_params(**flags)
                e.sweep(2)
                Wa, Ta, na = e.get_W(), e.get_T(), e.n_resets_used
            eng, red, stream = make_device_shard(n, d, k, dtype=np.float64, device_index=0)
            eng.upload_X(X), eng.set_W(W0), eng.set_T(T0), eng.set_params(**flags)
            drv = ShardedRRI(eng, red, k, stream=stream, row_lo=0, n_global=n)
            drv.sweep(1)
            first = [relfro(eng.get_W(), g[gW]), relfro(eng.get_T(), g[gT])]
            drv.sweep(1)
            res[key] = first + [relfro(eng.get_W(), Wa), relfro(eng.get_T(), Ta), float(drv.n_resets_used), float(na)]
            eng.close()
    finally:
        dist.destroy_process_group()
    with open(out, 'w') as f:
        json.dump(res, f)


if __name__ == '__main__':
    case, out, log = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(os.path.dirname(os.path.abspath(log)), exist_ok=True)
    logf = open(log, 'w')
    faulthandler.enable(file=logf, all_threads=True)
    faulthandler.dump_traceback_later(HANG_AFTER_S, exit=True, file=logf)
    try:
        args = sys.argv[4:]
        globals()['case_' + case](out, *args)
    except BaseException:
        import traceback
        traceback.print_exc(file=logf)
        logf.flush()
        raise
    faulthandler.cancel_dump_traceback_later()
    logf.write('ok\n')
    logf.close()
