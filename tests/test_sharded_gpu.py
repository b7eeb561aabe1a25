"""Row-sharded stepping on the device with MORE than one shard: two processes share the one GPU of the test box,
each holds a row block of X / W (and of the mask), and the reduce buffers are all-reduced over gloo (RCCL needs
one device per rank; the driver code is the same -- torch.distributed on the engine's stream).  Compared with
one engine holding all rows."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, relfro
from rri_nmf_amd.synthetic import planted_X, scaled_init, observed_mask

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(n, d, k, weighted, store):
    X = planted_X(n, d, k, seed=0, dtype=store)
    M = None
    if weighted:
        M = observed_mask(n, d, 0.3, seed=2, dtype=store)
        X = X * M
    W0, T0 = scaled_init(X, k, seed=1)
    return X, M, W0, T0


def _worker(rank, world, port, n, d, k, sweeps, weighted, store, flags, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from rri_nmf_amd.distributed import ShardedRRI, make_device_shard, shard_rows
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    try:
        X, M, W0, T0 = _problem(n, d, k, weighted, np.dtype(store))
        lo, hi = shard_rows(n, world, rank)
        eng, red, stream = make_device_shard(hi - lo, d, k, dtype=np.dtype(store), device_index=0, weighted=weighted)
        if weighted == 'sparse':
            import scipy.sparse as sp
            A = sp.csr_matrix(M[lo:hi])
            A.data = np.asarray(X[lo:hi][M[lo:hi] > 0], dtype=np.dtype(store))
            eng.upload_observed_csr(A)
        else:
            eng.upload_X(np.ascontiguousarray(X[lo:hi]))
            if weighted:
                eng.upload_mask(np.ascontiguousarray(M[lo:hi]))
        eng.set_W(W0[lo:hi])
        eng.set_T(T0)
        eng.set_params(**flags)
        drv = ShardedRRI(eng, red, k, stream=stream, row_lo=lo, n_global=n,
                         reset_topic_method=flags.get('reset_topic_method', 'max_resid_document'))
        drv.sweep(sweeps - 1)
        drv.sweep(1)
        obj = drv.objective()
        np.savez(os.path.join(out_dir, 'r%d.npz' % rank), W=eng.get_W(), T=eng.get_T(), obj=obj,
                 resets=drv.n_resets_used, calls=drv.allreduce_calls)
        eng.close()
    finally:
        dist.destroy_process_group()


CASES = {
    'plain_f32': (1501, 700, 6, 3, False, 'float32', dict()),
    'topic_model_f64': (900, 333, 5, 3, False, 'float64', dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0)),
    'weighted_f32': (1201, 515, 5, 3, True, 'float32', dict(t_row_sum=1.0, reset_topic_method=None)),
    'weighted_f64_regs': (700, 260, 4, 3, True, 'float64', dict(t_row_sum=1.0, reset_topic_method=None, reg_w_l1=0.01,
                                                               reg_t_l1=0.02)),
    'pattern_only_f64': (1201, 515, 5, 3, 'sparse', 'float64', dict(t_row_sum=1.0, reset_topic_method=None)),
    'pattern_only_f32_resets_T': (600, 200, 4, 2, 'sparse', 'float32', dict(t_row_sum=1.0, reg_t_l1=1e6)),
    'resets_W': (600, 200, 4, 2, False, 'float64', dict(t_row_sum=1.0, reg_w_l1=1e6)),
    'weighted_resets_T': (600, 200, 4, 2, True, 'float64', dict(t_row_sum=1.0, reg_t_l1=1e6)),
}


@pytest.mark.timeout(240)
@pytest.mark.parametrize('name', sorted(CASES))
def test_two_shards_on_one_gpu_match_one_engine(name, tmp_path):
    import torch.multiprocessing as mp
    from rri_nmf_amd.engine import RRIEngine
    n, d, k, sweeps, weighted, store, flags = CASES[name]
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n, d, k, sweeps, weighted, store, flags, str(tmp_path)),
             nprocs=world, join=True)
    X, M, W0, T0 = _problem(n, d, k, weighted, np.dtype(store))
    with RRIEngine(n, d, k, dtype=np.dtype(store), weighted=bool(weighted)) as e:     # all rows, dense arrays
        e.upload_X(X)
        if weighted:
            e.upload_mask(M)
        e.set_W(W0), e.set_T(T0)
        e.set_params(**flags)
        e.sweep(sweeps)
        Wa, Ta, obja, na = e.get_W(), e.get_T(), e.objective(), e.n_resets_used
    parts = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
    W = np.vstack([p['W'] for p in parts])
    assert np.array_equal(parts[0]['T'], parts[1]['T'])              # replicated, bit for bit
    tol = 1e-10 if store == 'float64' or not weighted else 1e-4    # fp32 residual: the schedules round it differently
    assert relfro(W, Wa) < tol and relfro(parts[0]['T'], Ta) < tol, (relfro(W, Wa), relfro(parts[0]['T'], Ta))
    assert int(parts[0]['resets']) == int(parts[1]['resets']) == na
    if 'resets' in name:
        assert na >= k
    # the regularisation terms are the caller's to add (ShardedRRI.objective's arguments); compare the data term
    if not any(flags.get(r) for r in ('reg_w_l1', 'reg_t_l1')):
        assert abs(float(parts[0]['obj']) - obja) <= tol * abs(obja)
