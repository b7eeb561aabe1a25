"""World-size-2 test of the row-sharded driver over gloo on the CPU.  The device engine is replaced
by a numpy stand-in that implements the same step protocol (partial sums -> all-reduce -> finish)
with the oracle's arithmetic, so what is under test is the driver: shard bounds, the reduce-buffer
layout [w^T X | w^T W | ||w||^2 | sum W[:,t-1]], the order of collectives, the deferred column check
and the assembly of the global objective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from oracle import rri_oracle as orc
from rri_nmf_amd.distributed import ShardedRRI, shard_rows
from rri_nmf_amd.synthetic import planted_X, scaled_init


class NumpyShardEngine(object):
    """test double of RRIEngine's row-sharded stepping (unweighted flavour, both halves free), including the
    halt-on-reset behaviour of the device queue: after an event every step is a no-op until it is resolved"""

    def __init__(self, X, W, T, red, reg_w_l1=0.0, resets=True):
        self.X, self.W, self.T, self.red = X, W.copy(), T.copy(), red
        self.n, self.d, self.k = X.shape[0], X.shape[1], W.shape[1]
        self.reg_w_l1 = reg_w_l1
        self.resets = resets
        self.pending = None        # (topic, local sum of the last updated W column)
        self.event = None          # (kind, topic, resume_topic)
        self.failed = None
        self.n_resets_used = 0

    def topic_reduce_local(self, t):
        if self.event:
            return
        w = self.W[:, t]
        buf = np.zeros(self.red.numel())
        buf[:self.d] = w @ self.X
        buf[self.d:self.d + self.k] = w @ self.W
        buf[self.d + self.k] = w @ w
        buf[self.d + self.k + 1] = 0.0 if self.pending is None else self.pending[1]
        self.red.copy_(torch.from_numpy(buf))

    def _check(self, r, pos):
        if self.pending is None:
            return True
        tprev, _ = self.pending
        sw = r[self.d + self.k + 1]
        self.pending = None
        if sw <= 1e-10 and self.resets:
            self.event = (2, tprev, pos)
            return False
        if not sw > 0:
            self.failed = 'W[:, t] sums to 0'
        return True

    def topic_finish(self, t):
        if self.event:
            return
        r = self.red.numpy()
        if not self._check(r, max(t, 0)) or t < 0:
            return
        z, g, nw = r[:self.d].copy(), r[self.d:self.d + self.k].copy(), float(r[self.d + self.k])
        g[t] = 0
        self.T[t, :], _ = orc.qf_min(-(z - g @ self.T), nw, s=None, ub=None)
        self.topic_finish_w(t)

    def topic_finish_w(self, t):
        if self.event:
            return
        Rt, nt = orc.residual_products_W(self.X, self.W, self.T, t)
        self.W[:, t], _ = orc.qf_min(-(Rt - self.reg_w_l1), nt, s=None, ub=None)
        self.pending = (t, float(self.W[:, t].sum()))

    def poll(self):
        if self.failed:
            raise AssertionError(self.failed)
        return 1 if self.event else 0

    def pending_event(self):
        return self.event

    def resid_row_argmax(self):
        Rp = np.maximum(self.X - self.W @ self.T, 0)
        norms = (Rp ** 2).sum(1)
        i = int(np.argmax(norms))
        return float(norms[i]), i

    def reset_row(self, i):
        return np.maximum(self.X[i] - self.W[i] @ self.T, 0)

    def apply_reset_vectors(self, t, T_row, W_col):
        self.T[t, :] = T_row
        self.W[:, t] = W_col
        self.event = None
        self.n_resets_used += 1

    def objective_parts(self):
        R = self.X - self.W @ self.T
        return [0.5 * float((R ** 2).sum()), float((self.W ** 2).sum()), float(np.abs(self.W).sum())]

    def t_norms(self):
        return float((self.T ** 2).sum()), float(np.abs(self.T).sum())


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, k, sweeps, out_dir, reg_w_l1=0.0):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    lo, hi = shard_rows(n, world, rank)
    red = torch.zeros(d + k + 2 + 2, dtype=torch.float64)
    eng = NumpyShardEngine(X[lo:hi], W0[lo:hi], T0, red, reg_w_l1=reg_w_l1)
    drv = ShardedRRI(eng, red, k, row_lo=lo, n_global=n)
    drv.sweep(sweeps)
    obj = drv.objective(reg_w_l1=0.1, reg_w_l2=0.2, reg_t_l1=0.3, reg_t_l2=0.4)
    np.savez(os.path.join(out_dir, 'r%d.npz' % rank), W=eng.W, T=eng.T, lo=lo, hi=hi, obj=obj,
             calls=drv.allreduce_calls, resets=drv.n_resets_used)
    dist.destroy_process_group()


def test_shard_rows_cover_everything():
    for n, w in [(10, 3), (7, 8), (100000, 8), (5, 1)]:
        cuts = [shard_rows(n, w, r) for r in range(w)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sweeps_match_single_process(tmp_path):
    n, d, k, sweeps, world = 301, 64, 5, 3, 2
    mp.spawn(_worker, args=(world, _free_port(), n, d, k, sweeps, str(tmp_path)), nprocs=world, join=True)
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    ref = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=sweeps, eps_stop=-1)
    parts = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
    W = np.vstack([p['W'] for p in parts])
    assert np.linalg.norm(W - ref['W']) / np.linalg.norm(ref['W']) < 1e-10
    for p in parts:     # T is replicated and identical on every rank
        assert np.linalg.norm(p['T'] - ref['T']) / np.linalg.norm(ref['T']) < 1e-10
    assert np.array_equal(parts[0]['T'], parts[1]['T'])
    want = orc.true_objective(X, ref['W'], ref['T'], reg_w_l1=0.1, reg_w_l2=0.2, reg_t_l1=0.3, reg_t_l2=0.4)
    assert abs(float(parts[0]['obj']) - want) < 1e-9 * abs(want)
    # one all-reduce per topic step, plus the one the final column check rides on
    assert int(parts[0]['calls']) == sweeps * k + 1


@pytest.mark.timeout(300)
def test_two_rank_resets_are_resolved_collectively(tmp_path):
    """a huge l1 penalty kills every W column: each topic step ends in a 'max_resid_document' reset
    (nmf.py:804-810) whose winning row lives on one rank and must reach both"""
    n, d, k, world = 157, 40, 4, 2
    mp.spawn(_worker, args=(world, _free_port(), n, d, k, 1, str(tmp_path), 1e6), nprocs=world, join=True)
    X = planted_X(n, d, k, seed=0, dtype=np.float64)
    W0, T0 = scaled_init(X, k, seed=1)
    ref = orc.nmf(X, k, W_in=W0.copy(), T_in=T0.copy(), max_iter=1, eps_stop=-1, reg_w_l1=1e6)
    assert ref['n_resets_used'] == k
    parts = [np.load(os.path.join(str(tmp_path), 'r%d.npz' % r)) for r in range(world)]
    W = np.vstack([p['W'] for p in parts])
    assert int(parts[0]['resets']) == k and int(parts[1]['resets']) == k
    assert np.linalg.norm(W - ref['W']) <= 1e-10 * np.linalg.norm(ref['W'])
    assert np.linalg.norm(parts[0]['T'] - ref['T']) <= 1e-10 * np.linalg.norm(ref['T'])
    assert np.array_equal(parts[0]['T'], parts[1]['T'])
