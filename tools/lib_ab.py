#!/usr/bin/env python3
"""Two builds of librri_hip.so against each other INSIDE one process (the read-modify-write pass has per-process modes that
drown an A/B of two processes): engines of the explicit-residual schedule are made alternately from library A and library B on
the same resident X, the pass timed by HIP events.
    python3 tools/lib_ab.py <libA.so> <libB.so> [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd import _capi                   # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402


def main():
    pa, pb = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    libs = {'A': _capi.load_library(pa), 'B': _capi.load_library(pb)}
    n, d, k = 100000, 10000, 50
    dev = torch.device('cuda', 0)
    X = device_planted_shard(n, d, k, 0, dev)
    rng = np.random.RandomState(0)
    a = (float(X[:20000].mean()) / k) ** 0.5
    W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
    torch.cuda.synchronize()
    for rnd in range(rounds):
        row = []
        for name in ('A', 'B'):
            _capi._lib = libs[name]
            eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule='residual')
            eng.bind_X_device(X.data_ptr(), X.stride(0))
            eng.set_W(W0), eng.set_T(T0), eng.set_params()
            eng.sweep(1)
            eng.synchronize()
            eng.timing_enable(True, every=4)
            eng.sweep(3)
            eng.synchronize()
            cnt, ms = eng.timing_read(3)
            row.append(ms / max(cnt, 1))
            Wn = eng.get_W()
            eng.close()
            row.append(float(np.linalg.norm(Wn)))
        print('round %d: A %.4f ms (%.3f of 8 TB/s)   B %.4f ms (%.3f)   |W| %.10e / %.10e' % (
            rnd, row[0], 8e9 / (row[0] * 1e-3) / 8e12, row[2], 8e9 / (row[2] * 1e-3) / 8e12, row[1], row[3]), flush=True)


if __name__ == '__main__':
    main()
