#!/usr/bin/env python3
"""Does something ELSE alive in the process set the mode of the read-modify-write pass?  bench.py's default run times the
explicit-residual schedule while the default schedule's handle is still alive (slow mode there in every line on file), its
--schedule residual run has one handle (fast in most).  One residual handle, measured: alone | with a second (idle) handle
alive | with only a second HIP stream alive | with only a 4 GB torch tensor alive | alone again.
    python3 tools/rmw_second_engine.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402


def measure(eng, kid=3):
    eng.sweep(1)
    eng.synchronize()
    eng.timing_enable(True, every=4)
    c0, m0 = eng.timing_read(kid)
    eng.sweep(2)
    eng.synchronize()
    c1, m1 = eng.timing_read(kid)
    eng.timing_enable(False)
    return (m1 - m0) / max(c1 - c0, 1)


def main():
    n, d, k = 100000, 10000, 50
    dev = torch.device('cuda', 0)
    X = device_planted_shard(n, d, k, 0, dev)
    rng = np.random.RandomState(0)
    a = (float(X[:20000].mean()) / k) ** 0.5
    W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
    torch.cuda.synchronize()

    def make(schedule):
        eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule=schedule)
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        eng.set_W(W0), eng.set_T(T0), eng.set_params()
        return eng

    r = make('residual')
    print('residual handle alone:                          %.4f ms' % measure(r), flush=True)
    g = make('gram')
    print('+ an idle default-schedule handle alive:         %.4f ms' % measure(r), flush=True)
    g.sweep(1); g.synchronize()
    print('+ after that handle ran a sweep:                 %.4f ms' % measure(r), flush=True)
    g.close()
    print('that handle destroyed:                           %.4f ms' % measure(r), flush=True)
    s = torch.cuda.Stream()
    print('+ only a second HIP stream alive:                %.4f ms' % measure(r), flush=True)
    del s
    t = torch.empty(4 * 2 ** 30, dtype=torch.uint8, device=dev)
    print('+ only a 4 GB tensor alive:                      %.4f ms' % measure(r), flush=True)
    del t
    torch.cuda.empty_cache()
    print('alone again:                                     %.4f ms' % measure(r), flush=True)
    r2 = make('residual')
    print('a second residual handle, the first alive:       %.4f ms (first: %.4f)' % (measure(r2), measure(r)), flush=True)
    r.close()
    print('the second alone:                                %.4f ms' % measure(r2), flush=True)
    r2.close()


if __name__ == '__main__':
    main()
