#!/usr/bin/env python3
"""tools/rmw_place.py found the two modes of the read-modify-write pass INSIDE one process: the first explicit-residual handle
made after X was generated runs the pass in ~1.49 ms, one made after a few handles have come and gone in 1.33 ms.  This
script is that sequence, short, for counter passes (rocprofv3 --pmc ... -- python3 tools/rmw_first_last.py) and for the
allocation switch RRI_MALLOC_CONTIGUOUS=1:   first handle | three more while it lives | all destroyed | last handle.
Prints the pass duration of each by HIP events and brackets the first and the last with marker kernels' worth of sweeps so
that the two can be told apart in a kernel trace (first: 2 sweeps, last: 3 sweeps)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402


def measure(eng, sweeps, kid=3):
    eng.sweep(1)
    eng.synchronize()
    eng.timing_enable(True, every=4)
    c0, m0 = eng.timing_read(kid)
    eng.sweep(sweeps)
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

    def make():
        eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule='residual')
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        eng.set_W(W0), eng.set_T(T0), eng.set_params()
        return eng

    first = make()
    print('first handle (right after X was generated): pass %.4f ms' % measure(first, 2), flush=True)
    more = [make() for _ in range(3)]
    print('three more while it lives: %s ms' % ' '.join('%.4f' % measure(e, 1) for e in more), flush=True)
    print('the first again: %.4f ms' % measure(first, 1), flush=True)
    for e in [first] + more:
        e.close()
    last = make()
    print('all destroyed, a last handle: pass %.4f ms' % measure(last, 3), flush=True)
    last.close()


if __name__ == '__main__':
    main()
