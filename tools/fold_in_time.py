#!/usr/bin/env python3
"""Sweeps with T fixed (the fold-in of new rows, sklearn_interface.py:327-333): the whole-sweep launch k_wsweep_rows against the
launch-per-topic W half (RRI_WSWEEP=0), engines made alternately in one process on the same resident X.
    python3 tools/fold_in_time.py [sweeps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402


def main():
    sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device('cuda', 0)
    os.environ['RRI_ONCHIP'] = '0'
    for n, d, k in ((100000, 10000, 50), (10000, 1000, 20), (2000, 1000, 20)):
        X = device_planted_shard(n, d, k, 0, dev)
        rng = np.random.RandomState(0)
        a = (float(X[:2000].mean()) / k) ** 0.5
        W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
        for rnd in range(2):
            row = []
            for sw in ('1', '0'):
                os.environ['RRI_WSWEEP'] = sw
                eng = RRIEngine(n, d, k, dtype=np.float32, device=0)
                eng.bind_X_device(X.data_ptr(), X.stride(0))
                eng.set_W(W0), eng.set_T(T0)
                eng.set_params(fix_T=True, t_row_sum=1.0, w_row_sum=1.0)
                t0 = time.perf_counter()
                eng.sweep(1)                 # X T^T and T T^T are taken here, once
                eng.synchronize()
                t1 = time.perf_counter()
                eng.sweep(sweeps)
                eng.synchronize()
                t2 = time.perf_counter()
                o = eng.objective()
                row.append((1e3 * (t1 - t0), 1e3 * (t2 - t1) / sweeps, o))
                eng.close()
            print('%d x %d k=%d round %d: whole-sweep launch: first sweep %.3f ms, then %.4f ms/sweep | launch per topic: %.3f ms, %.4f ms/sweep'
                  ' | objectives %.10g %.10g' % (n, d, k, rnd, row[0][0], row[0][1], row[1][0], row[1][1], row[0][2], row[1][2]), flush=True)
        del X
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
