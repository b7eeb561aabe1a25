#!/usr/bin/env python3
"""Two (or more) PROCESSES on one GPU, each sweeping a launch-bound problem through the persistent kernel: its grid wants every
CU, so two such grids dispatched at the same time can each hold a part of the chip and wait for the rest.  Inside one process
the launches are ordered (an event); across processes nothing orders them: the bounded polls give up and the call falls back.
This script counts how often that happens and what it costs.
    python3 tools/onchip_two_processes.py [processes calls sweeps_per_call]      (the parent starts the children)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(calls, sweeps):
    import numpy as np
    from rri_nmf_amd.engine import RRIEngine
    from rri_nmf_amd.synthetic import planted_X, scaled_init
    n, d, k = 10000, 1000, 20
    X = planted_X(n, d, k, seed=1, dtype=np.float32)
    W0, T0 = scaled_init(X, k, seed=2)
    fallbacks, handles, worst, onchip_calls = 0, 0, 0.0, 0
    t_all = time.perf_counter()
    eng = None
    W, T = W0, T0
    for c in range(calls):
        if c % 20 == 0:                                     # a new handle every 20 calls, as a process that calls nmf() again and
            if eng is not None:                             # again does; the run goes on from where the last handle stood
                fallbacks += eng.onchip_fallbacks()
                W, T = eng.get_W(), eng.get_T()
                eng.close()
            eng = RRIEngine(n, d, k, dtype=np.float32)
            eng.upload_X(X), eng.set_W(W), eng.set_T(T), eng.set_params()
            handles += 1
        before = eng.onchip_info()[1]
        t0 = time.perf_counter()
        eng.sweep(sweeps)
        worst = max(worst, time.perf_counter() - t0)
        onchip_calls += eng.onchip_info()[1] > before
    fallbacks += eng.onchip_fallbacks()
    wall = time.perf_counter() - t_all
    obj = eng.objective()
    eng.close()
    print('pid %d: %d calls of %d sweeps in %.2f s (%.1f sweeps/s), handles %d, calls on the persistent path %d, fallbacks %d, '
          'longest call %.3f s, objective %.12e' % (os.getpid(), calls, sweeps, wall, calls * sweeps / wall, handles, onchip_calls, fallbacks, worst, obj),
          flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        calls = int(sys.argv[2]) if len(sys.argv) > 2 else 100
        sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
        for nproc in (1, procs):
            print('--- %d process(es)' % nproc, flush=True)
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'child', str(calls), str(sweeps)]) for _ in range(nproc)]
            for p in ps:
                p.wait(timeout=600)
