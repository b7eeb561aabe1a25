#!/usr/bin/env python3
"""Does the mode of the read-modify-write pass follow the PHYSICAL memory behind the residual?  Round 2's test re-created the
handle in one process, but a freed 4 GB block is handed out again for the next request of that size -- every trial had the
same memory.  Here the handles are created while the earlier ones are still alive (every residual on other memory), then
all are destroyed and one more is made.  Pass duration by HIP events (every 4th launch), rotation 0 and 1 of the tile <-> XCD
map (tools/rmw_rot.py) for each.
    python3 tools/rmw_place.py [handles]"""
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
    eng.sweep(3)
    eng.synchronize()
    c1, m1 = eng.timing_read(kid)
    eng.timing_enable(False)
    return (m1 - m0) / max(c1 - c0, 1)


def main():
    handles = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n, d, k = 100000, 10000, 50
    dev = torch.device('cuda', 0)
    X = device_planted_shard(n, d, k, 0, dev)
    rng = np.random.RandomState(0)
    a = (float(X[:20000].mean()) / k) ** 0.5
    W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
    torch.cuda.synchronize()

    def make(rot):
        os.environ['RRI_PASS_ROT'] = str(rot)
        eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule='residual')
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        eng.set_W(W0), eng.set_T(T0), eng.set_params()
        return eng

    alive = []
    for h in range(handles):
        row = []
        for rot in (0, 1):
            eng = make(rot)
            row.append(measure(eng))
            alive.append(eng)
        free, total = torch.cuda.mem_get_info()
        print('handle pair %d (earlier ones alive, %.0f GB in use): rotation 0 %.4f ms, rotation 1 %.4f ms' % (h, (total - free) / 1e9, row[0], row[1]), flush=True)
    print('again, the same handles in the order they were made:', ' '.join('%.4f' % measure(e) for e in alive), flush=True)
    for e in alive:
        e.close()
    eng = make(0)
    print('all destroyed, one more: %.4f ms' % measure(eng), flush=True)
    eng.close()
    # a big spoiler first, so the next residual cannot take the block just freed
    spoiler = torch.empty(6 * 2 ** 30, dtype=torch.uint8, device=dev)
    eng = make(0)
    print('behind a 6 GB spoiler: %.4f ms' % measure(eng), flush=True)
    eng.close()
    del spoiler


if __name__ == '__main__':
    main()
