#!/usr/bin/env python3
"""Does the mode of the read-modify-write pass (k_pass<UPD=2>: 1.33 or 1.5 ms per 8 GB at C3, fixed for the life of a process,
not thermal: profiles/r03_rmw_timeline.log) depend on WHICH XCD works on which tile?  Workgroups go to the 8 XCDs round-robin by
blockIdx.x, starting from an XCD that is not fixed.  RRI_PASS_ROT=r makes every workgroup take the tile of its r-th neighbour in
its group of 8 -- the same tiles, the same kernel, every tile on another XCD.  One process, one resident X, a fresh handle
(and residual) per rotation; the pass by HIP events, the read-only pass of the default schedule beside it.
    python3 tools/rmw_rot.py [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n, d, k = 100000, 10000, 50
    dev = torch.device('cuda', 0)
    X = device_planted_shard(n, d, k, 0, dev)
    rng = np.random.RandomState(0)
    a = (float(X[:20000].mean()) / k) ** 0.5
    W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
    torch.cuda.synchronize()
    for rnd in range(rounds):
        for rot in range(8):
            os.environ['RRI_PASS_ROT'] = str(rot)
            row = []
            for schedule, kid in (('residual', 3), ('gram', 0)):
                eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule=schedule)
                eng.bind_X_device(X.data_ptr(), X.stride(0))
                eng.set_W(W0), eng.set_T(T0), eng.set_params()
                eng.sweep(1)
                eng.synchronize()
                eng.timing_enable(True, every=4)
                eng.sweep(3)
                eng.synchronize()
                cnt, ms = eng.timing_read(kid)
                row.append(ms / max(cnt, 1))
                eng.close()
            print('round %d  rotation %d:  read-modify-write pass %.4f ms (%.3f of 8 TB/s)   read-only pass %.4f ms (%.3f)'
                  % (rnd, rot, row[0], 8e9 / (row[0] * 1e-3) / 8e12, row[1], 4e9 / (row[1] * 1e-3) / 8e12), flush=True)
    os.environ.pop('RRI_PASS_ROT', None)


if __name__ == '__main__':
    main()
