#!/usr/bin/env python3
"""Does the per-process level of the read-only pass (0.640-0.678 ms at config 3) belong to the PROCESS or to the buffer X lives in?
Five copies of the same X in one process (torch allocations made one after the other, 4 GB each), an engine bound to each in turn.
    python3 tools/pass_placement_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402

n, d, k = 100000, 10000, 50
dev = torch.device('cuda', 0)
X0 = device_planted_shard(n, d, k, 0, dev)
copies = [X0]
pads = []
for i in range(4):
    pads.append(torch.empty((37 + 61 * i) * 1024 * 1024 // 4, device=dev))      # odd-sized spacers: other offsets inside the pool
    copies.append(X0.clone())
rng = np.random.RandomState(0)
a = (float(X0[:20000].mean()) / k) ** 0.5
W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
torch.cuda.synchronize()
for rnd in range(3):
    row = []
    for X in copies:
        eng = RRIEngine(n, d, k, dtype=np.float32, device=0)
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        eng.set_W(W0), eng.set_T(T0), eng.set_params()
        eng.sweep(1)
        eng.synchronize()
        eng.timing_enable(True, every=4)
        eng.sweep(2)
        eng.synchronize()
        cnt, ms = eng.timing_read(0)
        row.append(ms / max(cnt, 1))
        eng.close()
    print('round %d: ' % rnd + '   '.join('[copy %d @ %#x] %.4f ms' % (i, X.data_ptr(), v) for i, (X, v) in enumerate(zip(copies, row))), flush=True)
