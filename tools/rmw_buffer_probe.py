#!/usr/bin/env python3
"""Are the speeds of the read-modify-write pass a property of the PROCESS or of the residual BUFFER?  Six handles of the
explicit-residual schedule alive at once in one process (six residual buffers, 4 GB each, at six places), each calibrated over the
8 tile rotations (RRI_ROT_DEBUG prints the null-update time per rotation).     python3 tools/rmw_buffer_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402

os.environ['RRI_ROT_DEBUG'] = '1'
os.environ.setdefault('RRI_ROT_CAL', '8')
n, d, k = 100000, 10000, 50
dev = torch.device('cuda', 0)
X = device_planted_shard(n, d, k, 0, dev)
rng = np.random.RandomState(0)
a = (float(X[:20000].mean()) / k) ** 0.5
W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
torch.cuda.synchronize()
engines = []
for i in range(6):
    eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule='residual')
    eng.bind_X_device(X.data_ptr(), X.stride(0))
    eng.set_W(W0), eng.set_T(T0), eng.set_params()
    sys.stderr.write('--- handle %d\n' % i)
    sys.stderr.flush()
    eng.sweep(1)          # calibrates: eight lines on stderr
    eng.synchronize()
    eng.timing_enable(True, every=4)
    eng.sweep(1)
    eng.synchronize()
    cnt, ms = eng.timing_read(3)
    sys.stderr.write('    handle %d: the sweep after it ran its pass at %.4f ms\n' % (i, ms / max(cnt, 1)))
    engines.append(eng)
    if i == 2:
        spacer = torch.empty(3 * 1024 * 1024 * 1024 // 4 + 12345, device=dev)      # 3 GB between the third and the fourth buffer
for e in engines:
    e.close()
