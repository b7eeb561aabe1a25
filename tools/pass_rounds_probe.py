#!/usr/bin/env python3
"""The read-only pass at d = 10000, k = 50 over row counts that make whole and partial ROUNDS of resident workgroups (4 per CU,
1024 on the chip, 560 rows each): does a partly filled last round cost what config 3 (1.75 rounds) loses against config 4
(17 rounds)?   python3 tools/pass_rounds_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rri_nmf_amd.engine import RRIEngine        # noqa: E402

d, k = 10000, 50
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev)
g.manual_seed(0)
Xall = torch.rand(229376, d, device=dev, generator=g)
rng = np.random.RandomState(0)
for rnd in range(2):
    for n in (2240, 5600, 11200, 28560, 57120, 85680, 100000, 114240, 171360, 228480):       # 4 ... 408 row blocks of 560 rows x 10 panels
        X = Xall[:n]
        eng = RRIEngine(n, d, k, dtype=np.float32, device=0)
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        eng.set_W(0.1 * rng.rand(n, k)), eng.set_T(0.1 * rng.rand(k, d))
        eng.set_params()
        eng.sweep(1)
        eng.synchronize()
        eng.timing_enable(True, every=4)
        eng.sweep(2)
        eng.synchronize()
        cnt, ms = eng.timing_read(0)
        per = ms / max(cnt, 1)
        print('round %d  n %6d  (%5.2f rounds of 1024 workgroups)  pass %.4f ms  %.3f TB/s' % (rnd, n, -(-n // 560) * 10 / 1024.0, per, n * d * 4 / per / 1e9), flush=True)
        eng.close()
