#!/usr/bin/env python3
"""Is the per-process level of the passes the XCD the process's workgroups start on?  Prints, for the process: the XCC_ID of the
first workgroups of a launch on an engine's stream (twice), then the read-only pass and the rank-one update pass timed with the
tile rotation RRI_PASS_ROT = 0 .. 7 (engines made in turn on the same X).     python3 tools/xcc_mode_probe.py [gram|residual]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else 'gram'
var = 'RRI_PASS_ROT'
n, d, k = 100000, 10000, 50
dev = torch.device('cuda', 0)
X = device_planted_shard(n, d, k, 0, dev)
rng = np.random.RandomState(0)
a = (float(X[:20000].mean()) / k) ** 0.5
W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
torch.cuda.synchronize()
os.environ['RRI_ONCHIP'] = '0'
for rnd in range(2):
    row = []
    for rot in range(8):
        os.environ[var] = str(rot)
        eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule='residual' if what == 'residual' else 'gram')
        os.environ.pop(var, None)
        x1 = eng.debug_xcc(16)
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        eng.set_W(W0), eng.set_T(T0), eng.set_params()
        eng.sweep(1)
        eng.synchronize()
        eng.timing_enable(True, every=4)
        eng.sweep(2)
        eng.synchronize()
        cnt, ms = eng.timing_read(3 if what == 'residual' else 0)
        x2 = eng.debug_xcc(16)
        row.append((rot, ms / max(cnt, 1), x1[:8], x2[:8]))
        eng.close()
    print('round %d (%s, %s = 0..7): ' % (rnd, what, var) + '  '.join('%d: %.4f ms' % (r, v) for r, v, _, _ in row), flush=True)
    print('   XCC of workgroups 0..7 before / after, per engine: ' + ' | '.join('%s/%s' % (''.join(map(str, a)), ''.join(map(str, b))) for _, _, a, b in row), flush=True)
