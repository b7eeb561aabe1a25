#!/usr/bin/env python3
"""Two settings of the library's environment switches against each other INSIDE one process (the read-modify-write pass has
per-process modes that drown an A/B of two processes): the switches are read by rri_create, so engines made alternately under
setting A and setting B on the same resident X compare like with like.  Pass timed by HIP events.
    python3 tools/env_ab.py residual|gram|c5 "VAR=1,VAR2=0" "VAR=0" [rounds [timer id: 0 pass, 1 W column, 2 T-row chain, 3 updating pass]]
A setting may be a list "VAR=1|VAR=2|VAR=3": every alternative then takes its turn in every round (N-way, one process)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402


def main():
    what, sa, sb = sys.argv[1], sys.argv[2], sys.argv[3]
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    timer = int(sys.argv[5]) if len(sys.argv) > 5 else (0 if what == 'gram' else 3)
    n, d, k = 100000, 10000, 50
    dev = torch.device('cuda', 0)
    X = device_planted_shard(n, d, k, 0, dev)
    rng = np.random.RandomState(0)
    a = (float(X[:20000].mean()) / k) ** 0.5
    W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
    Mask = None
    if what == 'c5':
        g = torch.Generator(device=dev)
        g.manual_seed(2)
        Mask = (torch.rand(n, d, device=dev, generator=g) < 0.05).float()
        X.mul_(Mask)
    torch.cuda.synchronize()
    for rnd in range(rounds):
        row = []
        for setting in [x for part in (sa, sb) for x in part.split('|')]:
            pairs = [kv.split('=') for kv in setting.split(',') if kv]
            for kk, vv in pairs:
                os.environ[kk] = vv
            eng = RRIEngine(n, d, k, dtype=np.float32, device=0, weighted=(what == 'c5'), schedule='residual' if what == 'residual' else 'gram')
            for kk, vv in pairs:
                os.environ.pop(kk, None)
            eng.bind_X_device(X.data_ptr(), X.stride(0))
            if Mask is not None:
                eng.bind_mask_device(Mask.data_ptr(), Mask.stride(0))
            eng.set_W(W0), eng.set_T(T0)
            eng.set_params(**(dict(t_row_sum=1.0, reset_topic_method=None) if what == 'c5' else {}))
            eng.sweep(1)
            eng.synchronize()
            eng.timing_enable(True, every=4)
            eng.sweep(2)
            eng.synchronize()
            cnt, ms = eng.timing_read(timer)
            row.append(ms / max(cnt, 1))
            eng.close()
        names = [x for part in (sa, sb) for x in part.split('|')]
        print('round %d: ' % rnd + '   '.join('[%s] %.4f ms' % (nm, v) for nm, v in zip(names, row)), flush=True)


if __name__ == '__main__':
    main()
