"""Determinism soak: the same run twice (many sweeps, several flavours) must give the same bits."""
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, '.')
from rri_nmf_amd.engine import RRIEngine
from rri_nmf_amd.synthetic import planted_X, scaled_init
n, d, k = 10000, 1000, 20
X = planted_X(n, d, k, seed=0, dtype=np.float32)
W0, T0 = scaled_init(X.astype(np.float64), k, seed=1)
M = (np.random.RandomState(2).rand(n, d) < 0.1).astype(np.float32)
A = sp.csr_matrix(M); A.data = (X * M)[M > 0]
for name, weighted, flags, sweeps in (('plain', False, {}, 1500), ('topic model', False, dict(project_T_each_iter=True, t_row_sum=1.0, w_row_sum=1.0), 1000),
                                      ('T fixed (whole-sweep launches)', False, dict(fix_T=True, reg_w_l1=0.01), 1000),
                                      ('weighted dense', True, dict(t_row_sum=1.0, reset_topic_method=None), 200),
                                      ('weighted pattern-only', 'sparse', dict(t_row_sum=1.0, reset_topic_method=None), 400)):
    res = []
    for rep in range(2):
        with RRIEngine(n, d, k, dtype=np.float32, weighted=weighted) as e:
            if weighted == 'sparse':
                e.upload_observed_csr(A)
            else:
                e.upload_X(X * M if weighted else X)
                if weighted:
                    e.upload_mask(M)
            e.set_W(W0); e.set_T(T0); e.set_params(**flags)
            for _ in range(sweeps // 100):
                e.sweep(100)
            res.append((e.get_W(), e.get_T(), e.objective()))
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]
    print('%-32s %5d sweeps twice: %s   objective %.10e  finite %s' % (name, sweeps, 'bit-identical' if same else 'DIFFERENT', res[0][2], bool(np.isfinite(res[0][0]).all())))
    assert same
