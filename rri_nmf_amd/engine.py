"""RRIEngine: one device handle (= one nmf() call, or one row shard of it) behind a
small Python object.  All arithmetic happens in librri_hip.so; this class only moves
arguments, resolves the rare reset events and maps status codes to the exceptions
the reference raises (error conventions).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import Params, Event

EPS_DIV = float(np.spacing(10))  # nmf.py:52

_NP2RRI = {np.dtype(np.float32): _capi.RRI_F32, np.dtype(np.float64): _capi.RRI_F64}
_RESET_CODES = {None: _capi.RESET_NONE, 'max_resid_document': _capi.RESET_MAX_RESID_DOCUMENT,
                'random': _capi.RESET_RANDOM}


def _as_host(a, what):
    """C-contiguous float32/float64 view or copy of a matrix (never mutates the caller's)."""
    a = np.asarray(a)
    if a.dtype not in _NP2RRI:
        a = a.astype(np.float64)
    if a.ndim != 2:
        raise ValueError('%s must be a 2-d array' % what)
    return np.ascontiguousarray(a)


class RRIEngine(object):
    def __init__(self, n, d, k, dtype=np.float32, weighted=False, device=0, stream=None, schedule='gram'):
        """schedule (unweighted handles): 'gram' -- the residual is never formed, one read of X per topic step
        (the reference's form, nmf.py:670-676, 728-734) -- or 'residual' -- R = X - W T is kept in HBM and every topic
        step is one rank-one residual update pass fused with the residual products (RRI_UNWEIGHTED_RESIDUAL)"""
        self._lib = _capi.load_library()
        self.n, self.d, self.k = int(n), int(d), int(k)
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        if self.dtype not in _NP2RRI:
            raise ValueError('dtype must be float32 or float64')
        # weighted: False | True (dense W_mat) | 'sparse' (0/1 W_mat given as a CSR pattern, upload_observed_csr)
        self.sparse = weighted == 'sparse'
        self.weighted = bool(weighted)
        if schedule not in ('gram', 'residual'):
            raise ValueError("schedule must be 'gram' or 'residual'")
        if schedule == 'residual' and self.weighted:
            raise ValueError('the weighted flavour always keeps its (masked) residual; schedule applies to unweighted handles')
        self.schedule = schedule
        self._h = C.c_void_p()
        self.n_resets_used = 0
        self.reset_log = []
        self.fix_reset_seed = False
        self._reset_method = None
        self.group = None
        st = self._lib.rri_create(C.byref(self._h), self.n, self.d, self.k, _NP2RRI[self.dtype],
                                  3 if schedule == 'residual' else 2 if self.sparse else int(self.weighted), int(device),
                                  C.c_void_p(stream or 0))
        if st != _capi.RRI_OK:
            msg = self._lib.rri_last_error(None)
            self._h = C.c_void_p()
            raise _capi.RRIHipUnavailable('rri_create failed (%d): %s'
                                          % (st, msg.decode() if msg else '?'))

    def begin_run(self):
        """a handle kept from an earlier nmf() call starts another one: the per-run counters of the host side"""
        self.n_resets_used = 0
        self.reset_log = []

    # ---- plumbing -----------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.rri_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _err(self):
        m = self._lib.rri_last_error(self._h)
        return m.decode() if m else ''

    def _check(self, st):
        if st >= 0:
            return st
        msg = self._err()
        if st == _capi.RRI_ERR_UNBOUNDED:
            raise ValueError('Minimum objective is unbounded. ' + msg)  # optimization.py:105-107
        if st == _capi.RRI_ERR_W_COL_ZERO:
            raise AssertionError('W[:, t] sums to 0')                    # nmf.py:476
        if st == _capi.RRI_ERR_NOT_IMPLEMENTED:
            raise NotImplementedError(msg)                               # optimization.py:72-73
        if st == _capi.RRI_ERR_INVALID:
            raise ValueError(msg)
        if st == _capi.RRI_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        if st == _capi.RRI_ERR_COMM:
            raise RuntimeError('collective failed: ' + msg)
        raise RuntimeError('librri_hip: status %d: %s' % (st, msg))

    # ---- row-sharded runs -----------------------------------------------------------------
    def attach_group(self, group):
        """this handle holds rows [group.row_lo, group.row_lo + n) of the group's n_global-row problem: from here on
        sweep(), update_*(), objective() and the reset events are collective calls (distributed.RowGroup)"""
        if group is None:
            self._check(self._lib.rri_attach_comm(self._h, None, 0, 0))
            self.group = None
            return
        if group.closed:
            # NULL would DETACH (rri_attach_comm): a later objective() would silently be this rank's share only
            raise ValueError('the RowGroup has been closed: a group must outlive every handle and obj_calculator that uses it')
        if group.n_local != self.n:
            raise ValueError('the group registered %d rows for this rank, the handle has %d' % (group.n_local, self.n))
        self._check(self._lib.rri_attach_comm(self._h, group._comm, group.row_lo, group.n_global))
        self.group = group

    def comm_broadcast(self, values, root=0):
        """`values` (float64 array) of rank `root` on every rank; identity without a group"""
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(self._lib.rri_comm_broadcast(self._h, v.ctypes.data_as(C.POINTER(C.c_double)), v.size, int(root)))
        return v

    def comm_sum(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        self._check(self._lib.rri_comm_allreduce_sum(self._h, v.ctypes.data_as(C.POINTER(C.c_double)), v.size))
        return v

    def comm_stats(self):
        r, w, n = C.c_int32(0), C.c_int32(1), C.c_int64(0)
        self._check(self._lib.rri_comm_stats(self._h, C.byref(r), C.byref(w), C.byref(n)))
        return int(r.value), int(w.value), int(n.value)

    # ---- data ---------------------------------------------------------------------------
    def upload_X(self, X):
        X = _as_host(X, 'X')
        if X.shape != (self.n, self.d):
            raise ValueError('X has wrong dimensions')
        self._check(self._lib.rri_upload_X(self._h, X.ctypes.data, X.strides[0] // X.itemsize, _NP2RRI[X.dtype]))

    def upload_mask(self, M):
        M = _as_host(M, 'W_mat')
        if M.shape != (self.n, self.d):
            raise ValueError('W_mat has wrong dimensions')
        self._check(self._lib.rri_upload_mask(self._h, M.ctypes.data, M.strides[0] // M.itemsize, _NP2RRI[M.dtype]))

    def _csr_args(self, A):
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        if A.shape != (self.n, self.d):
            raise ValueError('sparse matrix has wrong dimensions')
        if not A.has_canonical_format:      # never reorder the caller's arrays in place
            A = A.copy()
            A.sum_duplicates()
        return self._csr_args_raw(A)

    @staticmethod
    def _csr_args_raw(A):
        data = A.data if A.data.dtype in _NP2RRI else A.data.astype(np.float64)
        indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(A.indices, dtype=np.int32)
        data = np.ascontiguousarray(data)
        keep = (indptr, indices, data)   # alive for the duration of the call
        return keep, (indptr.ctypes.data_as(C.POINTER(C.c_int64)), indices.ctypes.data_as(C.POINTER(C.c_int32)),
                      data.ctypes.data, int(A.nnz), _NP2RRI[data.dtype])

    def upload_X_csr(self, A):
        """X from a scipy sparse matrix, densified on the device (no host .toarray())"""
        keep, args = self._csr_args(A)
        self._check(self._lib.rri_upload_X_csr(self._h, *args))

    def upload_mask_csr_pattern(self, A):
        """W_mat = [A != 0] of a scipy sparse matrix, bit-packed on the device"""
        keep, args = self._csr_args(A)
        self._check(self._lib.rri_upload_mask_csr_pattern(self._h, *args))

    def upload_observed_csr(self, A):
        """sparse-pattern handles: the stored entries of the scipy sparse matrix A are the observed ones
        (W_mat = 1 there, explicit zeros included), A's values are X on them"""
        import scipy.sparse as sp
        A = sp.csr_matrix(A)
        if A.shape != (self.n, self.d):
            raise ValueError('sparse matrix has wrong dimensions')
        if not A.has_canonical_format:      # never reorder the caller's arrays in place
            A = A.copy()
            A.sum_duplicates()
        keep, args = self._csr_args_raw(A)
        self._check(self._lib.rri_upload_observed_csr(self._h, *args))

    def bind_X_device(self, ptr, ld):
        """X already on the device (row stride ld elements); the call synchronises the device once, later writes
        to that memory are the caller's to order against the handle's stream"""
        self._check(self._lib.rri_bind_X_device(self._h, C.c_void_p(ptr), int(ld)))

    def bind_mask_device(self, ptr, ld):
        self._check(self._lib.rri_bind_mask_device(self._h, C.c_void_p(ptr), int(ld)))

    def set_W(self, W):
        W = _as_host(W, 'W')
        if W.shape != (self.n, self.k):
            raise ValueError('W_in has wrong dimensions, must be n*k')   # nmf.py:853-854
        self._check(self._lib.rri_set_W(self._h, W.ctypes.data, W.strides[0] // W.itemsize, _NP2RRI[W.dtype]))

    def set_T(self, T):
        T = _as_host(T, 'T')
        if T.shape != (self.k, self.d):
            raise ValueError('T_in has wrong dimensions, must be k*d')   # nmf.py:858-859
        self._check(self._lib.rri_set_T(self._h, T.ctypes.data, T.strides[0] // T.itemsize, _NP2RRI[T.dtype]))

    def get_W(self, dtype=np.float64):
        out = np.empty((self.n, self.k), dtype=dtype)
        self._check(self._lib.rri_get_W(self._h, out.ctypes.data, self.k, _NP2RRI[out.dtype]))
        return out

    def get_T(self, dtype=np.float64):
        out = np.empty((self.k, self.d), dtype=dtype)
        self._check(self._lib.rri_get_T(self._h, out.ctypes.data, self.d, _NP2RRI[out.dtype]))
        return out

    def set_params(self, fix_W=False, fix_T=False, project_T_each_iter=False, t_row_sum=None,
                   w_row_sum=None, reset_topic_method='max_resid_document', n_resets=23,
                   reg_w_l1=0.0, reg_w_l2=0.0, reg_t_l1=0.0, reg_t_l2=0.0, fix_reset_seed=False):
        if reset_topic_method not in _RESET_CODES:
            raise ValueError('unknown reset_topic_method %r' % (reset_topic_method,))
        p = Params()
        p.fix_W, p.fix_T = int(bool(fix_W)), int(bool(fix_T))
        p.project_T_each_iter = int(bool(project_T_each_iter))
        p.has_t_row_sum = int(t_row_sum is not None)
        p.t_row_sum = float(t_row_sum) if t_row_sum is not None else 0.0
        scalar_w = w_row_sum is not None and np.isscalar(w_row_sum)
        p.has_w_row_sum = int(scalar_w)
        p.w_row_sum = float(w_row_sum) if scalar_w else 0.0
        p.reset_method = _RESET_CODES[reset_topic_method]
        p.resets_left = int(n_resets)
        p.reg_w_l1, p.reg_w_l2 = float(reg_w_l1), float(reg_w_l2)
        p.reg_t_l1, p.reg_t_l2 = float(reg_t_l1), float(reg_t_l2)
        p.eps_div = EPS_DIV
        self._params = p
        self._reset_method = reset_topic_method
        self.fix_reset_seed = bool(fix_reset_seed)
        self._check(self._lib.rri_set_params(self._h, C.byref(p)))

    # ---- the hot path -------------------------------------------------------------------
    def _resolve_event(self):
        ev = Event()
        self._check(self._lib.rri_pending_event(self._h, C.byref(ev)))
        t = ev.topic
        if self._reset_method == 'max_resid_document':       # nmf.py:770-776 / :804-810
            row = C.c_int64(-1)
            self._check(self._lib.rri_apply_reset_max_resid(self._h, t, C.byref(row)))
            self.reset_log.append((ev.kind, t, int(row.value)))
        elif self._reset_method == 'random' and self.group is not None:
            # row-sharded: rank 0 draws T[t,:] and the WHOLE column W[:,t] with numpy's global RNG, as the reference
            # does for the whole matrix, and broadcasts; every rank keeps its rows
            g = self.group
            buf = np.zeros(self.d + g.n_global)
            if g.rank == 0:
                if self.fix_reset_seed:
                    np.random.seed(t + int(np.argmax(self.get_T()[t, :])))
                trow = np.random.rand(1, self.d)
                buf[:self.d] = (trow / trow.sum()).ravel()
                buf[self.d:] = np.random.rand(g.n_global)
            buf = self.comm_broadcast(buf, 0)
            Trow = np.ascontiguousarray(buf[:self.d])
            Wcol = np.ascontiguousarray(buf[self.d + g.row_lo:self.d + g.row_lo + self.n])
            self._check(self._lib.rri_apply_reset_vectors(
                self._h, t, Trow.ctypes.data_as(C.POINTER(C.c_double)),
                Wcol.ctypes.data_as(C.POINTER(C.c_double))))
            self.reset_log.append((ev.kind, t, -1))
        elif self._reset_method == 'random':                 # nmf.py:778-783 / :811-816
            if self.fix_reset_seed:
                Trow_now = self.get_T()[t, :]
                np.random.seed(t + int(np.argmax(Trow_now)))
            Trow = np.random.rand(1, self.d)
            Trow = np.ascontiguousarray((Trow / Trow.sum()).ravel())
            Wcol = np.ascontiguousarray(np.random.rand(self.n))
            self._check(self._lib.rri_apply_reset_vectors(
                self._h, t, Trow.ctypes.data_as(C.POINTER(C.c_double)),
                Wcol.ctypes.data_as(C.POINTER(C.c_double))))
            self.reset_log.append((ev.kind, t, -1))
        else:
            self._check(self._lib.rri_skip_reset(self._h))
        self.n_resets_used += 1

    def _drive(self, st, done):
        while st == _capi.RRI_PAUSED:
            self._resolve_event()
            st = self._lib.rri_resume(self._h, C.byref(done))
        self._check(st)
        return int(done.value)

    def sweep(self, n_sweeps=1):
        """n_sweeps Gauss-Seidel sweeps (nmf.py:415-476) on the device."""
        done = C.c_int32(0)
        return self._drive(self._lib.rri_sweep(self._h, int(n_sweeps), C.byref(done)), done)

    def update_T_row(self, t):
        st = self._lib.rri_update_T_row(self._h, int(t))
        if st == _capi.RRI_PAUSED:
            self._resolve_event()
            st = self._lib.rri_resume(self._h, None)
        self._check(st)

    def update_W_col(self, t):
        st = self._lib.rri_update_W_col(self._h, int(t))
        if st == _capi.RRI_PAUSED:
            self._resolve_event()
            st = self._lib.rri_resume(self._h, None)
        self._check(st)

    # ---- the explicit residual (schedule='residual') ------------------------------------------
    def residual_rebuild(self):
        """R = X - W T for the factors now on the device"""
        self._check(self._lib.rri_residual_rebuild(self._h))

    def get_residual(self, dtype=None):
        out = np.empty((self.n, self.d), dtype=dtype or self.dtype)
        self._check(self._lib.rri_get_residual(self._h, out.ctypes.data, self.d, _NP2RRI[out.dtype]))
        return out

    def residual_update(self, a, b, trow, wcol, a2=None, b2=None):
        """R <- R - a b^T [- a2 b2^T]; returns (R_new @ trow, R_new.T @ wcol) of the updated stored residual"""
        vec = lambda v, m: None if v is None else np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(m))
        a, a2, wcol = vec(a, self.n), vec(a2, self.n), vec(wcol, self.n)
        b, b2, trow = vec(b, self.d), vec(b2, self.d), vec(trow, self.d)
        ptr = lambda v: None if v is None else v.ctypes.data_as(C.POINTER(C.c_double))
        y, z = np.empty(self.n), np.empty(self.d)
        self._check(self._lib.rri_residual_update(self._h, ptr(a), ptr(b), ptr(a2), ptr(b2), ptr(trow), ptr(wcol),
                                                  ptr(y), ptr(z)))
        return y, z

    # ---- around the loop ----------------------------------------------------------------
    def project_W_rows(self, s):
        if np.isscalar(s):
            self._check(self._lib.rri_project_W_rows(self._h, float(s), None))
        else:
            v = np.ascontiguousarray(np.asarray(s, dtype=np.float64).ravel())
            if v.size != self.n:
                raise AssertionError('proj_mat_to_simplex: expected s to have size %d but s has size %d'
                                     % (self.n, v.size))
            self._check(self._lib.rri_project_W_rows(self._h, 0.0, v.ctypes.data_as(C.POINTER(C.c_double))))

    def objective(self):
        out = C.c_double(0.0)
        self._check(self._lib.rri_objective(self._h, C.byref(out)))
        return float(out.value)

    def objective_parts(self):
        out = (C.c_double * 3)()
        self._check(self._lib.rri_objective_parts(self._h, out))
        return [float(v) for v in out]

    def t_norms(self):
        """(sum T^2, sum |T|) of the replicated T, for the sharded objective"""
        T = self.get_T()
        return float((T ** 2).sum()), float(np.abs(T).sum())

    def argmax_rows(self):
        out = np.empty(self.n, dtype=np.int32)
        self._check(self._lib.rri_argmax_rows(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def masked_rmse(self, I, J, vals, lo, hi):
        """clipped RMSE of W T on the listed entries; row-sharded: local rows, collective, the global score on every rank"""
        ij = np.ascontiguousarray(np.stack([np.asarray(I, dtype=np.int64), np.asarray(J, dtype=np.int64)], 1).reshape(-1, 2))
        v = np.ascontiguousarray(np.asarray(vals, dtype=np.float64))
        out = C.c_double(0.0)
        self._check(self._lib.rri_masked_rmse(self._h, ij.ctypes.data_as(C.POINTER(C.c_int64)),
                                              v.ctypes.data_as(C.POINTER(C.c_double)), v.size,
                                              float(lo), float(hi), C.byref(out)))
        return float(out.value)

    def snapshot(self):
        self._check(self._lib.rri_snapshot(self._h))

    def rollback(self):
        self._check(self._lib.rri_rollback(self._h))

    # ---- products with the resident X (initialisation) ----------------------------------------
    def X_times(self, B):
        """X @ B for a host (d, m) matrix, float64"""
        B = np.ascontiguousarray(B, dtype=np.float64)
        if B.ndim != 2 or B.shape[0] != self.d:
            raise ValueError('operand must be (d, m)')
        if self.sparse and B.shape[1] > 64:     # pattern-only handles take up to 64 columns per call
            return np.hstack([self.X_times(B[:, lo:lo + 64]) for lo in range(0, B.shape[1], 64)])
        out = np.empty((self.n, B.shape[1]))
        self._check(self._lib.rri_X_times(self._h, B.ctypes.data_as(C.POINTER(C.c_double)), B.shape[1],
                                          out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def Xt_times(self, Q):
        """X.T @ Q for a host (n, m) matrix, float64"""
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        if Q.ndim != 2 or Q.shape[0] != self.n:
            raise ValueError('operand must be (n, m)')
        if self.sparse and Q.shape[1] > 64:
            return np.hstack([self.Xt_times(Q[:, lo:lo + 64]) for lo in range(0, Q.shape[1], 64)])
        out = np.empty((self.d, Q.shape[1]))
        self._check(self._lib.rri_Xt_times(self._h, Q.ctypes.data_as(C.POINTER(C.c_double)), Q.shape[1],
                                           out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def range_finder(self, Q0, n_iter, transpose=False):
        """(Q, B) of the randomized range finder on the resident X, panels on the device (rri_range_finder): A = X, or X.T when
        `transpose`; Q0 is the (columns of A, m) test matrix, Q the orthonormal (rows of A, m) basis of range(A (A^T A)^n_iter Q0),
        B = Q^T A.  m <= 64, dense handles."""
        Q0 = np.ascontiguousarray(Q0, dtype=np.float64)
        rows, cols = (self.d, self.n) if transpose else (self.n, self.d)
        if Q0.ndim != 2 or Q0.shape[0] != cols:
            raise ValueError('test matrix must be (%d, m)' % cols)
        m = Q0.shape[1]
        Q, B = np.empty((rows, m)), np.empty((m, cols))
        self._check(self._lib.rri_range_finder(self._h, Q0.ctypes.data_as(C.POINTER(C.c_double)), m, int(n_iter), int(bool(transpose)),
                                               Q.ctypes.data_as(C.POINTER(C.c_double)), B.ctypes.data_as(C.POINTER(C.c_double))))
        return Q, B

    # ---- preprocessing of the resident X (matrixops.py:124-179) ---------------------------
    def column_positive_counts(self):
        """df[j] = #{i: X[i, j] > 0}"""
        out = np.empty(self.d, dtype=np.float64)
        self._check(self._lib.rri_column_positive_counts(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def scale_X(self, col_scale=None, normalize_rows=False):
        """X <- X * col_scale (per column), then rows divided by their sums when normalize_rows (in place)"""
        ptr = None
        if col_scale is not None:
            col_scale = np.ascontiguousarray(col_scale, dtype=np.float64).ravel()
            if col_scale.size != self.d:
                raise ValueError('col_scale must have d entries')
            ptr = col_scale.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self._lib.rri_scale_X(self._h, ptr, int(bool(normalize_rows))))

    def preprocess(self, tfidf=False, normalize=False):
        """tf-idf and/or row normalisation of the resident X, as matrixops.tfidf / normalize produce them.
        tfidf: True (idf from this X), an idf vector (transform of new documents), or False.  Returns the idf used."""
        idf = None
        if tfidf is True:
            df = self.column_positive_counts()
            n_docs = self.n
            grp = getattr(self, 'group', None)
            if grp is not None:        # row-sharded: document frequencies are sums over the ranks, the corpus is all rows
                df = self.comm_sum(df)
                n_docs = grp.n_global
            idf = np.log(n_docs / (df + np.spacing(1)))       # matrixops.py:169-170
        elif tfidf is not False and tfidf is not None:
            idf = np.asarray(tfidf, dtype=np.float64).ravel()
        if idf is not None or normalize:
            self.scale_X(idf, normalize)
        return idf

    # ---- row-sharded stepping -----------------------------------------------------------
    def reduce_buffer(self):
        ptr, cnt = C.c_void_p(), C.c_int64()
        self._check(self._lib.rri_reduce_buffer(self._h, C.byref(ptr), C.byref(cnt)))
        return ptr.value, int(cnt.value)

    def bind_reduce_buffer(self, ptr, n_elems):
        self._check(self._lib.rri_bind_reduce_buffer(self._h, C.c_void_p(ptr), int(n_elems)))

    def topic_reduce_local(self, t):
        self._check(self._lib.rri_topic_reduce_local(self._h, int(t)))

    def topic_finish(self, t):
        self._check(self._lib.rri_topic_finish(self._h, int(t)))

    def reduce_read(self, count):
        out = np.empty(int(count), dtype=np.float64)
        self._check(self._lib.rri_reduce_read(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), int(count)))
        return out

    def reduce_write(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._lib.rri_reduce_write(self._h, v.ctypes.data_as(C.POINTER(C.c_double)), int(v.size)))

    def _stepping_event(self):
        """polls; a reset event is resolved as rri_sweep's are.  Returns its (kind, topic) or None."""
        if self.poll() != _capi.RRI_PAUSED:
            return None
        kind, topic, _ = self.pending_event()
        self._resolve_event()
        return kind, topic

    def sweep_with_T_noise(self, draw):
        """One sweep in which every T-row step sees wR + noise and max(nw + noise, 0): the Gaussian mechanism of
        nmf.py:422-435.  `draw(m)` returns m samples; it is called in the reference's order (wR first, then nw,
        after any reset of the previous column has drawn its own numbers)."""
        self.sweep_stepwise(draw=draw)

    def sweep_stepwise(self, draw=None, observe=None):
        """One sweep with the host between the two device stages of every T-row step: the sums of a topic step are
        taken on the device (rri_topic_reduce_local), read -- and, with `draw`, perturbed -- in the reduce buffer on the
        host, and the step finishes on the device (rri_topic_finish): the split stepping of the row-sharded path with
        the host in the place of the all-reduce.
          observe(t, wR, nw)  sees the sums of _compute_update_T (nmf.py:670-676, 687-701) before any noise:
                              store_gradients (nmf.py:454-456)
          draw(m)             the Gaussian mechanism (see sweep_with_T_noise)"""
        d, k = self.d, self.k
        ld = -(-d // (16 // self.dtype.itemsize)) * (16 // self.dtype.itemsize)
        for t in range(k):
            self.topic_reduce_local(t)
            self.topic_finish(-1)                       # the column check of topic t-1 (nmf.py:471-476)
            if self._stepping_event() is not None:      # dead column: reset, then the sums again
                self.topic_reduce_local(t)
            r, wR, nw, trow = self._topic_sums(t, ld, with_wR=observe is not None or self.weighted)
            if observe is not None:
                observe(t, np.array(wR), np.array(nw) if self.weighted else nw)
            if draw is not None and self.weighted:
                wR = wR + draw(d)
                nw2 = np.maximum(nw + draw(d), 0)
                r[:d] = wR - trow * nw2
                r[ld:ld + d] = nw2
            elif draw is not None:
                at = [ld + g * (k + 2) + k for g in range(_capi.RRI_GRAM_SLICES)]
                r[:d] += draw(d)                        # wR = w^T X - (w^T W) T: the noise passes through
                nw = nw + float(np.asarray(draw(1)).ravel()[0])
                for i in at:
                    r[i] = 0.0
                r[at[0]] = max(nw, 0.0)
            if draw is not None:
                self.reduce_write(r)
            self.topic_finish(t)
            if self._stepping_event() is not None:      # the new T row was (numerically) zero: reset, then its W half
                self.topic_finish_w(t)
        self.topic_reduce_local(0)                      # the last column's check rides on topic 0's sums
        self.topic_finish(-1)
        self._stepping_event()

    def _topic_sums(self, t, ld, with_wR=True):
        """(reduce buffer, wR, nw, T[t,:] or None) after rri_topic_reduce_local(t): the sums _compute_update_T returns
        (nmf.py:670-676 plain, :687-701 weighted)"""
        d, k = self.d, self.k
        if self.weighted:                               # red = [a | nw]; wR = a + t .* nw (rri_wrri_kernels.hpp)
            r = self.reduce_read(2 * ld)
            trow = self.get_T()[t, :]
            nw = r[ld:ld + d]
            return r, r[:d] + trow * nw, nw, trow
        r = self.reduce_read(ld + _capi.RRI_GRAM_SLICES * (k + 2))   # red = [w^T X | slices of (w^T W, ||w||^2, .)]
        nw = float(sum(r[ld + g * (k + 2) + k] for g in range(_capi.RRI_GRAM_SLICES)))
        wR = None
        if with_wR:
            wW = sum(r[ld + g * (k + 2):ld + g * (k + 2) + k] for g in range(_capi.RRI_GRAM_SLICES))
            wW[t] = 0.0                                 # nmf.py:672
            wR = r[:d] - wW.dot(self.get_T())
        return r, wR, nw, None

    def topic_sums(self, t):
        """(wR, nw) of topic t for the factors now on the device: w_t^T (X - sum_{j != t} w_j t_j) and ||w_t||^2, or
        their weighted counterparts (d-vectors)"""
        ld = -(-self.d // (16 // self.dtype.itemsize)) * (16 // self.dtype.itemsize)
        self.topic_reduce_local(t)
        _, wR, nw, _ = self._topic_sums(t, ld)
        return np.array(wR), (np.array(nw) if self.weighted else nw)

    def topic_finish_w(self, t):
        self._check(self._lib.rri_topic_finish_w(self._h, int(t)))

    def poll(self):
        """synchronises and returns RRI_OK or RRI_PAUSED (errors raise)"""
        return self._check(self._lib.rri_poll(self._h))

    def pending_event(self):
        ev = Event()
        self._check(self._lib.rri_pending_event(self._h, C.byref(ev)))
        return ev.kind, ev.topic, ev.resume_topic

    def resid_row_argmax(self):
        val, row = C.c_double(0.0), C.c_int64(-1)
        self._check(self._lib.rri_resid_row_argmax(self._h, C.byref(val), C.byref(row)))
        return float(val.value), int(row.value)

    def reset_row(self, local_row):
        out = np.empty(self.d, dtype=np.float64)
        self._check(self._lib.rri_reset_row(self._h, int(local_row), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def apply_reset_vectors(self, t, T_row, W_col):
        T_row = np.ascontiguousarray(T_row, dtype=np.float64)
        W_col = np.ascontiguousarray(W_col, dtype=np.float64)
        self._check(self._lib.rri_apply_reset_vectors(
            self._h, int(t), T_row.ctypes.data_as(C.POINTER(C.c_double)),
            W_col.ctypes.data_as(C.POINTER(C.c_double))))
        self.n_resets_used += 1

    # ---- measurement --------------------------------------------------------------------
    def timing_enable(self, on=True, every=1):
        """HIP-event timing of the streaming kernels; every=N samples each N-th launch"""
        self._check(self._lib.rri_timing_enable(self._h, int(every) if on else 0))

    def timing_read(self, kernel_id):
        cnt, ms = C.c_int64(0), C.c_double(0.0)
        self._check(self._lib.rri_timing_read(self._h, int(kernel_id), C.byref(cnt), C.byref(ms)))
        return int(cnt.value), float(ms.value)

    def onchip_info(self):
        """(would the next sweep() run as the register-resident persistent launch, how many such launches so far)"""
        el, n = C.c_int32(0), C.c_int64(0)
        self._check(self._lib.rri_onchip_info(self._h, C.byref(el), C.byref(n)))
        return bool(el.value), int(n.value)

    def debug_xcc(self, count=32):
        """diagnostics: the XCD each of `count` workgroups of a launch on this handle's stream lands on"""
        out = (C.c_int32 * count)()
        self._check(self._lib.rri_debug_xcc(self._h, out, count))
        return list(out)

    def onchip_fallbacks(self):
        """how many persistent launches of this handle gave up (workgroups not co-resident) and were rerun launch by launch"""
        n = C.c_int64(0)
        self._check(self._lib.rri_onchip_fallbacks(self._h, C.byref(n)))
        return int(n.value)

    def sweep_until(self, n_sweeps, obj_prev, stop_scale):
        """Up to n_sweeps sweeps of the register-resident kernel with every sweep's objective kept and the stop rule of
        nmf.py:510 applied on the device (rri_sweep_until): the run ends after the first sweep whose objective moved by no more
        than stop_scale (= eps_stop |o_0 - o_1|; negative: never).  Returns (sweeps run, their objectives -- NaN where the
        kernel left none: take objective() there), or None when the handle does not take the persistent path."""
        n_sweeps = int(n_sweeps)
        if not self.onchip_info()[0]:
            return None
        hist = np.full(n_sweeps, np.nan, dtype=np.float64)
        done = C.c_int32(0)
        st = self._lib.rri_sweep_until(self._h, n_sweeps, float(obj_prev), float(stop_scale),
                                       hist.ctypes.data_as(C.POINTER(C.c_double)), C.byref(done))
        if st == _capi.RRI_ERR_UNSUPPORTED:
            return None
        n_done = self._drive(st, done)
        return n_done, hist[:n_done]

    def synchronize(self):
        self._check(self._lib.rri_synchronize(self._h))

    def bench_rank1_update(self, reps=5):
        ms = C.c_double(0.0)
        self._check(self._lib.rri_bench_rank1_update(self._h, int(reps), C.byref(ms)))
        return float(ms.value)

    def bench_stream_copy(self, reps=5):
        ms = C.c_double(0.0)
        self._check(self._lib.rri_bench_stream_copy(self._h, int(reps), C.byref(ms)))
        return float(ms.value)
