"""Row-sharded RRI over the GPUs of one node: one process per GPU, X and W blocked by rows,
T replicated.

Two ways to run it:

RowGroup (the product path)  the communicator lives INSIDE librri_hip.so (rri_comm_create: RCCL over xGMI, resolved
    at run time): every rank attaches it to its handle and calls the ordinary entry points -- engine.sweep(),
    nmf(X_rows, k, ..., group=grp) -- collectively.  A sweep is one C call; the all-reduce of a topic step is enqueued
    on the handle's stream between two kernels, no Python and no torch per topic.

ShardedRRI (below)  the same step protocol with the collective in the caller's hands (torch.distributed on a
    caller-owned buffer): what hosts with their own collectives use, and how the protocol is tested over gloo.

A topic step needs ONE cross-row reduction: [w_t^T X (d) | w_t^T W (k) | ||w_t||^2 | sum W[:,t-1]].
Every rank reduces its shard into the engine's reduce buffer (rri_topic_reduce_local), the ranks
all-reduce that buffer (float64; RCCL over xGMI through torch.distributed -- the payload is
(d + 8(k+2))*8 bytes: the Gram part travels as 8 slice sums; latency- not link-bound), and the rest of the step is rank-local
(rri_topic_finish): identical T on every rank, own rows of W.

The driver only needs the `step engine` protocol (topic_reduce_local / topic_finish / poll /
objective_parts / norms), so the CPU tests run it over gloo with a numpy stand-in engine.
"""
import numpy as np


def shard_rows(n, world_size, rank):
    """[lo, hi) rows of `rank`: contiguous blocks, sizes differ by at most one"""
    base, extra = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


PAUSED = 1
EVENT_RESET_T, EVENT_RESET_W = 1, 2


class RowGroup(object):
    """The communicator of a row-sharded run and where this rank's rows sit in the global matrix.

        grp = RowGroup.rccl(n_local)                  # under torchrun: id and block sizes travel over torch.distributed
        out = nmf(X_rows, k, W_in=W_rows, T_in=T, group=grp)          # every rank, same arguments
        grp.close()

    `sizes[r]` = rows of rank r (contiguous blocks in rank order).  The RCCL communicator is created once per process
    and group and attached to as many handles as wanted, one after the other."""

    def __init__(self, comm, rank, world, sizes, keep=()):
        self._comm_handle, self._parent, self.rank, self.world = comm, None, int(rank), int(world)
        self.sizes = [int(v) for v in sizes]
        self.row_lo = int(sum(self.sizes[:self.rank]))
        self.n_local = self.sizes[self.rank]
        self.n_global = int(sum(self.sizes))
        self._keep = keep          # ctypes callbacks must outlive the communicator

    @staticmethod
    def _exchange_default():
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError('RowGroup needs torch.distributed initialised (or an `exchange` function)')

        def exchange(obj):                      # every rank's object, in rank order
            out = [None] * dist.get_world_size()
            dist.all_gather_object(out, obj)
            return out
        return exchange, dist.get_rank(), dist.get_world_size()

    @classmethod
    def rccl(cls, n_local, device=0, exchange=None, rank=None, world=None):
        """RCCL communicator inside the library.  `exchange(obj) -> [obj of rank 0, ..., obj of rank world-1]` is any
        all-gather of small Python objects the host program has (default: torch.distributed's, whatever its backend);
        it carries the RCCL id of rank 0 and the block sizes, nothing else."""
        import ctypes as C
        import os
        from . import _capi
        if exchange is None:
            exchange, rank, world = cls._exchange_default()
        lib = _capi.load_library()
        _capi.share_rccl_with_torch()
        # one node (rendezvous on 127.0.0.1 by contract): RCCL's bootstrap socket stays on loopback whatever interface
        # the container's hostname resolves to -- the one step of a single-rank or single-node start that depends on
        # the box's network set-up
        if int(os.environ.get('LOCAL_WORLD_SIZE', world)) == world:
            os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
        ident = (C.c_uint8 * _capi.RRI_COMM_ID_BYTES)()
        if rank == 0:
            st = lib.rri_comm_unique_id(ident)
            if st != _capi.RRI_OK:
                msg = lib.rri_last_error(None)
                raise _capi.RRIHipUnavailable('rri_comm_unique_id failed (%d): %s' % (st, msg.decode() if msg else '?'))
        got = exchange((bytes(bytearray(ident)) if rank == 0 else None, int(n_local)))
        ident = (C.c_uint8 * _capi.RRI_COMM_ID_BYTES)(*bytearray(got[0][0]))
        comm = C.c_void_p()
        st = lib.rri_comm_create(C.byref(comm), ident, int(rank), int(world), int(device))
        if st != _capi.RRI_OK:
            msg = lib.rri_last_error(None)
            raise _capi.RRIHipUnavailable('rri_comm_create failed (%d): %s' % (st, msg.decode() if msg else '?'))
        return cls(comm, rank, world, [g[1] for g in got])

    @classmethod
    def over_torch(cls, n_local, group=None):
        """The same protocol with torch.distributed as the transport of the library's host-callback communicator
        (rri_comm_create_host): every collective drains the stream and goes through host memory.  For tests with
        several ranks on ONE GPU (RCCL wants one device per rank) over gloo; not a measured path."""
        import ctypes as C
        import torch
        import torch.distributed as dist
        from . import _capi
        lib = _capi.load_library()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        as_tensor = lambda ptr, count: torch.from_numpy(np.ctypeslib.as_array(ptr, shape=(int(count),)))

        def allreduce(user, buf, count):
            try:
                dist.all_reduce(as_tensor(buf, count), op=dist.ReduceOp.SUM, group=group)
                return 0
            except Exception:  # noqa: BLE001  (the library turns a non-zero return into RRI_ERR_COMM)
                return 1

        def allgather(user, send, count, recv):
            try:
                out = as_tensor(recv, count * world)
                dist.all_gather(list(out.split(int(count))), as_tensor(send, count).clone(), group=group)
                return 0
            except Exception:  # noqa: BLE001
                return 1

        def broadcast(user, buf, count, root):
            try:
                dist.broadcast(as_tensor(buf, count), src=dist.get_global_rank(group, root) if group is not None else root,
                               group=group)
                return 0
            except Exception:  # noqa: BLE001
                return 1

        cbs = (_capi.ALLREDUCE_FN(allreduce), _capi.ALLGATHER_FN(allgather), _capi.BROADCAST_FN(broadcast))
        comm = C.c_void_p()
        st = lib.rri_comm_create_host(C.byref(comm), rank, world, cbs[0], cbs[1], cbs[2], None)
        if st != _capi.RRI_OK:
            raise RuntimeError('rri_comm_create_host failed (%d)' % st)
        sizes = [None] * world
        dist.all_gather_object(sizes, int(n_local), group=group)
        return cls(comm, rank, world, sizes, keep=cbs)

    @property
    def _comm(self):
        """the library's communicator, or None once the group that OWNS it has been closed: a view made by resized() looks it
        up in its owner every time, so a view of a closed group is closed too (it used to keep a copy of the raw pointer: a
        use-after-free at the next attach)"""
        return self._parent._comm if self._parent is not None else self._comm_handle

    @property
    def closed(self):
        c = self._comm
        return c is None or not c

    def resized(self, sizes):
        """the same communicator for another problem: `sizes[r]` rows on rank r (the caller makes sure every rank passes
        the same list).  The view does not own the communicator: close the group it came from."""
        view = RowGroup(None, self.rank, self.world, sizes, keep=self._keep)
        view._parent = self if self._parent is None else self._parent
        return view

    def gather(self, values, device=0):
        """every rank's `values` (a short float64 vector, the same length everywhere) as a world x len array on every rank:
        a sum over one-hot rows through the communicator (a scratch handle of one row per rank carries the call)"""
        from .engine import RRIEngine
        v = np.atleast_1d(np.asarray(values, dtype=np.float64)).ravel()
        buf = np.zeros((self.world, v.size))
        buf[self.rank] = v
        with RRIEngine(1, 1, 1, dtype=np.float64, device=device) as eng:
            eng.attach_group(self.resized([1] * self.world))
            out = eng.comm_sum(buf.ravel())
        return np.asarray(out).reshape(self.world, v.size)

    def close(self):
        if self._parent is not None:
            return
        if self._comm_handle is not None and self._comm_handle:
            from . import _capi
            _capi.load_library().rri_comm_destroy(self._comm_handle)
            self._comm_handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ShardedRRI(object):
    """Drives the sweep loop of one rank.  `engine` holds this rank's rows [row_lo, row_lo + n_local) of the
    n_global-row problem; `red` is a torch tensor (float64, the engine's reduce buffer) that
    torch.distributed can all-reduce in place.

    Reset events (nmf.py:762-783, 796-816) are decided from all-reduced / replicated quantities, so every rank
    pauses at the same half step; they are resolved collectively:
      'max_resid_document'  every rank offers its largest row-residual norm, the global winner (lowest global
                            row index on ties, as np.argmax) broadcasts its reset row; T[t,:] is set on every
                            rank, W[:,t] becomes the unit vector of the winning row
      'random'              rank 0 draws T[t,:] and the whole W[:,t] with numpy's global RNG and broadcasts
    """

    def __init__(self, engine, red, k, group=None, stream=None, row_lo=0, n_global=None,
                 reset_topic_method='max_resid_document', fix_reset_seed=False):
        import torch.distributed as dist
        self.dist = dist
        self.eng, self.red, self.k, self.group = engine, red, int(k), group
        self.stream = stream       # torch.cuda.Stream the engine's kernels run on (None on the CPU)
        self.row_lo = int(row_lo)
        self.n_global = n_global
        self.reset_topic_method = reset_topic_method
        self.fix_reset_seed = fix_reset_seed
        self._reduced_for = None   # topic whose reduced sums currently sit in `red`
        self.allreduce_calls = 0
        self.n_resets_used = 0

    # ---- collectives -----------------------------------------------------------------------------
    def _on_stream(self):
        import contextlib
        if self.stream is None:
            return contextlib.nullcontext()
        import torch
        # RCCL is ordered against the CURRENT torch stream: make that the engine's stream, so the
        # collective waits for k_reduce and k_trow_numer waits for the collective
        return torch.cuda.stream(self.stream)

    def _sync(self):
        """the host is about to read what a collective on the engine's stream produced: RCCL only orders the
        collective against that stream, not against the host or torch's default stream"""
        if self.stream is not None:
            self.stream.synchronize()

    def _allreduce(self):
        with self._on_stream():
            self.dist.all_reduce(self.red, op=self.dist.ReduceOp.SUM, group=self.group)
        self.allreduce_calls += 1

    def _prepare(self, t):
        if self._reduced_for != t:
            self.eng.topic_reduce_local(t)
            self._allreduce()
            self._reduced_for = t

    # ---- resets ------------------------------------------------------------------------------------
    def _resolve(self, t):
        # every torch operation below runs on the ENGINE's stream (fills, copies, collectives): tensors made on
        # torch's default stream would race with the collectives, which are ordered against the engine's stream only
        with self._on_stream():
            self._resolve_on_stream(t)

    def _resolve_on_stream(self, t):
        import torch
        dist, dev = self.dist, self.red.device
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        n_local, d = self.eng.n, self.eng.d
        method = self.reset_topic_method
        if method == 'max_resid_document':
            val, idx = self.eng.resid_row_argmax()
            mine = torch.tensor([val, float(self.row_lo + idx)], dtype=torch.float64, device=dev)
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine, group=self.group)
            self._sync()
            cand = [(float(v[0]), int(v[1]), r) for r, v in enumerate(allv)]
            best = max(c[0] for c in cand)
            winner = min((c for c in cand if c[0] == best), key=lambda c: c[1])   # first index of the maximum
            row = torch.zeros(d, dtype=torch.float64, device=dev)
            if rank == winner[2]:
                row.copy_(torch.from_numpy(self.eng.reset_row(idx)))
            dist.broadcast(row, src=winner[2], group=self.group)
            self._sync()
            wcol = np.zeros(n_local)
            if rank == winner[2]:
                wcol[idx] = 1.0
            self.eng.apply_reset_vectors(t, row.cpu().numpy(), wcol)
        elif method == 'random':
            n_global = self.n_global if self.n_global is not None else n_local
            buf = torch.zeros(d + n_global, dtype=torch.float64, device=dev)
            if rank == 0:
                if self.fix_reset_seed:
                    np.random.seed(t + int(np.argmax(self.eng.get_T()[t, :])))
                trow = np.random.rand(1, d)
                trow = (trow / trow.sum()).ravel()
                buf.copy_(torch.from_numpy(np.concatenate([trow, np.random.rand(n_global)])))
            dist.broadcast(buf, src=0, group=self.group)
            self._sync()
            h = buf.cpu().numpy()
            self.eng.apply_reset_vectors(t, h[:d], h[d + self.row_lo:d + self.row_lo + n_local])
        else:
            raise RuntimeError('reset event with reset_topic_method=None')
        self.n_resets_used += 1
        self._reduced_for = None

    # ---- sweeps ------------------------------------------------------------------------------------
    def _run_topics(self, t0, w_half_only_first=False):
        for t in range(t0, self.k):
            if t == t0 and w_half_only_first:
                self.eng.topic_finish_w(t)
            else:
                self._prepare(t)
                self.eng.topic_finish(t)
            self._reduced_for = None

    def _settle(self):
        """polls; while a reset is pending: resolve it collectively and redo what the halted queue skipped"""
        st = self.eng.poll()
        while st == PAUSED:
            kind, topic, resume_topic = self.eng.pending_event()
            self._resolve(topic)
            if kind == EVENT_RESET_T:
                # raised inside topic `topic` after its T row was written: its W half and all later topics
                # of the sweep were skipped by the halted queue
                self._run_topics(topic, w_half_only_first=True)
            elif self._in_sweep:
                # a dead W column is noticed by the NEXT T-row step (`resume_topic`, possibly topic 0 for the
                # last column of the previous sweep): that step and everything after it were skipped
                self._run_topics(resume_topic)
            # else: raised by the stand-alone check after the last sweep: nothing was skipped
            st = self.eng.poll()

    def sweep(self, n_sweeps=1, check=True):
        for _ in range(int(n_sweeps)):
            self._in_sweep = True
            self._run_topics(0)
            self._settle()          # one synchronisation per sweep
        self._in_sweep = False
        if check:
            # the last column's sum-to-zero check needs the global column sum: it rides on the
            # reduction of topic 0 of the next sweep, which is then already in place
            self._prepare(0)
            self.eng.topic_finish(-1)
            self._settle()

    def objective(self, reg_w_l1=0.0, reg_w_l2=0.0, reg_t_l1=0.0, reg_t_l2=0.0, t_norms=None):
        """true_objective (nmf.py:71-94) of the global problem: row terms summed over the ranks"""
        import torch
        with self._on_stream():
            parts = torch.tensor(self.eng.objective_parts(), dtype=torch.float64, device=self.red.device)
            self.dist.all_reduce(parts, op=self.dist.ReduceOp.SUM, group=self.group)
            self._sync()
        base, w2, w1 = [float(v) for v in parts.cpu()]
        t2, t1 = t_norms if t_norms is not None else self.eng.t_norms()
        return base + 0.5 * reg_w_l2 * w2 + 0.5 * reg_t_l2 * t2 + reg_t_l1 * t1 + reg_w_l1 * w1


def make_device_shard(n_local, d, k, dtype=np.float32, device_index=0, weighted=False):
    """Engine on a dedicated torch stream of `device_index` with a torch-owned reduce buffer.
    Returns (engine, red tensor, torch stream); pass the stream to ShardedRRI."""
    import torch
    from .engine import RRIEngine
    torch.cuda.set_device(device_index)
    stream = torch.cuda.Stream(device=device_index)   # a real (non-default) HIP stream handle
    eng = RRIEngine(n_local, d, k, dtype=dtype, weighted=weighted, device=device_index, stream=stream.cuda_stream)
    _, n_elems = eng.reduce_buffer()
    red = torch.zeros(n_elems, dtype=torch.float64, device='cuda:%d' % device_index)
    torch.cuda.synchronize(device_index)
    eng.bind_reduce_buffer(red.data_ptr(), n_elems)
    return eng, red, stream
