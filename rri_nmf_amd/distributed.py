"""Row-sharded RRI over the GPUs of one node: one process per GPU, X and W blocked by rows,
T replicated.

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
