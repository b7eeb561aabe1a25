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


class ShardedRRI(object):
    """Drives the sweep loop of one rank.  `engine` holds this rank's rows; `red` is a torch tensor
    (float64, the engine's reduce buffer) that torch.distributed can all-reduce in place."""

    def __init__(self, engine, red, k, group=None, stream=None):
        import torch.distributed as dist
        self.dist = dist
        self.eng, self.red, self.k, self.group = engine, red, int(k), group
        self.stream = stream       # torch.cuda.Stream the engine's kernels run on (None on the CPU)
        self._reduced_for = None   # topic whose reduced sums currently sit in `red`
        self.allreduce_calls = 0

    def _allreduce(self):
        if self.stream is not None:
            import torch
            # RCCL is ordered against the CURRENT torch stream: make that the engine's stream, so the
            # collective waits for k_reduce and k_trow_numer waits for the collective
            with torch.cuda.stream(self.stream):
                self.dist.all_reduce(self.red, op=self.dist.ReduceOp.SUM, group=self.group)
        else:
            self.dist.all_reduce(self.red, op=self.dist.ReduceOp.SUM, group=self.group)
        self.allreduce_calls += 1

    def _prepare(self, t):
        if self._reduced_for != t:
            self.eng.topic_reduce_local(t)
            self._allreduce()
            self._reduced_for = t

    def sweep(self, n_sweeps=1, check=True):
        for _ in range(int(n_sweeps)):
            for t in range(self.k):
                self._prepare(t)
                self.eng.topic_finish(t)
                self._reduced_for = None
        if check:
            # the last column's sum-to-zero check needs the global column sum: it rides on the
            # reduction of topic 0 of the next sweep, which is then already in place
            self._prepare(0)
            self.eng.topic_finish(-1)
            self.eng.poll()

    def objective(self, reg_w_l1=0.0, reg_w_l2=0.0, reg_t_l1=0.0, reg_t_l2=0.0, t_norms=None):
        """true_objective (nmf.py:71-94) of the global problem: row terms summed over the ranks"""
        import torch
        parts = torch.tensor(self.eng.objective_parts(), dtype=torch.float64, device=self.red.device)
        self.dist.all_reduce(parts, op=self.dist.ReduceOp.SUM, group=self.group)
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
