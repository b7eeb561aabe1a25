#!/usr/bin/env python3
"""bench.py -- RRI sweeps/sec and achieved HBM GB/s of the MI355X path (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c4|c5|c5s] [--schedule gram|residual]

A step is one RRI sweep (k topic steps = k T-row + k W-column updates, nmf.py:415-476) over a
synthetic dense fp32 X that is already resident in HBM when the timed region starts.

Workloads (BASELINE.json configs):
    c3 (default)  100000 x 10000, k=50: the roofline run the north_star target is quoted on.
                  N > 1: WEAK scaling -- every rank holds its own 100000-row shard of an
                  (N*100000) x 10000 problem (T replicated, one RCCL all-reduce of d+8(k+2) doubles per
                  topic step, inside rri_sweep).  `value` is the whole job's rate in sweeps/s of a 100000-row
                  shard, i.e. N * (global sweeps/s): it equals plain sweeps/s at N = 1.
    c2            10000 x 1000, k=20 (X is Infinity-Cache resident: latency-, not HBM-bound).
    c4            1000000 x 10000, k=50 split by rows over the N ranks (STRONG scaling).
    c5            WRRI: c3's shape with a dense fp32 5 %-observed 0/1 mask, recommender flags (T clipped to
                  [0,1], no resets); single GPU.  Algorithmic bytes 5*n*d*4 per topic step.
    c5s           the same problem on the observed pattern only (sparse-index formulation: its own byte figure).
    --schedule residual   the explicit-residual form (one rank-one residual update pass per topic step, 2*n*d*4 B).

Prints ONE JSON line on rank 0 (contract in the task description) including
    roofline      the dominant kernel -- algorithmic bytes per launch over its average duration, measured with HIP
                  events on the kernel's stream inside the timed region, against 8 TB/s
    cpu_baseline  the float64 numpy restatement (oracle/) timed on this box's host cores: c3 at FULL size (every
                  row, `--cpu-sweeps` timed sweeps), the other workloads on the stated sample
    parity_sample the device path against that CPU run after equal sweeps (same X, same start).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    'c2': dict(n=10000, d=1000, k=20, scaling='weak', name='synthetic dense fp32 X 10000x1000 k=20'),
    'mid': dict(n=20000, d=5000, k=20, scaling='weak', name='synthetic dense fp32 X 20000x5000 k=20 (not a BASELINE config: a mid-size point)'),
    'c3': dict(n=100000, d=10000, k=50, scaling='weak', name='synthetic dense fp32 X 100000x10000 k=50'),
    'c4': dict(n=1000000, d=10000, k=50, scaling='strong', name='synthetic dense fp32 X 1000000x10000 k=50'),
    'c5': dict(n=100000, d=10000, k=50, scaling='weak', weighted=True,
               name='elementwise-weighted WRRI (Algorithm 10), dense fp32 X 100000x10000 k=50, 5% observed 0/1 mask'),
    'c5s': dict(n=100000, d=10000, k=50, scaling='weak', weighted=True, sparse=True,
                name='elementwise-weighted WRRI (Algorithm 10) 100000x10000 k=50, 5% observed 0/1 mask, SPARSE-INDEX '
                     'formulation (fp32 residual on the observed pattern, CSR + CSC copies): a different, smaller byte '
                     'figure than the dense c5 (SURVEY 8d)'),
}
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
PROFILE_ROUND = 'r04'
# rows of the weighted workloads the CPU oracle runs (one sweep: about a CPU-minute on 16 threads; a full-size sweep of the
# reference's two n*d*k GEMMs per topic takes five) -- the same slice tests/test_full_size_gpu.py::test_c5_slice_against_the_cpu_oracle asserts
WEIGHTED_CPU_ROWS = 20000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--schedule', default='gram', choices=['gram', 'residual'],
                    help="gram: X is read once per topic step (default); residual: the explicit residual R = X - WT is "
                         "updated by rank-one terms, one read-modify-write pass per topic step (unweighted)")
    ap.add_argument('--cpu-sweeps', type=int, default=2)
    ap.add_argument('--cpu-rows', type=int, default=0,
                    help='rows of X in the CPU baseline (0 = the workload default: all rows for c2 / c3, 100000 for c4)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--storage', default='f32', choices=['f32', 'f64'],
                    help='storage type of X / mask / residual in HBM (the BASELINE configs are fp32; arithmetic is float64 either way)')
    ap.add_argument('--force-sharded', action='store_true',
                    help='use the row-sharded path (an all-reduce per topic step) even with one rank')
    ap.add_argument('--collective', default='library', choices=['library', 'torch'],
                    help='row-sharded runs: library = RCCL inside librri_hip.so, the all-reduce enqueued by rri_sweep itself '
                         '(default); torch = the caller-owned protocol, torch.distributed per topic step from Python')
    ap.add_argument('--comm-probe', default='', help=argparse.SUPPRESS)    # child mode of probe_library_collective()
    return ap.parse_args()


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child job.  Done before
    anything touches the GPU; the child is waited for, never exec'ed over this process."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


PROBE_SECONDS = 240


def comm_probe_child(ident_hex):
    """Child mode (`--comm-probe <RCCL id of rank 0>`): this rank's share of ONE in-library all-reduce among the ranks of
    the job -- communicator created by rri_comm_create, a tiny handle attached, sum of (rank + 1) checked -- in a
    process of its own that gives up (stack trace, exit 1) instead of stalling."""
    import faulthandler
    faulthandler.dump_traceback_later(PROBE_SECONDS - 30, exit=True)
    import numpy as np
    from rri_nmf_amd.distributed import RowGroup
    from rri_nmf_amd.engine import RRIEngine
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev = int(os.environ.get('RRI_PROBE_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    ident = bytes.fromhex(ident_hex)
    grp = RowGroup.rccl(64, device=dev, exchange=lambda obj: [(ident, 64)] * world, rank=rank, world=world)
    eng = RRIEngine(64, 32, 2, device=dev)
    eng.attach_group(grp)
    got = eng.comm_sum(np.array([rank + 1.0, 1.0]))
    want = [world * (world + 1) / 2.0, float(world)]
    eng.close()
    grp.close()
    if list(got) != want:
        print('comm probe: sum %r, expected %r' % (list(got), want))
        return 1
    print('comm probe ok')
    return 0


def probe_library_collective(rank, world, device_index):
    """Before this process creates the in-library RCCL communicator it lets a child per rank do exactly that once
    (comm_probe_child).  A bootstrap that cannot complete on this box then costs a bounded wait and the run falls
    back to torch.distributed's communicator, saying so in the line, instead of stalling every rank inside
    ncclCommInitRank where nothing can interrupt it.  Returns (usable on every rank, reason)."""
    import ctypes as C
    import torch.distributed as dist
    from rri_nmf_amd import _capi
    lib = _capi.load_library()
    _capi.share_rccl_with_torch()
    ident = ''
    if rank == 0:
        buf = (C.c_uint8 * _capi.RRI_COMM_ID_BYTES)()
        if lib.rri_comm_unique_id(buf) == _capi.RRI_OK:
            ident = bytes(bytearray(buf)).hex()
    got = [None] * world
    dist.all_gather_object(got, ident)
    ident = got[0]
    ok, why = False, 'rri_comm_unique_id failed'
    if ident:
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), RRI_PROBE_DEVICE=str(device_index))
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--comm-probe', ident], env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        try:
            log, _ = child.communicate(timeout=PROBE_SECONDS)
            ok = child.returncode == 0 and 'comm probe ok' in log
            why = '' if ok else 'rank %d: exit %s: %s' % (rank, child.returncode, log[-300:].replace('\n', ' | '))
        except subprocess.TimeoutExpired:
            child.kill()                     # the exact PID started above
            child.communicate()
            why = 'rank %d: no answer in %d s' % (rank, PROBE_SECONDS)
    verdicts = [None] * world
    dist.all_gather_object(verdicts, (ok, why))
    return all(v[0] for v in verdicts), '; '.join(v[1] for v in verdicts if v[1])


def source_stamp():
    """sha of the kernel sources: PMC traffic files carry the stamp of the build they were measured on"""
    h = hashlib.sha256()
    cs = os.path.join(ROOT, 'rri_nmf_amd', 'csrc')
    for f in sorted(os.listdir(cs)):
        if f.endswith(('.hip', '.hpp')):
            h.update(open(os.path.join(cs, f), 'rb').read())
    return h.hexdigest()[:16]


def device_planted_shard(n_rows, d, k, seed, device):
    """X = W* T* + 0.01 U with 30 %-dense uniform factors, generated on the device in row
    chunks (T* is common to all shards, W* and the noise are per shard)."""
    import torch
    gt = torch.Generator(device=device)
    gt.manual_seed(0)
    Ts = torch.rand(k, d, device=device, generator=gt) * (torch.rand(k, d, device=device, generator=gt) < 0.3)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    X = torch.empty(n_rows, d, device=device, dtype=torch.float32)
    step = 25000
    for lo in range(0, n_rows, step):
        hi = min(n_rows, lo + step)
        Ws = torch.rand(hi - lo, k, device=device, generator=g) * (torch.rand(hi - lo, k, device=device, generator=g) < 0.3)
        torch.matmul(Ws, Ts, out=X[lo:hi])
        X[lo:hi].add_(torch.rand(hi - lo, d, device=device, generator=g), alpha=0.01)
    return X


def relfro(a, b):
    import numpy as np
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def cpu_plain(X_dev, W0, T0, rows, sweeps, threads, n_full, want_factors=True):
    """the numpy float64 restatement (oracle/rri_oracle.py: the reference's operation order) on rows [0, rows) of the
    resident X, `sweeps` timed sweeps one by one on `threads` BLAS threads.  Returns (rates per sweep, factors after
    each sweep)."""
    import numpy as np
    from oracle import rri_oracle as orc
    from threadpoolctl import threadpool_limits
    X = np.empty((rows, X_dev.shape[1]), dtype=np.float64)
    for lo in range(0, rows, 20000):                      # float64 on the host, chunk by chunk (no fp32 host copy of X)
        hi = min(rows, lo + 20000)
        X[lo:hi] = X_dev[lo:hi].cpu().numpy()
    W, T = W0[:rows].astype(np.float64).copy(), T0.astype(np.float64).copy()
    times, facs = [], []
    with threadpool_limits(limits=threads):
        warm = min(rows, 2000)
        orc.plain_sweeps(X[:warm].copy(), W[:warm].copy(), T.copy(), 1)       # BLAS threads up, pages of the code touched
        for _ in range(sweeps):
            t0 = time.perf_counter()
            orc.plain_sweeps(X, W, T, 1)
            times.append(time.perf_counter() - t0)
            if want_factors:
                facs.append((W.copy(), T.copy()))
    return times, facs, X


def main():
    args = parse()
    if args.comm_probe:
        sys.exit(comm_probe_child(args.comm_probe))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world == 1:
        sys.exit(relaunch_under_torchrun(args))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner on stdout when a communicator
    # comes up, so whatever the libraries write to descriptor 1 during the run goes to stderr instead
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    # Rehearsal of the multi-rank path on a ONE-GPU box (never the measured configuration): RRI_BENCH_REHEARSAL=1 puts
    # every rank on device 0 and runs the collectives over gloo (RCCL wants one device per rank)
    rehearsal = os.environ.get('RRI_BENCH_REHEARSAL', '0') == '1'
    if rehearsal:
        local_rank = 0

    import numpy as np
    import torch
    import torch.distributed as dist
    from rri_nmf_amd.distributed import RowGroup, ShardedRRI, make_device_shard, shard_rows
    from rri_nmf_amd.engine import RRIEngine

    cfg = CONFIGS[args.config]
    d, k = cfg['d'], cfg['k']
    weighted = bool(cfg.get('weighted'))
    sparse = bool(cfg.get('sparse'))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    sharded = world > 1 or args.force_sharded
    if args.schedule == 'residual' and (weighted or (sharded and args.collective != 'library')):
        sys.exit('--schedule residual is the unweighted flavour; row-sharded it needs the in-library collective')
    if sharded:
        import faulthandler
        faulthandler.dump_traceback_later(600, repeat=True)     # a rank that waits for its peers says where, every 10 min
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            os.environ['MASTER_PORT'] = str(free_port())
        # one node by contract (rendezvous on 127.0.0.1): RCCL's bootstrap socket stays on loopback too, whatever the
        # container's hostname resolves to; the data path is xGMI / shared memory either way
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
        if rehearsal:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', device_id=device, rank=rank, world_size=world)
    if cfg['scaling'] == 'weak':
        n_local, n_global = cfg['n'], cfg['n'] * world
    else:
        lo, hi = shard_rows(cfg['n'], world, rank)
        n_local, n_global = hi - lo, cfg['n']

    X = device_planted_shard(n_local, d, k, seed=(0 if world == 1 else 1000 + rank), device=device)
    mean = torch.zeros((), dtype=torch.float64, device=device)
    for lo_ in range(0, n_local, 100000):
        mean += X[lo_:lo_ + 100000].sum(dtype=torch.float64)
    if world > 1:
        dist.all_reduce(mean)
    mean = float(mean) / (float(n_global) * d)
    a = (mean / k) ** 0.5
    gi = torch.Generator(device=device)
    gi.manual_seed(1)
    T0 = (a * torch.rand(k, d, device=device, generator=gi, dtype=torch.float64)).cpu().numpy()   # same on all ranks
    gw = torch.Generator(device=device)
    gw.manual_seed(2000 + rank)
    W0 = (a * torch.rand(n_local, k, device=device, generator=gw, dtype=torch.float64)).cpu().numpy()
    torch.cuda.synchronize()

    Mask = None
    if weighted:
        gm = torch.Generator(device=device)
        gm.manual_seed(2 if world == 1 else 3000 + rank)
        Mask = (torch.rand(n_local, d, device=device, generator=gm) < 0.05).to(torch.float32)
        X.mul_(Mask)
        torch.cuda.synchronize()
    sdt = np.float32 if args.storage == 'f32' else np.float64
    es = 4 if args.storage == 'f32' else 8

    # ---- the handle, and for row-sharded runs who owns the collective --------------------------------------------
    collective, group, drv, red, stream = 'none', None, None, None, None
    flavour = 'sparse' if sparse else weighted
    if sharded and args.collective == 'library' and not rehearsal:
        try:
            usable, why = probe_library_collective(rank, world, local_rank)
            if not usable:
                raise RuntimeError('probe: ' + why)
            group = RowGroup.rccl(n_local, device=local_rank)
            collective = 'RCCL inside librri_hip.so: one ncclAllReduce per topic step enqueued by rri_sweep on the handle\'s stream'
        except Exception as e:  # noqa: BLE001  (measurement robustness: say so in the line and use the caller-owned protocol)
            group = None
            collective = 'torch.distributed (library communicator unavailable: %s)' % str(e)[:200]
    if sharded and group is None:
        if rehearsal and args.collective == 'library':
            group = RowGroup.over_torch(n_local)
            collective = 'REHEARSAL: host-callback transport over gloo (all ranks on one GPU; not a measurement)'
        else:
            if collective == 'none':
                collective = 'torch.distributed all_reduce per topic step from Python (caller-owned protocol)'
    if group is not None:
        eng = RRIEngine(n_local, d, k, dtype=sdt, weighted=flavour, device=local_rank, schedule=args.schedule)
        eng.attach_group(group)
        # one all-reduce of the size a topic step sends, checked: the ranks really are connected, and the connections of
        # that message size exist before the timed region whatever --warmup is
        ones = eng.comm_sum(np.ones((2 * d + 2) if weighted else (d + 8 * (k + 2))))
        if not np.all(ones == float(world)):
            raise SystemExit('in-library all-reduce returned %r on rank %d, expected %d everywhere' % (ones[:4], rank, world))
    elif sharded:
        if args.schedule == 'residual':
            sys.exit('--schedule residual row-sharded needs the in-library collective: ' + collective)
        eng, red, stream = make_device_shard(n_local, d, k, dtype=sdt, device_index=local_rank, weighted=flavour)
    else:
        eng = RRIEngine(n_local, d, k, dtype=sdt, weighted=flavour, device=local_rank, schedule=args.schedule)
    nnz, t_up = 0, 0.0
    if sparse:
        import scipy.sparse as sp
        nz = Mask.nonzero()                                   # row-major order = CSR order
        counts = torch.bincount(nz[:, 0], minlength=n_local)
        indptr = np.concatenate([[0], np.cumsum(counts.cpu().numpy())]).astype(np.int64)
        A = sp.csr_matrix((X[Mask > 0].cpu().numpy(), nz[:, 1].to(torch.int32).cpu().numpy(), indptr), shape=(n_local, d))
        nnz = int(A.nnz)
        del nz, counts
        t_up = time.perf_counter()
        eng.upload_observed_csr(A)
        t_up = time.perf_counter() - t_up
        Xs_keep, Ms_keep = X[:WEIGHTED_CPU_ROWS].cpu().numpy().astype(np.float64), Mask[:WEIGHTED_CPU_ROWS].cpu().numpy().astype(np.float64)
        del X, Mask, A
        torch.cuda.empty_cache()
    else:
        if args.storage == 'f64':
            X = X.double()
            if weighted:
                Mask = Mask.double()
            torch.cuda.synchronize()   # the handle's stream does not wait for torch's: the tensors must be complete
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        if weighted:
            eng.bind_mask_device(Mask.data_ptr(), Mask.stride(0))
    eng.set_W(W0)
    eng.set_T(T0)
    flags = dict(t_row_sum=1.0, reset_topic_method=None) if weighted else {}
    eng.set_params(**flags)   # plain RRI: no constraints, default resets; WRRI: the RS-fit flags
    row_lo = rank * n_local if cfg['scaling'] == 'weak' else shard_rows(cfg['n'], world, rank)[0]
    if sharded and group is None:
        drv = ShardedRRI(eng, red, k, stream=stream, row_lo=row_lo, n_global=n_global)

    def run(steps):
        if drv is None:
            eng.sweep(steps)          # one C call: the kernels (and, row-sharded, the all-reduces) of `steps` sweeps
        else:
            drv.sweep(steps)

    def fence():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup)
    fence()
    # HIP events around every 8th launch of each kernel: an event record leaves a bubble of a few
    # microseconds on the stream, sampling keeps the timed region representative
    eng.timing_enable(True, every=8)
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    eng.timing_enable(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)

    resid_sched = args.schedule == 'residual'
    launches, pass_ms = eng.timing_read(3 if (weighted or resid_sched) else 0)
    n1, wcol_ms = eng.timing_read(1)
    n2, trow_ms = eng.timing_read(2)
    pass_avg_ms = pass_ms / max(launches, 1)
    # plain: one fused pass reads X once.  weighted: two passes per topic step over (E, M): B reads E and M, C reads
    # E and M and writes E = 5 fp32 arrays with a dense fp32 mask (SURVEY 8d's figure); the 0/1 mask of this workload
    # is bit-packed by the library (1/32 of an array per read), so the schedule ACTUALLY moves 3 + 2/32 arrays per
    # topic step: the roofline line is priced on what is moved, never on the larger formula.
    # explicit-residual schedule: one read + one write of R per topic step (2 n d s)
    mask_packed = weighted and os.environ.get('RRI_MASK_BITS', '1') != '0'
    # round 4: the dense weighted flavour takes ONE read-modify-write pass per topic step (E read + written, the mask read:
    # 2 + 1/32 arrays bit-packed, 3 with an fp32 mask) plus, once per sweep, the read-only pass that takes the first column
    # sums after the rebuild (1 + 1/32 or 2 arrays); the timed launches are these k + 1 per sweep.  The pass over the mask
    # alone between two steps (k_wmcorr_cols / k_wmcorr, 1/32 of an array) is its own kernel, in the T-row chain's segment.
    wpass_one = weighted and not sparse and os.environ.get('RRI_WPASS_ONE', '1') != '0'
    if wpass_one:
        m_arr = 1.0 / 32.0 if mask_packed else 1.0
        arrays_per_launch = (k * (2.0 + m_arr) + (1.0 + m_arr)) / (k + 1.0)
    else:
        arrays_per_launch = (((3.0 + 2.0 / 32.0) if mask_packed else 5.0) / 2.0) if weighted else (2.0 if resid_sched else 1.0)
    bytes_per_launch = float(n_local) * d * es * arrays_per_launch
    sp_step_bytes = 2.0 * (2.0 + 2.0 * es)     # fp32: 20 B per observed entry and topic step
    if sparse:
        # per topic step and observed entry: ONE read-modify-write pass over each of the two copies of the pattern
        # residual, (uint16 offset + value read, value written) = 2 + 4 + 4 B each at fp32 -> 20 B per step, 10 B per
        # timed launch (the factor tables, staged in LDS once per workgroup, are not counted)
        bytes_per_launch = sp_step_bytes / 2.0 * nnz
    # launch-bound sizes whose X fits the chip's registers run whole calls as ONE persistent launch (rri_onchip_kernels.hpp):
    # that launch processes steps * k topic steps, each worth n*d*s algorithmic bytes -- none of which moves through HBM
    onchip = (not weighted) and (not sharded) and eng.onchip_info()[1] > 0
    if onchip and launches:
        bytes_per_launch *= float(args.steps) * k / launches
    achieved = bytes_per_launch / (pass_avg_ms * 1e-3) / 1e9 if launches else 0.0
    sweeps_per_s = args.steps / elapsed
    shards = (n_global / float(cfg['n'])) if cfg['scaling'] == 'weak' else 1.0
    value = sweeps_per_s * shards
    sweep_bytes = (sp_step_bytes * k * nnz if sparse else (2.0 if not weighted else 4.0) * k * n_local * d * es)

    out = {
        'metric': 'RRI sweeps/sec and achieved HBM GB/s on dense X (n x d, rank k)',
        'value': value, 'unit': 'sweeps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': cfg['scaling'],
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': cfg['name'] + (' per GPU (row shard), %d x %d global' % (n_global, d)
                                              if world > 1 and cfg['scaling'] == 'weak' else ''),
                   'n_global': n_global, 'n_per_gpu': n_local, 'd': d, 'k': k,
                   'x_storage': '%s on the observed pattern (%d entries, CSR + CSC), upload %.2f s' % (args.storage, nnz, t_up) if sparse else '%s in HBM' % args.storage,
                   'arithmetic': ('float64 sums, W, T; the factor tables of the pattern-only passes are rounded to the storage type'
                                  if sparse and args.storage == 'f32' else 'float64 (W, T, all sums)'),
                   'flavour': 'WRRI (W_mat)' if weighted else 'plain RRI',
                   'schedule': ('explicit residual: R <- R - dw t^T - w dt^T, one read-modify-write pass per topic step, R rebuilt '
                                'once per sweep' if resid_sched else
                                'maintained masked residual, ONE read-modify-write pass per topic step (both pending rank-one terms, row '
                                'products, next column sums) + a launch over the bit-packed mask alone for the term the W update leaves '
                                'pending (a sparse 0/1 mask: a walk over its set bits that takes (w^2)^T M as well); rebuilt once per sweep'
                                if wpass_one else
                                'maintained masked residual, two passes per topic step' if weighted
                                else 'Gram form: one fused read of X per topic step (row dots + next column sums)'
                                     + ('; at this size ONE persistent launch per call with X resident in registers' if
                                        (not weighted and not sharded and eng.onchip_info()[0]) else '')),
                   'parallelism': ('row-sharded, %d rank(s), 1 all-reduce of %d doubles per topic step; %s'
                                   % (world, (2 * d + 2) if weighted else (d + 8 * (k + 2)), collective)) if sharded else 'single GPU'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBPS, 'traffic': None,
                     'kernel': ('k_sp_blk<float,...>: one read-modify-write pass per topic step over the row copy (both pending rank-one '
                                'terms, row products) and one over the column copy (column sums), averaged; %d observed entries' % nnz if sparse else
                                'k_wpass<float,Y,Z,UPD2,WRITE>: E <- E - M.(dw t^T) - M.(w dt^T) read + written, row products and next column '
                                'sums of the new E, mask ' + ('bit-packed' if mask_packed else 'fp32') + ' (k per sweep; + the read-only '
                                'column-sum pass after the rebuild, 1 per sweep, in the average)' if wpass_one else
                                'k_wpass<float,...> passes B (read E, mask) and C (read E, mask; write E), averaged; mask '
                                + ('bit-packed' if mask_packed else 'fp32') if weighted
                                else 'k_pass<float,Y,Z,UPD=2> (rank-one residual update R <- R - a b^T - a2 b2^T, read + write, fused '
                                     'with the row dots and column sums of the new R)' if resid_sched
                                else 'k_onchip_sweeps: ONE persistent launch for the whole call, X resident in registers (one 512-thread '
                                     'workgroup per CU), two exchanges between workgroups per topic step in which the data is its own hand-over; `achieved` is algorithmic bytes '
                                     '(n*d*4 per topic step) over time and no HBM figure: X is read from HBM once per call' if onchip
                                else 'k_pass<float,Y,Z> (fused row-dot + column-sum pass over X)'),
                     'bytes_per_launch': bytes_per_launch, 'launches': launches, 'avg_ms': pass_avg_ms},
        'sweep_level': {'global_sweeps_per_s': sweeps_per_s,
                        'timed_launch_samples': launches,
                        'survey_formula': ('%d*k*nnz B per sweep (sparse-index formulation; NOT the dense 4*k*n*d*4 figure)' % int(sp_step_bytes) if sparse
                                           else '4*k*n*d*4 B per sweep (dense fp32 mask, residual not rewritten)' if weighted
                                           else '2*k*n*d*4 B per sweep (two BLAS2 passes per topic step, or one read + one write of R)'),
                        'algorithmic_GBps_2knd': sweep_bytes * sweeps_per_s / 1e9,
                        'frac_of_8TBps_2knd': sweep_bytes * sweeps_per_s / 1e9 / HBM_PEAK_GBPS,
                        'kernel_avg_ms': {'pass': pass_avg_ms, 'wcol': wcol_ms / max(n1, 1),
                                          'trow_chain_segment': trow_ms / max(n2, 1)}},
    }

    if sharded:
        # What the communicator itself saw, per rank, gathered to rank 0: the record that the run really had `world` ranks
        # connected and how many all-reduces each of them enqueued (timed region + warm-up + the checked one before it).
        # transport: 'rccl' = ncclAllReduce inside librri_hip.so; 'torch' = the caller-owned protocol over torch.distributed.
        if group is not None:
            crank, cworld, ccalls = eng.comm_stats()
            mine = [float(crank), float(cworld), float(ccalls)]
            transport = 'rccl' if 'RCCL inside' in collective else 'host-callback (rehearsal)'
        else:
            mine = [float(rank), float(dist.get_world_size()), float(drv.allreduce_calls)]
            transport = 'torch.distributed'
        if world > 1:
            tt = torch.tensor(mine, dtype=torch.float64, device=device)
            allr = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(allr, tt)
            rows = [[float(v) for v in t.cpu()] for t in allr]
        else:
            rows = [mine]
        out['rccl'] = {'transport': transport, 'world': int(rows[0][1]), 'worlds_seen_by_rank': [int(r[1]) for r in rows],
                       'ranks': [int(r[0]) for r in rows], 'allreduce_calls': int(rows[0][2]),
                       'allreduce_calls_by_rank': [int(r[2]) for r in rows],
                       'expected_allreduce_calls_timed_region': int(args.steps * k),
                       'doubles_per_allreduce': int((2 * d + 2) if weighted else (d + 8 * (k + 2)))}

    # HBM bytes of the dominant kernel(s): PMC counters collected by this command under rocprofv3 --pmc (separate
    # FETCH_SIZE / WRITE_SIZE passes, tools/pmc_summary.py) -- a STATIC profile of the build whose source stamp it
    # carries; a stamp that differs from the sources of this run is flagged.  gfx950: FETCH_SIZE counts half the bytes
    # of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section), hence 2 x FETCH + WRITE.
    pmc_tag = args.config + ('_residual' if resid_sched else '')
    pmc_file = os.path.join(ROOT, 'profiles', '%s_pmc_hbm_traffic_%s.json' % (PROFILE_ROUND, pmc_tag))
    if os.path.exists(pmc_file):
        try:
            pm = json.load(open(pmc_file))
            if sparse:      # the timed launches: k_sp_blk on the row copy and on the column copy, alternating
                keys = [kk for kk in pm if 'k_sp_blk<float' in kk and pm[kk]['launches'] > 10]
            elif weighted:  # passes B and C (the prologue variant runs 3 times per call: left out)
                keys = [kk for kk in pm if 'k_wpass' in kk and '<float' in kk and pm[kk]['launches'] > 10]
            elif resid_sched:
                keys = [kk for kk in pm if 'k_pass<float, true, true, 2' in kk]
            elif onchip:    # one launch per call, of whatever --steps was: a per-launch counter of another command says nothing here
                keys = []
            else:
                keys = [kk for kk in pm if 'k_pass<float, true, true, 0' in kk]
            # average over the timed launches: every instantiation weighs with the launches it had
            nl = sum(pm[kk]['launches'] for kk in keys)
            tot = sum((2.0 * pm[kk]['FETCH_SIZE_KB_avg'] + pm[kk]['WRITE_SIZE_KB_avg']) * pm[kk]['launches'] for kk in keys) * 1024.0 / max(nl, 1)
            per_launch = 1.0
            if keys:
                out['roofline']['traffic'] = tot * per_launch
                stamp = pm.get('_source_stamp')
                out['roofline']['traffic_source'] = (
                    'static profile: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the same kernels and shape, '
                    'profiles/%s, kernel sources %s; FETCH_SIZE doubled per the gfx950 correction' % (os.path.basename(pmc_file), stamp))
                out['roofline']['traffic_stale'] = stamp != source_stamp()
        except Exception:  # noqa: BLE001
            pass

    thr = int(min(16, os.cpu_count() or 16))       # the box's CPU share for one GPU
    if rank == 0 and world == 1 and weighted and not args.no_cpu_baseline:
        rows = min(WEIGHTED_CPU_ROWS, n_local)
        if sparse:
            Xs, Ms = Xs_keep[:rows], Ms_keep[:rows]
        else:
            Xs, Ms = X[:rows].cpu().numpy().astype(np.float64), Mask[:rows].cpu().numpy().astype(np.float64)
        from oracle import rri_oracle as orc
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=thr):
            t1 = time.perf_counter()
            ref = orc.nmf(Xs, k, W_in=W0[:rows].copy(), T_in=T0.copy(), W_mat=Ms, max_iter=1, eps_stop=-1,
                          compute_obj_each_iter=True, **flags)
            dt1 = time.perf_counter() - t1
        out['cpu_baseline'] = dict(value=(1.0 / dt1) * rows / float(n_local), unit='sweeps/s', cores=thr, kind='port',
                                   sample='first %d of %d rows, 1 sweep (two n*d*k GEMMs per topic step, as the reference), numpy '
                                          'float64 on %d threads; sample rate %.4f sweeps/s, the value is that rate times %.4g (rows '
                                          'of the sample / rows of the workload; a full-size sweep of this flavour takes ~5 CPU-minutes)'
                                          % (rows, n_local, thr, 1.0 / dt1, rows / float(n_local)))
        with RRIEngine(rows, d, k, dtype=sdt, weighted='sparse' if sparse else True, device=local_rank) as e2:
            if sparse:
                import scipy.sparse as sp
                As = sp.csr_matrix(Ms)
                As.data = Xs[Ms > 0]
                e2.upload_observed_csr(As)
            else:
                e2.upload_X(Xs); e2.upload_mask(Ms)
            e2.set_W(W0[:rows]); e2.set_T(T0); e2.set_params(**flags)
            e2.sweep(1)
            Wg, Tg, og = e2.get_W(), e2.get_T(), e2.objective()
        rec = lambda W_, T_: Ms * (W_ @ T_)
        out['parity_sample'] = {'sweeps': 1, 'rows': rows,
                                'relfro_W': relfro(Wg, ref['W']), 'relfro_T': relfro(Tg, ref['T']),
                                'relfro_masked_WT': relfro(rec(Wg, Tg), rec(ref['W'], ref['T'])),
                                'rel_objective': abs(og - ref['obj_history'][-1]) / abs(ref['obj_history'][-1]),
                                'statement': 'the first %d rows of the workload -- same X, mask and start -- as a problem of its own, one '
                                             'sweep, against the CPU oracle (the reference\'s operation order).  The residual is stored in '
                                             '%s; at the FULL size the first sweep from this random start amplifies a storage rounding '
                                             'of 6e-8 to percents in W, T while objective and masked reconstruction M.(WT) agree '
                                             '(tests/test_full_size_gpu.py, DESIGN.md 7)' % (rows, args.storage)}
        out['gpu_over_cpu'] = value / out['cpu_baseline']['value']
    if rank == 0 and world == 1 and not weighted:
        # the explicit rank-one residual update R <- R - a b^T (read + write, fused residual products), with the handle's
        # own factors W[:,0], T[0,:] as a, b (tests/test_residual_gpu.py checks the same kernel against numpy / torch)
        try:
            r1_ms = eng.bench_rank1_update(5)
            rw_bytes = 2.0 * float(n_local) * d * es
            out['rank1_update'] = {'bound': 'hbm', 'achieved': rw_bytes / (r1_ms * 1e-3) / 1e9,
                                   'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                   'frac': rw_bytes / (r1_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                   'bytes_per_launch': rw_bytes, 'avg_ms': r1_ms,
                                   'kernel': 'k_pass<float,Y,Z,UPD=1>: R <- R - w_0 t_0^T on a scratch copy of X, row dots and column sums of the new R'}
            # (rounds 1-3 also quoted a fraction of a plain 16-byte copy measured in the run: that copy reaches 4.7-5.7 TB/s here,
            # below what the passes themselves reach, so the fraction said nothing -- dropped; the fractions are of 8 TB/s)
        except Exception as e:  # noqa: BLE001
            out['rank1_update'] = {'error': str(e)}
        if not resid_sched and args.config in ('c3', 'c2', 'mid'):
            # the whole explicit-residual schedule on the same X, start and sweeps: its rate, its kernel, and how far its
            # factors are from the default schedule's after the same sweeps
            try:
                with RRIEngine(n_local, d, k, dtype=sdt, device=local_rank, schedule='residual') as e3:
                    e3.bind_X_device(X.data_ptr(), X.stride(0))
                    e3.set_W(W0), e3.set_T(T0), e3.set_params()
                    e3.sweep(max(args.warmup, 1))
                    e3.synchronize()
                    e3.timing_enable(True, every=8)
                    t3 = time.perf_counter()
                    e3.sweep(args.steps)
                    e3.synchronize()
                    dt3 = time.perf_counter() - t3
                    c3n, c3ms = e3.timing_read(3)
                    Wr, Tr = e3.get_W(), e3.get_T()
                Wd, Td = eng.get_W(), eng.get_T()
                avg3 = c3ms / max(c3n, 1)
                out['rank1_update']['schedule'] = {
                    'what': 'RRIEngine(schedule=\'residual\'): every topic step is ONE rank-one residual update pass (k_pass UPD=2: '
                            'R <- R - dw t^T - w dt^T, read + write) fused with R t and R^T w; R rebuilt once per sweep',
                    'sweeps_per_s': args.steps / dt3, 'kernel_avg_ms': avg3,
                    'achieved_GBps': rw_bytes / (avg3 * 1e-3) / 1e9 if c3n else None,
                    'frac': rw_bytes / (avg3 * 1e-3) / 1e9 / HBM_PEAK_GBPS if c3n else None,
                    'sweep_frac_of_8TBps_2knd': sweep_bytes * (args.steps / dt3) / 1e9 / HBM_PEAK_GBPS,
                    'vs_default_schedule_after_%d_sweeps' % (args.warmup + args.steps): {'relfro_W': relfro(Wr, Wd), 'relfro_T': relfro(Tr, Td)}}
            except Exception as e:  # noqa: BLE001
                out['rank1_update']['schedule'] = {'error': str(e)}
        if onchip:
            # the same problem, start and sweeps through the launch-per-phase schedule (RRI_ONCHIP=0, read by rri_create):
            # three launches per topic step with X re-read from the caches
            try:
                saved = os.environ.get('RRI_ONCHIP')
                os.environ['RRI_ONCHIP'] = '0'
                try:
                    e4 = RRIEngine(n_local, d, k, dtype=sdt, device=local_rank)
                finally:
                    if saved is None:
                        os.environ.pop('RRI_ONCHIP', None)
                    else:
                        os.environ['RRI_ONCHIP'] = saved
                with e4:
                    e4.bind_X_device(X.data_ptr(), X.stride(0))
                    e4.set_W(W0), e4.set_T(T0), e4.set_params()
                    e4.sweep(max(args.warmup, 1))
                    e4.synchronize()
                    t4 = time.perf_counter()
                    e4.sweep(args.steps)
                    e4.synchronize()
                    dt4 = time.perf_counter() - t4
                    W4, T4 = e4.get_W(), e4.get_T()
                Wd, Td = eng.get_W(), eng.get_T()
                out['launch_per_phase_schedule'] = {
                    'what': 'RRI_ONCHIP=0: k_trow_small + k_pass + k_wcol per topic step, no HIP-event timing',
                    'sweeps_per_s': args.steps / dt4, 'us_per_topic_step': 1e6 * dt4 / args.steps / k,
                    'persistent_launch_us_per_topic_step': 1e6 * elapsed / args.steps / k,
                    'vs_persistent_launch_after_%d_sweeps' % (max(args.warmup, 1) + args.steps): {'relfro_W': relfro(Wd, W4), 'relfro_T': relfro(Td, T4)}}
            except Exception as e:  # noqa: BLE001
                out['launch_per_phase_schedule'] = {'error': str(e)}
        if not args.no_cpu_baseline:
            full = args.config in ('c2', 'c3', 'mid')
            rows = args.cpu_rows or (n_local if full else min(100000, n_local))
            rows = min(rows, n_local)
            times, facs, X64 = cpu_plain(X, W0, T0, rows, args.cpu_sweeps, thr, n_local)
            rate = len(times) / sum(times)
            out['cpu_baseline'] = dict(
                value=rate * rows / float(n_local), unit='sweeps/s', cores=thr, kind='port',
                sample=('ALL %d rows of the same X' % rows if rows == n_local else 'first %d of %d rows of the same X (rate times %.4g)'
                        % (rows, n_local, rows / float(n_local))) +
                       ', %d timed sweeps (%s s), numpy float64 + OpenBLAS on %d threads (%d host cpus), oracle/rri_oracle.py'
                       % (len(times), ', '.join('%.2f' % t for t in times), thr, os.cpu_count() or 0))
            # one BLAS thread, on a bounded row sample (a full-size sweep on one thread takes minutes)
            rows1 = min(rows, 10000)
            t1s, _, _ = cpu_plain(X, W0, T0, rows1, 1, 1, n_local, want_factors=False)
            out['cpu_baseline']['one_thread'] = dict(
                value=(1.0 / t1s[0]) * rows1 / float(n_local), cores=1,
                sample='first %d rows, 1 sweep on 1 BLAS thread: %.3f sweeps/s on the sample, times %.4g' % (rows1, 1.0 / t1s[0], rows1 / float(n_local)))
            # parity in the same run: the device path on the same rows, same start, after 1 .. cpu_sweeps sweeps
            par = {'rows': rows, 'what': 'device (fp32 X in HBM, float64 arithmetic) vs the CPU run above on the same rows and start'}
            with RRIEngine(rows, d, k, dtype=sdt, device=local_rank, schedule=args.schedule) as e2:
                Xr = X[:rows]
                e2.bind_X_device(Xr.data_ptr(), Xr.stride(0))
                e2.set_W(W0[:rows]), e2.set_T(T0), e2.set_params()
                srows = min(rows, 2000)
                for s_, (Wc, Tc) in enumerate(facs, 1):
                    e2.sweep(1)
                    Wg, Tg = e2.get_W(), e2.get_T()
                    par['after_%d_sweeps' % s_] = {'relfro_W': relfro(Wg, Wc), 'relfro_T': relfro(Tg, Tc),
                                                   'relfro_WT_first_%d_rows' % srows: relfro(Wg[:srows] @ Tg, Wc[:srows] @ Tc)}
            last = par['after_%d_sweeps' % len(facs)]
            par.update(sweeps=len(facs), relfro_W=last['relfro_W'], relfro_T=last['relfro_T'])
            # control: the CPU restatement against itself with every entry of W0 moved to the next double -- the
            # iteration's own sensitivity, which bounds what any two implementations (or two BLAS builds) can show
            from oracle import rri_oracle as orc
            from threadpoolctl import threadpool_limits
            crow = min(rows, 20000)
            with threadpool_limits(limits=thr):
                Wa, Ta = W0[:crow].astype(np.float64).copy(), T0.astype(np.float64).copy()
                Wb, Tb = np.nextafter(Wa, np.inf), Ta.copy()
                orc.plain_sweeps(X64[:crow], Wa, Ta, len(facs))
                orc.plain_sweeps(X64[:crow], Wb, Tb, len(facs))
            par['reference_self_sensitivity'] = {'rows': crow, 'sweeps': len(facs), 'relfro_W': relfro(Wb, Wa), 'relfro_T': relfro(Tb, Ta),
                                                 'what': 'CPU restatement vs itself, every entry of W0 one ulp up (np.nextafter), first %d rows' % crow}
            del X64
            out['parity_sample'] = par
            if args.config == 'c3' and not resid_sched:
                # PCIe-inclusive: bringing X to the device once from pageable host memory (never part of `value`)
                Xh = X[:10000].cpu().numpy()
                with RRIEngine(10000, d, k, dtype=np.float32, device=local_rank) as e4:
                    e4.upload_X(Xh)
                    tu = time.perf_counter()
                    e4.upload_X(Xh)
                    tu = time.perf_counter() - tu
                h2d = Xh.nbytes / tu / 1e9
                out['pcie_inclusive'] = {'h2d_GBps_pageable': h2d, 'x_upload_ms': 1e3 * bytes_per_launch / (h2d * 1e9),
                                         'note': 'rri_upload_X of a 10000-row sample from pageable host memory; uploading the whole X once '
                                                 'costs x_upload_ms = %.1f sweeps; never part of `value`' % (bytes_per_launch / (h2d * 1e9) * sweeps_per_s)}
            out['gpu_over_cpu'] = value / out['cpu_baseline']['value'] if out['cpu_baseline']['value'] else None
    if sharded and world > 1 and not weighted and not args.no_cpu_baseline:
        # N > 1: parity THROUGH the N-rank path and the CPU baseline from ONE host run.  A problem of the workload's own
        # shape (min(n, 100000) x d, same k, same generator) is row-sharded over the same ranks and communicator, two sweeps;
        # rank 0 runs the CPU oracle on ALL its rows (timed: that is the baseline) and compares its block of W and the
        # replicated T.  Every rank takes part in the sharded run.
        rows_s = min(cfg['n'], 100000)
        Xs_dev = device_planted_shard(rows_s, d, k, seed=77, device=device)          # the same matrix on every rank
        lo_s, hi_s = shard_rows(rows_s, world, rank)
        xsum = torch.zeros((), dtype=torch.float64, device=device)
        for lo_ in range(0, rows_s, 100000):
            xsum += Xs_dev[lo_:lo_ + 100000].sum(dtype=torch.float64)
        as_ = (float(xsum) / (float(rows_s) * d) / k) ** 0.5
        gs = torch.Generator(device=device)
        gs.manual_seed(5)
        Ws0 = (as_ * torch.rand(rows_s, k, device=device, generator=gs, dtype=torch.float64)).cpu().numpy()
        Ts0 = (as_ * torch.rand(k, d, device=device, generator=gs, dtype=torch.float64)).cpu().numpy()
        torch.cuda.synchronize()
        Xs_loc = Xs_dev[lo_s:hi_s]
        sizes_s = [shard_rows(rows_s, world, r)[1] - shard_rows(rows_s, world, r)[0] for r in range(world)]
        if group is not None:
            e5 = RRIEngine(hi_s - lo_s, d, k, dtype=np.float32, device=local_rank)
            e5.attach_group(group.resized(sizes_s))
            e5.bind_X_device(Xs_loc.data_ptr(), Xs_loc.stride(0))
            e5.set_W(Ws0[lo_s:hi_s]), e5.set_T(Ts0), e5.set_params()
            e5.sweep(2)
        else:
            e5, red5, st5 = make_device_shard(hi_s - lo_s, d, k, dtype=np.float32, device_index=local_rank)
            e5.bind_X_device(Xs_loc.data_ptr(), Xs_loc.stride(0))
            e5.set_W(Ws0[lo_s:hi_s]), e5.set_T(Ts0), e5.set_params()
            ShardedRRI(e5, red5, k, stream=st5, row_lo=lo_s, n_global=rows_s).sweep(2)
        Wg5, Tg5 = e5.get_W(), e5.get_T()
        e5.close()
        if rank == 0:
            times, facs, _ = cpu_plain(Xs_dev, Ws0, Ts0, rows_s, 2, thr, rows_s)
            Wc, Tc = facs[-1]
            out['parity_sample'] = {'what': 'a %d x %d, k=%d problem (the workload\'s generator) row-sharded over the same %d ranks and collective: rank '
                                            '0\'s rows of W and the replicated T against the CPU oracle on all rows' % (rows_s, d, k, world),
                                    'sweeps': 2, 'rows': rows_s, 'relfro_W': relfro(Wg5, Wc[lo_s:hi_s]), 'relfro_T': relfro(Tg5, Tc)}
            rate_s = len(times) / sum(times)                          # sweeps/s of a rows_s-row problem on this host
            shard_rate = rate_s * rows_s / float(n_local)             # ... of ONE n_local-row shard
            out['cpu_baseline'] = dict(value=shard_rate * (1.0 if cfg['scaling'] == 'weak' else 1.0 / world), unit='sweeps/s', cores=thr, kind='port',
                                       sample='the %d x %d problem above, ALL its rows, 2 timed sweeps (%s s), numpy float64 on %d threads (%.3f sweeps/s), '
                                              'times %.4g (rows / rows of a shard); in the unit of `value` ONE host like this one does %s' % (
                                                  rows_s, d, ', '.join('%.2f' % t for t in times), thr, rate_s, rows_s / float(n_local),
                                                  'this many shard-sweeps per second (the job has %d shards)' % world
                                                  if cfg['scaling'] == 'weak' else 'the whole %d-row problem at this rate (shard rate / %d)' % (n_global, world)))
            out['gpu_over_cpu'] = value / out['cpu_baseline']['value']
        del Xs_dev
    eng.close()
    if group is not None:
        group.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + '\n').encode())
    os.close(line_fd)


if __name__ == '__main__':
    main()
