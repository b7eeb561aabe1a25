#!/usr/bin/env python3
"""bench.py -- RRI sweeps/sec and achieved HBM GB/s of the MI355X path (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c4]

A step is one RRI sweep (k topic steps = k T-row + k W-column updates, nmf.py:415-476) over a
synthetic dense fp32 X that is already resident in HBM when the timed region starts.

Workloads (BASELINE.json configs):
    c3 (default)  100000 x 10000, k=50: the roofline run the north_star target is quoted on.
                  N > 1: WEAK scaling -- every rank holds its own 100000-row shard of an
                  (N*100000) x 10000 problem (T replicated, one RCCL all-reduce of d+8(k+2) doubles per
                  topic step).  `value` is the whole job's rate in sweeps/s of a 100000-row shard,
                  i.e. N * (global sweeps/s): it equals plain sweeps/s at N = 1.
    c2            10000 x 1000, k=20 (X is Infinity-Cache resident: latency-, not HBM-bound).
    c4            1000000 x 10000, k=50 split by rows over the N ranks (STRONG scaling).
    c5            WRRI: c3's shape with a dense fp32 5 %-observed 0/1 mask, recommender flags (T clipped to
                  [0,1], no resets); single GPU.  Algorithmic bytes 5*n*d*4 per topic step.

Prints ONE JSON line on rank 0 (contract in the task description) including
    roofline     : the dominant kernel (the fused X pass) -- algorithmic bytes n_local*d*4 per launch
                   over its average duration, measured with HIP events on the kernel's stream
                   inside the timed region, against 8 TB/s
    cpu_baseline : the float64 numpy restatement (oracle/) timed on this box's host cores on a
                   bounded row sample of the same X, scaled linearly in n to the full workload.
"""
import argparse
import json
import os
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c3', choices=sorted(CONFIGS))
    ap.add_argument('--cpu-sweeps', type=int, default=2)
    ap.add_argument('--cpu-rows', type=int, default=10000, help='rows of X in the CPU baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--storage', default='f32', choices=['f32', 'f64'],
                    help='storage type of X / mask / residual in HBM (the BASELINE configs are fp32; arithmetic is float64 either way)')
    ap.add_argument('--force-sharded', action='store_true',
                    help='use the row-sharded driver (RCCL all-reduce per topic step) even with one rank')
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child job.  Done before
    anything touches the GPU; the child is waited for, never exec'ed over this process."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(29500 + os.getpid() % 2000),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


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


def cpu_baseline(Xs_host, W0s, T0, n_full, sweeps=2, threads=16):
    """the numpy float64 restatement on the host cores: 1 warm-up + `sweeps` timed sweeps on the sample.
    BLAS threads are capped at the box's CPU share for one GPU (16).  Also returns how far the SAME
    restatement moves when its start is perturbed by one ulp: the iteration's own sensitivity, which
    bounds the agreement any two implementations (or two BLAS builds) can show."""
    import numpy as np
    from oracle import rri_oracle as orc
    from threadpoolctl import threadpool_limits
    threads = int(min(threads, os.cpu_count() or threads))
    X = Xs_host.astype(np.float64)
    W, T = W0s.astype(np.float64).copy(), T0.astype(np.float64).copy()
    with threadpool_limits(limits=threads):
        orc.plain_sweeps(X, W, T, 1)
        t0 = time.perf_counter()
        orc.plain_sweeps(X, W, T, sweeps)
        dt = time.perf_counter() - t0
        Wp = W0s.astype(np.float64) * (1.0 + 2.0 ** -52 * np.sign(np.random.RandomState(7).randn(*W0s.shape)))
        Tp = T0.astype(np.float64).copy()
        orc.plain_sweeps(X, Wp, Tp, 1 + sweeps)
    rate_sample = sweeps / dt
    frac = X.shape[0] / float(n_full)
    sens = {'relfro_W': float(np.linalg.norm(Wp - W) / np.linalg.norm(W)),
            'relfro_T': float(np.linalg.norm(Tp - T) / np.linalg.norm(T)),
            'what': 'CPU restatement vs itself with W0 perturbed by 1 ulp, same sweeps'}
    return dict(value=rate_sample * frac, unit='sweeps/s', cores=threads, kind='port',
                sample='first %d of %d rows of the same X, %d timed sweeps after 1 warm-up, numpy float64 + '
                       'OpenBLAS on %d threads (%d host cpus); sample rate %.3f sweeps/s scaled by %.4g '
                       '(work is linear in n)' % (X.shape[0], n_full, sweeps, threads, os.cpu_count() or 0,
                                                  rate_sample, frac)), W, T, sens


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and world == 1:
        sys.exit(relaunch_under_torchrun(args))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # Rehearsal of the multi-rank path on a ONE-GPU box (never the measured configuration): RRI_BENCH_REHEARSAL=1 puts
    # every rank on device 0 and runs the collectives over gloo (RCCL wants one device per rank)
    rehearsal = os.environ.get('RRI_BENCH_REHEARSAL', '0') == '1'
    if rehearsal:
        local_rank = 0

    import numpy as np
    import torch
    import torch.distributed as dist
    from rri_nmf_amd.distributed import ShardedRRI, make_device_shard, shard_rows

    cfg = CONFIGS[args.config]
    d, k = cfg['d'], cfg['k']
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
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
    mean = X.sum(dtype=torch.float64)
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

    weighted = bool(cfg.get('weighted'))
    Mask = None
    if weighted:
        gm = torch.Generator(device=device)
        gm.manual_seed(2 if world == 1 else 3000 + rank)
        Mask = (torch.rand(n_local, d, device=device, generator=gm) < 0.05).to(torch.float32)
        X.mul_(Mask)
        torch.cuda.synchronize()
    sparse = bool(cfg.get('sparse'))
    sdt = np.float32 if args.storage == 'f32' else np.float64
    es = 4 if args.storage == 'f32' else 8
    eng, red, stream = make_device_shard(n_local, d, k, dtype=sdt, device_index=local_rank,
                                         weighted='sparse' if sparse else weighted)
    nnz = 0
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
        Xs_keep, Ms_keep = X[:2000].cpu().numpy().astype(np.float64), Mask[:2000].cpu().numpy().astype(np.float64)
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
    drv = ShardedRRI(eng, red, k, stream=stream, row_lo=row_lo, n_global=n_global) if sharded else None

    def run(steps):
        if drv is None:
            eng.sweep(steps)
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

    launches, pass_ms = eng.timing_read(3 if weighted else 0)
    n1, wcol_ms = eng.timing_read(1)
    n2, trow_ms = eng.timing_read(2)
    pass_avg_ms = pass_ms / max(launches, 1)
    # plain: one fused pass reads X once.  weighted: two passes per topic step over (E, M): B reads E and M, C reads
    # E and M and writes E = 5 fp32 arrays with a dense fp32 mask (SURVEY 8d's figure); the 0/1 mask of this workload
    # is bit-packed by the library (1/32 of an array per read), so the schedule ACTUALLY moves 3 + 2/32 arrays per
    # topic step: the roofline line is priced on what is moved, never on the larger formula
    mask_packed = weighted and os.environ.get('RRI_MASK_BITS', '1') != '0'
    arrays_per_launch = (((3.0 + 2.0 / 32.0) if mask_packed else 5.0) / 2.0) if weighted else 1.0
    bytes_per_launch = float(n_local) * d * es * arrays_per_launch
    if sparse:
        # per topic step and observed entry: pass B reads (uint16 offset, fp32 value) of the row copy = 6 B; pass C
        # reads and rewrites both copies = 2 x (2 + 4 + 4) B; two timed passes per step -> 13 B per entry and pass on
        # average (the factor tables, staged in LDS once per workgroup, are not counted)
        bytes_per_launch = (6.0 + 5.0 * es) / 2.0 * nnz   # fp32: 13 B
    achieved = bytes_per_launch / (pass_avg_ms * 1e-3) / 1e9 if launches else 0.0
    sweeps_per_s = args.steps / elapsed
    shards = (n_global / float(cfg['n'])) if cfg['scaling'] == 'weak' else 1.0
    value = sweeps_per_s * shards

    out = {
        'metric': 'RRI sweeps/sec and achieved HBM GB/s on dense X (n x d, rank k)',
        'value': value, 'unit': 'sweeps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': cfg['scaling'],
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': cfg['name'] + (' per GPU (row shard), %d x %d global' % (n_global, d)
                                              if world > 1 and cfg['scaling'] == 'weak' else ''),
                   'n_global': n_global, 'n_per_gpu': n_local, 'd': d, 'k': k,
                   'x_storage': '%s on the observed pattern (%d entries, CSR + CSC), upload %.2f s' % (args.storage, nnz, t_up) if sparse else '%s in HBM' % args.storage, 'arithmetic': 'float64 (W, T, all sums)', 'flavour': 'WRRI (W_mat)' if weighted else 'plain RRI',
                   'parallelism': 'row-sharded, %d rank(s), 1 all-reduce of %d doubles per topic step'
                                  % (world, (2 * d + 2) if weighted else (d + 8 * (k + 2))) if world > 1 else 'single GPU'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBPS, 'traffic': None,
                     'kernel': ('k_sp_blk<float,...> passes B (row copy: read) and C (row + column copies: read, write), '
                                'averaged; %d observed entries' % nnz if sparse else
                                'k_wpass<float,...> passes B (read E, mask) and C (read E, mask; write E), averaged; mask '
                                + ('bit-packed' if mask_packed else 'fp32') if weighted
                                else 'k_pass<float,Y,Z> (fused row-dot + column-sum pass over X)'),
                     'bytes_per_launch': bytes_per_launch, 'launches': launches, 'avg_ms': pass_avg_ms},
        'sweep_level': {'global_sweeps_per_s': sweeps_per_s,
                        'timed_launch_samples': launches,
                        'survey_formula': ('26*k*nnz B per sweep (sparse-index formulation; NOT the dense 4*k*n*d*4 figure)' if sparse
                                           else '4*k*n*d*4 B per sweep (dense fp32 mask, residual not rewritten)' if weighted
                                           else '2*k*n*d*4 B per sweep (two BLAS2 passes per topic step)'),
                        'algorithmic_GBps_2knd': ((6.0 + 5.0 * es) * k * nnz if sparse else (2.0 if not weighted else 4.0) * k * n_local * d * es) * sweeps_per_s / 1e9,
                        'frac_of_8TBps_2knd': ((6.0 + 5.0 * es) * k * nnz if sparse else (2.0 if not weighted else 4.0) * k * n_local * d * es) * sweeps_per_s / 1e9 / HBM_PEAK_GBPS,
                        'kernel_avg_ms': {'pass': pass_avg_ms, 'wcol': wcol_ms / max(n1, 1),
                                          'trow_chain_segment': trow_ms / max(n2, 1)}},
    }

    # HBM bytes of the dominant kernel(s) from PMC counters collected by the same command under rocprofv3 --pmc
    # (separate FETCH_SIZE / WRITE_SIZE passes, tools/pmc_summary.py).  gfx950: FETCH_SIZE counts half the bytes of
    # a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section), hence 2 x FETCH + WRITE.
    pmc_file = os.path.join(ROOT, 'profiles', 'r01_pmc_hbm_traffic_%s.json' % args.config)
    if os.path.exists(pmc_file):
        try:
            pm = json.load(open(pmc_file))
            if sparse:      # one timed pass = pass B, or pass C on both copies
                keys = [kk for kk in pm if 'k_sp_blk<float' in kk]
                per_launch = 0.5
            elif weighted:  # passes B and C (the prologue variant runs 3 times per call: left out)
                keys = [kk for kk in pm if 'k_wpass<float' in kk and pm[kk]['launches'] > 10]
                per_launch = 0.5
            else:
                keys = [kk for kk in pm if 'k_pass<float, true, true, false' in kk]
                per_launch = 1.0
            tot = sum(2.0 * pm[kk]['FETCH_SIZE_KB_avg'] + pm[kk]['WRITE_SIZE_KB_avg'] for kk in keys) * 1024.0
            out['roofline']['traffic'] = tot * per_launch
            out['roofline']['traffic_source'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the same '
                                                 'kernels and shape, profiles/%s; FETCH_SIZE doubled per the gfx950 '
                                                 'correction' % os.path.basename(pmc_file))
        except Exception:  # noqa: BLE001
            pass

    if rank == 0 and world == 1 and weighted and not args.no_cpu_baseline:
        rows = min(2000, n_local)
        if sparse:
            Xs, Ms = Xs_keep[:rows], Ms_keep[:rows]
        else:
            Xs, Ms = X[:rows].cpu().numpy().astype(np.float64), Mask[:rows].cpu().numpy().astype(np.float64)
        from oracle import rri_oracle as orc
        from threadpoolctl import threadpool_limits
        thr = int(min(16, os.cpu_count() or 16))
        with threadpool_limits(limits=thr):
            t1 = time.perf_counter()
            ref = orc.nmf(Xs, k, W_in=W0[:rows].copy(), T_in=T0.copy(), W_mat=Ms, max_iter=1, eps_stop=-1, **flags)
            dt1 = time.perf_counter() - t1
        out['cpu_baseline'] = dict(value=(1.0 / dt1) * rows / float(n_local), unit='sweeps/s', cores=thr, kind='port',
                                   sample='first %d of %d rows, 1 sweep (two n*d*k GEMMs per topic step), numpy float64 on %d '
                                          'threads; sample rate %.4f sweeps/s scaled by %.4g' % (rows, n_local, thr, 1.0 / dt1, rows / float(n_local)))
        from rri_nmf_amd.engine import RRIEngine
        with RRIEngine(rows, d, k, dtype=np.float32, weighted='sparse' if sparse else True, device=local_rank) as e2:
            if sparse:
                import scipy.sparse as sp
                As = sp.csr_matrix(Ms)
                As.data = Xs[Ms > 0]
                e2.upload_observed_csr(As)
            else:
                e2.upload_X(Xs); e2.upload_mask(Ms)
            e2.set_W(W0[:rows]); e2.set_T(T0); e2.set_params(**flags)
            e2.sweep(1)
            Wg, Tg = e2.get_W(), e2.get_T()
        rec = lambda W_, T_: Ms * (W_ @ T_)
        out['parity_sample'] = {'sweeps': 1, 'rows': rows,
                                'relfro_W': float(np.linalg.norm(Wg - ref['W']) / np.linalg.norm(ref['W'])),
                                'relfro_T': float(np.linalg.norm(Tg - ref['T']) / np.linalg.norm(ref['T'])),
                                'relfro_masked_WT': float(np.linalg.norm(rec(Wg, Tg) - rec(ref['W'], ref['T'])) / np.linalg.norm(rec(ref['W'], ref['T'])))}
        out['gpu_over_cpu'] = value / out['cpu_baseline']['value']
    if rank == 0 and world == 1 and not weighted:
        # the explicit rank-one residual update R <- R - a b^T (read + write, fused residual products)
        try:
            r1_ms = eng.bench_rank1_update(5)
            cp_ms = eng.bench_stream_copy(5)
            out['rank1_update'] = {'bound': 'hbm', 'achieved': 2 * bytes_per_launch / (r1_ms * 1e-3) / 1e9,
                                   'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                   'frac': 2 * bytes_per_launch / (r1_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                   'bytes_per_launch': 2 * bytes_per_launch, 'avg_ms': r1_ms,
                                   'stream_copy_GBps': 2 * bytes_per_launch / (cp_ms * 1e-3) / 1e9}
            # "report fraction of 8.0 and of [the achievable float4 copy]", here the copy measured in this run
            copy_gbps = out['rank1_update']['stream_copy_GBps']
            out['roofline']['frac_of_measured_copy'] = achieved / copy_gbps   # a read-only pass can exceed a read+write copy
            out['roofline']['measured_copy_GBps'] = copy_gbps
            out['rank1_update']['frac_of_measured_copy'] = out['rank1_update']['achieved'] / copy_gbps
        except Exception as e:  # noqa: BLE001
            out['rank1_update'] = {'error': str(e)}
        if not args.no_cpu_baseline:
            rows = min(args.cpu_rows, n_local)
            Xs = X[:rows].cpu().numpy()
            cb, Wc, Tc, sens = cpu_baseline(Xs, W0[:rows], T0, n_local)
            out['cpu_baseline'] = cb
            # parity in the same run: the device path on the same sample, equal sweeps (1 + 2)
            from rri_nmf_amd.engine import RRIEngine
            with RRIEngine(rows, d, k, dtype=np.float32, device=local_rank) as e2:
                e2.upload_X(Xs)                      # warm-up (allocation)
                tu = time.perf_counter()
                e2.upload_X(Xs)
                tu = time.perf_counter() - tu
                h2d = Xs.nbytes / tu / 1e9
                out['pcie_inclusive'] = {
                    'h2d_GBps_pageable': h2d, 'x_upload_ms': 1e3 * bytes_per_launch / (h2d * 1e9),
                    'note': 'rri_upload_X of the %d-row sample from pageable host memory; uploading the whole X once '
                            'costs x_upload_ms = %.1f sweeps; never part of `value`'
                            % (rows, bytes_per_launch / (h2d * 1e9) * sweeps_per_s)}
                e2.set_W(W0[:rows])
                e2.set_T(T0)
                e2.set_params()
                e2.sweep(3)
                Wg, Tg = e2.get_W(), e2.get_T()
            out['parity_sample'] = {
                'sweeps': 3, 'rows': rows,
                'relfro_W': float(np.linalg.norm(Wg - Wc) / np.linalg.norm(Wc)),
                'relfro_T': float(np.linalg.norm(Tg - Tc) / np.linalg.norm(Tc)),
                'relfro_WT': float(np.linalg.norm(Wg @ Tg - Wc @ Tc) / np.linalg.norm(Wc @ Tc)),
                'reference_self_sensitivity': sens}
            out['gpu_over_cpu'] = value / cb['value'] if cb['value'] else None
    eng.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
