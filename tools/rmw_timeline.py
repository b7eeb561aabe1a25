#!/usr/bin/env python3
"""Is the "slow mode" of the read-modify-write pass (k_pass<UPD=2>: 1.33 or 1.51 ms per 8 GB at C3) a STATE THE CARD MOVES INTO
UNDER LOAD rather than a property of a process?  Round 2 saw short runs (--steps 6) fast and long runs slow; PMC passes, which
leave gaps between the kernels, always fast (profiles/r03_rmw_pmc_by_process.txt).  One process: the explicit-residual schedule
sweeps continuously for `busy` seconds (pass duration by HIP events, averaged per 2 sweeps = 100 launches), idles for `idle`
seconds, sweeps again; the card's power, caps, clocks and temperatures are sampled from sysfs all along.  A second leg does the
same with the default schedule's read-only pass (k_pass, 4 GB per launch).
    python3 tools/rmw_timeline.py [busy_s idle_s]
"""
import glob
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import device_planted_shard          # noqa: E402
from rri_nmf_amd.engine import RRIEngine        # noqa: E402


def sysfs_base():
    base = [b for b in sorted(glob.glob('/sys/class/drm/card*/device')) if os.path.exists(os.path.join(b, 'pp_dpm_sclk'))]
    base = base[0] if base else None
    hw = (sorted(glob.glob(os.path.join(base, 'hwmon', 'hwmon*'))) or [None])[0] if base else None
    return base, hw


def read_num(path, scale):
    try:
        return float(open(path).read().strip()) / scale
    except (OSError, ValueError):
        return float('nan')


def active_level(path):
    try:
        for ln in open(path).read().splitlines():
            if ln.rstrip().endswith('*'):
                return ln.split(':')[1].replace('*', '').strip()
    except OSError:
        pass
    return '?'


class Sampler(threading.Thread):
    def __init__(self, t0):
        super().__init__(daemon=True)
        self.t0, self.stop_flag, self.rows = t0, False, []
        self.base, self.hw = sysfs_base()

    def run(self):
        while not self.stop_flag and self.base:
            row = {'t': time.time() - self.t0}
            for f in ('sclk', 'mclk', 'fclk'):
                row[f] = active_level(os.path.join(self.base, 'pp_dpm_' + f))
            if self.hw:
                p = read_num(os.path.join(self.hw, 'power1_average'), 1e6)
                if p != p:
                    p = read_num(os.path.join(self.hw, 'power1_input'), 1e6)
                row['W'] = p
                for t in sorted(glob.glob(os.path.join(self.hw, 'temp*_input'))):
                    lab = t.replace('_input', '_label')
                    name = open(lab).read().strip() if os.path.exists(lab) else os.path.basename(t)
                    row['T_' + name] = read_num(t, 1e3)
            self.rows.append(row)
            time.sleep(0.1)

    def near(self, t):
        if not self.rows:
            return {}
        return min(self.rows, key=lambda r: abs(r['t'] - t))


def leg(name, eng, kid, busy, idle, t0, smp, sweeps_per_point=2):
    print('--- %s: %g s busy, %g s idle, %g s busy' % (name, busy, idle, busy / 2), flush=True)
    for phase, dur in (('busy', busy), ('idle', idle), ('busy again', busy / 2)):
        if phase == 'idle':
            time.sleep(dur)
            r = smp.near(time.time() - t0)
            print('   idle %4.1f s -> %s' % (dur, fmt(r)), flush=True)
            continue
        t_end = time.time() + dur
        while time.time() < t_end:
            c0, m0 = eng.timing_read(kid)
            eng.sweep(sweeps_per_point)
            eng.synchronize()
            c1, m1 = eng.timing_read(kid)
            now = time.time() - t0
            print('   t %6.2f s  pass %.4f ms (%d samples)  %s' % (now, (m1 - m0) / max(c1 - c0, 1), c1 - c0, fmt(smp.near(now))), flush=True)


def fmt(r):
    return ' '.join('%s %s' % (k, ('%.0f' % v) if isinstance(v, float) else v) for k, v in r.items() if k != 't')


def main():
    busy = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
    idle = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    n, d, k = 100000, 10000, 50
    base, hw = sysfs_base()
    if hw:
        for f in ('power1_cap', 'power1_cap_max', 'power1_cap_default'):
            print('%s %.0f W' % (f, read_num(os.path.join(hw, f), 1e6)))
        for t in sorted(glob.glob(os.path.join(hw, 'temp*_crit'))) + sorted(glob.glob(os.path.join(hw, 'temp*_emergency'))):
            print('%s %.0f C' % (os.path.basename(t), read_num(t, 1e3)))
    dev = torch.device('cuda', 0)
    X = device_planted_shard(n, d, k, 0, dev)
    rng = np.random.RandomState(0)
    a = (float(X[:20000].mean()) / k) ** 0.5
    W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
    torch.cuda.synchronize()
    t0 = time.time()
    smp = Sampler(t0)
    smp.start()
    time.sleep(3.0)          # the card at rest first
    print('at rest: %s' % fmt(smp.near(time.time() - t0)), flush=True)
    for name, schedule, kid in (('explicit-residual schedule, k_pass<UPD=2> (8 GB per launch)', 'residual', 3),
                                ('default schedule, read-only k_pass (4 GB per launch)', 'gram', 0)):
        eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule=schedule)
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        eng.set_W(W0), eng.set_T(T0), eng.set_params()
        eng.timing_enable(True, every=1)
        leg(name, eng, kid, busy, idle, t0, smp)
        eng.close()
        time.sleep(idle)
    smp.stop_flag = True


if __name__ == '__main__':
    main()
