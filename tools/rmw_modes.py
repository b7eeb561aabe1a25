#!/usr/bin/env python3
"""Is the two-mode behaviour of the read-modify-write pass (k_pass<UPD=2>, 1.33 ms or 1.52 ms at C3 from process to process,
profiles/r02_residual_schedule_run_to_run.log) a property of WHERE the residual lands?  One process, one resident X, the
handle of the explicit-residual schedule created and destroyed several times: every creation allocates the 4 GB residual
afresh.  Between some trials an allocation of another size is made (and kept) so that the next residual cannot land where
the last one did.  Prints the average duration of the pass per trial, and of the stand-alone rank-one update kernel
(rri_bench_rank1_update, which allocates its own scratch copy per call)."""
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


class Sampler(threading.Thread):
    """the card's clock levels, power and temperatures from sysfs (readable by an ordinary user) while a trial runs"""

    def __init__(self):
        super().__init__(daemon=True)
        self.stop_flag = False
        self.rows = []
        base = [b for b in sorted(glob.glob('/sys/class/drm/card*/device')) if os.path.exists(os.path.join(b, 'pp_dpm_sclk'))]
        self.base = base[0] if base else None
        self.hw = (sorted(glob.glob(os.path.join(self.base, 'hwmon', 'hwmon*'))) or [None])[0] if self.base else None

    @staticmethod
    def active_level(path):
        try:
            for ln in open(path).read().splitlines():
                if ln.rstrip().endswith('*'):
                    return ln.split(':')[1].replace('*', '').strip()
        except OSError:
            pass
        return '?'

    @staticmethod
    def number(path, scale):
        try:
            return float(open(path).read().strip()) / scale
        except (OSError, ValueError):
            return float('nan')

    def run(self):
        while not self.stop_flag and self.base:
            row = {f: self.active_level(os.path.join(self.base, 'pp_dpm_' + f)) for f in ('sclk', 'mclk', 'fclk', 'socclk')}
            if self.hw:
                row['power_W'] = self.number(os.path.join(self.hw, 'power1_average'), 1e6)
                if row['power_W'] != row['power_W']:
                    row['power_W'] = self.number(os.path.join(self.hw, 'power1_input'), 1e6)
                for t in sorted(glob.glob(os.path.join(self.hw, 'temp*_input'))):
                    lab = t.replace('_input', '_label')
                    name = open(lab).read().strip() if os.path.exists(lab) else os.path.basename(t)
                    row['T_' + name] = self.number(t, 1e3)
            self.rows.append(row)
            time.sleep(0.05)

    def summary(self):
        if not self.rows:
            return 'no sysfs clock files on this box'
        out = []
        for key in self.rows[0]:
            vals = [r[key] for r in self.rows]
            if isinstance(vals[0], str):
                seen = sorted(set(vals))
                out.append('%s %s' % (key, '/'.join(seen)))
            else:
                out.append('%s %.0f..%.0f' % (key, min(vals), max(vals)))
        return ', '.join(out)


def main():
    n, d, k = 100000, 10000, 50
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device('cuda', 0)
    X = device_planted_shard(n, d, k, 0, dev)
    rng = np.random.RandomState(0)
    a = (float(X[:20000].mean()) / k) ** 0.5
    W0, T0 = a * rng.rand(n, k), a * rng.rand(k, d)
    torch.cuda.synchronize()
    keep = []
    for trial in range(trials):
        eng = RRIEngine(n, d, k, dtype=np.float32, device=0, schedule='residual')
        eng.bind_X_device(X.data_ptr(), X.stride(0))
        eng.set_W(W0)
        eng.set_T(T0)
        eng.set_params()
        eng.sweep(1)
        eng.synchronize()
        smp = Sampler()
        smp.start()
        eng.timing_enable(True, every=4)
        eng.sweep(8)
        eng.synchronize()
        smp.stop_flag = True
        smp.join()
        cnt, ms = eng.timing_read(3)
        eng.timing_enable(False)
        r1 = [eng.bench_rank1_update(5) for _ in range(2)]
        free, total = torch.cuda.mem_get_info()
        print('trial %d: pass %.4f ms (%d samples), rank-one update alone %s ms, free HBM %.1f GB'
              % (trial, ms / max(cnt, 1), cnt, ' / '.join('%.4f' % v for v in r1), free / 1e9), flush=True)
        print('   while the passes ran: ' + smp.summary(), flush=True)
        eng.close()
        if trial % 2 == 1:      # shift what the allocator hands out next
            keep.append(torch.empty(int((0.3 + 0.5 * rng.rand()) * 2 ** 30), dtype=torch.uint8, device=dev))
            print('   kept another %.2f GB' % (keep[-1].numel() / 1e9), flush=True)


if __name__ == '__main__':
    main()
