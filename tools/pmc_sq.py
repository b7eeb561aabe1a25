"""Per-kernel averages of the SQ counters of one `rocprofv3 --pmc A B C ... --output-format csv` run (several counters
in one pass):  python tools/pmc_sq.py <dir> [<dir> ...]   -> table on stdout (kernels of namespace rri only)"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row['Kernel_Name']
            if 'rri::' not in kn:
                continue
            a = acc[kn[:96]][row['Counter_Name']]
            a[0] += 1
            a[1] += float(row['Counter_Value'])
names = sorted({c for v in acc.values() for c in v})
top = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', [0, 0.0])[1])[:10]
for kn, v in top:
    print(kn)
    wc = v.get('SQ_WAVE_CYCLES', [1, 1.0])
    wavg = wc[1] / max(wc[0], 1)
    for c in names:
        if c in v:
            avg = v[c][1] / v[c][0]
            print('    %-26s launches %5d  avg %14.1f  (%.3f of WAVE_CYCLES)' % (c, v[c][0], avg, avg / wavg))
