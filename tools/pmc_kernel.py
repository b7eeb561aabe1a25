"""Counters and durations of ONE kernel from a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run:
    python3 tools/pmc_kernel.py <dir> <kernel-name-substring> [<substring> ...]
prints, per matching kernel name, the launches, the average duration from the kernel trace (under counter collection: longer
than unprofiled, but comparable between processes that collect the same counters) and the average of every counter."""
import collections
import csv
import glob
import os
import sys

d, subs = sys.argv[1], sys.argv[2:]
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row['Kernel_Name']
        if any(s in kn for s in subs):
            a = dur[kn[:110]]
            a[0] += 1
            a[1] += (float(row['End_Timestamp']) - float(row['Start_Timestamp'])) * 1e-3
ctr = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        kn = row['Kernel_Name']
        if any(s in kn for s in subs):
            a = ctr[kn[:110]][row['Counter_Name']]
            a[0] += 1
            a[1] += float(row['Counter_Value'])
for kn in sorted(set(dur) | set(ctr)):
    n, tot = dur.get(kn, [0, 0.0])
    print('%s\n    launches %d  avg duration %.1f us' % (kn, n, tot / max(n, 1)))
    for c, (m, s) in sorted(ctr.get(kn, {}).items()):
        print('    %-44s avg %16.1f  (%d samples)' % (c, s / max(m, 1), m))
