"""Counters of ONE kernel in dispatch order, averaged over consecutive groups of launches:
    python3 tools/pmc_timeline.py <dir> <kernel-name-substring> [group=50]
from a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run: one line per group with the average duration (kernel trace)
and the average of every counter -- for processes in which the same kernel changes its behaviour over time."""
import collections
import csv
import glob
import os
import sys

d, sub = sys.argv[1], sys.argv[2]
group = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dur = {}
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        if sub in row['Kernel_Name']:
            dur[int(row['Dispatch_Id'])] = (float(row['End_Timestamp']) - float(row['Start_Timestamp'])) * 1e-3
ctr = collections.defaultdict(dict)
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(f)):
        if sub in row['Kernel_Name']:
            ctr[int(row['Dispatch_Id'])][row['Counter_Name']] = float(row['Counter_Value'])
ids = sorted(set(dur) | set(ctr))
names = sorted({c for v in ctr.values() for c in v})
print('launches %d; columns: first dispatch id, avg duration us, %s' % (len(ids), ', '.join(names)))
for i in range(0, len(ids), group):
    chunk = ids[i:i + group]
    dd = [dur[j] for j in chunk if j in dur]
    line = '%8d  %8.1f us' % (chunk[0], sum(dd) / max(len(dd), 1))
    for c in names:
        vv = [ctr[j][c] for j in chunk if c in ctr.get(j, {})]
        line += '  %14.0f' % (sum(vv) / max(len(vv), 1))
    print(line)
