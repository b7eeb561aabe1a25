"""Per-kernel summary of a rocprofv3 rocpd database (kernel-trace): python tools/kstats.py results.db [skip_fraction]"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = list(c.execute('select name, start, end from kernels order by start'))
rows = rows[int(len(rows) * skip):]
d = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    d[re.sub(r'\(.*', '', n)[:100]][0] += 1
    d[re.sub(r'\(.*', '', n)[:100]][1] += e - s
tot = rows[-1][2] - rows[0][1]
busy = sum(v[1] for v in d.values())
print('kernels %d  wall %.3f ms  busy %.3f ms (%.1f%%)' % (len(rows), tot / 1e6, busy / 1e6, 100.0 * busy / tot))
print('%-100s %8s %10s %7s' % ('kernel', 'calls', 'avg us', '% wall'))
for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%-100s %8d %10.2f %6.1f%%' % (k, v[0], v[1] / v[0] / 1e3, 100 * v[1] / tot))
