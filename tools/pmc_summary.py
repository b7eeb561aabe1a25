"""Per-kernel average of one PMC counter from rocprofv3 --pmc <C> --output-format csv runs, merged over counters:
python tools/pmc_summary.py out.json dir_FETCH_SIZE dir_WRITE_SIZE ...   (kernels of namespace rri only)"""
import collections
import csv
import glob
import json
import os
import sys

out = collections.defaultdict(dict)
for d in sys.argv[2:]:
    f = glob.glob(os.path.join(d, '*counter_collection.csv'))[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    name = None
    for row in csv.DictReader(open(f)):
        kn = row['Kernel_Name']
        if 'rri::' not in kn:
            continue
        name = row['Counter_Name']
        a = acc[kn[:72]]
        a[0] += 1
        a[1] += float(row['Counter_Value'])
    for kn, (cnt, tot) in acc.items():
        out[kn]['launches'] = cnt
        out[kn][name + '_KB_avg'] = tot / cnt
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402  (the stamp of the kernel sources this profile belongs to)
out['_source_stamp'] = bench.source_stamp()
json.dump(out, open(sys.argv[1], 'w'), indent=1)
del out['_source_stamp']
for kn, v in sorted(out.items(), key=lambda kv: -kv[1].get('FETCH_SIZE_KB_avg', 0))[:8]:
    print('%-74s %s' % (kn, {k: round(x, 1) if isinstance(x, float) else x for k, x in v.items()}))
