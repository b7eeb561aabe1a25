#!/bin/bash
# A/B of two builds of the library on one box: kernel averages of the residual-schedule bench under rocprofv3
cd /tmp && export TMPDIR=/tmp
for lib in old new old new; do
  rm -rf /tmp/ab_$lib
  if [ $lib = old ]; then export RRI_HIP_LIB=/root/repo/rri_nmf_amd/lib/librri_hip_old.so; else unset RRI_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$lib -o st -- python3 /root/repo/bench.py --schedule residual --steps 6 --warmup 2 --no-cpu-baseline > /tmp/ab_$lib.log 2>&1
  f=$(find /tmp/ab_$lib -name '*kernel_stats.csv' | head -1); python3 - "$f" $lib <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:4]:
    if 'k_resid' in r[0] or 'k_pass<float, true, true, 2' in r[0]:
        print('%s  %-52s calls %5s avg %10.1f us' % (sys.argv[2], r[0][:52], r[1], float(r[3]) / 1e3))
PY
done
