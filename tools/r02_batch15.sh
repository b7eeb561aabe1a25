#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_parity.py tests/test_residual_gpu.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py tests/test_nmf_gpu.py -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t14.log; tail -3 gpurun_out/r02_t14.log
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_r
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_r -o st -- python3 /root/repo/bench.py --schedule residual --steps 6 --warmup 2 --no-cpu-baseline > /tmp/st_r.log 2>&1
f=$(find /tmp/st_r -name '*kernel_stats.csv' | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:5]:
    print('%-60s calls %5s avg %10.1f us' % (r[0][:60], r[1], float(r[3]) / 1e3))
PY
rm -rf /tmp/st_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_w -o st -- python3 /root/repo/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/st_w.log 2>&1
f=$(find /tmp/st_w -name '*kernel_stats.csv' | head -1); python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:5]:
    print('%-60s calls %5s avg %10.1f us' % (r[0][:60], r[1], float(r[3]) / 1e3))
PY
