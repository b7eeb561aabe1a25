#!/bin/bash
# A/B of pass geometries at C3, interleaved and repeated (box-to-box variance is +-3 %: compare inside one call only)
for rep in 1 2 3; do
for cfg in "2048 8 1 32" "8192 4 0 16" "8192 8 1 16" "4096 4 0 32" "8192 4 0 32" "16384 4 0 16"; do
  set -- $cfg
  RRI_PASS_WGS=$1 RRI_PASS_UNROLL=$2 RRI_PASS_RS=$3 RRI_PASS_MIN_ROWS=$4 timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/r1.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.loads(open('/tmp/r1.json').read().strip().splitlines()[-1]); print('rep $rep wgs=$1 unroll=$2 rs=$3 minrows=$4  pass %.1f us  sweeps/s %.2f' % (1e3*j['roofline']['avg_ms'], j['value']))"
done; done
