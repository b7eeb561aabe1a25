#!/bin/bash
# geometry sweep of the explicit rank-one update kernel (k_pass<UPD>): workgroups, rows in flight, min rows per block
for cfg in "2048 8 1 32" "4096 8 1 32" "8192 8 1 32" "16384 8 1 16" "8192 4 0 16" "8192 4 1 16" "16384 4 1 16" "32768 4 1 8" "8192 16 1 16"; do
  set -- $cfg
  RRI_PASS_WGS=$1 RRI_PASS_UNROLL=$2 RRI_PASS_RS=$3 RRI_PASS_MIN_ROWS=$4 timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/r1.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.loads(open('/tmp/r1.json').read().strip().splitlines()[-1]); r=j['rank1_update']; print('wgs=$1 unroll=$2 rs=$3 minrows=$4  update %.3f ms  %.0f GB/s  (copy yardstick %.0f GB/s)   pass %.1f us  sweeps/s %.2f' % (r['avg_ms'], r['achieved'], r['stream_copy_GBps'], 1e3*j['roofline']['avg_ms'], j['value']))"
done
