#!/bin/bash
# rows per workgroup of the pass at launch-bound sizes (C2 = 10000 x 1000 k=20, mid = 20000 x 5000 k=20)
for rep in 1 2; do
for cfg in "c2 0 32" "c2 1024 16" "c2 4096 16" "c2 8192 16" "c2 2048 48" "mid 0 32" "mid 2048 16" "mid 4096 16"; do
  set -- $cfg
  if [ "$2" = "0" ]; then unset RRI_PASS_WGS; else export RRI_PASS_WGS=$2; fi
  RRI_PASS_MIN_ROWS=$3 timeout -k 10 200 python3 bench.py --config $1 --steps 100 --warmup 10 --no-cpu-baseline > /tmp/r1.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.loads(open('/tmp/r1.json').read().strip().splitlines()[-1]); print('rep $rep $1 wgs=$2 minrows=$3  pass %.1f us  sweeps/s %.1f' % (1e3*j['roofline']['avg_ms'], j['value']))"
done; done
