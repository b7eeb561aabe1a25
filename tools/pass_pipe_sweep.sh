#!/bin/bash
# workgroup counts of the pass at C3, two repetitions (used for the double-buffered variant recorded in DESIGN.md
# section 4; works on any build); compare inside one call only
for rep in 1 2; do
for wgs in 2048 1536 768 2304 3072 1024; do
  RRI_PASS_WGS=$wgs timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/r1.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.loads(open('/tmp/r1.json').read().strip().splitlines()[-1]); print('rep $rep wgs=$wgs  pass %.1f us  %.2f TB/s  sweeps/s %.2f' % (1e3*j['roofline']['avg_ms'], j['roofline']['achieved']/1e3, j['value']))"
done; done
