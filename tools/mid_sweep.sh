#!/bin/bash
# workgroups of the pass at mid sizes (repeats interleaved)
for rep in 1 2; do
for wgs in 256 384 512 768 1024; do
  RRI_PASS_WGS=$wgs timeout -k 10 200 python3 bench.py --config mid --steps 100 --warmup 10 --no-cpu-baseline > /tmp/m.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.loads(open('/tmp/m.json').read().strip().splitlines()[-1]); print('rep $rep mid wgs=$wgs  pass %.1f us  sweeps/s %.1f' % (1e3*j['roofline']['avg_ms'], j['value']))"
done; done
