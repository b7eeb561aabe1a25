#!/bin/bash
# workgroup size / table size / work items of the sparse-pattern passes (c5s)
for cfg in "1024 120 768" "512 60 768" "512 60 1536" "512 60 3072" "1024 60 1536" "512 120 768"; do
  set -- $cfg
  RRI_SP_THREADS=$1 RRI_SP_BLOCK_KB=$2 RRI_SP_ITEMS=$3 timeout -k 10 200 python3 bench.py --config c5s --steps 4 --warmup 1 --no-cpu-baseline > /tmp/sp.json 2>/tmp/sp.err || { tail -3 /tmp/sp.err; exit 1; }
  python3 -c "
import json; j=json.loads(open('/tmp/sp.json').read().strip().splitlines()[-1]); print('threads=$1 block_kb=$2 items=$3  sweeps/s %.2f  pass avg %.1f us' % (j['value'], 1e3*j['roofline']['avg_ms']))"
done
