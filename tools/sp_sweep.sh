#!/bin/bash
# geometry sweep of the sparse-pattern WRRI passes (c5s): lanes per segment and work items
for cfg in "64 64 768" "32 32 768" "16 16 768" "8 8 768" "16 32 768" "32 16 768" "16 16 1536"; do
  set -- $cfg
  RRI_SP_LANES_ROW=$1 RRI_SP_LANES_COL=$2 RRI_SP_ITEMS=$3 timeout -k 10 200 python3 bench.py --config c5s --steps 4 --warmup 1 --no-cpu-baseline > /tmp/sp.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.loads(open('/tmp/sp.json').read().strip().splitlines()[-1]); print('lanes_row=$1 lanes_col=$2 items=$3  sweeps/s %.2f  pass avg %.1f us' % (j['value'], 1e3*j['roofline']['avg_ms']))"
done
