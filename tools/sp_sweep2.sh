#!/bin/bash
# sparse-pattern passes (c5s): lanes per segment and work items, then a kernel trace of the default geometry
# (used for the software-pipelined variant recorded in DESIGN.md section 4; works on any build)
for cfg in "32 32 768" "16 16 768" "64 64 768" "32 32 512" "32 32 1024" "32 32 1536" "32 32 2048" "16 16 1536" "64 32 768"; do
  set -- $cfg
  RRI_SP_LANES_ROW=$1 RRI_SP_LANES_COL=$2 RRI_SP_ITEMS=$3 timeout -k 10 200 python3 bench.py --config c5s --steps 4 --warmup 1 --no-cpu-baseline > /tmp/sp.json 2>/dev/null || exit 1
  python3 -c "
import json; j=json.loads(open('/tmp/sp.json').read().strip().splitlines()[-1]); print('lanes_row=$1 lanes_col=$2 items=$3  sweeps/s %.2f  pass avg %.1f us' % (j['value'], 1e3*j['roofline']['avg_ms']))"
done
repo=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/spk
rocprofv3 --kernel-trace -d /tmp/spk -o sp -- python3 $repo/bench.py --config c5s --steps 6 --warmup 2 --no-cpu-baseline > /tmp/spk.log 2>&1
cd $repo
python3 tools/kstats.py $(find /tmp/spk -name '*.db' | head -1) 0.3
