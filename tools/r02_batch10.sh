#!/bin/bash
# k_pass (read-only, C3) geometry revisited with this round's knobs: interleaved chunks, workgroup count, loads
mkdir -p gpurun_out; : > gpurun_out/r02_pass_knobs_c3.log
for cfg in "0 8 1 -1 -1" "0 8 1 1 -1" "4096 8 1 1 -1" "8192 8 1 1 -1" "8192 8 1 0 -1" "4096 8 1 0 -1" "0 8 0 -1 -1" "8192 8 0 1 -1" "0 8 1 -1 0" "0 8 1 -1 -1"; do
  set -- $cfg
  env=""
  [ "$1" != "0" ] && env="$env RRI_PASS_WGS=$1 RRI_PASS_MIN_ROWS=16"
  [ "$4" != "-1" ] && env="$env RRI_PASS_IL=$4"
  [ "$5" != "-1" ] && env="$env RRI_PASS_NT=$5"
  env $env RRI_PASS_UNROLL=$2 RRI_PASS_RS=$3 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 --warmup 2 > /tmp/rk.json 2>/tmp/rk.err || { echo "failed: $cfg" >> gpurun_out/r02_pass_knobs_c3.log; continue; }
  python - "$cfg" <<'PY' >> gpurun_out/r02_pass_knobs_c3.log
import json, sys
j = json.loads(open('/tmp/rk.json').read().strip().splitlines()[-1])
r = j['roofline']
print('wgs,unroll,rs,il,nt = %-20s %.2f sweeps/s   k_pass %.4f ms %.0f GB/s (%.3f)' % (sys.argv[1], j['value'], r['avg_ms'], r['achieved'], r['frac']))
PY
done
cat gpurun_out/r02_pass_knobs_c3.log
