#!/bin/bash
# GPU call: full suite, c2 with and without the fused W update, geometry knobs of the residual schedule, SQ counters of
# the weighted passes (c5) next to the plain pass (c3)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t3.log; tail -3 gpurun_out/r02_t3.log
for f in 1 0; do
  RRI_FUSE_W=$f timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r02_c2_fuse$f.json 2> gpurun_out/r02_c2_fuse$f.err; echo "c2 fuse=$f rc=$?"
done
: > gpurun_out/r02_resid_knobs.log
for cfg in "0 8 1 -1" "4096 8 1 -1" "8192 8 1 -1" "16384 8 1 -1" "8192 8 1 1" "2048 8 1 1" "8192 16 0 -1" "8192 4 0 -1" "16384 4 0 -1"; do
  set -- $cfg
  env=""
  [ "$1" != "0" ] && env="$env RRI_PASS_WGS=$1"
  [ "$4" != "-1" ] && env="$env RRI_PASS_IL=$4"
  env $env RRI_PASS_UNROLL=$2 RRI_PASS_RS=$3 RRI_PASS_MIN_ROWS=16 timeout -k 10 120 python bench.py --schedule residual --no-cpu-baseline --steps 4 --warmup 1 > /tmp/rk.json 2>/tmp/rk.err || { echo "knob run failed: $cfg" >> gpurun_out/r02_resid_knobs.log; tail -3 /tmp/rk.err >> gpurun_out/r02_resid_knobs.log; continue; }
  python - "$cfg" <<'PY' >> gpurun_out/r02_resid_knobs.log
import json, sys
j = json.loads(open('/tmp/rk.json').read().strip().splitlines()[-1])
r = j['roofline']; u = j['rank1_update']
print('wgs,unroll,rs,il = %-16s residual sweep %.2f sweeps/s   UPD2 pass %.4f ms %.0f GB/s (%.3f)   UPD1 %.4f ms %.0f GB/s' % (sys.argv[1], j['value'], r['avg_ms'], r['achieved'], r['frac'], u['avg_ms'], u['achieved']))
PY
done
cat gpurun_out/r02_resid_knobs.log
bash tools/pmc_sq.sh c5 gpurun_out/r02_pmc_sq_c5.txt > /dev/null 2>&1; echo "pmc c5 rc=$?"
bash tools/pmc_sq.sh c3 gpurun_out/r02_pmc_sq_c3.txt > /dev/null 2>&1; echo "pmc c3 rc=$?"
