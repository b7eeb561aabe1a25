#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py tests/test_residual_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py tests/test_edge_cases_gpu.py -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t4.log; tail -3 gpurun_out/r02_t4.log
for f in 1 0; do
  for rep in 1 2; do
  RRI_FUSE_W=$f timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/r02_c2_fuse${f}_$rep.json 2> gpurun_out/r02_c2_fuse$f.err; echo "c2 fuse=$f rc=$?"
  done
done
python - <<'PY'
import json
for f in (1,0):
  for rep in (1,2):
    j=json.loads(open('gpurun_out/r02_c2_fuse%d_%d.json'%(f,rep)).read().strip().splitlines()[-1])
    print('fuse',f,'%.1f sweeps/s'%j['value'], j['sweep_level']['kernel_avg_ms'])
PY
: > gpurun_out/r02_resid_knobs2.log
for cfg in "0 8 1 -1" "1024 8 1 1" "4096 8 1 1" "8192 8 1 1" "16384 8 1 1" "2048 16 0 1" "8192 16 0 1" "2048 8 1 0"; do
  set -- $cfg
  env=""
  [ "$1" != "0" ] && env="$env RRI_PASS_WGS=$1"
  [ "$4" != "-1" ] && env="$env RRI_PASS_IL=$4"
  env $env RRI_PASS_UNROLL=$2 RRI_PASS_RS=$3 RRI_PASS_MIN_ROWS=16 timeout -k 10 120 python bench.py --schedule residual --no-cpu-baseline --steps 4 --warmup 1 > /tmp/rk.json 2>/tmp/rk.err || { echo "knob run failed: $cfg" >> gpurun_out/r02_resid_knobs2.log; tail -3 /tmp/rk.err >> gpurun_out/r02_resid_knobs2.log; continue; }
  python - "$cfg" <<'PY' >> gpurun_out/r02_resid_knobs2.log
import json, sys
j = json.loads(open('/tmp/rk.json').read().strip().splitlines()[-1])
r = j['roofline']; u = j['rank1_update']
print('wgs,unroll,rs,il = %-16s residual sweep %.2f sweeps/s   UPD2 pass %.4f ms %.0f GB/s (%.3f)   UPD1 %.4f ms %.0f GB/s' % (sys.argv[1], j['value'], r['avg_ms'], r['achieved'], r['frac'], u['avg_ms'], u['achieved']))
PY
done
cat gpurun_out/r02_resid_knobs2.log
