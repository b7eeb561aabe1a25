#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_hip_parity.py -m gpu -q --no-header -rf -p no:cacheprovider -k "fused or rare or plain or resum" > gpurun_out/r02_t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t5.log; tail -3 gpurun_out/r02_t5.log
run() { # label, env...
  lab=$1; shift
  for rep in 1 2; do
    env "$@" timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 300 --warmup 20 > /tmp/c2.json 2>/tmp/c2.err || { echo "$lab failed"; tail -2 /tmp/c2.err; continue; }
    python - "$lab" <<'PY'
import json, sys
j = json.loads(open('/tmp/c2.json').read().strip().splitlines()[-1])
print('%-44s %.1f sweeps/s   %s' % (sys.argv[1], j['value'], {k: round(1e3*v, 2) for k, v in j['sweep_level']['kernel_avg_ms'].items()}))
PY
  done
}
run "fused (default)" RRI_FUSE_W=1
run "unfused" RRI_FUSE_W=0
run "unfused, Gram row of T as own launch" RRI_FUSE_W=0 RRI_SIDE_JOBS=0
run "unfused, NT=0" RRI_FUSE_W=0 RRI_PASS_NT=0
run "fused, NT=0" RRI_FUSE_W=1 RRI_PASS_NT=0
run "unfused, U16 rows16" RRI_FUSE_W=0 RRI_PASS_UNROLL=16 RRI_PASS_RS=0 RRI_PASS_MIN_ROWS=16
run "unfused, rows 64" RRI_FUSE_W=0 RRI_PASS_MIN_ROWS=64
run "fused, rows 64" RRI_FUSE_W=1 RRI_PASS_MIN_ROWS=64
run "fused, rows 48" RRI_FUSE_W=1 RRI_PASS_MIN_ROWS=48
run "unfused, trow_small off" RRI_FUSE_W=0 RRI_TROW_SMALL=0
