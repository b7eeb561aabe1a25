#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_parity.py tests/test_residual_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py tests/test_edge_cases_gpu.py tests/test_nmf_gpu.py -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t10.log; tail -4 gpurun_out/r02_t10.log
show() { python - "$1" "$2" <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print('%-30s %.1f sweeps/s   %s' % (sys.argv[1], j['value'], {k: round(1e3*v, 2) for k, v in j['sweep_level']['kernel_avg_ms'].items()}))
PY
}
for rep in 1 2; do for deep in 8 16 32; do
  RRI_PASS_DEEP=$deep timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 300 --warmup 20 > /tmp/b.json 2>/tmp/b.err && show "c2 rows in flight=$deep" /tmp/b.json || tail -3 /tmp/b.err
done; done
RRI_PASS_DEEP=16 RRI_PASS_MIN_ROWS=16 timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 300 --warmup 20 > /tmp/b.json 2>/tmp/b.err && show "c2 deep=16 rows16" /tmp/b.json
RRI_PASS_DEEP=32 RRI_PASS_MIN_ROWS=64 timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 300 --warmup 20 > /tmp/b.json 2>/tmp/b.err && show "c2 deep=32 rows64" /tmp/b.json
