#!/bin/bash
mkdir -p gpurun_out
show() { python - "$1" "$2" <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j['roofline']
print('%-34s %.2f sweeps/s  kernel %.4f ms %.0f GB/s (%.3f)' % (sys.argv[1], j['value'], r['avg_ms'], r['achieved'], r['frac']))
PY
}
for rep in 1 2 3; do
for uc in 4 16 8; do
  RRI_WPASS_UC=$uc timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --steps 12 > /tmp/b.json 2>/tmp/b.err && show "c5 pass C rows in flight=$uc" /tmp/b.json || tail -3 /tmp/b.err
done
done
RRI_WPASS_UC=16 timeout -k 10 300 python -m pytest tests/test_hip_parity.py tests/test_edge_cases_gpu.py -m gpu -q --no-header -p no:cacheprovider -k "weighted or ragged" 2>&1 | tail -2
