#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_sparse_wrri_gpu.py tests/test_sharded_gpu.py tests/test_full_size_gpu.py tests/test_fuzz_gpu.py -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t11.log; tail -3 gpurun_out/r02_t11.log
show() { python - "$1" "$2" <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j['roofline']
print('%-30s %.2f sweeps/s  kernel %.4f ms (%.3f)  %s' % (sys.argv[1], j['value'], r['avg_ms'], r['frac'], {k: ('%.2e' % v) for k, v in j.get('parity_sample', {}).items() if isinstance(v, float)}))
PY
}
for rep in 1 2; do for dflag in 1 0; do
RRI_SP_DIRECT=$dflag timeout -k 10 300 python bench.py --config c5s --steps 20 > /tmp/b.json 2>/tmp/b.err && show "c5s direct=$dflag" /tmp/b.json || tail -5 /tmp/b.err
done; done
