#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_sparse_wrri_gpu.py tests/test_sharded_gpu.py tests/test_group_gpu.py tests/test_full_size_gpu.py tests/test_nmf_gpu.py tests/test_hip_parity.py -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t8.log; tail -3 gpurun_out/r02_t8.log
show() { python - "$1" "$2" <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j['roofline']
print('%-34s %.2f sweeps/s  kernel %.4f ms %.0f GB/s (%.3f)  %s  %s' % (sys.argv[1], j['value'], r['avg_ms'], r['achieved'], r['frac'], {k: round(1e3*v, 2) for k, v in j['sweep_level']['kernel_avg_ms'].items()}, {k: ('%.2e' % v) for k, v in j.get('parity_sample', {}).items() if isinstance(v, float)}))
PY
}
timeout -k 10 300 python bench.py --config c5s > gpurun_out/r02_bench_c5s_merged.json 2>/tmp/b.err && show "c5s merged" gpurun_out/r02_bench_c5s_merged.json || tail -5 /tmp/b.err
RRI_SP_MERGE=0 timeout -k 10 300 python bench.py --config c5s > /tmp/b.json 2>/tmp/b.err && show "c5s unmerged" /tmp/b.json
timeout -k 10 300 python bench.py --config c5 > gpurun_out/r02_bench_c5c.json 2>/tmp/b.err && show "c5 (8192 wgs default)" gpurun_out/r02_bench_c5c.json
RRI_PASS_WGS=16384 RRI_PASS_MIN_ROWS=16 timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --steps 8 > /tmp/b.json 2>/tmp/b.err && show "c5 wgs=16384" /tmp/b.json
timeout -k 10 200 python bench.py --schedule residual --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err && show "c3 residual" /tmp/b.json
