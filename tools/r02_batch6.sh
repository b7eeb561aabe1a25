#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_hip_parity.py tests/test_sharded_gpu.py tests/test_edge_cases_gpu.py tests/test_fuzz_gpu.py tests/test_nmf_gpu.py tests/test_full_size_gpu.py -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t7.log; tail -3 gpurun_out/r02_t7.log
show() { python - "$1" "$2" <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j['roofline']
print('%-34s %.2f sweeps/s  kernel %.4f ms %.0f GB/s (%.3f)  %s' % (sys.argv[1], j['value'], r['avg_ms'], r['achieved'], r['frac'], {k: round(1e3*v, 2) for k, v in j['sweep_level']['kernel_avg_ms'].items()}))
PY
}
for il in -1 0 1; do
  env=""; [ "$il" != "-1" ] && env="RRI_WPASS_IL=$il"
  env $env timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --steps 8 > /tmp/b.json 2>/tmp/b.err && show "c5 wpass_il=$il" /tmp/b.json
done
RRI_PASS_WGS=4096 timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --steps 8 > /tmp/b.json 2>/tmp/b.err && show "c5 wgs=4096" /tmp/b.json
RRI_PASS_WGS=8192 RRI_PASS_MIN_ROWS=16 timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --steps 8 > /tmp/b.json 2>/tmp/b.err && show "c5 wgs=8192" /tmp/b.json
for rep in 1 2; do timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 300 --warmup 20 > /tmp/b.json 2>/tmp/b.err && show "c2" /tmp/b.json; done
timeout -k 10 200 python bench.py --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err && show "c3" /tmp/b.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st5 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st5 -o st -- python3 /root/repo/bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/st5.log 2>&1
f=$(find /tmp/st5 -name '*kernel_stats.csv' | head -1) && head -8 "$f" | cut -c1-220
