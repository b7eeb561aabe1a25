#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t6.log; tail -3 gpurun_out/r02_t6.log
show() { python - "$1" "$2" <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = j['roofline']
print('%-28s %.2f sweeps/s  kernel %.4f ms %.0f GB/s (%.3f)  %s %s' % (sys.argv[1], j['value'], r['avg_ms'], r['achieved'], r['frac'], {k: round(1e3*v, 2) for k, v in j['sweep_level']['kernel_avg_ms'].items()}, ('rank1 %.4f ms' % j['rank1_update']['avg_ms']) if 'rank1_update' in j and 'avg_ms' in j['rank1_update'] else ''))
PY
}
for rep in 1 2; do timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 300 --warmup 20 > /tmp/b.json 2>/tmp/b.err && show "c2" /tmp/b.json; done
RRI_PASS_NT=1 timeout -k 10 120 python bench.py --config c2 --no-cpu-baseline --steps 300 --warmup 20 > /tmp/b.json 2>/tmp/b.err && show "c2 NT=1" /tmp/b.json
timeout -k 10 120 python bench.py --config mid --no-cpu-baseline --steps 100 --warmup 10 > /tmp/b.json 2>/tmp/b.err && show "mid" /tmp/b.json
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/r02_c3_nocpu.json 2>/tmp/b.err && show "c3" gpurun_out/r02_c3_nocpu.json
timeout -k 10 200 python bench.py --schedule residual --no-cpu-baseline > gpurun_out/r02_c3_residual2.json 2>/tmp/b.err && show "c3 residual" gpurun_out/r02_c3_residual2.json
timeout -k 10 300 python bench.py --config c5 > gpurun_out/r02_bench_c5b.json 2>/tmp/b.err && show "c5" gpurun_out/r02_bench_c5b.json
bash tools/pmc_sq.sh c5 gpurun_out/r02_pmc_sq_c5_after.txt > /dev/null 2>&1; echo "pmc c5 rc=$?"
grep -A10 "k_wpass<float, true, false" gpurun_out/r02_pmc_sq_c5_after.txt | grep -E "VMEM_RD|WAIT_ANY|WAVE_CYCLES"
