#!/bin/bash
# one GPU call: the failing group tests again, then bench lines (c3 default, c2, c5, c5s, c3 residual schedule) and
# the rocprofv3 kernel stats of c3 / c2.  Outputs under gpurun_out/r02_*.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_group_gpu.py -m gpu -q --no-header -rf -p no:cacheprovider > gpurun_out/r02_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_t2.log
timeout -k 10 400 python bench.py > gpurun_out/r02_bench_c3.json 2> gpurun_out/r02_bench_c3.err; echo "c3 rc=$?"
timeout -k 10 200 python bench.py --config c2 > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench_c2.err; echo "c2 rc=$?"
timeout -k 10 200 python bench.py --schedule residual --no-cpu-baseline > gpurun_out/r02_bench_c3_residual.json 2> gpurun_out/r02_bench_c3_residual.err; echo "c3r rc=$?"
timeout -k 10 300 python bench.py --config c5 > gpurun_out/r02_bench_c5.json 2> gpurun_out/r02_bench_c5.err; echo "c5 rc=$?"
timeout -k 10 300 python bench.py --config c5s > gpurun_out/r02_bench_c5s.json 2> gpurun_out/r02_bench_c5s.err; echo "c5s rc=$?"
tail -c 600 gpurun_out/r02_bench_c3.json
