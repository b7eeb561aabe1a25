#!/bin/bash
# round-2 profile artefacts (one box) + the cost of the row-sharded path on one rank
mkdir -p gpurun_out profiles
: > gpurun_out/profile_progress.log
bash tools/profile_cfg.sh c3
bash tools/profile_cfg.sh c3_residual --schedule residual
bash tools/profile_cfg.sh c2 --config c2
bash tools/profile_cfg.sh c5 --config c5
bash tools/profile_cfg.sh c5s --config c5s
timeout -k 10 300 python bench.py --config c4 > gpurun_out/profiles_r02/r02_bench_c4_one_gpu.json 2>/tmp/b.err; echo "c4 rc=$?"
ls -la gpurun_out/profiles_r02 | head -30
