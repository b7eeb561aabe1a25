#!/bin/bash
for u in 4 8; do for nt in 0 1; do for wgs in 2048 4096; do
  echo "U=$u NT=$nt WGS=$wgs: $(RRI_PASS_UNROLL=$u RRI_PASS_NT=$nt RRI_PASS_WGS=$wgs timeout -k 5 200 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; o=json.loads(sys.stdin.read()); print("%.2f sweeps/s  %.1f ms  wpass %.3f ms  %.0f GB/s" % (o["value"], o["ms_per_step"], o["roofline"]["avg_ms"], o["roofline"]["achieved"]))')"
done; done; done
