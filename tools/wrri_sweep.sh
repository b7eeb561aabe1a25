#!/bin/bash
for bits in 1 0; do for uw in 4 8; do for ur in 4 8; do
  echo "BITS=$bits UW=$uw UR=$ur: $(RRI_MASK_BITS=$bits RRI_WPASS_UW=$uw RRI_WPASS_UR=$ur timeout -k 5 200 python bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; o=json.loads(sys.stdin.read()); print("%.2f sweeps/s  %.1f ms  wpass %.3f ms" % (o["value"], o["ms_per_step"], o["roofline"]["avg_ms"]))')"
done; done; done
