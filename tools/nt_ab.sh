#!/bin/bash
# non-temporal against plain loads / stores in the read-modify-write passes (k_pass<UPD>, k_wpass C), same box, alternating
# processes:  bash tools/nt_ab.sh  -> gpurun_out/nt_ab.log
out=gpurun_out/nt_ab.log; : > $out
line() { python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); r=j['roofline']; print('$1: %.2f sweeps/s, dominant kernel %.4f ms (%.3f of 8 TB/s)' % (j['value'], r['avg_ms'], r['frac']))
"; }
for rep in 1 2; do
  for nt in default 0; do
    if [ $nt = default ]; then unset RRI_PASS_NT; else export RRI_PASS_NT=$nt; fi
    timeout -k 10 200 python3 bench.py --schedule residual --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | line "residual schedule, RRI_PASS_NT=$nt" >> $out || exit 1
    timeout -k 10 200 python3 bench.py --config c5 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | line "c5 dense weighted,  RRI_PASS_NT=$nt" >> $out || exit 1
  done
done
cat $out
