#!/bin/bash
# k_sp_blk's loads / stores in the REAL schedule (row copy and column copy alternate: 1 GB per topic step, no reuse):
# library variants built with -DRRI_SP_MEM=m (bit 0 non-temporal value loads, bit 1 offset loads, bit 2 stores), same box,
# two rounds:   bash tools/sp_mem_ab.sh -> gpurun_out/sp_mem_ab.log
out=gpurun_out/sp_mem_ab.log; : > $out
for rep in 1 2; do
  for m in plain 2 4 7; do
    if [ $m = plain ]; then unset RRI_HIP_LIB; else export RRI_HIP_LIB=$PWD/rri_nmf_amd/lib/librri_hip_spmem$m.so; fi
    timeout -k 10 200 python3 bench.py --config c5s --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); r=j['roofline']; print('RRI_SP_MEM=$m: %.2f sweeps/s, pass %.1f us (%.3f of 8 TB/s)' % (j['value'], 1e3*r['avg_ms'], r['frac']))
" >> $out || exit 1
  done
done
cat $out
