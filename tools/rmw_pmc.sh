#!/bin/bash
# The two modes of the read-modify-write pass (k_pass<UPD=2>: 1.33 or 1.51 ms per 8 GB at C3, per process), with counters:
# several processes started back to back, each ONE --pmc pass of the explicit-residual schedule; per process the kernel's
# average duration (same counters => comparable) and the memory-side counters.  bash tools/rmw_pmc.sh <outfile> <reps>
out=${1:-gpurun_out/rmw_pmc.txt}; reps=${2:-3}
repo=$PWD; mkdir -p $(dirname $out); : > $out
cd /tmp && export TMPDIR=/tmp
sets=("TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
      "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_REQUEST_sum"
      "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum"
      "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_NORMAL_WRITEBACK_sum")
for r in $(seq 1 $reps); do
  for i in 0 1 2 3; do
    rm -rf /tmp/rmwp
    timeout -k 10 200 rocprofv3 --pmc ${sets[$i]} --kernel-trace --output-format csv -d /tmp/rmwp -- python3 $repo/bench.py --schedule residual --steps 3 --warmup 1 --no-cpu-baseline > /tmp/rmwp.log 2>&1
    rc=$?
    echo "== process $r.$i rc=$rc: ${sets[$i]}" >> $repo/$out
    python3 $repo/tools/pmc_kernel.py /tmp/rmwp "2, 16, " >> $repo/$out 2>&1
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping" >> $repo/$out; exit 1; fi
  done
done
# the same schedule unprofiled, a few processes: which mode does this box give
for r in 1 2 3; do
  timeout -k 10 200 python3 $repo/bench.py --schedule residual --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('unprofiled process: value %.2f sweeps/s, roofline %r' % (j['value'], {k:j['roofline'].get(k) for k in ('kernel','avg_ms','achieved','frac')}))
" >> $repo/$out
done
