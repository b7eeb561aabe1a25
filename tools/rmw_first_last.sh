#!/bin/bash
# bash tools/rmw_first_last.sh : the in-process contrast of the read-modify-write pass (first handle / last handle), plain, with
# contiguous allocation, and under two counter passes (translation, memory-side queues)
out=gpurun_out/rmw_first_last.txt; repo=$PWD; : > $out
cd /tmp && export TMPDIR=/tmp
echo "== plain" >> $repo/$out
timeout -k 10 200 python3 $repo/tools/rmw_first_last.py >> $repo/$out 2>&1 || exit 1
echo "== RRI_MALLOC_CONTIGUOUS=1" >> $repo/$out
RRI_MALLOC_CONTIGUOUS=1 RRI_ONCHIP_DEBUG=1 timeout -k 10 200 python3 $repo/tools/rmw_first_last.py >> $repo/$out 2>&1 || exit 1
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1)); rm -rf /tmp/rfl$i
  echo "== rocprofv3 --pmc $set" >> $repo/$out
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/rfl$i -- python3 $repo/tools/rmw_first_last.py >> $repo/$out 2>&1 || exit 1
  python3 $repo/tools/pmc_timeline.py /tmp/rfl$i "2, 16, " 50 >> $repo/$out 2>&1
done
