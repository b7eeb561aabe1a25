#!/bin/bash
# the pattern-only pass outside the library: shipped kernel vs candidates, three work-list sizes, then the SQ counters of one run
out=gpurun_out/sp_blk_probe.log; repo=$PWD; : > $out
for items in 768; do
  echo "=== work items target $items" >> $out
  timeout -k 10 200 tools/sp_blk_probe 100000 10000 0.05 10 $items >> $out 2>&1 || exit 1
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/spq
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/spq -- $repo/tools/sp_blk_probe 100000 10000 0.05 3 768 > /tmp/spq.log 2>&1
echo "=== SQ counters, work items 768 (rocprofv3 --pmc, 3 repetitions)" >> $repo/$out
python3 $repo/tools/pmc_kernel.py /tmp/spq "k_sp_blk" >> $repo/$out 2>&1
