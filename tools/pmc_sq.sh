#!/bin/bash
# SQ stall / LDS counters of the dominant kernels: bash tools/pmc_sq.sh <config> <outfile>
set -e
cfg=${1:-c5s}
out=${2:-gpurun_out/pmc_sq_$cfg.txt}
repo=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcsq_$cfg
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/pmcsq_$cfg/a -- python3 $repo/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmcsq_$cfg.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES \
  --kernel-trace --output-format csv -d /tmp/pmcsq_$cfg/b -- python3 $repo/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline >> /tmp/pmcsq_$cfg.log 2>&1
cd $repo
python3 tools/pmc_sq.py /tmp/pmcsq_$cfg/a /tmp/pmcsq_$cfg/b > $out
tail -3 /tmp/pmcsq_$cfg.log >> $out
