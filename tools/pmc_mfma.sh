#!/bin/bash
# matrix-core / stall counters of the kernels of one bench configuration (the residual rebuild k_resid_mfma is what this is for):
#   bash tools/pmc_mfma.sh <config> <outfile> [bench args]
set -e
cfg=${1:-c5}
out=${2:-gpurun_out/pmc_mfma_$cfg.txt}
shift 2 || true
repo=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcmf_$cfg
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d /tmp/pmcmf_$cfg/a -- python3 $repo/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline "$@" > /tmp/pmcmf_$cfg.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES \
  --kernel-trace --output-format csv -d /tmp/pmcmf_$cfg/b -- python3 $repo/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline "$@" >> /tmp/pmcmf_$cfg.log 2>&1
cd $repo
python3 tools/pmc_sq.py /tmp/pmcmf_$cfg/a /tmp/pmcmf_$cfg/b > $out
tail -3 /tmp/pmcmf_$cfg.log >> $out
