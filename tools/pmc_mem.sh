#!/bin/bash
# memory-path counters of the dominant kernels: bash tools/pmc_mem.sh <config> <outfile>   (progress -> gpurun_out/pmc_mem.progress)
cfg=${1:-c5s}
out=${2:-gpurun_out/pmc_mem_$cfg.txt}
repo=$PWD
mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcm_$cfg
i=0
for set in "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  echo "pass $i start $(date +%T): $set" >> $repo/gpurun_out/pmc_mem.progress
  timeout -k 5 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcm_$cfg/p$i -- python3 $repo/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmcm_$cfg.$i.log 2>&1
  echo "pass $i rc=$? end $(date +%T): $(tail -c 300 /tmp/pmcm_$cfg.$i.log | tr '\n' ' ')" >> $repo/gpurun_out/pmc_mem.progress
done
cd $repo
python3 tools/pmc_sq.py /tmp/pmcm_$cfg/p* > $out
