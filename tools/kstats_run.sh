#!/bin/bash
# rocprofv3 kernel trace of one bench configuration, summarised: bash tools/kstats_run.sh <config> <outfile> [bench args]
cfg=$1; out=$2; shift 2
repo=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ks_$cfg
rocprofv3 --kernel-trace -d /tmp/ks_$cfg -o ks -- python3 $repo/bench.py --config $cfg --no-cpu-baseline "$@" > /tmp/ks_$cfg.log 2>&1
cd $repo
python3 tools/kstats.py $(find /tmp/ks_$cfg -name '*.db' | head -1) 0.3 > $out
tail -1 /tmp/ks_$cfg.log | cut -c1-300 >> $out
