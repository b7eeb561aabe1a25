#!/bin/bash
# All profile artefacts of the round (r04) of one bench configuration, from ONE box:
#   bash tools/profile_cfg.sh <tag> [bench.py arguments ...]        e.g.  c3   |   c3_residual --schedule residual   |   c5 --config c5
#   profiles/r04_<tag>_rocprofv3_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the command (top rows)
#   profiles/r04_pmc_hbm_traffic_<tag>.json         --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs), per kernel,
#                                                   stamped with the hash of the kernel sources (tools/pmc_summary.py)
#   profiles/r04_bench_<tag>.json                   the bench line of the same command, taken last (so it carries the traffic)
tag=$1; shift
repo=$PWD
out=$repo/gpurun_out/profiles_r04   # only gpurun_out/ comes back from the GPU box: copy into profiles/ afterwards
mkdir -p $out $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf_$tag
echo "[$tag] kernel stats $(date +%T)" >> $repo/gpurun_out/profile_progress.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$tag/st -o st -- python3 $repo/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline > /tmp/pf_$tag.st.log 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  echo "[$tag] pmc $ctr $(date +%T)" >> $repo/gpurun_out/profile_progress.log
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pf_$tag/$ctr -- python3 $repo/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pf_$tag.$ctr.log 2>&1
done
cd $repo
f=$(find /tmp/pf_$tag/st -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$out/r04_${tag}_rocprofv3_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], 'w', newline='') as out:
    w = csv.writer(out, quoting=csv.QUOTE_MINIMAL)
    for r in rows[:16]:
        w.writerow([r[0][:120]] + r[1:])
PY
python3 tools/pmc_summary.py $out/r04_pmc_hbm_traffic_$tag.json $(dirname $(find /tmp/pf_$tag/FETCH_SIZE -name '*counter_collection.csv' | head -1)) $(dirname $(find /tmp/pf_$tag/WRITE_SIZE -name '*counter_collection.csv' | head -1)) > $out/r04_pmc_hbm_traffic_$tag.txt 2>&1
cp $out/r04_pmc_hbm_traffic_$tag.json profiles/ 2>/dev/null   # so that the bench line below finds it
echo "[$tag] bench $(date +%T)" >> $repo/gpurun_out/profile_progress.log
timeout -k 10 400 python3 bench.py "$@" > /tmp/pf_$tag.json 2>/tmp/pf_$tag.err && tail -1 /tmp/pf_$tag.json > $out/r04_bench_$tag.json
tail -3 /tmp/pf_$tag.FETCH_SIZE.log > $out/r04_${tag}_pmc_fetch_tail.log; tail -3 /tmp/pf_$tag.err >> $out/r04_${tag}_pmc_fetch_tail.log
echo "[$tag] done $(date +%T)" >> $repo/gpurun_out/profile_progress.log
head -4 $out/r04_${tag}_rocprofv3_kernel_stats.csv | cut -c1-160
