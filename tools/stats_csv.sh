#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench configuration -> first 14 rows of kernel_stats.csv (names cut to 110
# characters) and, on the same box right after it, the bench line:  bash tools/stats_csv.sh <config> <csv out> <json out>
cfg=$1; csv=$2; js=$3
repo=$PWD
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st_$cfg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$cfg -o st -- python3 $repo/bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline > /tmp/st_$cfg.log 2>&1
cd $repo
f=$(find /tmp/st_$cfg -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$csv" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], 'w', newline='') as out:
    w = csv.writer(out, quoting=csv.QUOTE_MINIMAL)
    for r in rows[:15]:
        w.writerow([r[0][:110]] + r[1:])
PY
python3 bench.py --config $cfg > /tmp/st_$cfg.json 2>/dev/null && tail -1 /tmp/st_$cfg.json > $js
