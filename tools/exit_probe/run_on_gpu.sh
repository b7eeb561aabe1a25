#!/bin/bash
# One pass over the variants (never a loop): each under rocprofv3 --kernel-trace --stats, the program directly after `--`.
# A fault at exit is the expected outcome of some of them, so the script goes on after one; it stops after a timeout.
out=gpurun_out/exit_probe; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # tag, command...
    tag=$1; shift
    timeout -k 10 180 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ep_$tag -o t -- "$@" > $R/$out/$tag.log 2>&1
    rc=$?
    echo "$tag: exit code $rc" | tee -a $R/$out/summary.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping" | tee -a $R/$out/summary.txt; exit 1; fi
}
: > $R/$out/summary.txt
run min_plain_leak    $R/tools/exit_probe/coop_min plain leak    $R/$out/min_plain_leak.maps
run min_coop_release  $R/tools/exit_probe/coop_min coop  release $R/$out/min_coop_release.maps
run min_coop_leak     $R/tools/exit_probe/coop_min coop  leak    $R/$out/min_coop_leak.maps
run lib_plain_leak    python3 $R/tools/exit_probe/lib_exit_probe.py $R/$out/lib_plain_leak.maps plain leak
run lib_coop_release  python3 $R/tools/exit_probe/lib_exit_probe.py $R/$out/lib_coop_release.maps coop release
run lib_coop_leak     python3 $R/tools/exit_probe/lib_exit_probe.py $R/$out/lib_coop_leak.maps coop leak
# the same cooperative program with no tool around it
timeout -k 10 120 $R/tools/exit_probe/coop_min coop leak $R/$out/min_coop_notool.maps > $R/$out/min_coop_notool.log 2>&1; echo "min_coop_notool: exit code $?" | tee -a $R/$out/summary.txt
timeout -k 10 180 python3 $R/tools/exit_probe/lib_exit_probe.py $R/$out/lib_coop_notool.maps coop leak > $R/$out/lib_coop_notool.log 2>&1; echo "lib_coop_notool: exit code $?" | tee -a $R/$out/summary.txt
cat $R/$out/summary.txt
