( timeout -k 10 300 python3 tools/onchip_two_processes.py 2 400 2 2>&1 | grep -v amdgpu.ids; timeout -k 10 300 python3 tools/onchip_two_processes.py 4 400 2 2>&1 | grep -v amdgpu.ids ) > gpurun_out/onchip_two_processes.log 2>&1
tail -12 gpurun_out/onchip_two_processes.log
bash tools/onchip_sizes.sh > gpurun_out/onchip_sizes.log 2>&1
grep -c "sweeps in" gpurun_out/onchip_sizes.log
( for m in "" tm; do echo "== 10000 1000 20 100 $m (diagnostics build, 2000 topic steps; us total per section)"; RRI_ONCHIP_TIMING=1 timeout -k 10 120 python3 tools/onchip_probe.py 10000 1000 20 100 $m 2>&1 | grep "sections" | tail -2; done ) > gpurun_out/onchip_sections.log 2>&1
(for a in "5000 800 47 47 3" "5000 800 47 47 3 warm" "5000 1000 64 64 3" "5000 1000 64 64 3 warm" "5000 1000 64 20 3" "5000 1000 64 20 3 warm"; do timeout -k 10 300 python3 tools/onchip_large_k_check.py $a 2>&1 | grep sweeps; done) > gpurun_out/onchip_large_k.log
