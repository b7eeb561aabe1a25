#!/bin/bash
# the register-resident sweep against the launch-per-phase schedule over shapes and flag sets (tools/onchip_probe.py), one box:
#   bash tools/onchip_sizes.sh > gpurun_out/onchip_sizes.log
for cfg in "10000 1000 20 100" "10000 1000 20 100 tm" "2000 500 10 100" "2000 500 10 100 tm" "500 100 5 200" "500 100 5 200 tm" "20000 512 20 100" "5000 1000 22 100 tm" "10240 1024 8 100 tm" \
           "5000 1000 24 100" "5000 1000 32 100" "5000 1000 48 50" "5000 1000 64 50" "10000 1000 32 50" "5000 1000 32 50 tm" "5000 1000 64 50 tm" \
           "5000 2000 20 100" "2500 2048 10 100" "5000 1500 32 50"; do
  echo "== $cfg"
  timeout -k 10 120 python3 tools/onchip_probe.py $cfg 2>&1 | grep -v "eligible" || exit 1
done
