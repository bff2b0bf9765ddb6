#!/bin/bash
# Per-kernel register / LDS / occupancy report from the compiler (tuning aid, not part of the product).
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Irfs-slam_amd/csrc "$@" rfs-slam_amd/csrc/rfsgpu_engine.hip -o /tmp/rfsgpu_res.so -Rpass-analysis=kernel-resource-usage 2>/tmp/rfsgpu_res.txt
grep -E "Function Name|  VGPRs:|AGPRs|Occupancy|ScratchSize" /tmp/rfsgpu_res.txt | sed 's/.*remark: //;s/\[-Rpass.*//' | paste - - - - - | sed 's/Function Name: //' | awk '{print}' | cut -c1-200
