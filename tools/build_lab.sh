#!/bin/bash
# builds tools/lpn_lab (development harness for the config-5 kernels) and prints the ISA statistics of one kernel if asked
cd "$(dirname "$0")" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -Wno-unused-value -Wno-unused-lambda-capture -o lpn_lab lpn_lab.hip -ldl 2>&1 | grep -E "error" -A3 | head -30
if [ -n "$1" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-pass-failed -Wno-unused-value -Wno-unused-lambda-capture -S --cuda-device-only -o /tmp/lab.s lpn_lab.hip 2>/dev/null
  /tmp/isastat.sh /tmp/lab.s "$1"
fi
