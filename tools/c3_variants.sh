#!/bin/bash
# C3 (tools/c3_rate.py) under rocprofv3 in the builds named on the command line ("default" = the product): the rate and the float16 FIR's kernels.
# Usage (GPU box): bash tools/c3_variants.sh NAME...
export TMPDIR=/tmp
for v in "$@"; do
  lib=$PWD/dspeed_amd/libdspeed_hip_$v.so; [ "$v" = default ] && lib=$PWD/dspeed_amd/libdspeed_hip.so
  [ -f "$lib" ] || { echo "$v: no such build"; continue; }
  DSPEED_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3v_$v -- python3 tools/c3_rate.py 250000 1 > gpurun_out/c3v_$v.json 2>/dev/null
  echo "$v $(cut -c1-120 gpurun_out/c3v_$v.json)"
  find gpurun_out/c3v_$v -name '*kernel_stats.csv' | xargs grep -h "f16" | awk -F'",' '{split($2,a,","); printf "   %s avg %.3f ms x%s\n", substr($1,30,50), a[3]/1e6, a[1]}'
done
