#!/bin/bash
# the pole-zero rows kernel under rocprofv3 in the builds named on the command line (python -m dspeed_amd.build --variant NAME --define ...):
# its average duration per build.  Usage (GPU box): bash tools/pz_variants.sh NAME...
export TMPDIR=/tmp
for v in "$@"; do
  lib=$PWD/dspeed_amd/libdspeed_hip_$v.so
  [ "$v" = default ] && lib=$PWD/dspeed_amd/libdspeed_hip.so
  DSPEED_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pzv_$v -- python3 tools/icpc_rate.py 131072 5 > gpurun_out/pzv_$v.json 2> gpurun_out/pzv_$v.err
  echo "$v $(find gpurun_out/pzv_$v -name '*kernel_stats.csv' | xargs grep -h pz_rows | awk -F'",' '{print $2}' | cut -d, -f1-3)"
done
