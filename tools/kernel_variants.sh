#!/bin/bash
# one kernel of the Ge recipe under rocprofv3 in the builds named on the command line (python -m dspeed_amd.build --variant NAME --define ...):
# calls, total and average ns per build.  Usage (GPU box): bash tools/kernel_variants.sh 'KERNEL-NAME-PATTERN' NAME...   ("default" = the product)
export TMPDIR=/tmp
pat="$1"; shift
for v in "$@"; do
  lib=$PWD/dspeed_amd/libdspeed_hip_$v.so
  [ "$v" = default ] && lib=$PWD/dspeed_amd/libdspeed_hip.so
  DSPEED_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kv_$v -- python3 tools/icpc_rate.py 131072 5 > gpurun_out/kv_$v.json 2> gpurun_out/kv_$v.err
  find gpurun_out/kv_$v -name '*kernel_stats.csv' | xargs grep -h "$pat" | awk -v v="$v" -F'",' '{split($2,a,","); printf "%-16s %s  avg %.3f ms\n", v, substr($1,30,60), a[3]/1e6}'
done
