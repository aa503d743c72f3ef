#!/bin/bash
# What parts of the Ge recipe's main program cost: the pass and its longest interpreter launch (the main program) with groups of outputs left out.
#   tools/icpc_marginal.sh   (on the GPU box; writes gpurun_out/icpc_marginal.txt)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/icpc_marginal
mkdir -p "$OUT"
: > gpurun_out/icpc_marginal.txt
run() {
    local tag=$1 drop=$2
    ICPC_DROP="$drop" rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -- python3 tools/icpc_rate.py 131072 5 > "$OUT/$tag.json" 2> "$OUT/$tag.err"
    local f
    f=$(find "$OUT/$tag" -name '*kernel_stats.csv' | head -1)
    echo "== $tag (dropped: $drop)" >> gpurun_out/icpc_marginal.txt
    cat "$OUT/$tag.json" >> gpurun_out/icpc_marginal.txt
    grep -E "dsp_vm_kernel|dsp_scalar|dsp_fir_store_kernel|dsp_rows|dsp_current|dsp_fit" "$f" | cut -d, -f1-4,6,7 | cut -c1-200 >> gpurun_out/icpc_marginal.txt
}
run full ""
run no_raw_minmax "tp_min,tp_max,wf_min,wf_max"
run no_cusp "cuspEmax,cuspEftp"
run no_risetimes "tp_10,tp_50,tp_90,tp_99,tp_100"
run no_trap_energy "trapEmax,trapEftp"
run no_drift "QDrift,dt_eff"
cat gpurun_out/icpc_marginal.txt
