#!/bin/bash
# Everything profiles/<tag>_* is made from, in one gpurun call (about 4 minutes of box time):  tools/profile_round.sh r02
#   headline (bench.py): kernel-trace stats + PMC passes          -> gpurun_out/prof_<tag>/
#   phase split of the headline kernel (diagnostic library)       -> gpurun_out/prof_<tag>/phases.txt
#   C3 / C5 kernels: kernel-trace stats + PMC                     -> gpurun_out/prof_<tag>_c3, _c5
#   secondary configs, whole recipe                               -> gpurun_out/prof_<tag>/other_configs.jsonl, icpc_recipe.jsonl
#   the recipe's launches (stages + program), stored FIR, VM PMC  -> gpurun_out/prof_<tag>/icpc_trace, icpc_rate.json, fir_store_rate.json, prof_<tag>_vm
set -u
export TMPDIR=/tmp
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
tools/profile_bench.sh "$TAG" > "$OUT/profile_bench.log" 2>&1
python3 -m dspeed_amd.build --diag > "$OUT/diag_build.log" 2>&1
DSPEED_HIP_LIB=$PWD/dspeed_amd/libdspeed_hip_diag.so DSPEED_HIP_ABLATE=8 python3 bench.py --allow-variants --no-cpu --steps 10 --warmup 3 > "$OUT/phases.json" 2> "$OUT/phases.txt"
tools/pmc_kernel.sh gpurun_out/prof_${TAG}_c3 tools/c3_rate.py 250000 1 > /dev/null 2>&1
python3 tools/pmc_table.py gpurun_out/prof_${TAG}_c3 "" "$OUT/c3_pmc.json" > /dev/null
tools/pmc_kernel.sh gpurun_out/prof_${TAG}_c5 tools/c5_rate.py 1000000 1 > /dev/null 2>&1
python3 tools/pmc_table.py gpurun_out/prof_${TAG}_c5 "" "$OUT/c5_pmc.json" > /dev/null
python3 tools/c3_rate.py 250000 1 > "$OUT/c3_rate.json" 2>/dev/null
python3 tools/c5_rate.py 1000000 1 > "$OUT/c5_rate.json" 2>/dev/null
python3 tools/bench_configs.py 1000000 > "$OUT/other_configs.jsonl" 2> "$OUT/other_configs.err"
python3 tools/icpc_breakdown.py > "$OUT/icpc_recipe.jsonl" 2> "$OUT/icpc_recipe.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/icpc_trace" -- python3 tools/icpc_rate.py 131072 5 > "$OUT/icpc_rate_traced.json" 2> "$OUT/icpc_trace.err"
python3 tools/icpc_rate.py 131072 5 > "$OUT/icpc_rate.json" 2> /dev/null
python3 tools/fir_store_rate.py 100000 1 > "$OUT/fir_store_rate.json" 2> /dev/null
# the run-length FIR (piecewise-constant kernels: the t0 filter) in its three forms, the matrix-core FIR on the same rows, its counters
python3 tools/fir_runs_rate.py 131072 8192 5 > "$OUT/fir_runs_rate.jsonl" 2> /dev/null
DSPEED_HIP_NO_FIR_RUNS=1 python3 tools/fir_runs_rate.py 131072 8192 2 > "$OUT/fir_runs_rate_mfma.jsonl" 2> /dev/null
tools/pmc_kernel.sh gpurun_out/prof_${TAG}_runs tools/fir_runs_rate.py 65536 8192 2 > /dev/null 2>&1
python3 tools/pmc_table.py gpurun_out/prof_${TAG}_runs "dsp_fir_runs_kernel" "$OUT/fir_runs_pmc.json" > /dev/null
# the long FIRs on the float32 matrix instructions (the float16 form is the default) and the accuracy of both against float64
DSPEED_HIP_FIR_F32=1 python3 tools/c3_rate.py 250000 1 > "$OUT/c3_rate_f32.json" 2>/dev/null
DSPEED_HIP_FIR_F32=1 python3 tools/fir_store_rate.py 100000 1 > "$OUT/fir_store_rate_f32.json" 2> /dev/null
python3 tools/fir_f16_accuracy.py > "$OUT/fir_f16_accuracy.json" 2> /dev/null
# the recipe on pulses with a charge-collection time (the rise-time walks end where they do on detector pulses) and with the float32 FIRs
ICPC_RISE=6,60 python3 tools/icpc_rate.py 131072 5 > "$OUT/icpc_rate_rise.json" 2> /dev/null
DSPEED_HIP_FIR_F32=1 python3 tools/icpc_rate.py 131072 5 > "$OUT/icpc_rate_fir_f32.json" 2> /dev/null
tools/icpc_marginal.sh > /dev/null 2>&1
python3 tools/e2e_recipe_rate.py 400000 > "$OUT/e2e_recipe_rate.json" 2> /dev/null
for r in 16384 32768 65536; do python3 tools/icpc_rate.py $r 10; done > "$OUT/icpc_rate_small.jsonl" 2> /dev/null
DSPEED_HIP_NO_FUSED=1 tools/pmc_kernel.sh gpurun_out/prof_${TAG}_vm bench.py --allow-variants --no-cpu --rows 500000 --steps 5 --warmup 2 > /dev/null 2>&1
python3 tools/pmc_table.py gpurun_out/prof_${TAG}_vm "dsp_vm" "$OUT/vm_pmc.json" > /dev/null
python3 bench.py --wf-len 8192 --rows 500000 --no-cpu --steps 10 --warmup 5 > "$OUT/bench_8192.json" 2>/dev/null
tools/pmc_kernel.sh gpurun_out/prof_${TAG}_icpc tools/icpc_rate.py 131072 3 > /dev/null 2>&1
python3 tools/pmc_table.py gpurun_out/prof_${TAG}_icpc "" "$OUT/icpc_pmc.json" > /dev/null
python3 tools/c5_instruction_roofline.py "$OUT/c5_rate.json" "$OUT/c5_pmc.json" > "$OUT/c5_rate_roofline.json" 2> /dev/null
echo "round profile $TAG done"; tail -2 "$OUT/phases.txt"; cat "$OUT/c3_rate.json" "$OUT/c5_rate.json"
