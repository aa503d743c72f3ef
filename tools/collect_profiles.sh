#!/bin/bash
# Copies what tools/profile_round.sh <tag> left under gpurun_out/ into the tracked profiles/<tag>_* files (after python tools/summarise_profile.py <tag>).
set -u
TAG=${1:-r03}
S=gpurun_out/prof_$TAG
cp() { [ -s "$1" ] && command cp "$1" "$2"; }
cp $S/c3_rate.json profiles/${TAG}_c3_rate.json
cp $S/c3_rate_f32.json profiles/${TAG}_c3_rate_f32.json
cp $S/c3_pmc.json profiles/${TAG}_c3_pmc.json
cp $S/c5_rate_roofline.json profiles/${TAG}_c5_rate.json
cp $S/c5_pmc.json profiles/${TAG}_c5_pmc.json
cp $S/other_configs.jsonl profiles/${TAG}_other_configs.jsonl
cp $S/icpc_recipe.jsonl profiles/${TAG}_icpc_recipe.jsonl
cp $S/icpc_rate.json profiles/${TAG}_icpc_rate.json
cp $S/icpc_rate_rise.json profiles/${TAG}_icpc_rate_rise.json
cp $S/icpc_rate_fir_f32.json profiles/${TAG}_icpc_rate_fir_f32.json
cp $S/icpc_rate_small.jsonl profiles/${TAG}_icpc_rate_small.jsonl
cp $S/icpc_pmc.json profiles/${TAG}_icpc_pmc.json
cp $S/fir_store_rate.json profiles/${TAG}_fir_store_rate.json
cp $S/fir_store_rate_f32.json profiles/${TAG}_fir_store_rate_f32.json
cp $S/fir_runs_rate.jsonl profiles/${TAG}_fir_runs_rate.jsonl
cp $S/fir_runs_rate_mfma.jsonl profiles/${TAG}_fir_runs_rate_mfma.jsonl
cp $S/fir_runs_pmc.json profiles/${TAG}_fir_runs_pmc.json
cp $S/fir_f16_accuracy.json profiles/${TAG}_fir_f16_accuracy.json
cp $S/e2e_recipe_rate.json profiles/${TAG}_e2e_recipe_rate.json
cp $S/vm_pmc.json profiles/${TAG}_vm_pmc.json
cp $S/bench_8192.json profiles/${TAG}_bench_8192.json
cp $S/phases.txt profiles/${TAG}_headline_phases.txt
cp gpurun_out/icpc_marginal.txt profiles/${TAG}_icpc_marginal.txt
for d in c3 c5; do
    f=$(ls -t gpurun_out/prof_${TAG}_$d/trace/*/*_kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && command cp "$f" profiles/${TAG}_${d}_kernel_stats.csv
done
f=$(ls -t $S/icpc_trace/*/*_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && command cp "$f" profiles/${TAG}_icpc_kernel_stats.csv
ls -la profiles/${TAG}_* | wc -l
