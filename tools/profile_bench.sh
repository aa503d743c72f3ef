#!/bin/bash
# Profiles the default bench.py command on the GPU box (run through gpurun from the repo root):
#   tools/profile_bench.sh r01
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ ; tools/summarise_profile.py turns it into profiles/<tag>_*.
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
BENCH="python3 bench.py --no-cpu"  # the default command (30 timed launches after 10 warm-up ones) without the CPU leg
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$OUT/pmc_sq1" -- $BENCH > /dev/null 2> "$OUT/pmc_sq1.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA \
    --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > /dev/null 2> "$OUT/pmc_sq2.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > /dev/null 2> "$OUT/pmc_write.err"
# the shard a rank of a multi-GPU run processes (1 250 000 rows): HBM traffic of its launches
SHARD="python3 bench.py --no-cpu --rows 1250000 --steps 10 --warmup 5"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_shard" -- $SHARD > /dev/null 2> "$OUT/pmc_fetch_shard.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_shard" -- $SHARD > /dev/null 2> "$OUT/pmc_write_shard.err"
$SHARD > "$OUT/bench_shard.json" 2> "$OUT/bench_shard.err"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "profile $TAG done"; cat "$OUT/bench_default.json"
