#!/bin/bash
# PMC counters of one command's kernels (run through gpurun from the repo root):  tools/pmc_kernel.sh <out-dir> <python script and args...>
# Separate passes (counter slots), csv output; tools/pmc_table.py prints per-kernel averages.
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$OUT/sq1" -- python3 "$@" > "$OUT/sq1.out" 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA \
    --output-format csv -d "$OUT/sq2" -- python3 "$@" > "$OUT/sq2.out" 2> "$OUT/sq2.err"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/sq3" -- python3 "$@" > "$OUT/sq3.out" 2> "$OUT/sq3.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$@" > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$@" > /dev/null 2> "$OUT/write.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$@" > "$OUT/trace.out" 2> "$OUT/trace.err"
echo "pmc done: $OUT"
