#!/bin/bash
# LDS counters of the waveform VM, op by op:  tools/vm_probe_all.sh <out-dir>   (GPU box; about 3 minutes)
set -u
OUT=${1:-gpurun_out/vm_probe}
export TMPDIR=/tmp
mkdir -p "$OUT"
for p in load bl pz trap c2 minmax tpt; do
    rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS \
        --output-format csv -d "$OUT/$p/pmc" -- python3 tools/vm_probe.py $p 200000 > "$OUT/$p.json" 2> "$OUT/$p.err"
    python3 tools/pmc_table.py "$OUT/$p" dsp_vm > "$OUT/$p.pmc.json" 2>/dev/null
done
echo probes done
