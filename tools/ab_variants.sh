#!/bin/bash
# A/B runs of the energy-kernel variants on the GPU box:  tools/ab_variants.sh 6 8 1
#   DSPEED_HIP_VARIANT: 6 = register-resident (default), 8 = the same with two replay sub-chains per lane, 1 = classic kernel
# For each: one run with per-phase cycle stamps (DSPEED_HIP_ABLATE=8: stage, pass 1, pass 2, carries, pass 3, tail) and one clean run.
for v in "$@"; do
    DSPEED_HIP_VARIANT=$v DSPEED_HIP_ABLATE=8 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bi_$v.json 2> gpurun_out/bi_$v.err
    grep stamps gpurun_out/bi_$v.err | tail -1
    DSPEED_HIP_VARIANT=$v timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/bn_$v.json 2> gpurun_out/bn_$v.err
    python -c "import json;d=json.load(open('gpurun_out/bn_$v.json'));print('variant',$v,d['config']['kernel'],round(d['value']/1e6,1),'M wf/s',round(d['roofline']['frac'],3),d['parity_max_rel_vs_oracle'])"
done
