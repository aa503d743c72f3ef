#!/usr/bin/env python3
"""C5 (dsp_rows_kernel) against the bound that binds it: the kernel reads each row once (traffic / algorithmic = 1.0, HBM at 15 %) and is
limited by the vector instructions it issues.  Instruction roofline: a SIMD of gfx950 issues one wave64 VALU instruction per 2 cycles
(MI355X_MICROARCH.md, cycle constants: `v_fma_f32` 2 cycles on a SIMD-32; float64 operations run at half that rate), 4 SIMDs per CU, 256 CUs.
    python tools/c5_instruction_roofline.py <c5_rate.json> <c5_pmc.json>   ->  the rate record with the instruction-roofline fields added"""
import json
import sys

rate = json.load(open(sys.argv[1]))
pmc = json.load(open(sys.argv[2]))
k = next(v for name, v in pmc.items() if "dsp_rows_kernel" in name)
rows = rate["rows"]
valu, salu, lds = k["SQ_INSTS_VALU"] / rows, k["SQ_INSTS_SALU"] / rows, k["SQ_INSTS_LDS"] / rows
CUS, SIMDS, CLK = 256, 4, 2.4e9
issue_peak = CUS * SIMDS * CLK / 2.0  # wave64 VALU instructions per second, all float32
rate.update({"valu_wave_insts_per_row": valu, "salu_insts_per_row": salu, "lds_insts_per_row": lds,
             "valu_issue_peak_wave_insts_per_s": issue_peak, "valu_issue_bound_waveforms_per_s": issue_peak / valu,
             "frac_valu_issue": rate["waveforms_per_s"] * valu / issue_peak,
             "note": "about 40 % of the kernel's vector instructions are float64 (double_pole_zero, the trapezoid's divisions), which issue at half "
                     "the float32 rate: priced so, the issue bound is ~1.4x lower and the fraction correspondingly higher",
             "wave_cycles_waiting_frac": k.get("SQ_WAIT_ANY", 0.0) / max(k.get("SQ_WAVE_CYCLES", 1.0), 1.0)})
print(json.dumps(rate, indent=1))
