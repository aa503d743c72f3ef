"""BASELINE.json configs as dspeed recipes, shared by CPU (translation) and GPU (parity) tests."""

C1 = {
    "outputs": ["wf_trap"],
    "processors": {
        "wf_pz": {"function": "pole_zero", "module": "dspeed.processors", "args": ["waveform", "1716.28", "wf_pz"]},
        "wf_trap": {"function": "trap_filter", "module": "dspeed.processors", "args": ["wf_pz", "64", "16", "wf_trap"]},
    },
}

C2 = {
    "outputs": ["trapEftp"],
    "processors": {
        "wf_blsub": "dspeed.processors.bl_subtract(waveform, baseline, wf_blsub)",
        "wf_pz": {"function": "pole_zero", "module": "dspeed.processors", "args": ["wf_blsub", "db.pz.tau", "wf_pz"],
                  "defaults": {"db.pz.tau": "1716.28"}},
        "wf_trap": {"function": "trap_filter", "module": "dspeed.processors", "args": ["wf_pz", "625", "188", "wf_trap"]},
        "trapEftp": {"function": "fixed_time_pickoff", "module": "dspeed.processors", "args": ["wf_trap", "t_pick", "'l'", "trapEftp"]},
    },
}

# same chain written with time quantities, as LEGEND configs do (icpc-dsp-config.json:116-158)
C2_UNITS = {
    "outputs": ["trapEftp", "wf_trap"],
    "processors": {
        "wf_blsub": "dspeed.processors.bl_subtract(waveform, baseline, wf_blsub)",
        "wf_pz": {"function": "pole_zero", "module": "dspeed.processors", "args": ["wf_blsub", "db.pz.tau", "wf_pz"],
                  "defaults": {"db.pz.tau": "27460.5*ns"}},
        "wf_trap": {"function": "trap_filter", "module": "dspeed.processors", "args": ["wf_pz", "db.etrap.rise", "db.etrap.flat", "wf_trap"],
                    "defaults": {"db.etrap.rise": "10*us", "db.etrap.flat": "3.008*us"}},
        "trapEftp": {"function": "fixed_time_pickoff", "module": "dspeed.processors", "args": ["wf_trap", "t_pick", "'l'", "trapEftp"]},
    },
}

C3 = {
    "outputs": ["cuspEmax", "zacEmax"],
    "processors": {
        "wf_blsub": "dspeed.processors.bl_subtract(waveform, baseline, wf_blsub)",
        "cusp_kernel": {"function": "cusp_filter", "module": "dspeed.processors", "args": ["1250", "188", "28125", "cusp_kernel(5792, 'f')"]},
        "zac_kernel": {"function": "zac_filter", "module": "dspeed.processors", "args": ["1250", "188", "28125", "zac_kernel(5792, 'f')"]},
        "wf_cusp": {"function": "convolve_wf", "module": "dspeed.processors", "args": ["wf_blsub[:6092]", "cusp_kernel", "'v'", "wf_cusp(301, 'f')"]},
        "wf_zac": {"function": "convolve_wf", "module": "dspeed.processors", "args": ["wf_blsub[:6092]", "zac_kernel", "'v'", "wf_zac(301, 'f')"]},
        "cuspEmax": {"function": "amax", "module": "numpy", "args": ["wf_cusp", 1, "cuspEmax"], "kwargs": {"signature": "(n),()->()", "types": ["fi->f"]}},
        "zacEmax": {"function": "amax", "module": "numpy", "args": ["wf_zac", 1, "zacEmax"], "kwargs": {"signature": "(n),()->()", "types": ["fi->f"]}},
    },
}


def c3_small(n=512, m=129, lo=0, hi=400):
    """C3 at a size the oracle finishes quickly."""
    p = hi - lo - m + 1
    return {
        "outputs": ["cuspEmax", "zacEmax", "wf_zac"],
        "processors": {
            "wf_blsub": "dspeed.processors.bl_subtract(waveform, baseline, wf_blsub)",
            "cusp_kernel": {"function": "cusp_filter", "module": "dspeed.processors", "args": ["30", "10", "400", f"cusp_kernel({m}, 'f')"]},
            "zac_kernel": {"function": "zac_filter", "module": "dspeed.processors", "args": ["30", "10", "400", f"zac_kernel({m}, 'f')"]},
            "wf_cusp": {"function": "convolve_wf", "module": "dspeed.processors", "args": [f"wf_blsub[{lo}:{hi}]", "cusp_kernel", "'v'", f"wf_cusp({p}, 'f')"]},
            "wf_zac": {"function": "convolve_wf", "module": "dspeed.processors", "args": [f"wf_blsub[{lo}:{hi}]", "zac_kernel", "'v'", f"wf_zac({p}, 'f')"]},
            "cuspEmax": {"function": "amax", "module": "numpy", "args": ["wf_cusp", 1, "cuspEmax"]},
            "zacEmax": {"function": "amax", "module": "numpy", "args": ["wf_zac", 1, "zacEmax"]},
        },
    }


C5 = {
    "outputs": ["tp_0", "tp_min", "tp_max", "wf_min", "wf_max", "dwt_haar"],
    "processors": {
        "wf_pz": {"function": "double_pole_zero", "module": "dspeed.processors", "args": ["waveform", "1716.28", "62.5", "0.02", "wf_pz"]},
        "wf_atrap": {"function": "asym_trap_filter", "module": "dspeed.processors", "args": ["wf_pz", "8", "4", "125", "wf_atrap"]},
        "tp_min, tp_max, wf_min, wf_max": {"function": "min_max", "module": "dspeed.processors",
                                           "args": ["wf_atrap", "tp_min", "tp_max", "wf_min", "wf_max"]},
        "tp_0": {"function": "time_point_thresh", "module": "dspeed.processors", "args": ["wf_atrap", "thr", "tp_max", 0, "tp_0"]},
        "dwt_haar": {"function": "discrete_wavelet_transform", "module": "dspeed.processors",
                     "args": ["wf_pz", 5, "'h'", "'a'", "dwt_haar(256, 'f')"]},
    },
}
