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

# The whole Ge recipe with LEGEND's structure (icpc-dsp-config.json:1-347): every branch of it in ONE recipe, written in the
# reference's full argument language -- units on outputs, grids on declarations, expressions between per-event variables, rounding
# onto a waveform's grid.  Own parameter values; what matters is that each construct of that file appears.
_M = "dspeed.processors"
ICPC = {
    "outputs": ["tp_min", "tp_max", "wf_min", "wf_max", "bl_mean", "bl_std", "bl_slope", "bl_intercept", "pz_mean", "pz_std", "trapTmax",
                "tp_0_est", "tp_0_atrap", "tp_10", "tp_50", "tp_90", "tp_99", "tp_100", "A_max", "QDrift", "dt_eff", "tp_aoe_max",
                "tp_aoe_samp", "trapEmax", "trapEftp", "cuspEmax", "cuspEftp"],
    "processors": {
        "tp_min, tp_max, wf_min, wf_max": {"function": "min_max", "module": _M, "args": ["waveform", "tp_min", "tp_max", "wf_min", "wf_max"],
                                           "unit": ["ns", "ns", "ADC", "ADC"]},
        "wf_blsub": f"{_M}.bl_subtract(waveform, baseline, wf_blsub(unit='ADC'))",
        "bl_mean , bl_std, bl_slope, bl_intercept": {"function": "linear_slope_fit", "module": _M, "unit": ["ADC"] * 4,
                                                      "args": ["wf_blsub[0:700]", "bl_mean", "bl_std", "bl_slope", "bl_intercept"]},
        "wf_pz": {"function": "pole_zero", "module": _M, "args": ["wf_blsub", "db.pz.tau", "wf_pz"], "unit": "ADC",
                  "defaults": {"db.pz.tau": "27.46*us"}},
        "pz_mean , pz_std, pz_slope, pz_intercept": {"function": "linear_slope_fit", "module": _M, "unit": ["ADC"] * 4,
                                                      "args": ["wf_pz[1600:]", "pz_mean", "pz_std", "pz_slope", "pz_intercept"]},
        "t0_kernel": {"function": "t0_filter", "module": _M, "unit": "ADC",
                      "args": ["128*ns/wf_pz.period", "2*us/wf_pz.period", "t0_kernel(round((128*ns+2*us)/wf_pz.period), 'f')"]},
        "wf_t0_filter": {"function": "convolve_wf", "module": _M, "unit": "ADC",
                         "args": ["wf_pz", "t0_kernel", "'s'", "wf_t0_filter(len(wf_pz), 'f', grid=wf_pz.grid)"]},
        "wf_atrap": {"function": "asym_trap_filter", "module": _M, "args": ["wf_pz", "128*ns", "4", "2*us", "wf_atrap"], "unit": "ADC"},
        "conv_tmin ,tp_start, conv_min, conv_max": {"function": "min_max", "module": _M, "unit": ["ns", "ns", "ADC", "ADC"],
                                                    "args": ["wf_t0_filter", "conv_tmin", "tp_start", "conv_min", "conv_max"]},
        "tp_0_atrap": {"function": "time_point_thresh", "module": _M, "args": ["wf_atrap", "bl_std", "tp_start", 0, "tp_0_atrap"], "unit": "ns"},
        "tp_0_est": {"function": "time_point_thresh", "module": _M, "args": ["wf_t0_filter", "bl_std", "tp_start", 0, "tp_0_est(unit=ns)"],
                     "unit": "ns"},
        "wf_trap": {"function": "trap_norm", "module": _M, "args": ["wf_pz", "db.ttrap.rise", "db.ttrap.flat", "wf_trap"], "unit": "ADC",
                    "defaults": {"db.ttrap.rise": "10*us", "db.ttrap.flat": "3.008*us"}},
        "trapTmax": {"function": "amax", "module": "numpy", "args": ["wf_trap", 1, "trapTmax"], "unit": "ADC",
                     "kwargs": {"signature": "(n),()->()", "types": ["fi->f"]}},
        "wf_etrap": {"function": "trap_norm", "module": _M, "args": ["wf_pz", "db.etrap.rise", "db.etrap.flat", "wf_etrap"], "unit": "ADC",
                     "defaults": {"db.etrap.rise": "8*us", "db.etrap.flat": "2*us"}},
        "trapEmax": {"function": "amax", "module": "numpy", "args": ["wf_etrap", 1, "trapEmax"], "unit": "ADC"},
        "trapEftp": {"function": "fixed_time_pickoff", "module": _M, "unit": "ADC",
                     "args": ["wf_etrap", "round(tp_0_est+db.etrap.rise+db.etrap.flat*db.etrap.sample, wf_etrap.grid)", "'l'", "trapEftp"],
                     "defaults": {"db.etrap.rise": "8*us", "db.etrap.flat": "2*us", "db.etrap.sample": "0.8"}},
        "cusp_kernel": {"function": "cusp_filter", "module": _M, "unit": "ADC",
                        "args": ["db.cusp.sigma/wf_blsub.period", "round(db.cusp.flat/wf_blsub.period)", "db.pz.tau/wf_blsub.period",
                                 "cusp_kernel(round(len(wf_blsub)-(33.6*us/wf_blsub.period)-(4.8*us/wf_blsub.period)), 'f')"],
                        "defaults": {"db.cusp.sigma": "20*us", "db.cusp.flat": "3*us", "db.pz.tau": "450*us"}},
        "wf_cusp": {"function": "fft_convolve_wf", "module": _M, "unit": "ADC",
                    "args": ["wf_blsub[:round(len(wf_blsub)-(33.6*us/wf_blsub.period))]", "cusp_kernel", "'v'",
                             "wf_cusp(round((4.8*us/wf_blsub.period)+1), 'f')"]},
        "cuspEmax": "numpy.amax(wf_cusp, 1, cuspEmax)",
        "cuspEftp": {"function": "fixed_time_pickoff", "module": _M, "args": ["wf_cusp", "db.cusp.sample", "'i'", "cuspEftp"], "unit": "ADC",
                     "defaults": {"db.cusp.sample": "50"}},
        "tp_100": {"function": "time_point_thresh", "module": _M, "args": ["wf_pz", "trapTmax", "tp_0_est", 1, "tp_100"], "unit": "ns"},
        "tp_99": {"function": "time_point_thresh", "module": _M, "args": ["wf_pz", "0.99*trapTmax", "tp_0_est", 1, "tp_99"], "unit": "ns"},
        "tp_90": {"function": "time_point_thresh", "module": _M, "args": ["wf_pz", "trapTmax*0.9", "tp_99", 0, "tp_90"], "unit": "ns"},
        "tp_50": {"function": "time_point_thresh", "module": _M, "args": ["wf_pz", "trapTmax*0.5", "tp_90", 0, "tp_50"], "unit": "ns"},
        "tp_10": {"function": "time_point_thresh", "module": _M, "args": ["wf_pz", "trapTmax*0.1", "tp_50", 0, "tp_10"], "unit": "ns"},
        "wf_trap2": {"function": "trap_norm", "module": _M, "args": ["wf_pz", "4*us", "96*ns", "wf_trap2"], "unit": "ADC"},
        "trapQftp": {"function": "fixed_time_pickoff", "module": _M, "args": ["wf_trap2", "tp_0_est + 8.096*us", "'l'", "trapQftp"], "unit": "ADC"},
        "QDrift": "trapQftp * 16",
        "dt_eff": {"function": "QDrift/trapTmax", "unit": "ns"},
        "wf_le": {"function": "windower", "module": _M, "args": ["wf_pz", "tp_0_est", "wf_le(301, 'f')"], "unit": "ADC"},
        "curr": {"function": "avg_current", "module": _M, "args": ["wf_le", 1, "curr(len(wf_le)-1, 'f')"], "unit": "ADC/sample"},
        "curr_up": {"function": "upsampler", "module": _M, "args": ["curr", "16", "curr_up(4784, 'f')"], "unit": "ADC/sample"},
        "curr_av": {"function": "moving_window_multi", "module": _M, "args": ["curr_up", "48", 3, 0, "curr_av"], "unit": "ADC/sample"},
        "aoe_t_min, tp_aoe_max, A_min, A_max": {"function": "min_max", "module": _M, "unit": ["ns", "ns", "ADC/sample", "ADC/sample"],
                                                "args": ["curr_av", "aoe_t_min", "tp_aoe_max", "A_min", "A_max"]},
        "tp_aoe_samp": {"function": "add", "module": "numpy", "args": ["tp_0_est", "tp_aoe_max/16", "tp_aoe_samp"], "unit": "ns"},
    },
}


def icpc_with_reference_values():
    """ICPC above with the parameter VALUES and the full output list of the reference's own Ge test configuration
    (tests/configs/icpc-dsp-config.json: 43 processors, 34 outputs): the fit windows, the unit-less pole-zero constant (27 460.5 *samples*, as
    that file's default stands), the 10 us / 3.008 us energy trapezoid picked at rise + 0.8 x 3 us, the zero-area cusp beside the cusp, and the
    whole rise-time ladder 100 / 99 / 95 / 90 / 80 / 50 / 20 / 10 / 1 %.  Written as edits of ICPC so that what differs is in one place;
    tests/test_recipe_language_cpu.py checks (where the reference checkout is mounted) that this recipe and the reference's file translate into
    the same device programs op for op."""
    import copy

    r = copy.deepcopy(ICPC)
    p = r["processors"]
    p["bl_mean , bl_std, bl_slope, bl_intercept"]["args"][0] = "wf_blsub[0:750]"
    p["wf_pz"]["defaults"] = {"db.pz.tau": "27460.5"}
    p["pz_mean , pz_std, pz_slope, pz_intercept"]["args"][0] = "wf_pz[1500:]"
    p["wf_etrap"]["defaults"] = {"db.etrap.rise": "10*us", "db.etrap.flat": "3.008*us"}
    p["trapEftp"]["defaults"] = {"db.etrap.rise": "10*us", "db.etrap.flat": "3*us", "db.etrap.sample": "0.8"}
    cusp = p["cusp_kernel"]
    p["zac_kernel"] = {"function": "zac_filter", "module": _M, "unit": "ADC",
                       "args": [a.replace("db.cusp.", "db.zac.").replace("cusp_kernel", "zac_kernel") for a in cusp["args"]],
                       "defaults": {"db.zac.sigma": "20*us", "db.zac.flat": "3*us", "db.pz.tau": "450*us"}}
    p["wf_zac"] = {"function": "fft_convolve_wf", "module": _M, "unit": "ADC",
                   "args": [p["wf_cusp"]["args"][0], "zac_kernel", "'v'", p["wf_cusp"]["args"][3].replace("wf_cusp", "wf_zac")]}
    p["zacEmax"] = {"function": "numpy.amax(wf_zac, 1, zacEmax)", "kwargs": {"signature": "(n),()->()", "types": ["fi->f"]}, "unit": "ADC"}
    p["zacEftp"] = {"function": "fixed_time_pickoff", "module": _M, "args": ["wf_zac", "db.zac.sample", "'i'", "zacEftp"], "unit": "ADC",
                    "defaults": {"db.zac.sample": "50"}}
    ladder = [("tp_95", "0.95", "tp_99"), ("tp_90", "0.9", "tp_95"), ("tp_80", "0.8", "tp_90"), ("tp_50", "0.5", "tp_80"), ("tp_20", "0.2", "tp_50"),
              ("tp_10", "0.1", "tp_20"), ("tp_01", "0.01", "tp_10")]
    for name, frac, start in ladder:
        p[name] = {"function": "time_point_thresh", "module": _M, "args": ["wf_pz", f"trapTmax*{frac}", start, 0, name], "unit": "ns"}
    r["outputs"] = ["tp_min", "tp_max", "wf_min", "wf_max", "bl_mean", "bl_std", "bl_slope", "bl_intercept", "pz_slope", "pz_std", "pz_mean", "trapTmax",
                    "tp_0_est", "tp_0_atrap", "tp_10", "tp_20", "tp_50", "tp_80", "tp_90", "tp_99", "tp_100", "tp_01", "tp_95", "A_max", "QDrift", "dt_eff",
                    "tp_aoe_max", "tp_aoe_samp", "trapEmax", "trapEftp", "cuspEmax", "zacEmax", "zacEftp", "cuspEftp"]
    return r


ICPC_REF = icpc_with_reference_values()
# what the oracle-side restatement of a whole pass needs to know about a recipe (tests/test_gpu_icpc_recipe.py::_expected)
ICPC_PARAMS = {"bl_window": 700, "tau_samples": 27460.0 / 16.0, "pz_from": 1600, "etrap": (500, 125), "pick_ns": (8000.0, 2000.0 * 0.8), "zac": False,
               "ladder": [("tp_90", 0.9, "tp_99"), ("tp_50", 0.5, "tp_90"), ("tp_10", 0.1, "tp_50")]}
ICPC_REF_PARAMS = {"bl_window": 750, "tau_samples": 27460.5, "pz_from": 1500, "etrap": (625, 188), "pick_ns": (10000.0, 3000.0 * 0.8), "zac": True,
                   "ladder": [("tp_95", 0.95, "tp_99"), ("tp_90", 0.9, "tp_95"), ("tp_80", 0.8, "tp_90"), ("tp_50", 0.5, "tp_80"), ("tp_20", 0.2, "tp_50"),
                              ("tp_10", 0.1, "tp_20"), ("tp_01", 0.01, "tp_10")]}
