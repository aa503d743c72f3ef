#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 csv output of tools/profile_bench.sh) into the tracked files
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json, profiles/<tag>_pmc_traffic.json and profiles/<tag>_summary.md."""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)

def newest(pattern):
    """gpurun merges every run's files into the same directory: keep the most recent one"""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


stats = newest(f"{src}/trace/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
bench = json.load(open(f"{src}/bench_default.json"))
kernel = bench["config"]["kernel"]
rows, wf_len = bench["config"]["rows_per_gpu"], bench["config"]["wf_len"]

pmc = {}
for d in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in newest(f"{src}/{d}/*/*_counter_collection.csv"):
        acc = {}
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            pmc[k] = sum(v) / len(v)
json.dump({"kernel": kernel, "rows": rows, "wf_len": wf_len, "per_launch": pmc}, open(f"profiles/{tag}_pmc.json", "w"), indent=1)

summary = [f"# {tag}: `python bench.py` on MI355X ({bench['config']['device'].strip()})", "",
           f"* value: **{bench['value'] / 1e6:.1f} M waveforms/s**, {bench['ms_per_step']:.3f} ms per step ({rows} x {wf_len} float32 rows)",
           f"* roofline: {bench['roofline']['achieved']:.0f} GB/s algorithmic of {bench['roofline']['peak']:.0f} GB/s = "
           f"**{100 * bench['roofline']['frac']:.1f} %**; kernel `{kernel}` avg {bench['roofline']['kernel_ms_avg']:.3f} ms (HIP events)",
           f"* parity guard inside the bench: max |GPU - oracle| / |oracle| = {bench['parity_max_rel_vs_oracle']:.2e} over 4096 rows"]
if bench.get("cpu_baseline"):
    c = bench["cpu_baseline"]
    summary.append(f"* cpu_baseline (oracle, dspeed-style 16-row blocks): {c['value']:.0f} wf/s on 1 core; "
                   f"{c['all_cores']['value']:.0f} wf/s on {c['all_cores']['cores']} cores")
if stats:
    for r in csv.DictReader(open(stats[0])):
        if kernel in r["Name"]:
            summary.append(f"* rocprofv3 --kernel-trace --stats: {r['Calls']} calls, average {float(r['AverageNs']) / 1e6:.3f} ms "
                           f"(min {float(r['MinNs']) / 1e6:.3f}, max {float(r['MaxNs']) / 1e6:.3f})")
trace = newest(f"{src}/trace/*/*_kernel_trace.csv")
if trace:
    # the same command's timed launches alone (bench.py warms up first: the first launches after start-up run slower)
    btr = json.load(open(f"{src}/bench_trace.json"))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(trace[0])) if kernel in r["Kernel_Name"]]
    k = btr["steps"]
    if len(d) >= k:
        summary.append(f"* ... of which the {k} timed launches (after {btr['warmup']} warm-up launches): average {sum(d[-k:]) / k:.3f} ms under rocprofv3; "
                       f"HIP events in that same run: {btr['roofline']['kernel_ms_avg']:.3f} ms")
if "FETCH_SIZE" in pmc:
    fetch = pmc["FETCH_SIZE"] * 1024 * 2  # KiB -> B, x2: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
    write = pmc.get("WRITE_SIZE", 0.0) * 1024
    alg = rows * (wf_len * 4 + 12)
    summary.append(f"* HBM traffic per launch (PMC, separate passes): FETCH_SIZE x2 = {fetch / 1e9:.3f} GB, WRITE_SIZE = {write / 1e6:.2f} MB; "
                   f"algorithmic {alg / 1e9:.3f} GB -> traffic / algorithmic = {(fetch + write) / alg:.3f}")
    geometry = {k: bench["config"][k] for k in ("lds_bytes_per_wave", "waves_per_block", "blocks")}
    json.dump({"kernel": kernel, "rows": rows, "wf_len": wf_len, "hbm_bytes_per_launch": fetch + write,
               "kernel_source_hash": bench["config"].get("kernel_source_hash"),  # bench.py quotes the figure only for these sources
               "geometry": geometry,                                             # ... and this launch geometry
               "source": f"profiles/{tag}_pmc.json: FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (rocprofv3 --pmc, separate passes)"},
              open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    # the shard of a multi-GPU run (1 250 000 rows per rank): its own PMC passes, its own record
    shard = {}
    for d in ("pmc_fetch_shard", "pmc_write_shard"):
        for f in newest(f"{src}/{d}/*/*_counter_collection.csv"):
            acc = {}
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k, v in acc.items():
                shard[k] = sum(v) / len(v)
    shard_bench = f"{src}/bench_shard.json"
    if "FETCH_SIZE" in shard and os.path.exists(shard_bench) and os.path.getsize(shard_bench):
        sb = json.load(open(shard_bench))
        srows = sb["config"]["rows_per_gpu"]
        sbytes = shard["FETCH_SIZE"] * 1024 * 2 + shard.get("WRITE_SIZE", 0.0) * 1024
        salg = srows * (wf_len * 4 + 12)
        summary.append(f"* the same for the {srows}-row shard of a multi-GPU run (`--rows {srows}`): {sbytes / 1e9:.3f} GB per launch, "
                       f"traffic / algorithmic = {sbytes / salg:.3f}")
        json.dump({"kernel": kernel, "rows": srows, "wf_len": wf_len, "hbm_bytes_per_launch": sbytes,
                   "kernel_source_hash": sb["config"].get("kernel_source_hash"),
                   "geometry": {k: sb["config"][k] for k in ("lds_bytes_per_wave", "waves_per_block", "blocks")},
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of `python bench.py --no-cpu --rows %d` (separate passes): FETCH_SIZE*1024*2 + WRITE_SIZE*1024" % srows},
                  open(f"profiles/{tag}_shard_pmc_traffic.json", "w"), indent=1)
n = 1e6 * rows / 1e6
per = lambda k: pmc.get(k, float("nan")) / rows  # noqa: E731
summary += ["", "Per waveform (= per wavefront-iteration), SQ counters in quad-cycles:", "",
            "| VALU insts | SALU insts | LDS insts | wave cycles | active VALU | active LDS | wait any | LDS bank-conflict cycles | LDS active cycles |",
            "|---|---|---|---|---|---|---|---|---|",
            "| " + " | ".join(f"{per(k):.0f}" for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
                                                        "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")) + " |"]
# secondary measurements committed beside it (tools/bench_configs.py, tools/icpc_breakdown.py), so that one file tells the round
other = f"profiles/{tag}_other_configs.jsonl"
if os.path.exists(other):
    summary += ["", f"Other BASELINE configs on device-resident batches (`tools/bench_configs.py`, `{other}`):", "",
                "| config | kernel | rows | waveforms/s | of its bound |", "|---|---|---|---|---|"]
    for line in open(other):
        o = json.loads(line)
        what = f"{o.get('achieved_TFLOPs', 0):.1f} TFLOP/s" if o["bound"] != "hbm" else f"{o['achieved_GBps']:.0f} GB/s"
        summary.append(f"| {o['config']} | `{o['kernel']}` | {o['rows']} | {o['waveforms_per_s'] / 1e6:.2f} M | {what} = {100 * o['frac']:.1f} % of {o['bound']} |")
icpc = f"profiles/{tag}_icpc_recipe.jsonl"
if os.path.exists(icpc):
    rows_ = [json.loads(line) for line in open(icpc)]
    summary += ["", f"The whole Ge recipe as one device program (`tools/icpc_breakdown.py`, `{icpc}`; 8192-sample int16 rows):", "",
                "| outputs requested | ops | LDS per waveform | waveforms/s |", "|---|---|---|---|"]
    for o in rows_:
        if "waveforms_per_s" in o:
            summary.append(f"| {o['outputs']} | {o['ops']} | {o['lds_bytes_per_waveform']} B | {o['waveforms_per_s'] / 1e3:.0f} k |")
    for o in rows_:
        if o.get("per_op_profile"):
            top = ", ".join(f"{k} {100 * v:.0f} %" for k, v in list(o["share_by_opcode"].items())[:6])
            summary.append("")
            summary.append(f"In-kernel op timers (`dsp_chain_profile`), {o['recipe']}: {o['cycles_per_waveform']} shader cycles per waveform per wavefront; {top}.")
rate = f"{src}/icpc_rate.json"
if os.path.exists(rate) and os.path.getsize(rate):
    r = json.load(open(rate))
    shutil.copy(rate, f"profiles/{tag}_icpc_rate.json")
    summary += ["", f"The recipe's launches per pass over {r['rows']} rows (`tools/icpc_rate.py`: {r['waveforms_per_s'] / 1e6:.2f} M waveforms/s, {r['ms_per_pass']:.1f} ms per pass; "
                f"`rocprofv3 --kernel-trace --stats` of the same command, `profiles/{tag}_icpc_kernel_stats.csv`):", "", "| kernel | launches | average ms |", "|---|---|---|"]
    for f in newest(f"{src}/icpc_trace/*/*_kernel_stats.csv"):
        shutil.copy(f, f"profiles/{tag}_icpc_kernel_stats.csv")
        for row in csv.DictReader(open(f)):
            if "dsp_" in row["Name"] and "synth" not in row["Name"]:
                nm = row["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                summary.append(f"| `{nm}` | {row['Calls']} | {float(row['AverageNs']) / 1e6:.3f} |")
small = f"{src}/icpc_rate_small.jsonl"
if os.path.exists(small) and os.path.getsize(small):
    shutil.copy(small, f"profiles/{tag}_icpc_rate_small.jsonl")
    rs = [json.loads(line) for line in open(small) if line.strip()]
    summary += ["", "The recipe on smaller device-resident batches (what a piece of a host-resident batch is): " +
                ", ".join(f"{r['rows']} rows {r['waveforms_per_s'] / 1e6:.2f} M/s" for r in rs) + "."]
e2e = f"{src}/e2e_recipe_rate.json"
if os.path.exists(e2e) and os.path.getsize(e2e):
    shutil.copy(e2e, f"profiles/{tag}_e2e_recipe_rate.json")
    ee = json.load(open(e2e))
    summary += ["", f"From host memory (NumPy rows in, NumPy columns out, `tools/e2e_recipe_rate.py`, {ee['rows']} rows): " +
                ", ".join(f"{k}: {v['waveforms_per_s'] / 1e6:.2f} M waveforms/s = {v['GB_per_s_over_pcie']} GB/s over PCIe" for k, v in ee["results"].items()) + "."]
fsr = f"{src}/fir_store_rate.json"
if os.path.exists(fsr) and os.path.getsize(fsr):
    shutil.copy(fsr, f"profiles/{tag}_fir_store_rate.json")
frr = f"{src}/fir_runs_rate.jsonl"
if os.path.exists(frr) and os.path.getsize(frr):
    shutil.copy(frr, f"profiles/{tag}_fir_runs_rate.jsonl")
    rs = [json.loads(line) for line in open(frr) if line.strip().startswith("{")]
    summary += ["", f"The run-length FIR (`dsp_fir_runs_kernel`, the 133-tap t0 filter on {rs[0]['rows']} float32 rows of {rs[0]['samples']}; `tools/fir_runs_rate.py`): " +
                "; ".join(f"{r['form']}: {r['ms']:.2f} ms = {r['algorithmic_GBps'] / 1e3:.2f} TB/s of the rows it must move" for r in rs) + "."]
    other = f"{src}/fir_runs_rate_mfma.jsonl"
    if os.path.exists(other) and os.path.getsize(other):
        shutil.copy(other, f"profiles/{tag}_fir_runs_rate_mfma.jsonl")
        ro = [json.loads(line) for line in open(other) if line.strip().startswith("{")]
        summary += ["The same programs with `DSPEED_HIP_NO_FIR_RUNS=1`: " + "; ".join(f"{r['form']}: {r['kernel']} {r['ms']:.2f} ms" for r in ro) + "."]
open(f"profiles/{tag}_summary.md", "w").write("\n".join(summary) + "\n")
shutil.copy(f"{src}/bench_default.json", f"profiles/{tag}_bench.json")
print("\n".join(summary))
