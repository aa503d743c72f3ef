#!/usr/bin/env python3
"""Headline benchmark: waveforms/s of the 4096-sample fp32 Ge energy chain (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A *step* is one pass of the hot path (bl_subtract -> pole_zero -> trap_filter -> fixed_time_pickoff) over one synthetic batch that is
already resident in HBM.  The chain is built the way a dspeed user builds it -- build_processing_chain(<the recipe as a LEGEND configuration
writes it: module string "dspeed.processors", db.pz.tau with its default>, table of device-resident columns) -- and a pass is
ProcessingChain.execute: one launch of the energy kernel (the bench refuses to print a line if the recipe was given to another kernel).  The
same program handed to dsp_chain_create directly is timed beside it (roofline.raw_chain):

    N = 1   1 000 000 x 4096 float32 rows                      (BASELINE.json configs[1])
    N > 1   1 250 000 x 4096 rows per rank, batch-sharded,     (configs[3]: 10 M rows over 8 GPUs)
            one process per GPU, NO data-path collective

Rank 0 prints ONE JSON line.  `value` = rows processed by all ranks / wall time of the K timed steps
(max over ranks, bracketed by barrier + device sync).  `roofline` prices the single kernel of the chain with
the ALGORITHMIC bytes of SURVEY.md 8(d): 16 396 B per waveform (4096x4 read + baseline + pick-off time read
+ energy written) against the 8 TB/s HBM3E peak, from per-launch HIP-event durations on the launch stream.
`cpu_baseline` times the CPU oracle (C restatement of the reference's numba loops, run the way dspeed's
ProcessingChain runs them: 16-row blocks, one processor call per block) on a bounded sample of the same batch, on rank 0,
after the timed region, for every N.  `ge_recipe` (N = 1 only, after the timed region, not part of `value`) is the whole Ge recipe of
tests/recipes.py (ICPC: 27 outputs, eleven launches a pass) on 131 072 device-resident int16 rows of 8192 samples: ms per pass and waveforms/s.

Multi-process rendezvous uses torch.distributed with the gloo backend only for the barrier and the
max-over-ranks of the timing: the path has no exchange step, so no RCCL traffic exists to measure.

Launching.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` every process is one rank
(RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Started plainly as `python bench.py --gpus N` with N > 1 the
process becomes a launcher: BEFORE any HIP call (the library is not even loaded) it starts N fresh child processes of
this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays rank 0's JSON line and exits
non-zero if any child does.  A process that has touched the GPU is never re-executed.  `--dry-run` makes the workers
skip the device (rendezvous, barrier and the line's bookkeeping only): the CPU tests drive the launcher with it.

The bench refuses to print a number when the result of the timed launches deviates from the oracle by more than the
parity bar (exit code 3), when DSPEED_HIP_ABLATE / a non-default kernel variant is set without --allow-variants
(exit code 4), or when fewer GPUs are visible than ranks asked for (exit code 5).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WF_LEN = 4096
TAU = 1716.28          # 27 460.5 ns / 16 ns (icpc-dsp-config.json:63)
RISE, FLAT = 625, 188  # 10 us, 3.008 us at 16 ns
SIGMA = 5.0
SEED = 0xD5BEED
BYTES_PER_WF = WF_LEN * 4 + 4 + 4 + 4
HBM_PEAK_GBPS = 8000.0                 # MI355X_MICROARCH.md: 8.0 TB/s spec


PARITY_BAR = 1e-6                      # north_star: float32 filter outputs within 1e-6 relative of the reference arithmetic
# what the headline kernel is compiled from (the launch geometry, which dsp_host.cpp decides, is compared field by field instead)
KERNEL_SOURCES = ("dsp_energy.hip", "dsp_wave.h", "dsp_program.h")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)  # (3 ms each: a timed region of more than a second, long enough for the driver's own GPU-activity samples)
    ap.add_argument("--warmup", type=int, default=20)  # (the first ~6 launches after start-up run 5-20 % slower: clocks, first touch)
    ap.add_argument("--rows", type=int, default=0, help="rows per rank (default: 1 000 000 at N=1, 1 250 000 at N>1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the single-thread CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the legs beside the timed region: cpu_baseline and the Ge recipe (profiling runs)")
    ap.add_argument("--no-recipe", action="store_true", help="skip the whole-Ge-recipe leg beside the headline")
    ap.add_argument("--wf-len", type=int, default=0, help="experiment only: other waveform length (trap geometry scaled)")
    ap.add_argument("--dry-run", action="store_true", help="workers skip the device: launcher / rendezvous / bookkeeping only (CPU tests)")
    ap.add_argument("--allow-variants", action="store_true", help="A/B experiments: run although DSPEED_HIP_* kernel switches are set")
    ap.add_argument("--share-gpus", action="store_true", help="rehearsal only: let ranks share devices when fewer GPUs than ranks are visible")
    ap.add_argument("--fail-rank", type=int, default=-1, help="(dry run) this rank exits non-zero: tests the launcher's error path")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="launcher: seconds before the ranks are given up")
    return ap.parse_args(argv)


def energy_recipe(tau: float, rise: int, flat: int) -> dict:
    """BASELINE.json configs[1] / [3] in dspeed's recipe syntax: what an existing LEGEND configuration holds"""
    return {
        "outputs": ["trapEftp"],
        "processors": {
            "wf_blsub": "dspeed.processors.bl_subtract(waveform, baseline, wf_blsub)",
            "wf_pz": {"function": "pole_zero", "module": "dspeed.processors", "args": ["wf_blsub", "db.pz.tau", "wf_pz"],
                      "defaults": {"db.pz.tau": repr(float(tau))}},
            "wf_trap": {"function": "trap_filter", "module": "dspeed.processors", "args": ["wf_pz", str(int(rise)), str(int(flat)), "wf_trap"]},
            "trapEftp": {"function": "fixed_time_pickoff", "module": "dspeed.processors", "args": ["wf_trap", "t_pick", "'l'", "trapEftp"]},
        },
    }


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: a recorded PMC traffic figure is only quoted for the code it was measured on."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        path = os.path.join(ROOT, "dspeed_amd", "csrc", name)
        if os.path.exists(path):
            with open(path, "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def dspeed_env() -> dict:
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("DSPEED_HIP_")}


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher around it: start the N ranks.  Nothing in this process has loaded the HIP
    library or torch, so the children are ordinary fresh processes (never an exec of a process that initialised the GPU)."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.monotonic() + args.launch_timeout
    out0, failed = "", None
    try:
        pending = set(range(n))
        while pending and failed is None:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if r == 0:
                    out0 = procs[0].stdout.read()
                if rc != 0:
                    failed = (r, rc)
                    break
            if time.monotonic() > deadline:
                failed = (-1, 124)
            if pending and failed is None:
                time.sleep(0.05)
    finally:
        for p in procs:  # exactly the processes started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    if failed is not None:
        who = "the launch timeout" if failed[0] < 0 else f"rank {failed[0]} (exit code {failed[1]})"
        print(f"bench.py: {who} ended the {n}-rank run", file=sys.stderr)
        return failed[1] or 1
    line = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not line:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(line[-1])
    return 0


def main() -> int:
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    return worker(args)


def worker(args) -> int:
    global WF_LEN, RISE, FLAT, BYTES_PER_WF
    if args.wf_len:
        WF_LEN, RISE, FLAT = args.wf_len, 625 * args.wf_len // 4096, 188 * args.wf_len // 4096
        BYTES_PER_WF = WF_LEN * 4 + 12
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag must agree", file=sys.stderr)
        return 2
    n_gpus = world
    env_switches = dspeed_env()
    if env_switches and not args.allow_variants:
        print(f"bench.py: refusing to measure with kernel switches set: {env_switches} (--allow-variants for A/B experiments)", file=sys.stderr)
        return 4

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist  # gloo: barrier + max of a timing scalar only

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    def gather_floats(v: float) -> list:
        if dist is None:
            return [float(v)]
        t = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(t, torch.tensor([float(v)], dtype=torch.float64))
        return [float(x[0]) for x in t]

    rows = args.rows or (1_000_000 if n_gpus == 1 else 1_250_000)
    first_row = rank * rows  # disjoint shards of one global synthetic batch

    if args.dry_run:  # launcher / rendezvous / bookkeeping only: no library, no device, no number
        barrier()
        ranks_seen = sorted(int(r) for r in gather_floats(rank))
        firsts = [int(x) for x in gather_floats(first_row)]
        pids = [int(x) for x in gather_floats(os.getpid())]
        barrier()
        if rank == args.fail_rank:
            return 7
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": n_gpus, "ranks_seen": ranks_seen, "rows_per_gpu": rows, "first_rows": firsts,
                              "pids": pids, "launcher_pid": os.getppid(), "value": None}))
        if dist is not None:
            dist.destroy_process_group()
        return 0

    # the product path: load the HIP library first (fails loudly if it was not built)
    from dspeed_amd import _lib
    from dspeed_amd.chain import Chain, energy_chain_program
    from dspeed_amd.device import DeviceArray, Event, Stream, device_count, device_info, set_device, sync

    L = _lib.lib()
    ndev = device_count()
    if ndev < 1:
        print("bench.py needs a GPU: no HIP device visible", file=sys.stderr)
        return 5
    if ndev < world and not args.share_gpus:
        print(f"bench.py: {world} ranks but only {ndev} GPU(s) visible (one process per GPU; --share-gpus for a rehearsal)", file=sys.stderr)
        return 5
    device = local_rank % ndev
    set_device(device)

    synth_stream = Stream()
    wf = DeviceArray((rows, WF_LEN), np.float32)
    bl = DeviceArray((rows,), np.float32)
    tp = DeviceArray((rows,), np.float32)
    out = DeviceArray((rows,), np.float32)
    _lib.check(L.dsp_synth_waveforms(wf.ptr, _lib.F32, rows, WF_LEN, WF_LEN, bl.ptr, tp.ptr, SEED, first_row, TAU, SIGMA,
                                     RISE + 0.8 * FLAT, 9000.0, 11000.0, 500.0, 15000.0, synth_stream.ptr), what="synth")
    synth_stream.sync()

    # The boundary a dspeed user crosses: the recipe as the reference writes it (module string "dspeed.processors",
    # the pole-zero constant looked up as db.pz.tau with its default) through build_processing_chain (reference processing_chain.py:2363-2369);
    # the table's columns are device-resident, so a pass is the chain's launch and nothing else.
    from dspeed_amd import build_processing_chain

    recipe = energy_recipe(TAU, RISE, FLAT)
    tb_in = {"waveform": wf, "baseline": bl, "t_pick": tp}
    chain, _mask, tb_out = build_processing_chain(recipe, tb_in, db_dict={})
    tb_out["trapEftp"] = out
    chain.link(tb_in, tb_out)
    stream = chain.stream
    kernel_name = dict(chain.kernels())["program"]
    if WF_LEN in (1024, 2048, 4096, 8192) and kernel_name != "dsp_energy_rr_kernel":
        print(f"bench.py: the recipe was given to {kernel_name}, not to the energy kernel: {chain.kernel_notes()}", file=sys.stderr)
        return 6

    for _ in range(args.warmup):
        chain.execute(0, rows, wait=False)
    chain.wait()

    starts = [Event() for _ in range(args.steps)]
    stops = [Event() for _ in range(args.steps)]
    barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        starts[k].record(stream)
        chain.execute(0, rows, wait=False)
        stops[k].record(stream)
    sync()
    barrier()
    t1 = time.perf_counter()
    try:
        chain.wait()
    except Exception as exc:  # a DSPFatal of the timed passes: rows of this rank's shard
        print(f"bench.py: rank {rank} (rows from {first_row}): {exc}", file=sys.stderr)
        return 3
    elapsed = max(gather_floats(t1 - t0))
    kernel_ms = [starts[k].elapsed_ms(stops[k]) for k in range(args.steps)]
    avg_kernel_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    rank_kernel_ms = gather_floats(avg_kernel_ms)
    ranks_seen = sorted(int(r) for r in gather_floats(rank))
    devices_seen = [int(d) for d in gather_floats(device)]

    # ---- parity guard on the output of the TIMED launches (every rank checks its own shard's first rows against the oracle)
    sample_n = 4096 if rank == 0 else 512
    wf_s = wf.view_rows(0, sample_n).to_numpy()
    bl_s = bl.view_rows(0, sample_n).to_numpy()
    tp_s = tp.view_rows(0, sample_n).to_numpy()
    got_s = out.view_rows(0, sample_n).to_numpy()
    import oracle  # the checker and the reported CPU baseline; never part of the timed GPU path

    want_s, rc = oracle.chain_energy(wf_s, bl_s, tp_s, TAU, RISE, FLAT, "l")
    ok = ~np.isnan(want_s)
    parity = float(np.max(np.abs(got_s[ok] - want_s[ok]) / np.abs(want_s[ok]))) if rc == 0 and ok.any() else float("nan")
    if not np.array_equal(np.isnan(got_s), np.isnan(want_s)):
        parity = float("nan")
    parities = gather_floats(parity)
    parity = float("nan") if any(p != p for p in parities) else max(parities)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return 0 if parity <= PARITY_BAR else 3

    total_rows = rows * n_gpus
    value = total_rows * args.steps / elapsed
    avg_kernel_s = float(np.mean(rank_kernel_ms)) * 1e-3
    achieved = rows * BYTES_PER_WF / avg_kernel_s / 1e9
    geo = chain.geometry(rows)

    # the same program handed to dsp_chain_create directly (no recipe, no ProcessingChain): the number the boundary must not cost
    raw = None
    try:
        raw_chain = Chain(energy_chain_program(WF_LEN, TAU, RISE, FLAT, "l"), "energy_chain")
        raw_bufs = {"waveform": wf, "baseline": bl, "t_pick": tp, "trapEftp": out}
        r0, r1 = Event(), Event()
        n_raw = max(5, min(args.steps, 40))
        for k in range(3 + n_raw):
            if k == 3:
                r0.record(stream)
            raw_chain.execute(raw_bufs, rows, stream)
        r1.record(stream)
        sync()
        raw_chain.check(stream, row_offset=first_row)
        raw_ms = r0.elapsed_ms(r1) / n_raw
        raw = {"entry": "dsp_chain_create (Chain(energy_chain_program))", "kernel": raw_chain.kernel_name, "kernel_ms_avg": raw_ms,
               "frac": rows * BYTES_PER_WF / (raw_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launches": n_raw}
    except Exception as exc:  # a measurement extra: never fails the bench line
        print(f"[bench] raw-chain measurement skipped: {exc}", file=sys.stderr)

    # beside the headline, SURVEY 8(f)'s widest row: the whole Ge recipe (tests/recipes.py ICPC, 27 outputs, eleven launches a pass) on
    # 131 072 device-resident int16 rows of 8192 samples -- the figure DESIGN section 5 quotes, measured here so that the driver's own run holds it.
    # N=1, rank 0, after the timed region; an extra that never fails the bench line.
    ge = None
    if n_gpus == 1 and not (args.no_recipe or args.no_cpu or args.wf_len or args.rows):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import recipes
            from dspeed_amd.processing_chain import WaveformInput

            n_ge, len_ge = 131072, 8192
            wf_g = DeviceArray((n_ge, len_ge), np.int16)
            bl_g, tp_g = DeviceArray((n_ge,), np.float32), DeviceArray((n_ge,), np.float32)
            _lib.check(L.dsp_synth_waveforms(wf_g.ptr, _lib.I16, n_ge, len_ge, len_ge, bl_g.ptr, tp_g.ptr, SEED, 0, TAU, SIGMA, 625 + 0.8 * 188,
                                             -3000.0, 3000.0, 500.0, 15000.0, synth_stream.ptr), what="synth")
            synth_stream.sync()
            tb_g = {"waveform": WaveformInput(wf_g, 16.0, 48000.0), "baseline": bl_g}
            chain_g, _, _ = build_processing_chain(recipes.ICPC, tb_g)
            chain_g.link(tb_g, {k: DeviceArray((n_ge,), np.float32) for k in recipes.ICPC["outputs"]})
            g0, g1 = Event(), Event()
            n_pass = 20
            for k in range(5 + n_pass):
                if k == 5:
                    g0.record(chain_g.stream)
                chain_g.execute()
            g1.record(chain_g.stream)
            sync()
            ms_g = g0.elapsed_ms(g1) / n_pass
            ge = {"workload": f"tests/recipes.py ICPC ({len(recipes.ICPC['outputs'])} outputs) on {n_ge} x {len_ge} int16 rows, device-resident, through build_processing_chain",
                  "ms_per_pass": ms_g, "waveforms_per_s": n_ge / (ms_g * 1e-3), "passes": n_pass, "kernels": [k for _what, k in chain_g.kernels()]}
            del chain_g, wf_g
        except Exception as exc:
            print(f"[bench] Ge-recipe measurement skipped: {exc}", file=sys.stderr)

    cpu = None
    if not args.no_cpu:  # rank 0, for every N: the other ranks have left, the timed region is over
        t = time.perf_counter()
        oracle.chain_energy(wf_s[:1024], bl_s[:1024], tp_s[:1024], TAU, RISE, FLAT, "l", block_width=16, n_threads=1)
        per_wf = (time.perf_counter() - t) / 1024
        n_cpu = int(min(rows, max(2048, args.cpu_seconds / per_wf)))
        wf_c = wf.view_rows(0, n_cpu).to_numpy()
        bl_c, tp_c = bl.view_rows(0, n_cpu).to_numpy(), tp.view_rows(0, n_cpu).to_numpy()
        t = time.perf_counter()
        oracle.chain_energy(wf_c, bl_c, tp_c, TAU, RISE, FLAT, "l", block_width=16, n_threads=1)
        dt1 = time.perf_counter() - t
        # the GPU box exposes every host CPU, but one GPU's share is 16: use no more threads than that
        cores = max(1, min(oracle.max_threads(), len(os.sched_getaffinity(0)), 16))
        t = time.perf_counter()
        oracle.chain_energy(wf_c, bl_c, tp_c, TAU, RISE, FLAT, "l", block_width=16, n_threads=cores)
        dtn = time.perf_counter() - t
        cpu = {"value": n_cpu / dt1, "unit": "waveforms/s", "cores": 1, "kind": "port",
               "sample": f"first {n_cpu} rows of the same synthetic batch, 16-row blocks, one processor call per block (dspeed defaults), {dt1:.1f} s",
               "all_cores": {"value": n_cpu / dtn, "cores": cores, "seconds": round(dtn, 2)}}

    # HBM traffic per launch: RECORDED from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled as the gfx950 guide
    # prescribes; tools/profile_bench.sh + tools/summarise_profile.py write the file).  Quoted only when the record is for this
    # kernel, these rows and exactly these kernel sources (hash); otherwise null -- counters cannot be read from inside the run.
    traffic, traffic_src = None, None
    src_hash = kernel_source_hash()
    prof_dir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(prof_dir) if os.path.isdir(prof_dir) else [], reverse=True):
        if not name.endswith("_pmc_traffic.json"):
            continue
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            if (t.get("rows") == rows and t.get("wf_len") == WF_LEN and t.get("kernel") == kernel_name
                    and t.get("kernel_source_hash") == src_hash and t.get("geometry") == geo):
                traffic, traffic_src = t["hbm_bytes_per_launch"], f"recorded: profiles/{name} ({t.get('source')})"
                break
        except (OSError, ValueError, KeyError):
            pass

    # measured read-only streaming ceiling of this box: the same batch read once by a trivial kernel, outside the timed region
    stream_gbps = None
    try:
        nbytes = rows * WF_LEN * 4
        e0, e1 = Event(), Event()
        for it in range(4):  # first pass warms the code object
            if it == 1:
                e0.record(stream)
            _lib.check(L.dsp_stream_read(wf.ptr, nbytes, out.ptr, stream.ptr if stream else None))
        e1.record(stream)
        sync()
        stream_gbps = 3 * nbytes / (e0.elapsed_ms(e1) * 1e-3) / 1e9
    except Exception as exc:  # a measurement extra: never fails the bench line
        print(f"[bench] stream-read measurement skipped: {exc}", file=sys.stderr)

    info = device_info(device)
    line = {
        "metric": "waveforms/sec, 4096-sample fp32 trap-energy chain",
        "value": value,
        "unit": "waveforms/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (f"{rows} x {WF_LEN} float32 rows per GPU, fused bl_subtract->pole_zero(tau={TAU})->trap_filter({RISE},{FLAT})"
                                "->fixed_time_pickoff('l'); " + ("BASELINE configs[1]" if n_gpus == 1 else "BASELINE configs[3] shard")),
                   "rows_per_gpu": rows, "wf_len": WF_LEN, "sharding": "event axis, no collectives", "device": info["name"],
                   "entry": "build_processing_chain", "kernel": kernel_name, "lds_bytes_per_wave": geo["lds_bytes_per_wave"], "waves_per_block": geo["waves_per_block"],
                   "blocks": geo["blocks"], "ranks_seen": ranks_seen, "devices_seen": devices_seen, "env": env_switches,
                   "kernel_source_hash": src_hash},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": rows * BYTES_PER_WF,
                     "bytes_per_waveform": BYTES_PER_WF, "kernel_ms_avg": 1e3 * avg_kernel_s,
                     "kernel_ms_min": float(np.min(kernel_ms)), "kernel_ms_avg_per_rank_min": float(np.min(rank_kernel_ms)),
                     "kernel_ms_avg_per_rank_max": float(np.max(rank_kernel_ms)), "measured_stream_read_GBps": stream_gbps,
                     "frac_of_measured_stream_read": (achieved / stream_gbps) if stream_gbps else None, "raw_chain": raw},
        "cpu_baseline": cpu,
        "ge_recipe": ge,
        "parity_max_rel_vs_oracle": parity,
        "parity_bar": PARITY_BAR,
    }
    if dist is not None:
        dist.destroy_process_group()
    if not parity <= PARITY_BAR:
        print(f"bench.py: the timed launches' output deviates from the oracle by {parity} (bar {PARITY_BAR}): no number", file=sys.stderr)
        return 3
    print(json.dumps(line))
    return 0


if __name__ == "__main__":
    sys.exit(main())
