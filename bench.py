#!/usr/bin/env python3
"""Headline benchmark: waveforms/s of the 4096-sample fp32 Ge energy chain (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A *step* is one pass of the fused hot path (bl_subtract -> pole_zero -> trap_filter -> fixed_time_pickoff,
one launch of the waveform VM) over one synthetic batch that is already resident in HBM:

    N = 1   1 000 000 x 4096 float32 rows                      (BASELINE.json configs[1])
    N > 1   1 250 000 x 4096 rows per rank, batch-sharded,     (configs[3]: 10 M rows over 8 GPUs)
            one process per GPU, NO data-path collective

Rank 0 prints ONE JSON line.  `value` = rows processed by all ranks / wall time of the K timed steps
(max over ranks, bracketed by barrier + device sync).  `roofline` prices the single kernel of the chain with
the ALGORITHMIC bytes of SURVEY.md 8(d): 16 396 B per waveform (4096x4 read + baseline + pick-off time read
+ energy written) against the 8 TB/s HBM3E peak, from per-launch HIP-event durations on the launch stream.
`cpu_baseline` times the CPU oracle (C restatement of the reference's numba loops, run the way dspeed's
ProcessingChain runs them: 16-row blocks, one processor call per block) on a bounded sample of the same batch.

Multi-process rendezvous uses torch.distributed with the gloo backend only for the barrier and the
max-over-ranks of the timing: the path has no exchange step, so no RCCL traffic exists to measure.
"""
from __future__ import annotations

import argparse
import json
import os
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)  # (the first ~6 launches after start-up run 5-20 % slower: clocks, first touch)
    ap.add_argument("--rows", type=int, default=0, help="rows per rank (default: 1 000 000 at N=1, 1 250 000 at N>1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the single-thread CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg (profiling runs)")
    ap.add_argument("--wf-len", type=int, default=0, help="experiment only: other waveform length (trap geometry scaled)")
    return ap.parse_args()


def main():
    args = parse()
    global WF_LEN, RISE, FLAT, BYTES_PER_WF
    if args.wf_len:
        WF_LEN, RISE, FLAT = args.wf_len, 625 * args.wf_len // 4096, 188 * args.wf_len // 4096
        BYTES_PER_WF = WF_LEN * 4 + 12
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    n_gpus = world

    # the product path: load the HIP library first (fails loudly if it was not built)
    from dspeed_amd import _lib
    from dspeed_amd.chain import Chain, energy_chain_program
    from dspeed_amd.device import DeviceArray, Event, Stream, device_count, device_info, set_device, sync

    _lib.lib()
    ndev = device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: no HIP device visible")
    set_device(local_rank % ndev)

    dist = None
    if world > 1:
        import torch.distributed as dist  # gloo: barrier + max of a timing scalar only

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    rows = args.rows or (1_000_000 if n_gpus == 1 else 1_250_000)
    first_row = rank * rows  # disjoint shards of one global synthetic batch

    L = _lib.lib()
    stream = Stream()
    wf = DeviceArray((rows, WF_LEN), np.float32)
    bl = DeviceArray((rows,), np.float32)
    tp = DeviceArray((rows,), np.float32)
    out = DeviceArray((rows,), np.float32)
    _lib.check(L.dsp_synth_waveforms(wf.ptr, _lib.F32, rows, WF_LEN, WF_LEN, bl.ptr, tp.ptr, SEED, first_row, TAU, SIGMA,
                                     RISE + 0.8 * FLAT, 9000.0, 11000.0, 500.0, 15000.0, stream.ptr), what="synth")
    stream.sync()

    chain = Chain(energy_chain_program(WF_LEN, TAU, RISE, FLAT, "l"), "energy_chain")
    bufs = {"waveform": wf, "baseline": bl, "t_pick": tp, "trapEftp": out}

    for _ in range(args.warmup):
        chain.execute(bufs, rows, stream)
    chain.check(stream)

    starts = [Event() for _ in range(args.steps)]
    stops = [Event() for _ in range(args.steps)]
    barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        starts[k].record(stream)
        chain.execute(bufs, rows, stream)
        stops[k].record(stream)
    sync()
    barrier()
    t1 = time.perf_counter()
    chain.check(stream, row_offset=first_row)
    elapsed = t1 - t0
    if dist is not None:
        import torch

        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    kernel_ms = [starts[k].elapsed_ms(stops[k]) for k in range(args.steps)]
    avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3 if kernel_ms else float("nan")

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_rows = rows * n_gpus
    value = total_rows * args.steps / elapsed
    achieved = rows * BYTES_PER_WF / avg_kernel_s / 1e9
    geo = chain.geometry(rows)

    # ---- parity guard + CPU baseline on a bounded sample of the same batch (rank 0, N = 1 only for the baseline)
    sample_n = 4096
    wf_s = wf.view_rows(0, sample_n).to_numpy()
    bl_s = bl.view_rows(0, sample_n).to_numpy()
    tp_s = tp.view_rows(0, sample_n).to_numpy()
    got_s = out.view_rows(0, sample_n).to_numpy()
    import oracle  # the checker and the reported CPU baseline; never part of the timed GPU path

    want_s, rc = oracle.chain_energy(wf_s, bl_s, tp_s, TAU, RISE, FLAT, "l")
    ok = ~np.isnan(want_s)
    parity = float(np.max(np.abs(got_s[ok] - want_s[ok]) / np.abs(want_s[ok]))) if rc == 0 else float("nan")

    cpu = None
    if n_gpus == 1 and not args.no_cpu:
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

    # HBM traffic per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled as the gfx950 guide prescribes),
    # recorded under profiles/ by tools/profile_bench.sh; null if no recorded measurement matches this workload
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            t = json.load(f)
        if t.get("rows") == rows and t.get("wf_len") == WF_LEN and t.get("kernel") == chain.kernel_name:
            traffic, traffic_src = t["hbm_bytes_per_launch"], t.get("source")
    except (OSError, ValueError, KeyError):
        pass

    # measured read-only streaming ceiling of this box: the same batch read once by a trivial kernel, outside the timed region
    stream_gbps = None
    try:
        from dspeed_amd import _lib
        from dspeed_amd.device import Event

        nbytes = rows * WF_LEN * 4
        e0, e1 = Event(), Event()
        for it in range(4):  # first pass warms the code object
            if it == 1:
                e0.record(stream)
            _lib.check(_lib.lib().dsp_stream_read(wf.ptr, nbytes, out.ptr, stream.ptr if stream else None))
        e1.record(stream)
        sync()
        stream_gbps = 3 * nbytes / (e0.elapsed_ms(e1) * 1e-3) / 1e9
    except Exception as exc:  # a measurement extra: never fails the bench line
        print(f"[bench] stream-read measurement skipped: {exc}", file=sys.stderr)

    info = device_info(local_rank % ndev)
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
                   "kernel": chain.kernel_name, "lds_bytes_per_wave": geo["lds_bytes_per_wave"], "waves_per_block": geo["waves_per_block"],
                   "blocks": geo["blocks"]},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": rows * BYTES_PER_WF,
                     "bytes_per_waveform": BYTES_PER_WF, "kernel_ms_avg": 1e3 * avg_kernel_s,
                     "kernel_ms_min": float(np.min(kernel_ms)), "measured_stream_read_GBps": stream_gbps,
                     "frac_of_measured_stream_read": (achieved / stream_gbps) if stream_gbps else None},
        "cpu_baseline": cpu,
        "parity_max_rel_vs_oracle": parity,
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
