"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): the event axis is split with shard_rows, every rank processes
only its slice, outputs are concatenated -- and no collective touches the data path.  The per-shard compute here is the CPU
oracle (tests may use it); on the GPU box each rank runs the device chain on its slice instead (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    import oracle
    from dspeed_amd.processing_chain import shard_rows

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(1234)  # every rank derives the same global batch, then keeps only its rows
    wf = (10000 + 3000 * (np.arange(1024)[None, :] > 500) + 5 * rng.standard_normal((n_rows, 1024))).astype(np.float32)
    bl = np.full(n_rows, 10000, dtype=np.float32)
    tp = np.full(n_rows, 700.25, dtype=np.float32)
    lo, hi = shard_rows(n_rows, world, rank)
    mine, rc = oracle.chain_energy(wf[lo:hi], bl[lo:hi], tp[lo:hi], 1716.28, 64, 16, "l")
    assert rc == 0
    # timing protocol of bench.py: barrier, then max over ranks of a scalar -- the only communication there is
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t[0]) == world
    # result assembly = plain concatenation of disjoint slices (gathered here only to check it)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, mine))
    if rank == 0:
        full, rc = oracle.chain_energy(wf, bl, tp, 1716.28, 64, 16, "l")
        cat = np.concatenate([p[2] for p in sorted(parts, key=lambda p: p[0])])
        ret["ok"] = bool(np.array_equal(cat, full)) and parts[0][0] == 0 and sorted(parts)[-1][1] == n_rows
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_concatenates_to_the_single_rank_result():
    import torch.multiprocessing as mp

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, 101, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        assert ret.get("ok") is True
