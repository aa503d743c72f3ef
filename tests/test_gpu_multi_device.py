"""Several chain handles at once -- the multi-GPU path of the product API (one chain handle + stream per device, rows
sharded, no collective) rehearsed on the one GPU of the test box: ``devices=[0, 0]`` gives two worker threads, two handles, two sets of
streams and staging buffers on device 0, and every result must equal the single-handle run bit for bit.  Plus what went in with it:
friend inputs of a recipe, binding offsets on the specialised kernels, ``proc_chain(tb_in, tb_out, begin, end)`` on LGDO tables,
WaveformTables of variable-length waveforms."""
import threading

import numpy as np
import pytest

import oracle
import recipes

pytestmark = pytest.mark.gpu


def _table(rng, n, wf_len=4096, dtype=np.float32):
    from dspeed_amd.processing_chain import WaveformInput

    i = np.arange(wf_len)[None, :]
    start = np.floor(rng.uniform(0.45, 0.55, (n, 1)) * wf_len)
    bl = rng.uniform(9000, 11000, (n, 1))
    x = bl + rng.uniform(500, 15000, (n, 1)) * np.exp(-(i - start) / 1716.28) * (i >= start) + 5 * rng.standard_normal((n, wf_len))
    x = np.rint(x).astype(dtype) if np.dtype(dtype).kind in "iu" else x.astype(dtype)
    return {"waveform": WaveformInput(x, 16.0, 0.0), "baseline": bl[:, 0].astype(np.float32),
            "t_pick": (start[:, 0] + 775.4).astype(np.float32)}, x


def test_two_handles_on_two_threads_share_a_device():
    """include/dspeed_hip.h: "independent handles may be used from different threads/devices" -- two chains of the same program, each
    executed and checked from its own thread on its own stream and buffers, at the same time, against one chain run alone"""
    from dspeed_amd.chain import Chain, energy_chain_program
    from dspeed_amd.device import DeviceArray, Stream, set_device

    rng = np.random.default_rng(21)
    tb, x = _table(rng, 6000)
    prog = energy_chain_program(4096, 1716.28, 625, 188, "l")

    def run(rows, out, rounds):
        set_device(0)
        ch, st = Chain(prog, "thread chain"), Stream()
        bufs = {"waveform": DeviceArray.from_numpy(x[rows]), "baseline": DeviceArray.from_numpy(tb["baseline"][rows]),
                "t_pick": DeviceArray.from_numpy(tb["t_pick"][rows]), "trapEftp": DeviceArray((len(x[rows]),), np.float32)}
        for _ in range(rounds):
            ch.execute(bufs, len(x[rows]), st)
            ch.check(st)
        out.append(bufs["trapEftp"].to_numpy())

    alone = []
    run(slice(0, 6000), alone, 1)
    halves = [[], []]
    threads = [threading.Thread(target=run, args=(slice(3000 * k, 3000 * (k + 1)), halves[k], 5)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert len(halves[0]) == 1 and len(halves[1]) == 1
    assert np.array_equal(np.concatenate([halves[0][0], halves[1][0]]), alone[0])
    want, rc = oracle.chain_energy(x, tb["baseline"], tb["t_pick"], 1716.28, 625, 188, "l")
    assert rc == 0 and np.max(np.abs(alone[0] - want) / np.abs(want)) <= 1e-6


def test_build_dsp_over_two_handles_equals_one(monkeypatch):
    from lgdo_standins import Array, LH5Iterator, Table, WaveformTable

    from dspeed_amd import build_dsp
    from dspeed_amd.errors import DSPFatal

    rng = np.random.default_rng(22)
    t1, x1 = _table(rng, 1001)
    t2, _ = _table(rng, 37)
    raw = {"raw/ch1": t1, "raw/ch2": t2}
    one = build_dsp(raw, dsp_config=recipes.C2, buffer_len=200)
    two = build_dsp(raw, dsp_config=recipes.C2, buffer_len=200, devices=[0, 0])
    three = build_dsp(raw, dsp_config=recipes.C2, devices=[0, 0, 0])
    for name in ("dsp/ch1", "dsp/ch2"):
        assert list(two[name]) == list(one[name])
        for k in one[name]:
            assert np.array_equal(two[name][k], one[name][k]) and np.array_equal(three[name][k], one[name][k]), (name, k)
    # the environment variable is the same switch
    monkeypatch.setenv("DSPEED_HIP_DEVICES", "0,0")
    env = build_dsp(t1, dsp_config=recipes.C2, i_start=7, n_entries=500)
    monkeypatch.delenv("DSPEED_HIP_DEVICES")
    assert np.array_equal(env["trapEftp"], one["dsp/ch1"]["trapEftp"][7:507])
    # a DSPFatal of the second shard carries rows of the table
    bad = {"outputs": ["tp"], "processors": {"tp": "dspeed.processors.time_point_thresh(waveform, 9500, t_pick, 0, tp)"}}
    t3 = dict(t1)
    t3["t_pick"] = np.floor(t1["t_pick"])
    t3["t_pick"][900] += 0.5
    with pytest.raises(DSPFatal, match="starting index must be an integer") as e:
        build_dsp(t3, dsp_config=bad, i_start=100, devices=[0, 0])
    lo, hi = (int(v) for v in str(e.value.wf_range).split("-"))
    assert lo <= 900 < hi and lo >= 100 + (1001 - 100) // 2
    # chunks of an iterator dealt to two handles arrive in file order; an LGDO table in memory is split like an array table
    n = 1000
    wf = np.rint(x1[:n]).astype(np.uint16)
    lg = Table(waveform=WaveformTable(wf, 16.0, np.zeros(n)), baseline=Array(t1["baseline"][:n]), t_pick=Array(t1["t_pick"][:n]))
    ref = np.asarray(build_dsp(lg, dsp_config=recipes.C2)["trapEftp"])
    it = LH5Iterator(lg, buffer_len=130)
    got = np.asarray(build_dsp(it, dsp_config=recipes.C2, devices=[0, 0])["trapEftp"])
    assert got.shape == (n,) and np.array_equal(got, ref)
    assert np.array_equal(np.asarray(build_dsp(lg, dsp_config=recipes.C2, devices=[0, 0])["trapEftp"]), ref)
    tp_bad = np.floor(t1["t_pick"][:n])
    tp_bad[640] += 0.5
    lg_bad = Table(lg)
    lg_bad["t_pick"] = Array(tp_bad)
    rec_i = {"outputs": ["e"], "processors": dict(recipes.C2["processors"])}
    rec_i["processors"]["e"] = {"function": "fixed_time_pickoff", "module": "dspeed.processors", "args": ["wf_trap", "t_pick", "'i'", "e"]}
    with pytest.raises(DSPFatal, match="integer t_in") as ei:
        build_dsp(LH5Iterator(lg_bad, buffer_len=130), dsp_config=rec_i, devices=[0, 0])
    assert ei.value.wf_range == "520-650"


def test_friend_inputs(tmp_path):
    """a recipe's ``inputs``: columns of another file / group joined to the table under a prefix (reference build_dsp.py:268-330)"""
    from lgdo_standins import Array, LH5Iterator, Table, WaveformTable

    from dspeed_amd import build_dsp, lgdo_io

    rng = np.random.default_rng(23)
    t1, x1 = _table(rng, 300)
    main = {k: v for k, v in t1.items() if k != "t_pick"}
    aux = str(tmp_path / "aux.npz")
    np.savez(aux, **{"hit/ch1/t_pick": t1["t_pick"], "hit/ch1/other": np.zeros(300)})
    rec = {"outputs": list(recipes.C2["outputs"]), "inputs": {"file": "db.aux.file", "group": "db.aux.group", "prefix": "aux_"},
           "processors": {k: (v if not isinstance(v, dict) else {**v, "args": ["aux_t_pick" if a == "t_pick" else a for a in v["args"]]})
                          for k, v in recipes.C2["processors"].items()}}
    assert any("aux_t_pick" in str(v) for v in rec["processors"].values())
    want = build_dsp(t1, dsp_config=recipes.C2)["trapEftp"]
    got = build_dsp({"raw/ch1": main}, dsp_config=rec, database={"ch1": {"aux": {"file": aux, "group": "hit/ch1"}}})
    assert np.array_equal(got["dsp/ch1"]["trapEftp"], want)
    # LGDO side: an iterator gets the friend through add_friend, a table in memory through join; the opener hook stands in for lgdo.lh5
    n = 300
    wf = x1.astype(np.float32)
    friend_tb = Table(t_pick=Array(t1["t_pick"]))
    calls = []

    def opener(file, group, iterator=False, n_rows=None, **sel):
        if file != "hits.lh5":
            return None
        calls.append((group, iterator, n_rows, sel.get("buffer_len")))
        return LH5Iterator(friend_tb, buffer_len=sel["buffer_len"]) if iterator else Table(t_pick=Array(t1["t_pick"][:n_rows]))

    lgdo_io.FRIEND_OPENERS.append(opener)
    try:
        rec2 = dict(rec, inputs=[{"file": "hits.lh5", "group": "ch1/hit", "prefix": "aux_"}])
        lg = Table(waveform=WaveformTable(wf, 16.0, np.zeros(n)), baseline=Array(t1["baseline"]))
        got_it = np.asarray(build_dsp(LH5Iterator(lg, buffer_len=128), dsp_config=rec2)["trapEftp"])
        got_tb = np.asarray(build_dsp(Table(lg), dsp_config=rec2)["trapEftp"])
    finally:
        lgdo_io.FRIEND_OPENERS.remove(opener)
    assert np.array_equal(got_it, want) and np.array_equal(got_tb, want)
    assert calls == [("ch1/hit", True, None, 128), ("ch1/hit", False, 300, None)]


def test_binding_offsets_reach_the_specialised_kernels():
    """every scalar / output binding of a program may start at an element offset inside its buffer; the register-resident energy kernel,
    the lane-per-waveform kernel and the VM must read and write the same elements"""
    from dspeed_amd import _lib
    from dspeed_amd.chain import Chain, Program, Scalar
    from dspeed_amd.device import DeviceArray

    rng = np.random.default_rng(24)
    n, L = 512, 4096
    tb, x = _table(rng, n)
    # interleaved scalar columns: baseline at element 1 and t_pick at element 2 of rows of 4 floats; the energy goes to element 3 of rows of 5
    side = np.zeros((n, 4), dtype=np.float32)
    side[:, 1], side[:, 2] = tb["baseline"], tb["t_pick"]
    results = {}
    for fused in (1, 0):
        p = Program()
        s, r = p.add_slot(L), p.add_sregs(1)
        io_wf = p.add_io("waveform", _lib.IO_WF_IN, np.float32, L)
        io_bl = p.add_io("baseline", _lib.IO_SCALAR_IN, np.float32, 1, 1, 4)
        io_tp = p.add_io("t_pick", _lib.IO_SCALAR_IN, np.float32, 1, 2, 4)
        io_e = p.add_io("e", _lib.IO_SCALAR_OUT, np.float32, 1, 3, 5)
        p.add_op(_lib.OP_LOAD, dst=s, io=io_wf)
        p.add_op(_lib.OP_BL_SUBTRACT, dst=s, src=s, sp=(Scalar.input(io_bl),))
        p.add_op(_lib.OP_POLE_ZERO, dst=s, src=s, sp=(Scalar.const(1716.28),))
        p.add_op(_lib.OP_TRAP_PICKOFF, dst=r, src=s, io=ord("l"), ip=(625, 188, 0, _lib.OP_TRAP_FILTER), sp=(Scalar.input(io_tp),))
        p.add_op(_lib.OP_STORE_SCALAR, io=io_e, ip=(r,))
        ch = Chain(p, "offsets")
        assert ch.set_fused(fused) == bool(fused)
        d_side, d_out = DeviceArray.from_numpy(side), DeviceArray.zeros((n, 5), np.float32)
        ch.execute({"waveform": DeviceArray.from_numpy(x), "baseline": d_side, "t_pick": d_side, "e": d_out}, n)
        ch.check()
        results[fused] = d_out.to_numpy()
    want, _ = oracle.chain_energy(x, tb["baseline"], tb["t_pick"], 1716.28, 625, 188, "l")
    for fused, got in results.items():
        assert np.all(got[:, [0, 1, 2, 4]] == 0), fused
        assert np.max(np.abs(got[:, 3] - want) / np.abs(want)) <= 1e-6, fused
    # the lane-per-waveform kernel: threshold column and the outputs at offsets
    L2 = 1024
    wf16 = np.rint(x[:, 1500:1500 + L2] - 10000).astype(np.int16)
    thr = np.full((n, 3), np.nan, dtype=np.float32)
    thr[:, 2] = 40.0
    outs = {}
    for fused in (1, 0):
        p = Program()
        s = p.add_slot(L2)
        r = p.add_sregs(5)
        io_wf = p.add_io("waveform", _lib.IO_WF_IN, np.int16, L2)
        io_thr = p.add_io("thr", _lib.IO_SCALAR_IN, np.float32, 1, 2, 3)
        io_o = [p.add_io(f"o{k}", _lib.IO_SCALAR_OUT, np.float32, 1, k + 1, 8) for k in range(5)]
        p.add_op(_lib.OP_LOAD, dst=s, io=io_wf)
        p.add_op(_lib.OP_POLE_ZERO, dst=s, src=s, sp=(Scalar.const(1716.28),))
        p.add_op(_lib.OP_TRAP_REDUCE, dst=r, src=s, io=r + 4, ip=(8, 4, 125, _lib.OP_ASYM_TRAP),
                 sp=(Scalar.input(io_thr), Scalar.reg(r + 1), Scalar.const(0.0)))
        for k in range(5):
            p.add_op(_lib.OP_STORE_SCALAR, io=io_o[k], ip=(r + k,))
        ch = Chain(p, "rows offsets")
        assert ch.set_fused(fused) == bool(fused)
        if fused:
            assert "rows" in ch.kernel_name
        d_o = DeviceArray.zeros((n, 8), np.float32)
        bufs = {"waveform": DeviceArray.from_numpy(wf16), "thr": DeviceArray.from_numpy(thr)}
        bufs.update({f"o{k}": d_o for k in range(5)})
        ch.execute(bufs, n)
        ch.check()
        outs[fused] = d_o.to_numpy()
    # (the rows kernel walks the reference's recurrence, the VM replays its rounding: values agree to the filter bar, indices nearly always)
    for got in outs.values():
        assert np.all(got[:, [0, 6, 7]] == 0) and np.all(got[:, 4] > 0) and np.all(got[:, 2] > got[:, 1] - L2)
    peak = np.abs(outs[0][:, 4:5])  # (the filter bar is relative to the row's peak)
    assert np.all(np.abs(outs[1][:, 3:5] - outs[0][:, 3:5]) <= 1e-6 * peak) and np.mean(outs[1][:, [1, 2, 5]] == outs[0][:, [1, 2, 5]]) > 0.98


def test_proc_chain_call_with_a_row_range_writes_only_those_rows():
    from lgdo_standins import Array, Table, WaveformTable

    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(25)
    t1, x1 = _table(rng, 64)
    lg = Table(waveform=WaveformTable(x1, 16.0, np.zeros(64)), baseline=Array(t1["baseline"]), t_pick=Array(t1["t_pick"]))
    chain, _mask, _ = build_processing_chain(recipes.C2, lg)
    full = Table(trapEftp=Array(np.zeros(0, np.float32)))
    chain(lg, full)
    part = Table(trapEftp=Array(np.full(64, -1.0, np.float32)))
    chain(lg, part, 10, 30)
    assert np.array_equal(part["trapEftp"].nda[10:30], full["trapEftp"].nda[10:30])
    assert np.all(part["trapEftp"].nda[:10] == -1) and np.all(part["trapEftp"].nda[30:] == -1) and len(part["trapEftp"].nda) == 64


def test_waveform_table_of_variable_length_waveforms():
    """WaveformTable.values as a VectorOfVectors (reference processing_chain.py:2327-2328): padded rows on the table's grid + len(<name>)"""
    from lgdo_standins import Table, VectorOfVectors, WaveformTable

    from dspeed_amd import lgdo_io
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(26)
    n = 40
    lens = rng.integers(20, 100, n)
    rows = [rng.normal(0, 5, m).astype(np.float32) + 100 for m in lens]
    vov = VectorOfVectors(np.concatenate(rows), np.cumsum(lens))
    tb = Table(waveform=WaveformTable(vov, 16.0, np.arange(n) * 32.0))
    cols = lgdo_io.table_columns(tb)
    assert cols["waveform"].values.shape == (n, 2 * lens.max()) and np.array_equal(cols["len(waveform)"], lens) and cols["waveform"].dt == 16.0
    rec = {"outputs": ["first", "n"], "processors": {"first": "waveform[0]", "n": "len(waveform)"}}
    chain, _, out = build_processing_chain(rec, tb)
    chain.execute()
    assert np.array_equal(out["first"], np.array([r[0] for r in rows])) and np.array_equal(out["n"], lens)


def test_two_different_gpus_when_the_box_has_them():
    """The first box with more than one GPU exercises what ``devices=[0, 0]`` cannot: hipSetDevice(1) on a worker thread, chain handles,
    streams, staging and stage buffers of a second device, ``dsp_chain_execute`` putting the caller's device back.  Skipped on the one-GPU
    boxes of the test pool."""
    from lgdo_standins import Array, LH5Iterator, Table, WaveformTable

    from dspeed_amd import build_dsp, build_processing_chain
    from dspeed_amd.device import device_count, set_device

    if device_count() < 2:
        pytest.skip("one GPU visible: devices=[0, 1] needs two")
    rng = np.random.default_rng(23)
    t1, x1 = _table(rng, 1203)
    one = build_dsp({"raw/ch1": t1}, dsp_config=recipes.C2, buffer_len=200)
    two = build_dsp({"raw/ch1": t1}, dsp_config=recipes.C2, buffer_len=200, devices=[0, 1])
    rev = build_dsp({"raw/ch1": t1}, dsp_config=recipes.C2, devices=[1, 0])
    for k in one["dsp/ch1"]:
        assert np.array_equal(two["dsp/ch1"][k], one["dsp/ch1"][k]) and np.array_equal(rev["dsp/ch1"][k], one["dsp/ch1"][k]), k
    # a chain bound to device 1, driven from a thread whose current device is 0: results equal, and the caller's device comes back
    chain, _, out = build_processing_chain(recipes.C2, t1, device=1)
    set_device(0)
    chain.execute()
    from dspeed_amd import _lib
    import ctypes

    cur = ctypes.c_int(-1)
    _lib.check(_lib.lib().dsp_get_device(ctypes.byref(cur)))
    assert np.array_equal(out["trapEftp"], one["dsp/ch1"]["trapEftp"])
    # chunks of an iterator dealt to the two devices arrive in file order
    n = 1000
    lg = Table(waveform=WaveformTable(np.rint(x1[:n]).astype(np.uint16), 16.0, np.zeros(n)), baseline=Array(t1["baseline"][:n]), t_pick=Array(t1["t_pick"][:n]))
    ref = np.asarray(build_dsp(lg, dsp_config=recipes.C2)["trapEftp"])
    got = np.asarray(build_dsp(LH5Iterator(lg, buffer_len=130), dsp_config=recipes.C2, devices=[0, 1])["trapEftp"])
    assert np.array_equal(got, ref)
