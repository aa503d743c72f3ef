"""build_dsp: the table loop of the reference (src/dspeed/build_dsp.py:27-452) over in-memory / .npz tables -- channels with wildcards,
one recipe per channel pattern, per-channel database, row selection, output file modes -- against the chain run directly."""
import numpy as np
import pytest

import oracle
import recipes

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _table(rng, n, wf_len=4096, t0=0.0):
    from dspeed_amd.processing_chain import WaveformInput

    i = np.arange(wf_len)[None, :]
    start = np.floor(rng.uniform(0.45, 0.55, (n, 1)) * wf_len)
    bl = rng.uniform(9000, 11000, (n, 1))
    x = bl + rng.uniform(500, 15000, (n, 1)) * np.exp(-(i - start) / 1716.28) * (i >= start) + 5 * rng.standard_normal((n, wf_len))
    return {"waveform": WaveformInput(x.astype(np.float32), 16.0, t0), "baseline": bl[:, 0].astype(np.float32),
            "t_pick": (start[:, 0] + 775.4).astype(np.float32)}, x.astype(np.float32)


def test_tables_channels_database_and_row_selection(tmp_path):
    from dspeed_amd import build_dsp

    rng = np.random.default_rng(8)
    t1, x1 = _table(rng, 700)
    t2, x2 = _table(rng, 300)
    t3, x3 = _table(rng, 100)
    raw = {"raw/ch1": t1, "raw/ch2": t2, "raw/aux7": t3}
    short = {"outputs": ["wf_max"], "processors": {
        "tp_min, tp_max, wf_min, wf_max": {"function": "min_max", "module": "dspeed.processors",
                                           "args": ["waveform", "tp_min", "tp_max", "wf_min", "wf_max"]}}}
    db = {"ch2": {"pz": {"tau": "1500.25"}}}
    out = build_dsp(raw, dsp_config=recipes.C2, chan_config={"*aux*": short}, database=db, lh5_tables=["ch*", "aux*"], buffer_len=128)
    assert sorted(out) == ["dsp/aux7", "dsp/ch1", "dsp/ch2"]
    want1 = oracle.chain_energy(x1, t1["baseline"], t1["t_pick"], 1716.28, 625, 188, "l")[0]
    want2 = oracle.chain_energy(x2, t2["baseline"], t2["t_pick"], 1500.25, 625, 188, "l")[0]  # the channel's own database entry
    assert np.max(np.abs(out["dsp/ch1"]["trapEftp"] - want1) / np.abs(want1)) <= 1e-6
    assert np.max(np.abs(out["dsp/ch2"]["trapEftp"] - want2) / np.abs(want2)) <= 1e-6
    assert np.array_equal(out["dsp/aux7"]["wf_max"], x3.max(axis=1)) and list(out["dsp/aux7"]) == ["wf_max"]
    # one table, a row range, other outputs; then the same rows through entry_list
    part = build_dsp(t1, dsp_config=recipes.C2, i_start=100, n_entries=250, outputs=["trapEftp"])
    assert np.array_equal(part["trapEftp"], out["dsp/ch1"]["trapEftp"][100:350])
    picked = build_dsp(t1, dsp_config=recipes.C2, entry_list=[5, 17, 300, 699])
    assert np.array_equal(picked["trapEftp"], out["dsp/ch1"]["trapEftp"][[5, 17, 300, 699]])
    mask = np.zeros(700, dtype=bool)
    mask[::50] = True
    assert np.array_equal(build_dsp(t1, dsp_config=recipes.C2, entry_mask=mask)["trapEftp"], out["dsp/ch1"]["trapEftp"][::50])
    # buffers are pipelined, not a different computation: any buffer_len gives the same values
    assert np.array_equal(build_dsp(t1, dsp_config=recipes.C2, buffer_len=37)["trapEftp"], out["dsp/ch1"]["trapEftp"])

    # file in, file out, write modes
    raw_file, dsp_file = str(tmp_path / "raw.npz"), str(tmp_path / "dsp.npz")
    flat = {}
    for name, tb in raw.items():
        flat[f"{name}/waveform/values"], flat[f"{name}/waveform/dt"] = tb["waveform"].values, np.full(len(tb["baseline"]), 16.0)
        flat[f"{name}/waveform/t0"] = np.zeros(len(tb["baseline"]), dtype=np.float32)
        flat[f"{name}/baseline"], flat[f"{name}/t_pick"] = tb["baseline"], tb["t_pick"]
    np.savez(raw_file, **flat)
    assert build_dsp(raw_file, dsp_file, dsp_config=recipes.C2, lh5_tables="ch*", database=db) is None
    with np.load(dsp_file) as z:
        assert sorted(z.files) == ["dsp/ch1/trapEftp", "dsp/ch2/trapEftp"]
        assert np.array_equal(z["dsp/ch1/trapEftp"], out["dsp/ch1"]["trapEftp"]) and np.array_equal(z["dsp/ch2/trapEftp"], out["dsp/ch2"]["trapEftp"])
    with pytest.raises(FileExistsError):
        build_dsp(raw_file, dsp_file, dsp_config=recipes.C2, lh5_tables="ch1")
    build_dsp(raw_file, dsp_file, dsp_config=recipes.C2, lh5_tables="ch1", write_mode="a", n_entries=10)
    with np.load(dsp_file) as z:
        assert len(z["dsp/ch1/trapEftp"]) == 710 and len(z["dsp/ch2/trapEftp"]) == 300
    build_dsp(raw_file, dsp_file, dsp_config=recipes.C2, lh5_tables="ch1", write_mode="r")
    with np.load(dsp_file) as z:
        assert z.files == ["dsp/ch1/trapEftp"] and len(z["dsp/ch1/trapEftp"]) == 700


def test_fatal_rows_and_argument_errors():
    from dspeed_amd import build_dsp
    from dspeed_amd.errors import DSPFatal

    rng = np.random.default_rng(9)
    t1, _ = _table(rng, 64)
    bad = {"outputs": ["tp"], "processors": {"tp": "dspeed.processors.time_point_thresh(waveform, 9500, t_pick, 0, tp)"}}
    with pytest.raises(DSPFatal) as e:  # t_pick is fractional: "The starting index must be an integer", with the rows of the table
        build_dsp(t1, dsp_config=bad, i_start=10)
    assert "-" in str(e.value.wf_range) and int(str(e.value.wf_range).split("-")[0]) >= 10
    with pytest.raises(RuntimeError):
        build_dsp({"raw/ch1": t1}, dsp_config=recipes.C2, lh5_tables=["nothing*"])
    with pytest.raises(RuntimeError):
        build_dsp(42, dsp_config=recipes.C2)
    assert build_dsp({"raw/ch1": t1}, dsp_config=None, chan_config={"*ch9*": recipes.C2}) == {}  # no recipe for the channel: skipped


def test_whole_ge_recipe_through_the_table_loop(tmp_path):
    """the ICPC-structured recipe, written to a JSON file, over two channels with per-row t0 and a row selection: the same numbers as
    the chain run directly on those rows"""
    import json

    from dspeed_amd import build_dsp
    from dspeed_amd.processing_chain import WaveformInput, build_processing_chain
    from test_gpu_icpc_recipe import _synth

    rng = np.random.default_rng(3)
    cfg = str(tmp_path / "ge-recipe.json")
    with open(cfg, "w") as f:
        json.dump(recipes.ICPC, f)
    tables = {}
    for name, n in (("raw/ch1", 40), ("raw/ch2", 24)):
        wf, bl = _synth(rng, n)
        tables[name] = {"waveform": WaveformInput(wf, 16.0, (rng.integers(100, 200, n) * 16).astype(np.float32)), "baseline": bl}
    out = build_dsp(tables, dsp_config=cfg, i_start=4, n_entries=30, buffer_len=7)
    assert sorted(out) == ["dsp/ch1", "dsp/ch2"] and len(out["dsp/ch1"]["trapEmax"]) == 30 and len(out["dsp/ch2"]["trapEmax"]) == 20
    for name, rows in (("raw/ch1", slice(4, 34)), ("raw/ch2", slice(4, 24))):
        t = tables[name]
        tb = {"waveform": WaveformInput(t["waveform"].values[rows], 16.0, t["waveform"].t0[rows]), "baseline": t["baseline"][rows]}
        chain, _, direct = build_processing_chain(recipes.ICPC, tb)
        chain.execute()
        for k in recipes.ICPC["outputs"]:
            assert np.array_equal(out[name.replace("raw", "dsp")][k], direct[k], equal_nan=True), (name, k)


def test_lgdo_tables_and_lh5_iterators_through_the_adaptors():
    """build_dsp on an LGDO-protocol Table and on an LH5Iterator over it (stand-ins: tests/lgdo_standins.py) -- chunks read ahead on a
    thread, chain built from the first chunk, field mask pushed back to the iterator, DSPFatal annotated with the file position -- and
    proc_chain(tb_in, tb_out) writing into an LGDO output table (reference build_dsp.py:256-266, 399-432; processing_chain.py:675-716)."""
    from lgdo_standins import Array, LH5Iterator, Table, WaveformTable

    from dspeed_amd import lgdo_io
    from dspeed_amd.build_dsp import build_dsp
    from dspeed_amd.errors import DSPFatal
    from dspeed_amd.processing_chain import build_processing_chain

    rng = np.random.default_rng(8)
    n, wf_len = 1000, 4096
    i = np.arange(wf_len)[None, :]
    t0 = np.floor(rng.uniform(0.45, 0.55, (n, 1)) * wf_len)
    B = rng.uniform(9000, 11000, (n, 1))
    wf = np.rint(B + rng.uniform(500, 15000, (n, 1)) * np.exp(-(i - t0) / 1716.28) * (i >= t0) + 5 * rng.standard_normal((n, wf_len))).astype(np.uint16)
    bl = B[:, 0].astype(np.float32)
    tp = (t0[:, 0] + 625 + 150.4).astype(np.float32)
    raw = Table(waveform=WaveformTable(wf, 16.0, np.zeros(n)), baseline=Array(bl), t_pick=Array(tp), unused=Array(np.zeros((n, 100), np.float32)))
    want, rc = oracle.chain_energy(wf.astype(np.float32), bl, tp, 1716.28, 625, 188, "l")
    assert rc == 0

    def energies(res):
        col = res["trapEftp"]
        return np.asarray(col.nda if hasattr(col, "nda") else col)

    # one table in memory; a row range of it
    assert np.max(np.abs(energies(build_dsp(raw, dsp_config=recipes.C2)) - want) / np.abs(want)) <= TOL
    part = energies(build_dsp(raw, dsp_config=recipes.C2, i_start=100, n_entries=250))
    assert part.shape == (250,) and np.array_equal(part, energies(build_dsp(raw, dsp_config=recipes.C2))[100:350])
    # the same table behind an iterator of 300-row chunks in one refilled buffer
    it = LH5Iterator(raw, buffer_len=300)
    got = energies(build_dsp(it, dsp_config=recipes.C2))
    assert got.shape == (n,) and np.max(np.abs(got - want) / np.abs(want)) <= TOL
    assert sorted(it.field_mask) == ["baseline", "t_pick", "waveform"]  # the unused column is no longer read
    assert all("unused" not in keys for _pos, _n, keys in it.reads[1:])
    # a data-dependent DSPFatal names the rows of the chunk in the file
    bad = Table(raw)
    tp_bad = np.floor(tp)  # (mode 'i' wants whole samples: every row but one has them)
    tp_bad[640] += 0.5
    bad["t_pick"] = Array(tp_bad)
    rec_i = {"outputs": ["e"], "processors": dict(recipes.C2["processors"])}
    rec_i["processors"]["e"] = {"function": "fixed_time_pickoff", "module": "dspeed.processors", "args": ["wf_trap", "t_pick", "'i'", "e"]}
    with pytest.raises(DSPFatal, match="integer t_in") as ei:
        build_dsp(LH5Iterator(bad, buffer_len=300), dsp_config=rec_i)
    assert ei.value.wf_range == "600-900"
    # proc_chain(tb_in, tb_out) with LGDO tables on both sides
    chain, _mask, _ = build_processing_chain(recipes.C2, raw)
    out_tb = Table(trapEftp=Array(np.zeros(0, np.float32)))
    chain(raw, out_tb)
    assert np.array_equal(out_tb["trapEftp"].nda, energies(build_dsp(raw, dsp_config=recipes.C2)))
    assert lgdo_io.is_lgdo_table(out_tb)
