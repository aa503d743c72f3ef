"""LGDO / LH5 adaptors on stand-ins with the containers' protocol (no lgdo, no GPU): columns out of a Table, waveform units, ragged
vectors both ways, the read-ahead over an LH5Iterator, recipe translation from an LGDO table, the error when a real file is asked for."""
import numpy as np
import pytest

import recipes
from dspeed_amd import _lib, lgdo_io
from dspeed_amd.errors import DSPFatal
from dspeed_amd.processing_chain import WaveformInput, build_processing_chain
from lgdo_standins import Array, ArrayOfEqualSizedArrays, LH5Iterator, Table, VectorOfVectors, WaveformTable


def _raw_table(n=10, wf_len=4096, dt=16, dt_units="ns", t0_units="ns"):
    rng = np.random.default_rng(1)
    return Table(waveform=WaveformTable(rng.integers(0, 2000, (n, wf_len)).astype(np.uint16), dt, np.arange(n) * 16.0, dt_units, t0_units),
                 baseline=Array(rng.uniform(9, 11, n).astype(np.float32), {"units": "ADC"}),
                 t_pick=Array(np.full(n, 3000.5, np.float32)), timestamp=Array(np.arange(n, dtype=np.float64), {"units": "s"}))


def test_table_columns_and_waveform_units():
    tb = _raw_table(dt=0.016, dt_units="us", t0_units="ns")
    assert lgdo_io.is_lgdo_table(tb) and not lgdo_io.is_lgdo_table({"a": np.zeros(3)}) and not lgdo_io.is_chunk_iterator(tb)
    cols = lgdo_io.table_columns(tb)
    wf = cols["waveform"]
    assert isinstance(wf, WaveformInput) and wf.dt == pytest.approx(16.0) and np.array_equal(wf.t0, np.arange(10) * 16.0)
    assert wf.values is tb["waveform"].values.nda or np.shares_memory(wf.values, tb["waveform"].values.nda)  # a view, not a copy
    assert cols["baseline"].dtype == np.float32 and cols["timestamp"].dtype == np.float64
    only = lgdo_io.table_columns(tb, fields={"baseline"})
    assert list(only) == ["baseline"]
    # no usable time unit -> a plain array without a grid (reference processing_chain.py:2304-2308)
    nogrid = lgdo_io.column(WaveformTable(np.zeros((3, 8), np.float32), 1.0, 0.0, dt_units=None, t0_units=None))
    assert isinstance(nogrid, np.ndarray) and nogrid.shape == (3, 8)


def test_vector_of_vectors_round_trip_and_limits():
    flat = np.arange(9, dtype=np.float32)
    vov = VectorOfVectors(flat, [2, 2, 6, 9], {"units": "ns"})
    r = lgdo_io.RaggedColumn.from_vov(vov, max_len=5)
    assert r.padded.shape == (4, 5) and list(r.lengths) == [2, 0, 4, 3] and r.unit == "ns"
    assert np.array_equal(r.padded[0, :2], [0, 1]) and np.isnan(r.padded[0, 2:]).all() and np.isnan(r.padded[1]).all()
    assert np.array_equal(r.padded[2, :4], [2, 3, 4, 5]) and np.array_equal(r.padded[3, :3], [6, 7, 8])
    f2, cl2 = r.to_flat()
    assert np.array_equal(f2, flat) and list(cl2) == [2, 2, 6, 9]
    assert lgdo_io.RaggedColumn.from_vov(vov).padded.shape[1] == 8  # no length given: twice the longest vector (reference :2216-2226)
    ints = lgdo_io.RaggedColumn.from_vov(VectorOfVectors(np.arange(5, dtype=np.int32), [3, 5]), 4)
    assert ints.padded[1, 2] == 0  # integers are padded with 0
    with pytest.raises(DSPFatal, match="larger than array variable length"):
        lgdo_io.RaggedColumn.from_vov(vov, max_len=3)
    cols = lgdo_io.table_columns(Table(hits=vov, e=Array(np.zeros(4, np.float32))))
    assert set(cols) == {"hits", "len(hits)", "e"} and cols["len(hits)"].dtype == np.uint32


def test_recipe_translates_from_an_lgdo_table():
    chain, mask, tb_out = build_processing_chain(recipes.C2, _raw_table())
    assert [o[0] for o in chain.program.ops] == [_lib.OP_LOAD, _lib.OP_BL_SUBTRACT, _lib.OP_POLE_ZERO, _lib.OP_TRAP_PICKOFF, _lib.OP_STORE_SCALAR]
    assert sorted(mask) == ["baseline", "t_pick", "waveform"] and tb_out["trapEftp"].shape == (10,)
    assert chain.program.io[0][2] == _lib.U16  # uint16 rows select the float32 loop (processing_chain.py:1565-1572)
    # time quantities work off the WaveformTable's dt
    chain2, _, _ = build_processing_chain(recipes.C2_UNITS, _raw_table())
    assert chain2.program.ops[3][4][:2] == (625, 188)


def test_chunk_reader_reads_ahead_copies_and_keeps_order():
    tb = _raw_table(n=23, wf_len=64)
    it = LH5Iterator(tb, buffer_len=5)
    it.reset_field_mask(["waveform", "baseline"])
    got = list(lgdo_io.ChunkReader(it, fields={"waveform", "baseline"}))
    assert [(i, n) for i, n, _ in got] == [(0, 5), (5, 5), (10, 5), (15, 5), (20, 3)]
    assert all(set(c) == {"waveform", "baseline"} for _, _, c in got) and all(r[2] == ("waveform", "baseline") for r in it.reads)
    # the iterator refills one buffer: the chunks handed on must be copies that still hold THEIR rows
    for i, n, c in got:
        assert np.array_equal(c["waveform"].values, tb["waveform"].values.nda[i:i + n])
        assert np.array_equal(c["baseline"], tb["baseline"].nda[i:i + n]) and np.array_equal(c["waveform"].t0, np.arange(i, i + n) * 16.0)

    class Broken(LH5Iterator):
        def __iter__(self):
            yield from super().__iter__()
            raise OSError("disk gone")

    with pytest.raises(OSError, match="disk gone"):
        list(lgdo_io.ChunkReader(Broken(tb, buffer_len=10)))


def test_results_table_without_lgdo_and_write_back():
    res = lgdo_io.results_table({"e": np.arange(3, dtype=np.float32), "hits": np.arange(12, dtype=np.float32).reshape(3, 4)}, units={"e": "ADC"},
                                lengths={"hits": np.array([1, 4, 0], dtype=np.uint32)})
    if lgdo_io.lgdo_or_none() is None:
        assert isinstance(res["hits"], lgdo_io.RaggedColumn) and list(res["hits"].to_flat()[1]) == [1, 5, 5]
    out = Table(e=Array(np.zeros(2, np.float32)), wf=ArrayOfEqualSizedArrays(np.zeros((2, 4), np.float32)))
    lgdo_io.write_back(out, {"e": np.array([1, 2, 3], np.float32), "wf": np.ones((3, 4), np.float32), "other": np.zeros(3)}, start=2)
    assert len(out["e"]) == 5 and list(out["e"].nda) == [0, 0, 1, 2, 3] and out["wf"].nda.shape == (5, 4) and out["wf"].nda[2:].all()


def test_an_lh5_file_needs_the_package_and_says_so():
    from dspeed_amd.build_dsp import build_dsp

    if lgdo_io.lgdo_or_none() is not None:
        pytest.skip("lgdo is installed here")
    with pytest.raises(ImportError, match="lgdo"):
        build_dsp("run0001.lh5", dsp_config=recipes.C2)


def test_variable_index_into_a_variable_length_array_translates():
    """the reference's own case (tests/test_processing_chain.py:75-97): a VectorOfVectors input, its middle and its last element"""
    vov = VectorOfVectors(np.arange(150.0), [10, 30, 60, 100, 150], {"units": "ns"})
    rec = {"outputs": ["vals", "v_end"], "processors": {"vals": "vov_in(shape=50)[len(vov_in)//2]", "v_end": "vov_in(shape=50)[-1]",
                                                        "var_slice": "vov_in[indices:20]"}}
    chain, mask, out = build_processing_chain(rec, Table(vov_in=vov))
    ops = chain.program.ops + [o for st in chain._stages for o in st["program"].ops]  # (len(vov_in) // 2 needs no waveform: the scalar head's)
    picks = [o for o in ops if o[0] == _lib.OP_PICKOFF]
    assert len(picks) == 2 and all(o[4][1] == 2 for o in picks)  # get_default with a per-event index
    # len(vov_in) // 2: the lengths are uint32, so NumPy's 'II->I' loop (the float64 rows make this the float64 chain, which holds them)
    assert any(o[0] == _lib.OP_SCALAR_FUNC and o[4][0] == _lib.fn_int(_lib.FN_IFLOORDIV, np.uint32) for o in ops)
    assert chain.program.slots == [50] and sorted(mask) == ["vov_in"]  # (the lengths come with the VectorOfVectors itself)
    with pytest.raises(DSPFatal, match="larger than array variable length"):
        build_processing_chain({"outputs": ["v"], "processors": {"v": "vov_in(shape=40)[0]"}}, Table(vov_in=vov))
