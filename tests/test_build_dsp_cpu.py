"""CPU part of build_dsp (reference src/dspeed/build_dsp.py:27-452): file layout, table selection and argument errors -- everything up
to the point where a chain would run."""
import numpy as np
import pytest

import recipes
from dspeed_amd.build_dsp import _read_npz, build_dsp
from dspeed_amd.processing_chain import WaveformInput


def test_npz_layout_mirrors_an_lh5_waveform_table(tmp_path):
    f = str(tmp_path / "raw.npz")
    np.savez(f, **{"raw/ch1/waveform/values": np.zeros((5, 64), dtype=np.uint16), "raw/ch1/waveform/dt": np.full(5, 16.0),
                   "raw/ch1/waveform/t0": np.arange(5, dtype=np.float32), "raw/ch1/baseline": np.ones(5, dtype=np.float32),
                   "raw/ch2/waveform/values": np.zeros((3, 32), dtype=np.int16), "raw/ch2/energy": np.zeros(3)})
    t = _read_npz(f)
    assert sorted(t) == ["raw/ch1", "raw/ch2"] and sorted(t["raw/ch1"]) == ["baseline", "waveform"]
    w = t["raw/ch1"]["waveform"]
    assert isinstance(w, WaveformInput) and w.dt == 16.0 and np.array_equal(w.t0, np.arange(5)) and w.values.shape == (5, 64)
    assert t["raw/ch2"]["waveform"].dt == 1.0 and t["raw/ch2"]["waveform"].t0 == 0.0


def test_argument_errors_before_anything_runs(tmp_path):
    tb = {"waveform": WaveformInput(np.zeros((4, 4096), dtype=np.float32), 16.0), "baseline": np.zeros(4, dtype=np.float32),
          "t_pick": np.zeros(4, dtype=np.float32)}
    with pytest.raises(RuntimeError):
        build_dsp(42, dsp_config=recipes.C2)
    with pytest.raises(RuntimeError):
        build_dsp({"raw/ch1": tb}, dsp_config=recipes.C2, lh5_tables=["nothing*"])
    with pytest.raises(RuntimeError):
        build_dsp(tb, dsp_config=recipes.C2, lh5_tables=["a", "b"])
    with pytest.raises(ValueError):
        build_dsp({"raw/ch1": tb}, dsp_config=recipes.C2, database=[1, 2])
    out = str(tmp_path / "dsp.npz")
    np.savez(out, x=np.zeros(1))
    with pytest.raises(FileExistsError):
        build_dsp({"raw/ch1": tb}, out, dsp_config=recipes.C2)
    assert build_dsp({"raw/ch1": tb}, dsp_config=None, chan_config={"*ch9*": recipes.C2}) == {}


def test_recipe_book_rows_friends_and_devices(monkeypatch):
    from dspeed_amd.build_dsp import DeviceTeam, Friend, RecipeBook, RowSelection, _device_list, friends_of, shard_bounds
    from dspeed_amd.errors import ProcessingChainError

    book = RecipeBook({"default": 1}, {"*aux*": {"aux": 1}, "raw/ch1*": {"ch1x": 1}, "raw/ch12": {"never": 1}}, {"ch12": {"tau": 5}, "ch3": {"tau": 7}})
    assert book.recipe_for("raw/aux7") == {"aux": 1} and book.recipe_for("raw/ch12") == {"ch1x": 1} and book.recipe_for("raw/ch3") == {"default": 1}
    assert RecipeBook(None, {"*ch9*": {}}, None).recipe_for("raw/ch1") is None  # no default: the channel is skipped
    assert book.database_for("raw/ch12") == {"tau": 5} and book.database_for("ch3/raw") == {"tau": 7} and book.database_for("raw/ch4") is None
    assert book.database_for("") == book.database and book.database_for("raw") == book.database
    assert RecipeBook.channel_of("raw/ch7/extra") == "ch7" and RecipeBook.channel_of("raw") is None

    sel, first = RowSelection(None, None, 100, 250).of(700)
    assert (sel, first) == (slice(100, 350), 100)
    assert RowSelection(None, None, 900, None).of(700) == (slice(700, 700), 700)
    sel, first = RowSelection([5, 17, 300, 699], None, 1, 2).of(700)
    assert list(sel) == [17, 300] and first is None
    mask = np.zeros(10, dtype=bool)
    mask[[2, 4, 9]] = True
    assert list(RowSelection(None, mask, 0, None).of(10)[0]) == [2, 4, 9]

    db = {"aux": {"file": "hits.lh5", "group": "ch1/hit"}}
    fr = friends_of({"inputs": {"file": "db.aux.file", "group": "db.aux.group", "prefix": "hit_"}}, db)
    assert fr == [Friend("hits.lh5", "ch1/hit", "hit_", "")]
    fr = friends_of({"inputs": [{"file": "a.lh5", "group": "g"}, {"file": "b.lh5", "group": "h", "suffix": "_b"}]}, None)
    assert fr == [Friend("a.lh5", "g"), Friend("b.lh5", "h", "", "_b")] and friends_of({}, None) == []
    with pytest.raises(ProcessingChainError, match="did not find db.aux.nothing in database."):
        friends_of({"inputs": {"file": "db.aux.nothing", "group": "g"}}, db)

    assert shard_bounds(10, 3) == [(0, 3), (3, 6), (6, 10)] and shard_bounds(2, 2) == [(0, 1), (1, 2)]
    monkeypatch.delenv("DSPEED_HIP_DEVICES", raising=False)
    assert _device_list(None) == [] and _device_list(2) == [2] and _device_list([0, 0, 1]) == [0, 0, 1]
    monkeypatch.setenv("DSPEED_HIP_DEVICES", "0, 2,3")
    assert _device_list(None) == [0, 2, 3] and _device_list([1]) == [1]
    with pytest.raises(ValueError):
        _device_list([-1])
    assert not DeviceTeam([]).parallel and not DeviceTeam([0]).parallel and DeviceTeam([0, 0]).parallel and len(DeviceTeam([])) == 1


def test_tables_of_an_lh5_file_are_found_like_the_reference_finds_them():
    """``raw`` nested below the channel, wildcards, the default base group (reference build_dsp.py:147-186), against a listing stand-in"""
    from dspeed_amd.build_dsp import _ChunkSource

    groups = ["raw", "raw/ch1", "raw/ch1/waveform", "raw/ch2", "raw/ch2/raw", "raw/aux", "raw/aux/energy", "raw/empty"]

    def ls(_file, pattern):
        from fnmatch import fnmatchcase

        pattern = pattern.strip("/")
        if pattern.endswith("/*"):
            return [g for g in groups if fnmatchcase(g, pattern) and g.count("/") == pattern.count("/")]
        return [g for g in groups if fnmatchcase(g, pattern) and g.count("/") == pattern.count("/")]

    src = _ChunkSource.__new__(_ChunkSource)
    src.lh5 = type("lh5", (), {"ls": staticmethod(ls)})
    assert src._tables_of_file("f.lh5", None, None) == ["raw/ch1", "raw/ch2/raw", "raw/aux", "raw/empty"]
    assert src._tables_of_file("f.lh5", ["ch*"], "raw") == ["raw/ch1", "raw/ch2/raw"]
    with pytest.raises(RuntimeError, match="could not find any valid LH5 table"):
        src._tables_of_file("f.lh5", ["nothing*"], None)
