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
