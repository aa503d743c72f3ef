"""The table loop around the chain -- the role of the reference's ``build_dsp`` (src/dspeed/build_dsp.py:27-452) for tables that are
already arrays.

The reference reads raw-tier LH5 files through ``lgdo.lh5`` (absent in this environment); everything of its driver
that is not HDF5 is mirrored here with the same parameter names and meaning: the loop over tables with wildcards (``lh5_tables``),
one recipe per channel pattern (``chan_config``, first match wins, :225-230), the per-channel parameter database (:232-238), row
selection (``entry_list`` / ``entry_mask`` / ``i_start`` / ``n_entries``), default outputs from the recipe (:333-336), the
``raw`` -> ``dsp`` renaming of the output tables (:375), ``DSPFatal`` annotated with the row range (:400-404), the write modes of the
output file (:204-213).  ``buffer_len`` keeps its meaning -- rows moved per transfer; left at ``None`` the chain picks the size (256 MiB
of rows: the reference's 3200 rows are a fraction of a millisecond of device work, and a recipe's one-waveform-per-lane kernels take
milliseconds whatever the number of rows) -- but the buffers of one table are pipelined inside one ``execute``: the host-to-device copy of buffer k+1 overlaps the kernel and the device-to-host copy of buffer k
(``ProcessingChain.execute``), instead of read / process / write in turn.

Tables: a mapping ``column -> array | DeviceArray | WaveformInput``; ``raw_in`` is one table, a mapping of tables, or the name of an
``.npz`` file whose keys are ``<table>/<column>`` for plain columns and ``<table>/<column>/values``, ``.../dt``, ``.../t0`` for
waveforms (the group layout of an LH5 WaveformTable).  ``dsp_out`` is ``None`` (return the tables) or the name of an ``.npz``.

LGDO / LH5 (dspeed_amd/lgdo_io.py): ``raw_in`` may also be an ``lgdo.Table``, an ``lh5.LH5Iterator`` -- or anything with their
protocol -- or the name of an ``.lh5`` file (that needs the ``lgdo`` package).  Chunks are then read one ahead on a thread of their own
while the device works on the previous one, processed with the chain built from the first chunk (``reset_field_mask`` narrows the
later reads to the columns the recipe uses, build_dsp.py:369-370) and written as they finish (``LH5Store.write`` with the reference's
``wo_mode`` / ``write_start``, :416-424) or collected into one table.
"""
from __future__ import annotations

import json
import os
from collections.abc import Collection, Mapping
from fnmatch import fnmatch

import numpy as np

from . import lgdo_io
from .device import DeviceArray
from .errors import DSPFatal
from .processing_chain import WaveformInput, build_processing_chain


def _is_table(obj) -> bool:
    return isinstance(obj, Mapping) and all(isinstance(v, (np.ndarray, DeviceArray, WaveformInput)) for v in obj.values())


def _load_config(cfg):
    if isinstance(cfg, str):
        with open(os.path.expandvars(os.path.expanduser(cfg))) as f:
            text = f.read()
        try:
            return json.loads(text)
        except json.JSONDecodeError:
            import yaml

            return yaml.safe_load(text)
    return cfg


def _read_npz(path) -> dict:
    """``<table>/<column>[/values|/dt|/t0]`` keys -> {table: {column: array | WaveformInput}}"""
    tables: dict = {}
    with np.load(path) as z:
        keys = list(z.files)
        wf = {}
        for k in keys:
            parts = k.split("/")
            if parts[-1] in ("values", "dt", "t0") and len(parts) >= 2:
                wf.setdefault("/".join(parts[:-1]), {})[parts[-1]] = z[k]
        for k in keys:
            parts = k.split("/")
            if "/".join(parts[:-1]) in wf and parts[-1] in ("values", "dt", "t0"):
                continue
            tables.setdefault("/".join(parts[:-1]), {})[parts[-1]] = z[k]
        for full, d in wf.items():
            parts = full.split("/")
            if "values" not in d:
                raise ValueError(f"{path}: waveform '{full}' has no values")
            dt = float(np.asarray(d.get("dt", 1.0)).reshape(-1)[0])  # (one sampling period per table, like the reference: wf_table.dt[0])
            t0 = d.get("t0", 0.0)
            t0 = float(t0) if np.ndim(t0) == 0 else np.ascontiguousarray(t0)
            tables.setdefault("/".join(parts[:-1]), {})[parts[-1]] = WaveformInput(d["values"], dt, t0)
    return tables


def _rows(col):
    return len(col.values) if isinstance(col, WaveformInput) else len(col)


def _select(col, sel):
    """rows of a column: a slice is a view, an index array a copy (entry_list / entry_mask)"""
    if isinstance(col, WaveformInput):
        t0 = col.t0 if isinstance(col.t0, float) else _select(col.t0, sel)
        return WaveformInput(_select(col.values, sel), col.dt, t0)
    if isinstance(col, DeviceArray):
        if not isinstance(sel, slice):
            raise NotImplementedError("entry_list / entry_mask on device-resident columns: select on the host, or pass a row range")
        return col.view_rows(sel.start, sel.stop)
    return col[sel]


def _pick_config(tb, dsp_config, chan_config, database):
    this_config = dsp_config
    for pat, cfg in chan_config.items():
        if fnmatch(tb, pat):
            this_config = cfg
            break
    if tb not in ("", "raw"):
        chan_name = next(k for k in tb.split("/") if k not in ("", "raw"))
        db_dict = database.get(chan_name) if database else None
    else:
        db_dict = database
    return this_config, db_dict


def _run_chunks(tb, source, this_config, db_dict, outputs, i_start, n_entries, buffer_len, sink):
    """One table given as an LGDO table or an iterator of chunks: build the chain on the first chunk, then read ahead / process / hand the
    results of every chunk to ``sink(i_entry, n_rows, {name: ndarray})``.  Returns the number of rows processed."""
    if this_config.get("inputs"):
        raise NotImplementedError("auxiliary 'inputs' files (LH5Iterator.add_friend, build_dsp.py:268-330) are not supported")
    _outputs = this_config["outputs"] if outputs is None else outputs
    if lgdo_io.is_chunk_iterator(source):
        it = source
        if n_entries is not None and hasattr(it, "n_entries"):
            it.n_entries = min(int(n_entries), len(it))
        first = next(iter(it), None)
        if first is None:
            return 0
        chain, mask, _tb_out = build_processing_chain(this_config["processors"], lgdo_io.table_columns(first), db_dict=db_dict, outputs=list(_outputs))
        if hasattr(it, "reset_field_mask"):
            it.reset_field_mask(mask)
        chunks = lgdo_io.ChunkReader(it, fields=set(mask))
    else:  # one table in memory
        cols = lgdo_io.table_columns(source)
        n_all = _rows(next(iter(cols.values())))
        stop = n_all if n_entries is None else min(n_all, i_start + int(n_entries))
        cols = {k: _select(v, slice(min(i_start, n_all), stop)) for k, v in cols.items()}
        if _rows(next(iter(cols.values()))) == 0:
            return 0
        chain, mask, _tb_out = build_processing_chain(this_config["processors"], cols, db_dict=db_dict, outputs=list(_outputs))
        chunks = [(0, _rows(next(iter(cols.values()))), cols)]
    done = 0
    try:
        for i_entry, n, cols in chunks:
            row_bytes = sum(np.asarray(c.values if isinstance(c, WaveformInput) else c).nbytes // max(n, 1) for k, c in cols.items() if k in mask)
            if buffer_len is not None:
                chain.pipeline_bytes = max(1, int(buffer_len)) * max(row_bytes, 1)
            out = {name[4:] if name.startswith("out:") else name: np.empty((n,) if length is None else (n, length), dtype=getattr(var, "dtype", None) or chain.loop_dtype)
                   for name, (var, length) in chain._out_vars.items()}
            try:
                chain(cols, out)
            except DSPFatal as e:
                e.wf_range = f"{i_entry}-{i_entry + n}"  # the position in the file (build_dsp.py:408-410)
                raise
            for c in chain._copy_pars:
                if c in cols:
                    col = cols[c]
                    out[c] = np.asarray(col.values if isinstance(col, WaveformInput) else col)
            # variable-length outputs (declared with vector_len=len(<input>)) leave as VectorOfVectors: padded rows + their lengths
            sink(i_entry, n, out, {k: np.asarray(cols[src]) for k, src in chain.vector_lens.items() if k in out and src in cols})
            done += n
    finally:
        if isinstance(chunks, lgdo_io.ChunkReader):
            chunks.close()
    return done


def _build_dsp_lgdo(raw_in, dsp_out, dsp_config, lh5_tables, base_group, database, outputs, write_mode, entry_list, entry_mask, i_start,
                    n_entries, buffer_len, chan_config):
    dsp_config = _load_config(dsp_config)
    chan_config = {k: _load_config(v) for k, v in dict(_load_config(chan_config) or {}).items()}
    database = _load_config(database)
    if database and not isinstance(database, Mapping):
        raise ValueError("input database is not a valid JSON or YAML file or dict")
    lh5_file = isinstance(raw_in, str)
    if isinstance(lh5_tables, str):
        lh5_tables = [lh5_tables]
    store = None
    if lh5_file:
        lg, lh5 = lgdo_io.require_lh5(f"reading '{raw_in}'")
        if base_group is None:
            base_group = "raw" if lh5.ls(raw_in, "raw") else ""
        if lh5_tables is None:
            tables = lh5.ls(raw_in, f"{base_group}/*")
        else:
            tables = [t for wc in lh5_tables for t in lh5.ls(raw_in, f"{base_group}/{wc}")]
        fixed = []
        for tb in tables:  # 'raw' is sometimes nested, e.g. ch024/raw (build_dsp.py:176-183)
            if lh5.ls(raw_in, f"{tb}/*") == [f"{tb}/raw"]:
                fixed.append(f"{tb}/raw")
            elif lh5.ls(raw_in, tb):
                fixed.append(tb)
        tables = fixed
        if not tables:
            raise RuntimeError(f"could not find any valid LH5 table in {raw_in}")
    else:
        if lh5_tables is not None and len(lh5_tables) > 1:
            raise RuntimeError("Cannot have more than one value in lh5_tables for input of type Table or LH5Iterator")
        tables = [lh5_tables[0] if lh5_tables else ""]
    to_lh5 = dsp_out is not None and not str(dsp_out).endswith(".npz")
    if dsp_out is not None:
        if write_mode is None and os.path.isfile(dsp_out):
            raise FileExistsError(f"output file {dsp_out} exists. Set the 'write_mode' keyword")
        if write_mode == "r" and os.path.isfile(dsp_out):
            os.remove(dsp_out)
        if to_lh5:
            lg, lh5 = lgdo_io.require_lh5(f"writing '{dsp_out}'")
            store = lh5.LH5Store(keep_open=True)
    result, result_lens = {}, {}
    for tb in tables:
        this_config, db_dict = _pick_config(tb, dsp_config, chan_config, database)
        if this_config is None:
            continue
        dsp_name = tb.replace("raw", "dsp")
        parts = []
        if lh5_file:
            source = lh5.LH5Iterator(raw_in, tb, entry_list=entry_list, entry_mask=entry_mask, i_start=i_start, n_entries=n_entries,
                                     buffer_len=buffer_len if buffer_len is not None else 32768)  # (rows per read: a chunk is one execute)
        else:
            source = raw_in

        lens_parts = []

        def sink(i_entry, n, out, lens, _name=dsp_name, _parts=parts, _lens=lens_parts):
            if store is not None:
                store.write(obj=lgdo_io.results_table(out, lengths=lens), name=_name, lh5_file=dsp_out, wo_mode="o" if write_mode == "u" else "a",
                            write_start=i_start + i_entry, n_rows=n)
            else:
                _parts.append(out)
                _lens.append(lens)

        _run_chunks(tb, source, this_config, db_dict, outputs, i_start, n_entries, buffer_len, sink)
        if store is None:
            keys = list(parts[0]) if parts else []
            result[dsp_name] = {k: np.concatenate([p[k] for p in parts]) for k in keys}
            result_lens[dsp_name] = {k: np.concatenate([p[k] for p in lens_parts]) for k in (lens_parts[0] if lens_parts else {})}
    if store is not None:
        return None
    if dsp_out is None:
        tabs = {k: lgdo_io.results_table(v, lengths=result_lens.get(k)) for k, v in result.items()}
        return tabs[next(iter(tabs))] if not lh5_file and tabs else tabs
    flat = {f"{t}/{k}" if t else k: np.asarray(v) for t, cols in result.items() for k, v in cols.items()}
    np.savez(dsp_out + ".tmp.npz", **flat)
    os.replace(dsp_out + ".tmp.npz", dsp_out)
    return None


def build_dsp(raw_in, dsp_out: str | None = None, dsp_config=None, lh5_tables=None, base_group: str | None = None, database=None,
              outputs: Collection[str] | None = None, write_mode: str | None = None, entry_list=None, entry_mask=None, i_start: int = 0,
              n_entries: int | None = None, buffer_len: int | None = None, block_width: int = 16, chan_config=None):
    """Run recipes over tables of waveforms; returns ``{dsp table name: {parameter: ndarray}}`` (one table: the table itself) when
    ``dsp_out`` is None, else writes them to the ``.npz`` and returns None.  Parameters as in the reference (build_dsp.py:27-127)."""
    del block_width  # (the device processes whole buffers)
    if (isinstance(raw_in, str) and raw_in.lower().endswith((".lh5", ".h5", ".hdf5"))) or lgdo_io.is_chunk_iterator(raw_in) or \
            lgdo_io.is_lgdo_table(raw_in):
        return _build_dsp_lgdo(raw_in, dsp_out, dsp_config, lh5_tables, base_group, database, outputs, write_mode, entry_list, entry_mask,
                               i_start, n_entries, buffer_len, chan_config)
    if isinstance(lh5_tables, str):
        lh5_tables = [lh5_tables]
    single = False
    if isinstance(raw_in, str):
        tables = _read_npz(raw_in)
    elif _is_table(raw_in):
        if lh5_tables is not None and len(lh5_tables) > 1:
            raise RuntimeError("Cannot have more than one value in lh5_tables for input of type Table")
        tables, single = {(lh5_tables[0] if lh5_tables else ""): raw_in}, True
        lh5_tables = None
    elif isinstance(raw_in, Mapping) and all(_is_table(t) for t in raw_in.values()):
        tables = dict(raw_in)
    else:
        raise RuntimeError(f"raw_in was not a file name, a table or a mapping of tables: {type(raw_in).__name__}")

    names = list(tables)
    if base_group is None:
        base_group = "raw" if any(n == "raw" or n.startswith("raw/") for n in names) else ""
    if not single:
        in_base = [n for n in names if not base_group or n == base_group or n.startswith(base_group + "/")]
        if lh5_tables is None:
            names = in_base
        else:
            rel = lambda n: n[len(base_group) + 1:] if base_group and n.startswith(base_group + "/") else n  # noqa: E731
            names = [n for pat in lh5_tables for n in in_base if fnmatch(rel(n), pat) or fnmatch(n, pat)]
            names = list(dict.fromkeys(names))
        if not names:
            raise RuntimeError(f"could not find any valid table in {raw_in if isinstance(raw_in, str) else 'raw_in'}")

    dsp_config = _load_config(dsp_config)
    chan_config = dict(_load_config(chan_config) or {})
    for chan, cfg in chan_config.items():
        chan_config[chan] = _load_config(cfg)
    database = _load_config(database)
    if database and not isinstance(database, Mapping):
        raise ValueError("input database is not a valid JSON or YAML file or dict")
    if dsp_out is not None:
        if write_mode is None and os.path.isfile(dsp_out):
            raise FileExistsError(f"output file {dsp_out} exists. Set the 'write_mode' keyword")
        if write_mode not in (None, "r", "a", "u"):
            raise ValueError("write_mode must be None, 'r', 'a' or 'u'")

    result = {}
    for tb in names:
        this_config = dsp_config
        for pat, cfg in chan_config.items():
            if fnmatch(tb, pat):
                this_config = cfg
                break
        if this_config is None:  # (dsp_config may be None with chan_config: channels without a match are skipped)
            continue
        if tb not in ("", "raw"):
            chan_name = next(k for k in tb.split("/") if k not in ("", "raw"))
            db_dict = database.get(chan_name) if database else None
        else:
            db_dict = database
        table = tables[tb]
        n_all = _rows(next(iter(table.values())))
        if entry_list is not None or entry_mask is not None:
            idx = np.asarray(entry_list) if entry_list is not None else np.flatnonzero(np.asarray(entry_mask))
            idx = idx[i_start:]
            if n_entries is not None:
                idx = idx[:n_entries]
            sel, first_row = idx, None
        else:
            stop = n_all if n_entries is None else min(n_all, i_start + n_entries)
            sel, first_row = slice(min(i_start, n_all), stop), min(i_start, n_all)
        tb_in = {k: _select(v, sel) for k, v in table.items()}
        tot_n_rows = _rows(next(iter(tb_in.values())))
        _outputs = this_config["outputs"] if outputs is None else outputs
        proc_chain, _mask, tb_out = build_processing_chain(this_config["processors"], tb_in, db_dict=db_dict, outputs=list(_outputs))
        if tot_n_rows:
            # rows per transfer: the chain streams host columns through pairs of device buffers of this many rows
            row_bytes = sum((np.asarray(c.values if isinstance(c, WaveformInput) else c).nbytes // max(tot_n_rows, 1))
                            for k, c in tb_in.items() if k in _mask and not isinstance(c.values if isinstance(c, WaveformInput) else c, DeviceArray))
            if buffer_len is not None:
                proc_chain.pipeline_bytes = max(1, int(buffer_len)) * max(row_bytes, 1)
            try:
                proc_chain.execute(0, tot_n_rows)
            except DSPFatal as e:
                if isinstance(e.wf_range, range) and first_row is not None:
                    e.wf_range = f"{first_row + e.wf_range.start}-{first_row + e.wf_range.stop}"
                raise
        dsp_name = tb.replace("raw", "dsp")
        result[dsp_name] = tb_out

    if dsp_out is None:
        return result[next(iter(result))] if single and result else result
    flat = {f"{t}/{k}" if t else k: np.asarray(v) for t, cols in result.items() for k, v in cols.items()}
    if write_mode in ("a", "u") and os.path.isfile(dsp_out):
        with np.load(dsp_out) as z:
            old = {k: z[k] for k in z.files}
        for k, v in flat.items():
            old[k] = np.concatenate([old[k], v]) if (write_mode == "a" and k in old) else v
        flat = old
    tmp = dsp_out + ".tmp.npz"
    np.savez(tmp, **flat)
    os.replace(tmp, dsp_out)
    return None
