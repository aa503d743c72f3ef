"""The table loop around the chain -- the role of the reference's ``build_dsp`` (src/dspeed/build_dsp.py:27-452).

What is kept, with the reference's parameter names and meaning: tables chosen with wildcards (``lh5_tables``, ``base_group``), one recipe per
channel pattern (``chan_config``, first match wins) and one database block per channel, row selection (``entry_list`` / ``entry_mask`` /
``i_start`` / ``n_entries``), default outputs from the recipe, ``raw`` -> ``dsp`` in the names of the output tables, ``DSPFatal``
annotated with the rows of the table, the write modes of the output file, auxiliary ``inputs`` files joined to a table as friends.
``buffer_len`` keeps its meaning -- rows moved per transfer; left at ``None`` the chain picks the size (256 MiB of rows) -- but the
buffers of one table are pipelined inside one ``execute`` (copy of buffer k+1 over PCIe while buffer k is processed) instead of read /
process / write in turn.

How it is organised here:

* ``RecipeBook`` answers "which recipe and which database block does table X get";
* ``RowSelection`` turns the four row-selection parameters into a selector for one table;
* a *source* (``_ArraySource``, ``_ChunkSource``) lists the tables of ``raw_in`` and hands out their rows; a *sink* (``_MemorySink``,
  ``_NpzSink``, ``_Lh5Sink``) takes the finished columns;
* ``DeviceTeam`` fans the rows of a table over several GPUs: one worker thread per device, each with its own chain handle, streams
  and staging buffers on that device, contiguous row shards (arrays) or whole chunks dealt in turn (chunk iterators), no collective --
  events are independent.  ``devices=[0, 1, ...]`` or the environment variable ``DSPEED_HIP_DEVICES`` ("0,1,2,3" or "all").

Tables: a mapping ``column -> array | DeviceArray | WaveformInput``; ``raw_in`` is one table, a mapping of tables, or the name of an
``.npz`` file whose keys are ``<table>/<column>`` for plain columns and ``<table>/<column>/values``, ``.../dt``, ``.../t0`` for
waveforms (the group layout of an LH5 WaveformTable).  ``raw_in`` may also be an ``lgdo.Table``, an ``lh5.LH5Iterator`` -- or anything
with their protocol (dspeed_amd/lgdo_io.py) -- or the name of an ``.lh5`` file (that needs the ``lgdo`` package).  Chunks are read one
ahead on a thread of their own while the device works on the previous one.  ``dsp_out`` is ``None`` (return the tables), the name of
an ``.npz``, or of an ``.lh5`` file (written through ``LH5Store.write`` with the reference's ``wo_mode`` / ``write_start``).
"""
from __future__ import annotations

import json
import os
import queue
import threading
from collections.abc import Collection, Mapping
from dataclasses import dataclass
from fnmatch import fnmatch

import numpy as np

from . import lgdo_io
from .device import DeviceArray, device_count, set_device
from .errors import DSPFatal, ProcessingChainError
from .processing_chain import WaveformInput, build_processing_chain
from .recipe import _DB_REF


# ----------------------------------------------------------------------------------------------------------------------------------
# small helpers on array tables
# ----------------------------------------------------------------------------------------------------------------------------------
def _is_table(obj) -> bool:
    return isinstance(obj, Mapping) and all(isinstance(v, (np.ndarray, DeviceArray, WaveformInput)) for v in obj.values())


def _load_config(cfg):
    """a path to a JSON / YAML file -> its content; anything else as it is"""
    if not isinstance(cfg, str):
        return cfg
    with open(os.path.expandvars(os.path.expanduser(cfg))) as f:
        text = f.read()
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        import yaml

        return yaml.safe_load(text)


def _read_npz(path) -> dict:
    """``<table>/<column>[/values|/dt|/t0]`` keys -> {table: {column: array | WaveformInput}}"""
    tables: dict = {}
    with np.load(path) as z:
        members = {k: k.split("/") for k in z.files}
        waveforms: dict = {}
        for k, parts in members.items():
            if len(parts) >= 2 and parts[-1] in ("values", "dt", "t0"):
                waveforms.setdefault("/".join(parts[:-1]), {})[parts[-1]] = z[k]
        for k, parts in members.items():
            group = "/".join(parts[:-1])
            if not (group in waveforms and parts[-1] in ("values", "dt", "t0")):
                tables.setdefault(group, {})[parts[-1]] = z[k]
        for full, d in waveforms.items():
            if "values" not in d:
                raise ValueError(f"{path}: waveform '{full}' has no values")
            table, _, column = full.rpartition("/")
            dt = float(np.asarray(d.get("dt", 1.0)).reshape(-1)[0])  # (one sampling period per table, like the reference: wf_table.dt[0])
            t0 = d.get("t0", 0.0)
            tables.setdefault(table, {})[column] = WaveformInput(d["values"], dt, float(t0) if np.ndim(t0) == 0 else np.ascontiguousarray(t0))
    return tables


def _rows(col):
    return len(col.values) if isinstance(col, WaveformInput) else len(col)


def _table_rows(tb) -> int:
    return _rows(next(iter(tb.values()))) if tb else 0


def _values(col):
    return col.values if isinstance(col, WaveformInput) else col


def _select(col, sel):
    """rows of a column: a slice is a view, an index array a copy (entry_list / entry_mask)"""
    if isinstance(col, WaveformInput):
        w = WaveformInput(_select(col.values, sel), col.dt, col.t0 if isinstance(col.t0, float) else _select(col.t0, sel))
        if getattr(col, "lengths", None) is not None:
            w.lengths = _select(col.lengths, sel)
        return w
    if isinstance(col, DeviceArray):
        if not isinstance(sel, slice):
            raise NotImplementedError("entry_list / entry_mask on device-resident columns: select on the host, or pass a row range")
        return col.view_rows(sel.start, sel.stop)
    return col[sel]


# ----------------------------------------------------------------------------------------------------------------------------------
# which recipe, which database block, which rows
# ----------------------------------------------------------------------------------------------------------------------------------
class RecipeBook:
    """The recipes of one ``build_dsp`` call: the default one, the per-channel ones (pattern -> recipe, tried in the order given) and the
    parameter database (channel -> block)."""

    def __init__(self, dsp_config, chan_config, database):
        self.default = _load_config(dsp_config)
        self.by_pattern = [(pattern, _load_config(cfg)) for pattern, cfg in dict(_load_config(chan_config) or {}).items()]
        self.database = _load_config(database)
        if self.database and not isinstance(self.database, Mapping):
            raise ValueError("input database is not a valid JSON or YAML file or dict")

    def recipe_for(self, table: str):
        """the first per-channel recipe whose pattern matches the table's name, else the default (None: the table is skipped)"""
        return next((cfg for pattern, cfg in self.by_pattern if fnmatch(table, pattern)), self.default)

    @staticmethod
    def channel_of(table: str):
        """the first element of the table's path that is not the tier name: 'raw/ch3' and 'ch3/raw' -> 'ch3'; '' and 'raw' -> None"""
        return next((part for part in table.split("/") if part and part != "raw"), None)

    def database_for(self, table: str):
        channel = self.channel_of(table)
        if channel is None:  # a table that is not one channel's sees the whole database
            return self.database
        return self.database.get(channel) if self.database else None


@dataclass
class RowSelection:
    entry_list: object = None
    entry_mask: object = None
    i_start: int = 0
    n_entries: int | None = None

    def of(self, n_all: int):
        """-> (selector for a table of n_all rows, the table row of the first selected one or None when rows are picked by index)"""
        if self.entry_list is not None or self.entry_mask is not None:
            picked = np.asarray(self.entry_list) if self.entry_list is not None else np.flatnonzero(np.asarray(self.entry_mask))
            picked = picked[self.i_start:]
            return (picked if self.n_entries is None else picked[:self.n_entries]), None
        first = min(self.i_start, n_all)
        last = n_all if self.n_entries is None else min(n_all, self.i_start + int(self.n_entries))
        return slice(first, max(first, last)), first

    def iterator_arguments(self) -> dict:
        return {"entry_list": self.entry_list, "entry_mask": self.entry_mask, "i_start": self.i_start, "n_entries": self.n_entries}


# ----------------------------------------------------------------------------------------------------------------------------------
# friends: auxiliary inputs of a recipe ("inputs": {"file", "group", "prefix", "suffix"}), reference build_dsp.py:268-330
# ----------------------------------------------------------------------------------------------------------------------------------
@dataclass
class Friend:
    file: str
    group: str
    prefix: str = ""
    suffix: str = ""


def friends_of(recipe: Mapping, db_block) -> list[Friend]:
    """the auxiliary inputs a recipe declares; ``file`` and ``group`` may be ``db.a.b`` references into the channel's database block"""
    declared = recipe.get("inputs", [])
    if isinstance(declared, Mapping):
        declared = [declared]

    def resolved(text):
        if not (isinstance(text, str) and _DB_REF.fullmatch(text)):
            return text
        level = db_block
        try:
            for part in text.split(".")[1:]:
                level = level[part]
        except (KeyError, TypeError, IndexError):
            raise ProcessingChainError(f"did not find {text} in database.") from None
        return level

    return [Friend(resolved(d["file"]), resolved(d["group"]), d.get("prefix", ""), d.get("suffix", "")) for d in declared]


def _renamed(columns: Mapping, friend: Friend) -> dict:
    return {f"{friend.prefix}{k}{friend.suffix}": v for k, v in columns.items()}


# ----------------------------------------------------------------------------------------------------------------------------------
# several devices
# ----------------------------------------------------------------------------------------------------------------------------------
def _device_list(devices) -> list[int]:
    """``devices`` argument / DSPEED_HIP_DEVICES -> device ordinals; [] = stay on the calling thread's current device"""
    if devices is None:
        env = os.environ.get("DSPEED_HIP_DEVICES", "").strip()
        if not env:
            return []
        devices = list(range(device_count())) if env.lower() == "all" else [int(tok) for tok in env.replace(" ", ",").split(",") if tok]
    elif isinstance(devices, (int, np.integer)):
        devices = [int(devices)]
    devices = [int(d) for d in devices]
    if any(d < 0 for d in devices):
        raise ValueError(f"devices: negative ordinal in {devices}")
    return devices


def shard_bounds(n_rows: int, n_shards: int) -> list[tuple[int, int]]:
    """contiguous, near-equal row ranges; shard k is rows [k*n/G, (k+1)*n/G) -- the partitioning bench.py's ranks use"""
    return [(k * n_rows // n_shards, (k + 1) * n_rows // n_shards) for k in range(n_shards)]


class DeviceTeam:
    """One worker thread per entry of ``devices`` (an ordinal may appear twice: two handles and two streams on one GPU).  A worker is started
    when the team first runs something, makes its device current once and lives until ``close()``: what it creates -- chain handles, streams,
    the page-locked bounce buffer of its synchronous copies (thread-local, ``device._Bounce``) -- is created once per ``build_dsp`` call, and a
    chain is served by the thread that built it."""

    def __init__(self, devices: list[int]):
        self.devices = list(devices)
        self._workers = []  # (thread, queue of (callable(device), slot list, index, done semaphore) or None)

    def __len__(self):
        return max(1, len(self.devices))

    @property
    def parallel(self) -> bool:
        return len(self.devices) > 1

    def _start(self):
        def serve(dev, jobs):
            try:
                set_device(dev)
                failed = None
            except BaseException as e:  # noqa: BLE001 -- every job of this worker reports it
                failed = e
            while True:
                item = jobs.get()
                if item is None:
                    return
                job, results, errors, k, done = item
                try:
                    if failed is not None:
                        raise failed
                    results[k] = job(dev)
                except BaseException as e:  # noqa: BLE001 -- handed to the caller of run()
                    errors[k] = e
                finally:
                    done.release()

        for k, dev in enumerate(self.devices):
            q: queue.Queue = queue.Queue()
            t = threading.Thread(target=serve, args=(dev, q), name=f"dspeed-dev{dev}-{k}", daemon=True)
            t.start()
            self._workers.append((t, q))

    def submit(self, k: int, job, results: list, errors: list, index: int, done: threading.Semaphore) -> None:
        """queue ``job(device)`` on worker k; results[index] / errors[index] receive what it returns / raises, ``done`` is released after it"""
        if not self._workers:
            self._start()
        self._workers[k % len(self._workers)][1].put((job, results, errors, index, done))

    def run(self, jobs: list):
        """jobs[k] = callable(device) run on worker k; returns their results in order; the error of the lowest-numbered failed job is raised
        after every worker has finished (a DSPFatal of an earlier shard is the one the serial loop would have met first)"""
        if not self.devices:
            return [job(None) for job in jobs]
        results, errors = [None] * len(jobs), [None] * len(jobs)
        done = threading.Semaphore(0)
        for k, job in enumerate(jobs):
            self.submit(k, job, results, errors, k, done)
        for _ in jobs:
            done.acquire()
        for e in errors:
            if e is not None:
                raise e
        return results

    def close(self):
        workers, self._workers = self._workers, []
        for _t, q in workers:
            q.put(None)
        for t, _q in workers:
            t.join()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ----------------------------------------------------------------------------------------------------------------------------------
# running one table
# ----------------------------------------------------------------------------------------------------------------------------------
def _set_buffer_len(chain, buffer_len, columns, mask, n_rows):
    """``buffer_len`` rows per transfer -> bytes per pipelined piece of this chain"""
    if buffer_len is None or not n_rows:
        return
    per_row = sum(np.asarray(_values(c)).nbytes // n_rows for k, c in columns.items() if k in mask and not isinstance(_values(c), DeviceArray))
    chain.pipeline_bytes = max(1, int(buffer_len)) * max(per_row, 1)


def _run_array_table(table: Mapping, recipe, db_block, outputs, rows: RowSelection, buffer_len, team: DeviceTeam) -> dict:
    """One table whose columns are arrays: select the rows, build the chain(s), run.  Returns {output: ndarray}."""
    selector, first_row = rows.of(_table_rows(table))
    chosen = {k: _select(v, selector) for k, v in table.items()}
    n_rows = _table_rows(chosen)
    wanted = list(recipe["outputs"] if outputs is None else outputs)

    def run_rows(lo, hi, device):
        part = chosen if (lo, hi) == (0, n_rows) else {k: _select(v, slice(lo, hi)) for k, v in chosen.items()}
        chain, mask, out = build_processing_chain(recipe["processors"], part, db_dict=db_block, outputs=wanted, device=device)
        if hi > lo:
            _set_buffer_len(chain, buffer_len, part, mask, hi - lo)
            try:
                chain.execute(0, hi - lo)
            except DSPFatal as e:
                if isinstance(e.wf_range, range) and first_row is not None:  # rows of the table, not of the shard
                    e.wf_range = f"{first_row + lo + e.wf_range.start}-{first_row + lo + e.wf_range.stop}"
                raise
        return out

    n_shards = min(len(team), n_rows) if team.parallel else 1
    if n_shards <= 1:
        return team.run([lambda dev: run_rows(0, n_rows, dev)])[0] if team.devices else run_rows(0, n_rows, None)
    if any(isinstance(_values(c), DeviceArray) for c in chosen.values()) and len(set(team.devices)) > 1:
        raise NotImplementedError("device-resident columns belong to one GPU: give host arrays to a build_dsp over several devices")
    parts = team.run([lambda dev, lo=lo, hi=hi: run_rows(lo, hi, dev) for lo, hi in shard_bounds(n_rows, n_shards)])
    return {name: np.concatenate([np.asarray(p[name]) for p in parts]) for name in parts[0]}


class _ChunkWorker:
    """the chain of one device for the chunks of one table"""

    def __init__(self, recipe, db_block, outputs, buffer_len):
        self.recipe, self.db_block, self.outputs, self.buffer_len = recipe, db_block, outputs, buffer_len
        self.chain = self.mask = None

    def build(self, columns, device):
        self.chain, self.mask, _ = build_processing_chain(self.recipe["processors"], columns, db_dict=self.db_block, outputs=list(self.outputs),
                                                           device=device)
        return self.mask

    @property
    def attrs(self):
        return self.chain.output_attrs if self.chain is not None else {}

    def process(self, i_entry, n, columns):
        chain = self.chain
        _set_buffer_len(chain, self.buffer_len, columns, self.mask, n)
        out = {name[4:] if name.startswith("out:") else name:
               np.empty((n,) if length is None else (n, length), dtype=getattr(var, "dtype", None) or chain.loop_dtype)
               for name, (var, length) in chain._out_vars.items()}
        try:
            chain(columns, out)
        except DSPFatal as e:
            e.wf_range = f"{i_entry}-{i_entry + n}"  # the position in the file
            raise
        for c in chain._copy_pars:
            if c in columns:
                out[c] = np.asarray(_values(columns[c]))
        # variable-length outputs (declared with vector_len=len(<input>)) leave as VectorOfVectors: padded rows + their lengths
        lens = {k: np.asarray(columns[src]) for k, src in chain.vector_lens.items() if k in out and src in columns}
        return out, lens


def _run_chunks(source, recipe, db_block, outputs, rows: RowSelection, buffer_len, team: DeviceTeam, deliver) -> int:
    """One table given as an LGDO table or an iterator of chunks.  The chains (one per device) are built on the first chunk; then chunks
    are read ahead, dealt to the devices in turn and handed, in file order, to ``deliver(i_entry, n_rows, columns, lengths)``."""
    wanted = recipe["outputs"] if outputs is None else outputs
    workers = [_ChunkWorker(recipe, db_block, wanted, buffer_len) for _ in range(len(team))]
    if lgdo_io.is_chunk_iterator(source):
        if rows.n_entries is not None and hasattr(source, "n_entries"):
            source.n_entries = min(int(rows.n_entries), len(source))
        first = next(iter(source), None)
        if first is None:
            return 0
        first_cols = lgdo_io.table_columns(first)
        masks = team.run([lambda dev, w=w: w.build(first_cols, dev) for w in workers])
        deliver.attrs.update(workers[0].attrs)
        if hasattr(source, "reset_field_mask"):
            source.reset_field_mask(masks[0])
        chunks = lgdo_io.ChunkReader(source, fields=set(masks[0]))
    else:  # one table in memory: a single chunk (split over the devices like an array table)
        cols = lgdo_io.table_columns(source)
        selector, _first = RowSelection(None, None, rows.i_start, rows.n_entries).of(_table_rows(cols))  # (a row range, as the reference slices it)
        cols = {k: _select(v, selector) for k, v in cols.items()}
        n = _table_rows(cols)
        if n == 0:
            return 0
        team.run([lambda dev, w=w: w.build(cols, dev) for w in workers])
        deliver.attrs.update(workers[0].attrs)
        bounds = shard_bounds(n, min(len(workers), n)) if team.parallel else [(0, n)]
        chunks = [(lo, hi - lo, {k: _select(v, slice(lo, hi)) for k, v in cols.items()}) for lo, hi in bounds]
    done = 0
    try:
        if not team.parallel:
            for i_entry, n, cols in chunks:  # (one device: its worker thread, which built the chain, also serves it)
                deliver(i_entry, n, *team.run([lambda _dev, a=(i_entry, n, cols): workers[0].process(*a)])[0])
                done += n
            return done
        # the team's worker of a device (the thread that built its chain) takes that device's chunks one after the other; results come back
        # through one queue and are delivered in file order.  At most two chunks wait per device (read-ahead without holding the whole file).
        finished: queue.Queue = queue.Queue()
        room = [threading.Semaphore(2) for _ in workers]
        sink_r, sink_e, sink_done = [None], [None], threading.Semaphore(0)  # (the jobs report through `finished`)

        def job_of(k, seq, i_entry, n, cols):
            def job(_dev):
                try:
                    finished.put((seq, i_entry, n, workers[k].process(i_entry, n, cols), None))
                except BaseException as e:  # noqa: BLE001
                    finished.put((seq, i_entry, n, None, e))
                finally:
                    room[k].release()
            return job

        waiting, next_seq, sent, failure = {}, 0, 0, None

        def drain(block):
            nonlocal next_seq, done, failure
            while next_seq < sent:
                if next_seq not in waiting:
                    if not block:
                        return
                    seq, i_entry, n, result, err = finished.get()
                    waiting[seq] = (i_entry, n, result, err)
                    continue
                i_entry, n, result, err = waiting.pop(next_seq)
                next_seq += 1
                if err is not None:
                    failure = failure or err
                elif failure is None:
                    try:  # (a failing delivery -- a write error -- ends the table like a failing chunk: later chunks are collected, not written)
                        deliver(i_entry, n, *result)
                        done += n
                    except BaseException as e:  # noqa: BLE001
                        failure = e

        try:
            for item in chunks:
                if failure is not None:
                    break
                k = sent % len(workers)
                while not room[k].acquire(timeout=0.05):
                    drain(block=False)
                team.submit(k, job_of(k, sent, *item), sink_r, sink_e, 0, sink_done)
                sent += 1
                while not finished.empty():
                    seq, i_entry, n, result, err = finished.get()
                    waiting[seq] = (i_entry, n, result, err)
                drain(block=False)
        finally:
            drain(block=True)
        if failure is not None:
            raise failure
    finally:
        if isinstance(chunks, lgdo_io.ChunkReader):
            chunks.close()
    return done


# ----------------------------------------------------------------------------------------------------------------------------------
# sources
# ----------------------------------------------------------------------------------------------------------------------------------
class _ArraySource:
    """tables whose columns are arrays: one table, a mapping of tables, or an ``.npz`` file"""

    def __init__(self, raw_in, lh5_tables, base_group):
        self.single = False
        self.label = raw_in if isinstance(raw_in, str) else "raw_in"
        if isinstance(raw_in, str):
            self.tables = _read_npz(raw_in)
        elif _is_table(raw_in):
            if lh5_tables is not None and len(lh5_tables) > 1:
                raise RuntimeError("Cannot have more than one value in lh5_tables for input of type Table")
            self.tables, self.single = {(lh5_tables[0] if lh5_tables else ""): raw_in}, True
        elif isinstance(raw_in, Mapping) and all(_is_table(t) for t in raw_in.values()):
            self.tables = dict(raw_in)
        else:
            raise RuntimeError(f"raw_in was not a file name, a table or a mapping of tables: {type(raw_in).__name__}")
        self.names = list(self.tables)
        if self.single:
            return
        if base_group is None:
            base_group = "raw" if any(n == "raw" or n.startswith("raw/") for n in self.names) else ""
        inside = [n for n in self.names if not base_group or n == base_group or n.startswith(base_group + "/")]
        if lh5_tables is not None:
            below = {n: (n[len(base_group) + 1:] if base_group and n.startswith(base_group + "/") else n) for n in inside}
            inside = list(dict.fromkeys(n for pattern in lh5_tables for n in inside if fnmatch(below[n], pattern) or fnmatch(n, pattern)))
        if not inside:
            raise RuntimeError(f"could not find any valid table in {self.label}")
        self.names = inside

    def with_friends(self, name, friends: list[Friend]):
        table = dict(self.tables[name])
        for fr in friends:
            other = lgdo_io.open_friend(fr.file, fr.group, n_rows=_table_rows(table))
            if lgdo_io.is_lgdo_table(other):
                other = lgdo_io.table_columns(other)
            table.update(_renamed(other, fr))
        return table


class _ChunkSource:
    """an LGDO table in memory, a chunk iterator, or an LH5 file (that one needs the lgdo package)"""

    def __init__(self, raw_in, lh5_tables, base_group, rows: RowSelection, buffer_len):
        self.raw_in, self.rows, self.buffer_len = raw_in, rows, buffer_len
        self.from_file = isinstance(raw_in, str)
        self.lh5 = None
        if self.from_file:
            _, self.lh5 = lgdo_io.require_lh5(f"reading '{raw_in}'")
            self.names = self._tables_of_file(raw_in, lh5_tables, base_group)
        else:
            if lh5_tables is not None and len(lh5_tables) > 1:
                raise RuntimeError("Cannot have more than one value in lh5_tables for input of type Table or LH5Iterator")
            self.names = [lh5_tables[0] if lh5_tables else ""]  # (only names the output group)

    def _tables_of_file(self, path, patterns, base_group):
        ls = self.lh5.ls
        if base_group is None:
            base_group = "raw" if ls(path, "raw") else ""
        candidates = ls(path, f"{base_group}/*") if patterns is None else [hit for pattern in patterns for hit in ls(path, f"{base_group}/{pattern}")]
        found = []
        for group in candidates:
            nested = f"{group}/raw"  # the tier is sometimes below the channel: ch024/raw
            if ls(path, f"{group}/*") == [nested]:
                found.append(nested)
            elif ls(path, group):
                found.append(group)
        if not found:
            raise RuntimeError(f"could not find any valid LH5 table in {path}")
        return found

    def open(self, name, friends: list[Friend]):
        if self.from_file:
            source = self.lh5.LH5Iterator(self.raw_in, name, buffer_len=self.buffer_len if self.buffer_len is not None else 32768,
                                          **self.rows.iterator_arguments())  # (rows per read: a chunk is one execute)
        else:
            source = self.raw_in
        for fr in friends:
            if lgdo_io.is_chunk_iterator(source):
                if not hasattr(source, "add_friend"):
                    raise NotImplementedError("auxiliary 'inputs' need an iterator with add_friend (lh5.LH5Iterator)")
                other = lgdo_io.open_friend(fr.file, fr.group, iterator=True, buffer_len=getattr(source, "buffer_len", self.buffer_len),
                                            **self.rows.iterator_arguments())
                source.add_friend(other, prefix=fr.prefix, suffix=fr.suffix)
            else:
                other = lgdo_io.open_friend(fr.file, fr.group, n_rows=len(source))
                if hasattr(source, "join"):
                    source.join(other, prefix=fr.prefix, suffix=fr.suffix)
                else:
                    source = type(source)({**{k: source[k] for k in source.keys()}, **_renamed({k: other[k] for k in other.keys()}, fr)})
        return source


# ----------------------------------------------------------------------------------------------------------------------------------
# sinks
# ----------------------------------------------------------------------------------------------------------------------------------
def _check_output_file(dsp_out, write_mode):
    if write_mode not in (None, "r", "a", "u"):
        raise ValueError("write_mode must be None, 'r', 'a' or 'u'")
    if dsp_out is not None and write_mode is None and os.path.isfile(dsp_out):
        raise FileExistsError(f"output file {dsp_out} exists. Set the 'write_mode' keyword")


class _MemorySink:
    """results stay in memory: {table: {column: ndarray}} (+ per-row lengths of variable-length columns)"""

    def __init__(self):
        self.tables, self.lengths, self.attrs = {}, {}, {}

    def table_done(self, name, columns, lengths=None, attrs=None):
        self.tables[name] = columns
        if lengths:
            self.lengths[name] = lengths
        if attrs:
            self.attrs[name] = attrs

    def chunk_writer(self, name):
        parts, lens = [], []

        def deliver(i_entry, n, columns, lengths):
            parts.append(columns)
            lens.append(lengths)

        deliver.attrs = {}  # output -> attributes of its column: filled in once the chain is built

        def close():
            merged = {k: np.concatenate([p[k] for p in parts]) for k in (parts[0] if parts else [])}
            self.table_done(name, merged, {k: np.concatenate([p[k] for p in lens]) for k in (lens[0] if lens else {})}, deliver.attrs)

        return deliver, close


class _NpzSink(_MemorySink):
    """an ``.npz`` laid out like an LH5 file: 'a' appends rows to the columns the file has, 'u' replaces them, 'r' starts a new file"""

    def __init__(self, path, write_mode):
        super().__init__()
        self.path, self.mode = path, write_mode

    def finish(self):
        flat = {f"{t}/{k}" if t else k: np.asarray(v) for t, cols in self.tables.items() for k, v in cols.items()}
        if self.mode in ("a", "u") and os.path.isfile(self.path):
            with np.load(self.path) as z:
                merged = {k: z[k] for k in z.files}
            for k, v in flat.items():
                merged[k] = np.concatenate([merged[k], v]) if (self.mode == "a" and k in merged) else v
            flat = merged
        tmp = self.path + ".tmp.npz"
        np.savez(tmp, **flat)
        os.replace(tmp, self.path)


class _Lh5Sink:
    """rows go to an LH5 file as they finish (``LH5Store.write`` with the reference's ``wo_mode`` / ``write_start``, build_dsp.py:416-424)"""

    def __init__(self, path, write_mode, i_start):
        _, lh5 = lgdo_io.require_lh5(f"writing '{path}'")
        if write_mode == "r" and os.path.isfile(path):
            os.remove(path)
        self.path, self.i_start = path, i_start
        self.wo_mode = "o" if write_mode == "u" else "a"
        self.store = lh5.LH5Store(keep_open=True)

    def chunk_writer(self, name):
        def deliver(i_entry, n, columns, lengths):
            self.store.write(obj=lgdo_io.results_table(columns, lengths=lengths, attrs=deliver.attrs), name=name, lh5_file=self.path,
                             wo_mode=self.wo_mode, write_start=self.i_start + i_entry, n_rows=n)

        deliver.attrs = {}
        return deliver, (lambda: None)

    def table_done(self, name, columns, lengths=None):
        n = len(next(iter(columns.values()))) if columns else 0
        self.chunk_writer(name)[0](0, n, columns, lengths or {})


# ----------------------------------------------------------------------------------------------------------------------------------
def build_dsp(raw_in, dsp_out: str | None = None, dsp_config=None, lh5_tables=None, base_group: str | None = None, database=None,
              outputs: Collection[str] | None = None, write_mode: str | None = None, entry_list=None, entry_mask=None, i_start: int = 0,
              n_entries: int | None = None, buffer_len: int | None = None, block_width: int = 16, chan_config=None, devices=None):
    """Run recipes over tables of waveforms; returns ``{dsp table name: {parameter: ndarray}}`` (one table: the table itself) when
    ``dsp_out`` is None, else writes them to the file and returns None.  Parameters as in the reference (build_dsp.py:27-127), plus
    ``devices``: the GPUs to spread the rows of every table over (default: the current one; environment DSPEED_HIP_DEVICES)."""
    del block_width  # (the device processes whole buffers)
    if isinstance(lh5_tables, str):
        lh5_tables = [lh5_tables]
    rows = RowSelection(entry_list, entry_mask, int(i_start), n_entries)
    chunked = (isinstance(raw_in, str) and raw_in.lower().endswith((".lh5", ".h5", ".hdf5"))) or lgdo_io.is_chunk_iterator(raw_in) or \
        lgdo_io.is_lgdo_table(raw_in)
    source = _ChunkSource(raw_in, lh5_tables, base_group, rows, buffer_len) if chunked else _ArraySource(raw_in, lh5_tables, base_group)
    book = RecipeBook(dsp_config, chan_config, database)
    _check_output_file(dsp_out, write_mode)
    team = DeviceTeam(_device_list(devices))
    if dsp_out is None:
        sink = _MemorySink()
    elif str(dsp_out).endswith(".npz"):
        if write_mode == "r" and os.path.isfile(dsp_out):
            os.remove(dsp_out)
        sink = _NpzSink(dsp_out, write_mode)
    else:
        sink = _Lh5Sink(dsp_out, write_mode, rows.i_start)

    try:
        for name in source.names:
            recipe = book.recipe_for(name)
            if recipe is None:  # (dsp_config may be None with chan_config: channels without a match are skipped)
                continue
            db_block = book.database_for(name)
            friends = friends_of(recipe, db_block)
            dsp_name = name.replace("raw", "dsp")
            if chunked:
                deliver, close = sink.chunk_writer(dsp_name)
                _run_chunks(source.open(name, friends), recipe, db_block, outputs, rows, buffer_len, team, deliver)
                close()
            else:
                table = source.with_friends(name, friends) if friends else source.tables[name]
                sink.table_done(dsp_name, _run_array_table(table, recipe, db_block, outputs, rows, buffer_len, team))
    finally:
        team.close()

    if isinstance(sink, _Lh5Sink):
        return None
    if isinstance(sink, _NpzSink):
        sink.finish()
        return None
    if chunked:
        tabs = {k: lgdo_io.results_table(v, lengths=sink.lengths.get(k), attrs=sink.attrs.get(k)) for k, v in sink.tables.items()}
        return tabs[next(iter(tabs))] if not source.from_file and tabs else tabs
    return sink.tables[next(iter(sink.tables))] if source.single and sink.tables else sink.tables
