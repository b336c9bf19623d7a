"""Columnar state log of a batch.

The reference's StateManager appends one pandas row per plant-step -- `collect_states` walks every registered
provider's `get_state_dict()` and `pd.concat`s the row (simulator/state/state_manager.py:152-233), 790 columns named
`category.variable` (auto_register.py:83-165); with physics on a GPU that bookkeeping would be the whole run time.
Here a sample is one gather kernel (`npb_gather_fields`): the chosen members of every plant, widened to double, land
in a device buffer `[sample, field, plant]`; nothing touches the host until `table()` / `write_parquet()`.

Columns carry the reference's own log names: `state_names.json` (made by running the reference through three eventful
runs and matching whole series value for value; the generator script is named in DESIGN.md section 6) maps 265 of the reference's 784
numeric log columns onto 193 state members -- several log columns can show one member (the reference logs the total
feedwater flow three times), a few through a unit factor.  `StateLog(env)` without a field list records exactly those
members and `table()` emits every one of the 265 columns; members selected by name that the reference does not log come
out as `npb.<section>.<member>`.  Not reproduced: the reference's derived diagnostics (per-stage turbine conditions, SG
capacities and heat fluxes, system averages ...: 251 columns) and the 268 columns that never vary in any of the runs.

    log = StateLog(env, fields=["pump.oil_level", "sec.electrical_power_output"], every=12, capacity=64)
    for t in range(steps):
        env.step(...); log.maybe_record(t + 1, time_minutes=(t + 1) * env.dt)
    log.write_parquet("run.parquet")          # long format: one row per (sample, plant)
"""
from __future__ import annotations

import ctypes
import json
import os
import re
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .schema import SCHEMA

_NAMES_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "state_names.json")


def reference_names() -> Dict[str, str]:
    """schema label -> ONE of the reference's state-log column names, for the members that are logged there."""
    with open(_NAMES_PATH) as fh:
        return json.load(fh)["names"]


def reference_log_columns() -> Dict[str, tuple]:
    """the reference's log column -> (schema label, factor): every column of its log that is a state member (times a unit factor)"""
    with open(_NAMES_PATH) as fh:
        return {k: (v[0], float(v[1])) for k, v in json.load(fh)["log_columns"].items()}


def log_columns(fields: Optional[Sequence[str]] = None) -> List[tuple]:
    """[(kind, slot, schema label, log column name)] of the members a log would hold.  `fields`: schema names
    ("pump.oil_level" = every instance and element, or a full label such as "pump[2].oil_level"); None = every
    member the reference itself logs."""
    names = reference_names()
    cols = SCHEMA.columns()
    out = []
    if fields is None:
        wanted = {label for label, _f in reference_log_columns().values()}
        for kind, slot, label, _path in cols:
            if label in wanted:
                out.append((kind, slot, label, names.get(label, "npb." + label)))
        return out
    for want in fields:
        hit = False
        for kind, slot, label, _path in cols:
            if want == label or want == re.sub(r"\[\d+\]", "", label):
                out.append((kind, slot, label, names.get(label, "npb." + label)))
                hit = True
        if not hit:
            raise KeyError("no state member named %r" % want)
    return out


class StateLog:
    """Device-resident ring of samples of selected state members of every plant."""

    def __init__(self, env, fields: Optional[Sequence[str]] = None, every: int = 1, capacity: int = 256):
        self.env = env
        self.columns = log_columns(fields)
        # with no field list the table carries the reference's log columns (several per member, unit factors applied)
        self._reference_layout = fields is None
        if not self.columns:
            raise ValueError("no columns to log")
        self.every = max(1, int(every))
        self.capacity = int(capacity)
        nf = len(self.columns)
        self._kinds = (ctypes.c_int * nf)(*[0 if c[0] == "f64" else 1 for c in self.columns])
        self._slots = (ctypes.c_int * nf)(*[c[1] for c in self.columns])
        self._buf = torch.empty((self.capacity, nf, env.n), dtype=torch.float64, device=env.device)
        self._times: List[float] = []
        self._steps: List[int] = []

    def __len__(self) -> int:
        return len(self._times)

    def record(self, step: int, time_minutes: float) -> None:
        """Sample now (one kernel launch on the env's stream)."""
        row = len(self._times)
        if row >= self.capacity:
            raise RuntimeError("StateLog is full (%d samples): flush it with table() / write_parquet() and clear()" % self.capacity)
        out = self._buf[row]
        _lib.check(self.env.L.npb_gather_fields(self.env._h, len(self.columns), self._kinds, self._slots,
                                                ctypes.c_void_p(out.data_ptr()), self.env._stream()), self.env._h)
        self._times.append(float(time_minutes)); self._steps.append(int(step))

    def maybe_record(self, step: int, time_minutes: float) -> bool:
        if step % self.every:
            return False
        self.record(step, time_minutes)
        return True

    def clear(self) -> None:
        self._times.clear(); self._steps.clear()

    def array(self) -> np.ndarray:
        """[samples, fields, plants] on the host."""
        return self._buf[:len(self._times)].cpu().numpy()

    def table(self, plants: Optional[Sequence[int]] = None):
        """pyarrow Table in long format: step, time (minutes), plant, then one column per member."""
        import pyarrow as pa
        data = self.array()
        idx = np.arange(self.env.n) if plants is None else np.asarray(plants)
        data = data[:, :, idx]
        ns, nf, npl = data.shape
        cols = {"step": np.repeat(np.asarray(self._steps, dtype=np.int64), npl),
                "time": np.repeat(np.asarray(self._times, dtype=np.float64), npl),
                "plant": np.tile(idx.astype(np.int64), ns)}
        if self._reference_layout:
            index = {c[2]: f for f, c in enumerate(self.columns)}
            for name, (label, factor) in sorted(reference_log_columns().items()):
                v = data[:, index[label], :].reshape(-1)
                cols[name] = v * factor if factor != 1.0 else v
            return pa.table(cols)
        for f, (kind, _slot, _label, name) in enumerate(self.columns):
            v = data[:, f, :].reshape(-1)
            cols[name] = v.astype(np.int32) if kind == "i32" else v
        return pa.table(cols)

    def write_parquet(self, path: str, plants: Optional[Sequence[int]] = None) -> None:
        import pyarrow.parquet as pq
        pq.write_table(self.table(plants), path)
