"""Shared helpers: load a golden fixture and replay it through a stepper."""
import glob
import json
import os

import numpy as np

from nuclear_sim_amd.schema import SCHEMA

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# state columns excluded from parity (none: every schema column is checked)
EXEMPT_PREFIXES = ()
# fp64 tolerance of the parity contract (BASELINE.json north_star: 1e-6 relative on fp64 state)
RTOL = 1e-6
# the CPU restatement against the reference's fixtures: both are IEEE fp64 evaluations of the same expressions in the same order, so they
# agree far below the contract (measured over every fixture: worst column 3e-12; libm's pow / log10 against numpy's are the residual).
# The oracle tests hold it to THIS, not to 1e-6: a restatement error that moves a slow integrator's increment by a thousandth changes
# the state by 1e-9 of itself per step and would pass at 1e-6 (tools/mutate_oracle.py found exactly those survivors)
ORACLE_RTOL = float(os.environ.get("NPB_ORACLE_RTOL", "1e-10"))
# ... and most columns it reproduces to the BIT (tests/oracle_column_error.py: 645 of 810 fp64 columns identical in every sample of
# every fixture): those are held to two ulps when a caller asks for the oracle's tolerance
try:
    _exact = json.load(open(os.path.join(GOLDEN_DIR, "oracle_exact_columns.json")))
    ORACLE_EXACT_COLUMNS = frozenset(_exact["bit_identical"]); ORACLE_EXACT_NOT_ON = frozenset(_exact.get("not_held_on", ()))
except OSError:
    ORACLE_EXACT_COLUMNS = frozenset(); ORACLE_EXACT_NOT_ON = frozenset()
# absolute floor under the relative tolerance: far below any column's working magnitude (deposit layers start at 1e-11 mm: with the 1e-12 of
# rounds 1-3 a 0.1 % error in their growth rate was invisible -- found by tools/mutate_device.py) ...
ATOL_SMALL = float(os.environ.get("NPB_ATOL_SMALL", "1e-18"))
# ... except for the two columns that are differences of nearly equal numbers (a steam generator's TSP fouling fraction = 1 - area ratio ~ 1e-8
# and the heat-transfer degradation computed from it): one ulp of the 1.0 is 1e-8 of them, so the ORACLE's 1e-10 needs a floor there
CANCELLATION_COLUMNS = ("tsp_fouling_fraction", "tsp_ht_degradation")
CANCELLATION_FLOOR = 1e-15


def fixture_names():
    """trajectory fixtures (ic_*.npz hold initial states only, tests/test_scenarios.py)"""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith(("ic_", "log_", "counts_"))]     # log_*: the reference's own state log of a fixture's run


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.name = name
        self.meta = json.loads(str(z["meta"]))
        for k in ("action", "magnitude", "setpoint", "cooling", "noise_z", "obs", "reward", "done", "info",
                  "state_steps", "labels", "kinds", "paths"):
            setattr(self, k, z[k])
        self.T = len(self.action)
        # columns are matched by the reference attribute path, so a schema re-ordering does not
        # invalidate the fixtures; schema columns the fixture does not know become NaN (unchecked)
        cols = SCHEMA.columns()
        idx = {str(p): j for j, p in enumerate(self.paths) if str(p)}
        raw = z["state"]
        st = np.full((raw.shape[0], len(cols)), np.nan)
        for j, c in enumerate(cols):
            if c[3] in idx:
                st[:, j] = raw[:, idx[c[3]]]
        self.state = st
        self.cols = cols
        # the scalar keys of info["secondary_system"] per step (fixtures made before they were recorded have none)
        self.sec_keys = [str(k) for k in z["sec_keys"]] if "sec_keys" in z.files else []
        self.sec = z["sec"] if "sec" in z.files else None
        # info["reactivity_components"] per step, in the reference dict's own order (reactor heat source; newer fixtures)
        self.rc_keys = [str(k) for k in z["rc_keys"]] if "rc_keys" in z.files else []
        self.rc = z["rc"] if "rc_keys" in z.files and len(z["rc_keys"]) else None
        # NuclearPlantSimulator.reset() calls recorded in the run: {step: (start_at_steady_state, obs, state row)}
        self.resets = {}
        if "reset_steps" in z.files:
            for k, t in enumerate(z["reset_steps"]):
                rs = np.full(len(cols), np.nan)
                for j, c in enumerate(cols):
                    if c[3] in idx:
                        rs[j] = z["reset_state"][k, idx[c[3]]]
                self.resets[int(t)] = (bool(z["reset_modes"][k]), z["reset_obs"][k], rs)
        self.pokes = {int(k): v for k, v in self.meta.get("pokes_schema", {}).items()}

    def split_state(self, row):
        """fixture state row -> (f64[total_f64], i32[total_i32]); NaN (no reference leaf) -> None mask."""
        f = np.zeros(SCHEMA.total_f64); i = np.zeros(SCHEMA.total_i32, dtype=np.int64)
        fm = np.zeros(SCHEMA.total_f64, dtype=bool); im = np.zeros(SCHEMA.total_i32, dtype=bool)
        for (kind, slot, label, _p), v in zip(self.cols, row):
            if np.isnan(v):
                continue
            if kind == "f64":
                f[slot] = v; fm[slot] = True
            else:
                i[slot] = int(v); im[slot] = True
        return f, i, fm, im

    def label_slot(self, path):
        """(kind, slot) of the schema column whose reference attribute path is `path`."""
        for kind, slot, _lab, p in self.cols:
            if p == path:
                return kind, slot
        raise KeyError(path)


def compare_state(g, f64, i32, row, where, rtol=None, loose=()):
    """Assert a stepper's (f64, i32) state against a fixture row (rtol: default = the parity contract's RTOL; loose: label
    prefixes held to RTOL whatever rtol says)."""
    rtol = RTOL if rtol is None else rtol
    bad = []
    # fixtures of the other action-test scenarios: the reference's maintenance control plane raises work orders of its
    # own there (action types no component knows; no plant state changes), which the restated rule does not produce
    skip_maint = bool(g.meta.get("maint_unchecked"))
    for (kind, slot, label, _p), v in zip(g.cols, row):
        if np.isnan(v) or (EXEMPT_PREFIXES and label.startswith(EXEMPT_PREFIXES)) or (skip_maint and label.startswith("maint.")):
            continue
        if kind == "i32":
            if int(i32[slot]) != int(v):
                bad.append((label, int(i32[slot]), int(v)))
        else:
            mine = float(f64[slot])
            floor = CANCELLATION_FLOOR if label.endswith(CANCELLATION_COLUMNS) else ATOL_SMALL
            if loose and label.startswith(loose):
                tol = RTOL * abs(v) + floor
            elif rtol < RTOL and label in ORACLE_EXACT_COLUMNS and g.name not in ORACLE_EXACT_NOT_ON:
                tol = 4.5e-16 * abs(v)         # the oracle against the reference on a column it reproduces bit for bit
            else:
                tol = rtol * abs(v) + floor
            if not (abs(mine - v) <= tol):
                bad.append((label, mine, float(v)))
    assert not bad, "%s %s: %d mismatching columns, first: %s" % (g.name, where, len(bad), bad[:5])


class Config4Counts:
    """tests/golden/counts_c4_64seeds.npz (oracle/ref_harness/make_golden.py make_c4_counts): 64 simulators as the data-gen
    runner builds them for the randomised oil_top_off action test, run by the REFERENCE for 48 steps of 5 minutes -- BASELINE
    config 4's per-plant quantities: work orders created / maintenance actions performed after every step, and the initial
    and final value of every schema column (oil levels, executions by action, open orders, cooldown stamps)."""

    def __init__(self):
        z = np.load(os.path.join(GOLDEN_DIR, "counts_c4_64seeds.npz"), allow_pickle=False)
        self.name = "counts_c4_64seeds"
        self.meta = json.loads(str(z["meta"]))
        self.seeds = [int(s) for s in z["seeds"]]
        self.setpoint, self.noise_z = z["setpoint"], z["noise_z"]
        self.created, self.performed = z["created"], z["performed"]
        self.final_obs = z["final_obs"]
        self.T = self.setpoint.shape[1]
        self.cols = SCHEMA.columns()
        idx = {str(p): j for j, p in enumerate(z["paths"]) if str(p)}

        def rows(raw):
            st = np.full((raw.shape[0], len(self.cols)), np.nan)
            for j, c in enumerate(self.cols):
                if c[3] in idx:
                    st[:, j] = raw[:, idx[c[3]]]
            return st
        self.initial_state, self.final_state = rows(z["initial_state"]), rows(z["final_state"])
