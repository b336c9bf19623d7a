"""Host-side mirror of the reference's simulator interface on top of the HIP stepper.

``BatchedPlantEnv`` steps N independent plants per call (one wavefront lane per plant);
``NuclearPlantSimulator`` / ``NuclearPlantEnv`` are single-plant facades with the reference's
scalar signatures (simulator/core/sim.py:27-258, 911-940) so that existing loops
(maintenance_scenario_runner.py:383-411, data/gen_training_data.py:249-290) run unchanged.

PyTorch is only the device-array container: every tensor handed to the library is passed as
a raw device pointer (``tensor.data_ptr()``).
"""
from __future__ import annotations

import ctypes
import os
import enum
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .schema import SCHEMA


class ControlAction(enum.Enum):
    """systems/primary/__init__.py:28-45"""
    CONTROL_ROD_INSERT = 0
    CONTROL_ROD_WITHDRAW = 1
    INCREASE_COOLANT_FLOW = 2
    DECREASE_COOLANT_FLOW = 3
    OPEN_STEAM_VALVE = 4
    CLOSE_STEAM_VALVE = 5
    INCREASE_FEEDWATER = 6
    DECREASE_FEEDWATER = 7
    NO_ACTION = 8
    DILUTE_BORON = 9
    BORATE_COOLANT = 10
    START_FEEDWATER_PUMP = 11
    STOP_FEEDWATER_PUMP = 12
    INCREASE_FEEDWATER_PUMP_SPEED = 13
    DECREASE_FEEDWATER_PUMP_SPEED = 14


INFO_COLUMNS = ("thermal_power", "reactivity", "electrical_power", "thermal_efficiency", "steam_flow",
                "steam_pressure", "condenser_pressure", "condenser_heat_rejection", "time", "feedwater_flow",
                "sg_heat_transfer", "turbine_power", "feedwater_power", "primary_thermal_power",
                "turbine_efficiency", "turbine_hp_power", "turbine_lp_power")


assert len(INFO_COLUMNS) == _lib.INFO_DIM     # the info block's width, checked against the library itself in _lib.load()


class HeatSourceNoise:
    """Per-plant pre-drawn heat-source noise, bit-identical to the reference's
    ``np.random.RandomState(seed).normal(0, sigma)`` stream (constant_heat_source.py:58-62,178):
    numpy's legacy normal is loc + scale * gauss, so the standard-normal stream of the same
    RandomState reproduces it exactly.  Drawn on the host a block of steps at a time and uploaded as ONE
    [block, n] tensor, so a step costs no host-to-device copy (BASELINE config 3 seeds every plant differently,
    42 + i: 65 536 generators; the draw is ~8 s per 256 steps on one core and happens once per block)."""

    def __init__(self, seeds: Sequence[int], block: int = 256, device=None):
        # one generator per DISTINCT seed (the data-gen runner seeds every plant's heat source with 42, so a
        # 262 144-plant batch needs one stream, not 262 144 generator objects)
        uniq, self._index = np.unique(np.asarray(seeds, dtype=np.int64), return_inverse=True)
        self._rngs = [np.random.RandomState(int(s)) for s in uniq]
        self._block = block
        self._device = device
        self._buf = None
        self._pos = block

    def next(self):
        if self._pos >= self._block:
            draws = np.stack([r.standard_normal(self._block) for r in self._rngs])          # [distinct seeds, block]
            host = np.ascontiguousarray(draws[self._index].T)                                 # [block, n]
            self._buf = host if self._device is None else torch.from_numpy(host).to(self._device)
            self._pos = 0
        out = self._buf[self._pos]
        self._pos += 1
        return out


def equilibrium_state(power_level=100.0, control_rod_position=95.0) -> Dict[str, object]:
    """create_equilibrium_state(power_level, control_rod_position, auto_balance=True)  reactivity_model.py:443-529, as a field
    dict for ``BatchedPlantEnv.set_fields``.  Scalars, or arrays [n] for one state per plant (BASELINE config 2: power ~ U[60, 100] %,
    rods ~ U[80, 100] % per plant): temperatures and precursors follow the asked power level, the boron concentration is the
    critical one for the asked rod position at those temperatures (calculate_critical_boron_concentration :390-416 on
    calculate_total_reactivity :77-125 with boron = 0) -- and the last two lines of the reference put flux and power level back to
    exactly 100 % whatever was asked (:518-520), which is reproduced."""
    p = np.asarray(power_level, dtype=np.float64); rods = np.asarray(control_rod_position, dtype=np.float64)
    p, rods = np.broadcast_arrays(p, rods)
    beta = [0.000215, 0.001424, 0.001274, 0.002568, 0.000748, 0.000273]
    lam = [0.077, 0.311, 1.40, 3.87, 1.40, 0.195]
    flux = 1e13 * (p / 100.0)
    prec = [(beta[i] / lam[i]) * (flux / 1e-5) for i in range(6)]
    fuel_t = 575.0 + (p - 100.0) * 2.0
    cool_t = 293.0 + (17.0 * (p / 100.0))
    # total reactivity without boron (ReactivityModel.calculate_total_reactivity, boron = 0), the dict's ten terms in their order
    pos = np.minimum(np.maximum(rods / 100.0, 0.0), 1.0)
    comps = [3000.0 * (pos - 0.5), -10.0 * 0.0, -2.5e-5 * (fuel_t - 575.0) * 1e5, -3.0e-5 * (cool_t - 280.0) * 1e5,
             -1000.0 * 0.0, 0.5 * (15.5 - 15.5), (1.0e15 / 1.0e15) * -1800.0, (5.0e14 / 5.0e14) * -600.0,
             3340.0 + -0.15 * 15000.0, 0.0 * float(np.exp(-0.0002 * 15000.0))]
    total = 0
    for c in comps:
        total = total + c
    boron = np.maximum(0, (total - 0.0) / -10.0)
    one = np.ones_like(p)
    d = {"prim.neutron_flux": 1e13 * one, "prim.power_level": 100.0 * one, "prim.control_rod_position": rods * one,
         "prim.xenon_concentration": 1.0e15 * one, "prim.iodine_concentration": 1.5e16 * one,
         "prim.samarium_concentration": 5.0e14 * one, "prim.fuel_temperature": fuel_t,
         "prim.coolant_temperature": cool_t, "prim.boron_concentration": boron}
    for i in range(6):
        d[("prim.precursors", 0, i)] = prec[i]
    if p.ndim == 0:
        d = {k: float(v) for k, v in d.items()}
    return d


def config2_draws(n: int):
    """BASELINE config 2's per-plant (power level, rod position) draws: U[60, 100] % and U[80, 100] % from
    numpy.random.default_rng(1234) -- one generator per quantity (1234, 1235), each indexed by plant, so that plant i's state does not
    depend on how many plants the batch has"""
    return np.random.default_rng(1234).uniform(60.0, 100.0, n), np.random.default_rng(1235).uniform(80.0, 100.0, n)


class BatchedPlantEnv:
    """N plants behind the reference's step()/reset()/observation contract, as columns.

    step() returns ``(obs[N,22], reward[N], done[N], info)`` where ``info`` holds column tensors
    (``electrical_power``, ``trip_flags``, ...) instead of one dict per plant.

    Every tensor step() hands out -- obs, reward, done and each column of info, ``info["maintenance_event_count"]`` included (the
    column the step kernels keep current, npb_set_maintenance_count_buffer) -- is one of the env's own device buffers, written again
    by the next step: ``clone()`` what is to be kept (a list that appends them step after step holds N aliases of the latest values).
    """

    action_space_size = 15       # NuclearPlantEnv sim.py:916
    observation_space_size = 22  # sim.py:917-918

    def __init__(self, n_envs: int, dt: float = 1.0, heat_source: str = "constant", noise_enabled: bool = False,
                 noise_std_percent: float = 0.1, noise_seeds: Optional[Sequence[int]] = None,
                 mode: str = "full", device: int = 0, params: Optional[dict] = None, maintenance: bool = False,
                 storage: str = "f64", maintenance_thresholds: Optional[dict] = None, reactivity_components: bool = False,
                 integrator: str = "reference"):
        if not torch.cuda.is_available():
            raise _lib.NpbError("BatchedPlantEnv needs a HIP device (torch.cuda.is_available() is False); "
                                "there is no CPU fallback")
        self.L = _lib.load()
        self.n = int(n_envs)
        self.device = torch.device("cuda", device)
        p = _lib.default_params()
        p.dt = float(dt)
        # "external": a heat source the caller computes (the reference's HeatSource plugin interface, heat_source_interface.py:23-112):
        # step(thermal_power_mw=..., power_percent=...) takes the plugin's result as two input columns (include/npb_params.h)
        p.heat_source = {"constant": _lib.HEAT_CONSTANT, "reactor": _lib.HEAT_REACTOR, "external": _lib.HEAT_EXTERNAL}[heat_source]
        self.heat_source = heat_source
        p.hs_noise_enabled = int(bool(noise_enabled))
        p.hs_noise_std_percent = float(noise_std_percent)
        # "primary": NuclearPlantSimulator(enable_secondary=False) -- the primary side alone, obs[:, :12] (sim.py:155,333)
        p.mode = {"full": _lib.MODE_FULL, "primary_sg": _lib.MODE_PRIMARY_SG, "primary": _lib.MODE_PRIMARY}[mode]
        self.mode = mode
        # info["reactivity_components"] (sim.py:205) exists under the reactor heat source only; asked for, the step writes the
        # ten terms behind the info columns (include/npb.h NPB_RHO_*)
        # integrator="rk4" (BASELINE config 2; reactor heat source): the point-kinetics equations by fourth-order Runge-Kutta sub-steps
        # of 2 ms inside the step kernel instead of the reference's clipped explicit Euler -- no reference counterpart.  Classical
        # explicit RK4 while it is stable on the plant's prompt mode (h |rho - beta| / Lambda < 2, i.e. above about -350 pcm); a
        # plant below that -- deep rod insertion, every scram -- takes the L-stable implicit method of the same order for that
        # step (npd_primary.h), so the mode is stable over the whole clipped reactivity range [-0.9, 0.1].
        p.kinetics_rk4_substeps = {"reference": 0, "rk4": max(1, int(np.ceil(float(dt) / 0.002)))}[integrator]
        self._with_rho = bool(reactivity_components) and heat_source == "reactor"
        p.info_reactivity_components = int(self._with_rho)
        # automatic oil_top_off maintenance after every step, as the data-gen runner's simulator has it
        # (maintenance_scenario_runner.py:210-244); thresholds/cadence via params["maint_*"]
        p.maint_enabled = int(bool(maintenance))
        for k, v in (params or {}).items():
            setattr(p, k, v)
        self.params = p
        self.dt = float(dt)
        self._h = ctypes.c_void_p()
        # storage="f32": carried state kept as float in HBM, arithmetic still fp64 (BASELINE config 5; include/npb.h)
        self.storage = storage
        kind = {"f64": _lib.STORAGE_F64, "f32": _lib.STORAGE_F32}[storage]
        _lib.check(self.L.npb_create_storage(ctypes.byref(p), self.n, device, kind, ctypes.byref(self._h)))
        if maintenance_thresholds is not None:   # the reference's thresholds dict for a feedwater pump, in its order
            table = _lib.maint_table_from_thresholds(maintenance_thresholds)
            _lib.check(self.L.npb_set_maintenance_table(self._h, ctypes.byref(table)), self._h)
        with torch.cuda.device(self.device):
            self._obs = torch.zeros((self.n, 22), dtype=torch.float64, device=self.device)
            self._reward = torch.zeros(self.n, dtype=torch.float64, device=self.device)
            self._done = torch.zeros(self.n, dtype=torch.uint8, device=self.device)
            self._flags = torch.zeros(self.n, dtype=torch.int32, device=self.device)
            self._info_buf = torch.zeros(self.n * (len(INFO_COLUMNS) + (_lib.INFO_NRHO if self._with_rho else 0)), dtype=torch.float64, device=self.device)
            self._info = self._info_buf[: self.n * len(INFO_COLUMNS)].view(self.n, len(INFO_COLUMNS))
            self._rho = self._info_buf[self.n * len(INFO_COLUMNS):].view(self.n, -1) if self._with_rho else None
        self._event_counts = None
        # for nuclear_sim_amd/statelog.py: how the reference names this plant's providers in its state log, and the log values its
        # constructor fixed from the initial conditions (action_test() replaces both with the data-gen composer's)
        self.log_naming = "default"
        self.log_side_columns = {"secondary.feedwater_SECONDARY-COMP-001-FW.diagnostics_total_wear": np.full(1, 60.0)}   # scenarios.log_side_columns(None)
        if p.maint_enabled and hasattr(self.L, "npb_set_maintenance_count_buffer"):
            with torch.cuda.device(self.device):
                self._event_counts = torch.zeros(self.n, dtype=torch.int32, device=self.device)
            _lib.check(self.L.npb_set_maintenance_count_buffer(self._h, ctypes.c_void_p(self._event_counts.data_ptr())), self._h)
        self._noise = None
        self._noise_seeds = None if noise_seeds is None else np.asarray(noise_seeds, dtype=np.int64).copy()
        if noise_enabled and noise_seeds is not None:
            self._noise = HeatSourceNoise(noise_seeds, device=self.device)
        self._keep = []

    @classmethod
    def action_test(cls, action: str, seeds: Sequence[int], dt: float = 5.0, device: int = 0, randomize: bool = True,
                    params: Optional[dict] = None) -> "BatchedPlantEnv":
        """One plant per seed, as data_gen's MaintenanceScenarioRunner builds them for
        ``compose_action_test_scenario(action, randomize=True, randomization_seed=seed)``
        (maintenance_scenario_runner.py:210-244): dt in minutes, ConstantHeatSource with 0.1 % noise seeded 42,
        automatic maintenance on, initial conditions from nuclear_sim_amd.scenarios (BASELINE config 4)."""
        from . import scenarios
        env = cls(len(seeds), dt=dt, heat_source="constant", noise_enabled=True, noise_std_percent=0.1,
                  noise_seeds=[42] * len(seeds), device=device, maintenance=True, params=params)
        eff = float(env.get_field("pump.lubrication_effectiveness")[0].item())
        env.set_fields(scenarios.action_test_fields(action, seeds, eff, randomize=randomize))
        # what a state log of these plants needs beside their state: the composer's provider names and the values the constructor
        # fixed from the initial conditions (nuclear_sim_amd/statelog.py)
        env.log_naming = "composed"
        env.log_side_columns = scenarios.log_side_columns(action, seeds, randomize=randomize)
        return env

    # ------------------------------------------------------------------ helpers
    def set_step_kernel(self, variant: int) -> None:
        """0 = by batch size (default), 1 = one-wave kernel, 2 = two-wave kernel, 3 = its 256-register build at any size, 4 = the
        one-wave kernel with streaming state stores (what 0 takes above 114 688 plants), 5 = four-wave kernel (what 0 takes up
        to 32 768 plants and between 45 057 and 114 688, where the handle's arena is segmented); same results to the last bit
        or two (include/npb.h)"""
        _lib.check(self.L.npb_set_step_kernel(self._h, int(variant)), self._h)

    def last_step_kernel(self) -> str:
        """the kernel the last step() actually launched, by the name rocprofv3 lists it under ("" before the first step)"""
        return self.L.npb_step_kernel_name(self.L.npb_debug_last_step_kernel(self._h)).decode()

    def enable_diagnostics(self, on: bool = True):
        """Have every following step also write the step-internal diagnostics (include/npb.h NPB_DIAG_*: per turbine stage inlet /
        outlet pressure and temperature, power output, loading factor) into ``self.diagnostics`` ([DIAG_DIM, n] on the device,
        rows in _lib.DIAG_STAGE_VALUES order x 14 stages).  The step then runs the diagnostics build of the one-wave kernel at
        every batch size: meant for state logging, not for throughput."""
        if on:
            pitch = (self.n + 63) // 64 * 64
            self._diag_buf = torch.zeros((_lib.DIAG_DIM, pitch), dtype=torch.float64, device=self.device)
            self._reset_carried_diagnostics(None)
            _lib.check(self.L.npb_set_diagnostics(self._h, ctypes.c_void_p(self._diag_buf.data_ptr()), pitch), self._h)
            self.diagnostics = self._diag_buf[:, : self.n]
        else:
            _lib.check(self.L.npb_set_diagnostics(self._h, None, 0), self._h)
            self._diag_buf = None; self.diagnostics = None
        return self.diagnostics

    def _reset_carried_diagnostics(self, mask) -> None:
        """The diagnostics rows the step carries from one step to the next (accumulators, latches, values kept while equipment
        rests: _lib.DIAG_CARRIED_ROWS) back to a freshly constructed plant's, for the masked plants: they are plant state that lives
        in this buffer instead of the arena, so every reset must take them along."""
        buf = getattr(self, "_diag_buf", None)
        if buf is None:
            return
        for row, value in _lib.DIAG_CARRIED_ROWS.items():
            if mask is None:
                buf[row].fill_(value)
            else:
                buf[row, : self.n].masked_fill_(mask.to(torch.bool), value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.L.npb_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _col(self, x, dtype):
        """None | scalar | array | tensor -> device tensor of shape [n] (kept alive until the next step)."""
        if x is None:
            return None
        if isinstance(x, torch.Tensor):
            t = x.to(device=self.device, dtype=dtype).expand(self.n).contiguous()
        else:
            a = np.asarray(x)
            if a.size == 1:      # the same value for every plant: filled on the device, nothing to copy
                t = torch.full((self.n,), a.reshape(()).item(), dtype=dtype, device=self.device)
            else:
                t = torch.as_tensor(np.array(np.broadcast_to(a, (self.n,))), dtype=dtype).to(self.device)
        self._keep.append(t)
        return t

    @staticmethod
    def _p(t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    # ------------------------------------------------------------------ state columns
    def get_field(self, name: str, instance: int = 0, k: int = 0) -> torch.Tensor:
        kind, slot = SCHEMA.slot(name, instance, k)
        t = torch.empty(self.n, dtype=torch.float64 if kind == "f64" else torch.int32, device=self.device)
        _lib.check(self.L.npb_get_field(self._h, 0 if kind == "f64" else 1, slot, self._p(t), 1, self._stream()), self._h)
        return t

    def set_field(self, name: str, value, instance: int = 0, k: int = 0) -> None:
        kind, slot = SCHEMA.slot(name, instance, k)
        t = self._col(value, torch.float64 if kind == "f64" else torch.int32)
        _lib.check(self.L.npb_set_field(self._h, 0 if kind == "f64" else 1, slot, self._p(t), 1, self._stream()), self._h)

    def _get_slot(self, kind: str, slot: int) -> torch.Tensor:
        t = torch.empty(self.n, dtype=torch.float64 if kind == "f64" else torch.int32, device=self.device)
        _lib.check(self.L.npb_get_field(self._h, 0 if kind == "f64" else 1, slot, self._p(t), 1, self._stream()), self._h)
        return t

    def _set_slot(self, kind: str, slot: int, value) -> None:
        t = self._col(value, torch.float64 if kind == "f64" else torch.int32)
        _lib.check(self.L.npb_set_field(self._h, 0 if kind == "f64" else 1, slot, self._p(t), 1, self._stream()), self._h)

    def set_fields(self, fields: dict) -> None:
        for key, v in fields.items():
            if isinstance(key, tuple):
                self.set_field(key[0], v, *key[1:])
            else:
                self.set_field(key, v)

    def state_arrays(self):
        """(f64[total_f64, n], i32[total_i32, n]) copies of the whole arena (testing / checkpointing)."""
        f = torch.empty((SCHEMA.total_f64, self.n), dtype=torch.float64, device=self.device)
        i = torch.empty((SCHEMA.total_i32, self.n), dtype=torch.int32, device=self.device)
        for s in range(SCHEMA.total_f64):
            _lib.check(self.L.npb_get_field(self._h, 0, s, ctypes.c_void_p(f[s].data_ptr()), 1, self._stream()), self._h)
        for s in range(SCHEMA.total_i32):
            _lib.check(self.L.npb_get_field(self._h, 1, s, ctypes.c_void_p(i[s].data_ptr()), 1, self._stream()), self._h)
        return f, i

    def load_state_arrays(self, f64, i32) -> None:
        f = torch.as_tensor(f64, dtype=torch.float64).to(self.device).contiguous()
        i = torch.as_tensor(i32, dtype=torch.int32).to(self.device).contiguous()
        for s in range(SCHEMA.total_f64):
            _lib.check(self.L.npb_set_field(self._h, 0, s, ctypes.c_void_p(f[s].data_ptr()), 1, self._stream()), self._h)
        for s in range(SCHEMA.total_i32):
            _lib.check(self.L.npb_set_field(self._h, 1, s, ctypes.c_void_p(i[s].data_ptr()), 1, self._stream()), self._h)
        torch.cuda.synchronize(self.device)

    @staticmethod
    def state_bytes_per_plant() -> int:
        return int(_lib.load().npb_state_bytes())

    @staticmethod
    def step_bytes_per_plant() -> int:
        return int(_lib.load().npb_step_bytes_per_plant())

    def handle_step_bytes_per_plant(self) -> int:
        """Algorithmic bytes of one plant-step for this handle's storage type."""
        return int(self.L.npb_handle_step_bytes_per_plant(self._h))

    # ------------------------------------------------------------------ reference API
    def reset(self, mask=None, reference: bool = False, start_at_steady_state: bool = True) -> torch.Tensor:
        """``reference=False``: back to the construction-time state, i.e. a freshly constructed simulator (the
        episode start of the data-gen runner, which never calls reset()).  ``reference=True``:
        NuclearPlantSimulator.reset(start_at_steady_state) with the reference's own semantics (sim.py:546-581): part of
        the state goes back to literals, part keeps its history, and with ``start_at_steady_state`` the secondary side
        is force-set to the reference's "steady state" (include/npb.h, npb_reset_reference).  Initial conditions set
        through ``set_fields`` are the caller's to re-apply.  Returns the observation, as the reference does."""
        m = self._col(mask, torch.uint8)
        self._reset_carried_diagnostics(m)
        if reference:
            _lib.check(self.L.npb_reset_reference(self._h, self._p(m), int(bool(start_at_steady_state)), self._stream()), self._h)
        else:
            _lib.check(self.L.npb_reset(self._h, self._p(m), self._stream()), self._h)
            # a freshly constructed simulator has a freshly seeded heat-source generator (the reference's own reset() keeps
            # drawing from the old one: constant_heat_source.py:185-194 does not touch the RNG).  Plants that share a seed
            # share one pre-drawn stream here, so only a reset of the whole batch can restart it.
            if mask is None and self._noise is not None and self._noise_seeds is not None:
                self._noise = HeatSourceNoise(self._noise_seeds, device=self.device)
        return self.get_observation()

    def get_observation(self) -> torch.Tensor:
        _lib.check(self.L.npb_observe(self._h, self._p(self._obs), self._stream()), self._h)
        return self._obs

    def secondary_result(self) -> Dict[str, torch.Tensor]:
        """info["secondary_system"] of the reference for every plant, as columns, for the step just taken (one gather launch)"""
        if getattr(self, "_sec_plan", None) is None:
            ks = [SCHEMA.slot(name) for name in SECONDARY_RESULT_MEMBERS]
            self._sec_plan = ((ctypes.c_int * len(ks))(*[0 if kd == "f64" else 1 for kd, _ in ks]), (ctypes.c_int * len(ks))(*[sl for _, sl in ks]))
            self._sec_buf = torch.empty((len(ks), self.n), dtype=torch.float64, device=self.device)
        _lib.check(self.L.npb_gather_fields(self._h, len(SECONDARY_RESULT_MEMBERS), self._sec_plan[0], self._sec_plan[1],
                                            ctypes.c_void_p(self._sec_buf.data_ptr()), self._stream()), self._h)
        members = {name: self._sec_buf[j] for j, name in enumerate(SECONDARY_RESULT_MEMBERS)}
        info = {name: self._info[:, j] for j, name in enumerate(INFO_COLUMNS)}
        return secondary_result(info, members)

    def step(self, action=None, magnitude=None, power_setpoint=None, cooling_water_temp=None, noise_z=None,
             thermal_power_mw=None, power_percent=None):
        self._keep = []
        if self.heat_source == "external":
            # the plugin's heat_result['thermal_power_mw'] / ['power_percent'] for this step travel in the noise_z / power_setpoint
            # columns of the C ABI (include/npb_params.h, NPB_HEAT_EXTERNAL); power_percent None = thermal power / rated x 100
            if thermal_power_mw is None:
                if noise_z is None:
                    raise ValueError("heat_source='external': step() needs the heat source's thermal_power_mw for this step")
                thermal_power_mw, power_percent = noise_z, power_setpoint       # (a caller that speaks the C ABI's column names)
            noise_z, power_setpoint = thermal_power_mw, power_percent
        elif thermal_power_mw is not None or power_percent is not None:
            raise ValueError("thermal_power_mw / power_percent are the inputs of heat_source='external'")
        a = self._col(None if action is None else action, torch.int32)
        m = self._col(magnitude, torch.float64)
        sp = self._col(power_setpoint, torch.float64)
        cw = self._col(cooling_water_temp, torch.float64)
        if noise_z is None and self._noise is None and self.params.hs_noise_enabled:
            # ConstantHeatSource(noise_enabled=True, noise_seed=None) draws from an unseeded generator
            # (constant_heat_source.py:58-62): one fresh, unseeded stream per plant
            self._noise = HeatSourceNoise(np.random.SeedSequence().generate_state(self.n, dtype=np.uint32), device=self.device)
        if noise_z is None and self._noise is not None:
            noise_z = self._noise.next()
        z = self._col(noise_z, torch.float64)
        _lib.check(self.L.npb_step(self._h, self._p(a), self._p(m), self._p(sp), self._p(z), self._p(cw),
                                   self._p(self._obs), self._p(self._reward), self._p(self._done), self._p(self._flags),
                                   self._p(self._info_buf), self._stream()), self._h)
        info = {name: self._info[:, j] for j, name in enumerate(INFO_COLUMNS)}
        if self._with_rho:
            info["reactivity_components"] = {name: self._rho[:, j] for j, name in enumerate(_lib.REACTIVITY_COMPONENTS)}
        info["trip_flags"] = self._flags
        info["scram_activated"] = self._done
        if self.params.maint_enabled:  # bit-exact counterpart of AutoMaintenanceSystem.maintenance_actions_performed
            # a column the step keeps current (npb_set_maintenance_count_buffer): no gather launch per step
            info["maintenance_event_count"] = self._event_counts if self._event_counts is not None else self.get_field("maint.maintenance_actions_performed")
        return self._obs, self._reward, self._done, info


def secondary_result(info: Dict[str, torch.Tensor], members: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The scalar keys of the reference's info["secondary_system"] (SecondaryReactorPhysics.update_system's result dict,
    secondary/__init__.py:922-1010) as columns, including its heat-flow and chemistry-flow tracker outputs
    (heat_flow_tracker.py:248-351, chemistry_flow_tracker.py:472-666; consumed at secondary/__init__.py:979-994).
    ``info`` = the step's fp64 info columns, ``members`` = the state members named in SECONDARY_RESULT_MEMBERS.
    The trackers are bookkeeping on top of quantities the step already has: the heat-flow state is closed-form in the SG
    heat transfer, the turbine power and the feedwater pump power (secondary/__init__.py:679-744); the chemistry-flow tracker
    looks its providers' states up under ChemicalSpecies names that none of them uses ("ph", "iron" ... vs
    "water_chemistry_ph" ...), so every lookup takes its default and its seven outputs are constants."""
    f = {}
    total_heat_transfer = info["sg_heat_transfer"]; turbine_gross = info["turbine_power"]; fw_power = info["feedwater_power"]
    electrical = info["electrical_power"]
    zero = torch.zeros_like(electrical); one = torch.ones_like(electrical)
    f["electrical_power_mw"] = members["sec.electrical_power_output"]
    f["thermal_efficiency"] = members["sec.thermal_efficiency"]
    f["heat_rate_kj_kwh"] = torch.where(electrical > 0, (total_heat_transfer / 1000.0) / (electrical * 1000.0) * 3600.0, zero)
    f["total_steam_flow"] = info["steam_flow"]; f["total_heat_transfer"] = total_heat_transfer; f["sg_total_heat_transfer"] = total_heat_transfer
    f["total_feedwater_flow"] = info["feedwater_flow"]; f["feedwater_total_flow"] = info["feedwater_flow"]
    f["sg_avg_pressure"] = info["steam_pressure"]
    # reported (not used by the turbine) as max(SG average, saturation temperature at the average pressure)
    # secondary/__init__.py:551-552 with _saturation_temperature :1455-1491
    pr = info["steam_pressure"] / 0.101325
    sat = 1.0 / (1.0 / (100.0 + 273.15) - (0.4615 / 2257.0) * torch.log(pr.clamp_min(1e-300))) - 273.15
    sat = torch.where(info["steam_pressure"] <= 0.001, torch.full_like(sat, 10.0), sat.clamp(10.0, 374.0))
    f["sg_avg_temperature"] = torch.maximum(members["sec.sg_avg_temperature"], sat); f["sg_avg_steam_quality"] = members["sec.sg_avg_quality"]
    mechanical = turbine_gross / 0.985
    f["turbine_mechanical_power"] = mechanical; f["turbine_electrical_power_gross"] = turbine_gross
    f["turbine_electrical_power_net"] = turbine_gross * 0.98
    f["turbine_steam_rate"] = torch.where(turbine_gross > 0, info["steam_flow"] / (turbine_gross * 1000) * 3600, zero)
    f["condenser_heat_rejection"] = members["cond.heat_rejection_rate"]; f["condenser_pressure"] = info["condenser_pressure"]
    f["condenser_vacuum_efficiency"] = members["cond.vacuum_system_efficiency"]
    f["total_system_heat_rejection"] = info["condenser_heat_rejection"]
    f["feedwater_total_power"] = fw_power
    mask = members["fw.running_mask"].to(torch.int64)
    f["feedwater_num_running_pumps"] = ((mask & 1) + ((mask >> 1) & 1) + ((mask >> 2) & 1) + ((mask >> 3) & 1)).to(torch.float64)
    f["feedwater_system_available"] = members["fw.system_availability"]
    # ---- HeatFlowTracker: component flows secondary/__init__.py:686-736, system state heat_flow_tracker.py:248-322
    sg_in = total_heat_transfer / 1e6; sg_out = total_heat_transfer * 0.98 / 1e6; sg_loss = total_heat_transfer * 0.02 / 1e6
    turb_loss = mechanical * 0.05
    fw_loss = fw_power * 0.1
    total_losses = (sg_loss + turb_loss + fw_loss)
    required_rejection = sg_in - mechanical - total_losses
    cond_loss = required_rejection * 0.01
    gen_out = mechanical * 0.985; gen_loss = mechanical - gen_out
    aux = gen_out * 0.02; net = gen_out - aux
    total_in = (sg_in + fw_power)
    total_out = (net + required_rejection + sg_loss + turb_loss + cond_loss + fw_loss + gen_loss + 0.0)
    err = total_in - total_out
    pct = torch.where(total_in > 0, (err / total_in) * 100.0, zero)
    f["heat_flow_energy_balance_error"] = err; f["heat_flow_energy_balance_percent"] = pct
    f["heat_flow_balance_ok"] = (pct.abs() < (0.01 * 100)).to(torch.float64)
    f["heat_flow_condenser_heat_rejection"] = required_rejection; f["heat_flow_net_electrical_output"] = net
    f["heat_flow_overall_efficiency"] = torch.where(total_in > 0, net / total_in, zero)
    # ---- ChemistryFlowTracker: every provider lookup falls through to its default (see the docstring)
    f["chemistry_flow_balance_error"] = zero; f["chemistry_flow_balance_ok"] = one; f["chemistry_flow_ph"] = one * 9.2
    f["chemistry_flow_iron_concentration"] = one * 0.1; f["chemistry_flow_tsp_fouling_rate"] = zero
    f["chemistry_flow_treatment_efficiency"] = one; f["chemistry_flow_stability"] = one * 0.5
    # ---- shared WaterChemistry and the pH controller
    f["water_chemistry_ph"] = members["chem.ph"]; f["water_chemistry_iron_concentration"] = one * 0.1
    f["water_chemistry_aggressiveness"] = members["chem.water_aggressiveness"]
    f["water_chemistry_treatment_efficiency"] = members["chem.treatment_efficiency"]
    f["ph_control_output"] = members["ph.controller_output"]; f["ph_control_ammonia_dose"] = members["ph.pending_ammonia_dose"]
    # ph_control_system.py:243: setpoint - measured, computed every step (previous_error only follows it while the controller is enabled)
    f["ph_control_error"] = 9.2 - members["ph.measured_ph"]
    f["load_demand"] = members["sec.load_demand"]; f["feedwater_temperature"] = one * 227.0
    f["cooling_water_inlet_temp"] = members["sec.cooling_water_temperature"]; f["cooling_water_outlet_temp"] = members["cond.cooling_water_outlet_temp"]
    # condenser/physics.py:692-693: outlet = inlet + rise, so the rise is their difference (to 1e-15 of the temperatures)
    f["condenser_cooling_water_temp_rise"] = members["cond.cooling_water_outlet_temp"] - members["sec.cooling_water_temperature"]
    # feedwater/physics.py:834: the configuration's auto_level_control, True unless a control command the step never issues clears it
    f["feedwater_auto_control"] = torch.ones_like(members["sec.cooling_water_temperature"])
    # the three values left over from inside the turbine step (secondary/__init__.py:955-958; include/npb.h NPB_INFO_TURBINE_*)
    f["turbine_efficiency"] = info["turbine_efficiency"]; f["turbine_hp_power"] = info["turbine_hp_power"]; f["turbine_lp_power"] = info["turbine_lp_power"]
    # condenser/physics.py:852-859: area factor (active / initial tubes, :122) x fouling factor x vacuum system efficiency, all end-of-step state
    area_factor = members["cond.active_tube_count"] / 84000.0
    fouling_factor = (1.0 - members["cond.total_fouling_resistance"] * 5).clamp_min(0.3)
    f["condenser_thermal_performance"] = area_factor * fouling_factor * members["cond.vacuum_system_efficiency"]
    return f


SECONDARY_RESULT_MEMBERS = ("sec.electrical_power_output", "sec.thermal_efficiency", "sec.sg_avg_temperature", "sec.sg_avg_quality",
                            "cond.heat_rejection_rate", "cond.vacuum_system_efficiency", "fw.running_mask", "fw.system_availability",
                            "chem.ph", "chem.water_aggressiveness", "chem.treatment_efficiency", "ph.controller_output",
                            "ph.pending_ammonia_dose", "ph.measured_ph", "sec.load_demand", "sec.cooling_water_temperature",
                            "cond.cooling_water_outlet_temp", "cond.active_tube_count", "cond.total_fouling_resistance")


FACTORY_DEFAULT_PUMP_THRESHOLDS = {"oil_level": {"threshold": 30.0, "comparison": "less_than", "action": "oil_top_off",
                                                 "cooldown_hours": 24.0, "priority": "HIGH"}}


class ConstantHeatSource:
    """systems/primary/reactor/heat_sources/constant_heat_source.py:29-102 (constructor and setpoint surface)."""

    def __init__(self, rated_power_mw: float = 3000.0, noise_enabled: bool = False, noise_std_percent: float = 5.0,
                 noise_seed: Optional[int] = None, noise_filter_time_constant: float = 30.0):
        self.rated_power_mw = rated_power_mw
        self.noise_enabled = noise_enabled
        self.noise_std_percent = noise_std_percent
        self.noise_seed = noise_seed
        self.noise_filter_time_constant = noise_filter_time_constant
        self.power_setpoint_percent = 100.0
        self._pending = None

    def set_power_setpoint(self, power_percent: float) -> None:
        self.power_setpoint_percent = float(np.clip(power_percent, 0.0, 150.0))
        self._pending = float(power_percent)


class HeatSource:
    """The reference's plugin interface for heat sources (heat_sources/heat_source_interface.py:23-112): subclass it and hand the
    object to NuclearPlantSimulator -- every step calls ``update(dt=..., reactor_state=..., control_action=...)`` on the host and
    feeds the returned ``thermal_power_mw`` / ``power_percent`` to the step as input columns (keys beyond those two --
    ``neutron_flux``, ``reactivity_pcm``, ``reactivity_components`` -- are not supported and raise).  Batches: compute the two
    columns yourself and call ``BatchedPlantEnv(heat_source="external").step(thermal_power_mw=..., power_percent=...)``."""

    def __init__(self, rated_power_mw: float = 3000.0):
        self.rated_power_mw = rated_power_mw
        self.current_power_mw = 0.0
        self.power_setpoint_percent = 100.0

    def update(self, dt: float, **kwargs) -> dict:
        raise NotImplementedError

    def set_power_setpoint(self, power_percent: float) -> None:
        self.power_setpoint_percent = power_percent

    def reset(self) -> None:
        pass


class ReactorHeatSource:
    """systems/primary/reactor/heat_sources/reactor_heat_source.py (point-kinetics heat source)."""

    def __init__(self, rated_power_mw: float = 3000.0):
        self.rated_power_mw = rated_power_mw
        self._pending = None


class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _PathProxy:
    """Read / write the plant's state through the reference's own attribute paths:
    ``sim.secondary_physics.feedwater_system.pump_system.pumps['FWP-1'].lubrication_system.oil_level`` resolves, step by
    step, against the reference attribute path every schema member carries (include/npb_fields.h), so code that reads
    or pokes the reference's object tree runs unchanged.  Only attributes that are state members exist; anything else
    raises AttributeError (there is no Python object tree behind this)."""

    def __init__(self, env, prefix, extras=None):
        object.__setattr__(self, "_env", env)
        object.__setattr__(self, "_prefix", prefix)
        object.__setattr__(self, "_extras", extras or {})

    @staticmethod
    def _index():
        idx = getattr(_PathProxy, "_paths", None)
        if idx is None:
            idx = {}
            for kind, slot, label, path in SCHEMA.columns():
                if path and not path.startswith("="):
                    idx[path] = (kind, slot)
            _PathProxy._paths = idx
        return idx

    def _resolve(self, path):
        idx = self._index()
        if path in idx:
            kind, slot = idx[path]
            v = self._env._get_slot(kind, slot)[0].item()
            return v if kind == "f64" else int(v)
        if any(p.startswith(path + ".") or p.startswith(path + "[") for p in idx):
            return _PathProxy(self._env, path)
        raise AttributeError("%s is not a state member of the plant" % path)

    def __getattr__(self, name):
        extras = object.__getattribute__(self, "_extras")
        if name in extras:
            return extras[name]
        return self._resolve("%s.%s" % (self._prefix, name) if self._prefix else name)

    def __getitem__(self, key):
        return self._resolve("%s[%r]" % (self._prefix, key))

    def _assign(self, path, value):
        idx = self._index()
        if path not in idx:
            raise AttributeError("%s is not a state member of the plant" % path)
        kind, slot = idx[path]
        self._env._set_slot(kind, slot, [value])

    def __setattr__(self, name, value):
        self._assign("%s.%s" % (self._prefix, name), value)

    def __setitem__(self, key, value):
        self._assign("%s[%r]" % (self._prefix, key), value)


class NuclearPlantSimulator:
    """Single-plant facade with the reference's signatures (simulator/core/sim.py:27-258), so that loops written
    against the reference -- ``sim.primary_physics.heat_source.set_power_setpoint(p); sim.step(action=...)``
    (maintenance_scenario_runner.py:383-411, 651-671) -- run unchanged on one lane of the HIP stepper.

    Differences, all explicit: ``enable_state_management=True`` (the reference's default, sim.py:31) turns on what the
    path needs of it -- the automatic maintenance of the feedwater pumps, with the priority delays of the configuration's
    ``maintenance_system.maintenance_mode`` (aggressive / ultra_aggressive: none; anything else: 1 h / 4 h / 24 h,
    sim.py:97-128, auto_maintenance.py:187-198) and the feedwater thresholds of
    ``maintenance_system.component_configs.feedwater.thresholds`` when the configuration has them -- not the pandas state
    log; ``secondary_config`` is honoured for the
    initial-condition keys nuclear_sim_amd.scenarios can map (others are listed in ``ignored_initial_conditions``);
    ``reset()`` follows the reference's own reset (default configuration; pinned by tests/golden/r1_*.npz)."""

    def __init__(self, dt: float = 1.0, heat_source=None, enable_secondary: bool = True,
                 enable_state_management: bool = True, max_state_rows: int = 100000, secondary_config=None,
                 secondary_config_file: Optional[str] = None, device: int = 0):
        if secondary_config is None and secondary_config_file is not None and enable_secondary:
            secondary_config = self._load_config_file(secondary_config_file)
        if heat_source is None or heat_source == "reactor":
            heat_source = ReactorHeatSource()            # sim.py:41-44: the default heat source is the reactor model
        elif heat_source == "constant":
            heat_source = ConstantHeatSource(noise_std_percent=0.1)
        constant = isinstance(heat_source, ConstantHeatSource)
        # anything else must speak the reference's HeatSource plugin interface; an object this facade cannot map is refused
        # rather than silently run as the reactor model
        plugin = not constant and not isinstance(heat_source, ReactorHeatSource)
        if plugin and not (callable(getattr(heat_source, "update", None)) and hasattr(heat_source, "rated_power_mw")):
            raise TypeError("heat_source must be a ConstantHeatSource, a ReactorHeatSource, 'constant', 'reactor' or an object with the "
                            "reference's HeatSource interface (update(dt, **kwargs) -> {'thermal_power_mw', 'power_percent'}, "
                            "rated_power_mw): got %r" % (heat_source,))
        self._plugin = heat_source if plugin else None
        self.dt = dt
        self.enable_secondary = bool(enable_secondary)
        self.enable_state_management = enable_state_management
        # StateManager's clock (state_manager.py:51-52,82-109): a random start date drawn from the `random` module, advanced
        # by dt minutes per step, never put back by reset(); without state management info["datetime"] is None
        self._datetime = None
        if enable_state_management:
            import datetime as _dt
            import random as _random
            self._datetime = _dt.datetime(_random.randint(2020, 2030), _random.randint(1, 12), _random.randint(1, 28),
                                          _random.randint(0, 23), _random.randint(0, 59), 0)
        params = {"rated_power_mw": float(heat_source.rated_power_mw)}
        if constant:
            params["hs_noise_filter_tau"] = float(heat_source.noise_filter_time_constant)
        maint_cfg = (secondary_config or {}).get("maintenance_system", {}) if isinstance(secondary_config, dict) else {}
        # sim.py:97-128, auto_maintenance.py:187-198: anything but an aggressive mode in the configuration -- no configuration at
        # all included -- delays execution by priority
        if maint_cfg.get("maintenance_mode") not in ("aggressive", "ultra_aggressive") and enable_state_management:
            params.update(maint_start_delay_hours=1.0, maint_medium_delay_hours=4.0, maint_low_delay_hours=24.0)
        thresholds = ((maint_cfg.get("component_configs") or {}).get("feedwater") or {}).get("thresholds")
        if thresholds is None:
            # no maintenance configuration: the state manager's factory default (state_manager.py
            # _create_default_maintenance_config) gives a feedwater pump this one threshold -- not the data-gen action-test
            # table, which would top a pump off at 58 % (fixture m14_default_configuration_maintenance)
            thresholds = dict(FACTORY_DEFAULT_PUMP_THRESHOLDS)
        self._env = BatchedPlantEnv(1, dt=dt, heat_source="constant" if constant else ("external" if plugin else "reactor"),
                                    noise_enabled=bool(constant and heat_source.noise_enabled),
                                    noise_std_percent=float(heat_source.noise_std_percent) if constant else 0.1,
                                    noise_seeds=[heat_source.noise_seed] if (constant and heat_source.noise_enabled and
                                                                             heat_source.noise_seed is not None) else None,
                                    device=device, maintenance=bool(enable_state_management and enable_secondary), params=params,
                                    maintenance_thresholds=thresholds, mode="full" if enable_secondary else "primary",
                                    reactivity_components=not constant and not plugin)
        # the reference's object tree, as far as it is plant state: attribute paths resolve against the schema
        self.primary_physics = _PathProxy(self._env, "primary_physics",
                                          extras={"heat_source": heat_source, "rated_power_mw": heat_source.rated_power_mw})
        self.secondary_physics = _PathProxy(self._env, "secondary_physics") if enable_secondary else None
        self.ignored_initial_conditions = []
        if secondary_config is not None and enable_secondary:
            self._apply_secondary_config(secondary_config)
        self.load_demand = 100.0
        self.cooling_water_temp = 25.0

    def _apply_secondary_config(self, cfg: dict) -> None:
        """initial_conditions of the composed configuration -> state columns (nuclear_sim_amd.scenarios)"""
        from . import scenarios
        sec = cfg.get("secondary_system", cfg)
        fields = {}
        fw_ic = dict(sec.get("feedwater", {}).get("initial_conditions", {}) or {})
        known = set(scenarios.FEEDWATER_IC_DEFAULTS)
        self.ignored_initial_conditions += ["feedwater." + k for k in fw_ic if k not in known]
        self._feedwater_ic = {k: v for k, v in fw_ic.items() if k in known}
        if fw_ic:
            eff = float(self._env.get_field("pump.lubrication_effectiveness")[0].item())
            fields.update(scenarios.feedwater_fields({k: v for k, v in fw_ic.items() if k in known}, 1, eff))
        sg_ic = sec.get("steam_generator", {}).get("initial_conditions", {}) or {}
        if "sg_steam_flows" in sg_ic:
            for k in range(scenarios.NUM_SG):
                fields[("sg.steam_flow_rate", k)] = np.array([float(sg_ic["sg_steam_flows"][k])])
        self.ignored_initial_conditions += ["steam_generator." + k for k in sg_ic if k != "sg_steam_flows"]
        tb_ic = sec.get("turbine", {}).get("initial_conditions", {}) or {}
        if "rotor_temperature" in tb_ic:
            fields["turb.rotor_temperature"] = np.array([float(tb_ic["rotor_temperature"])])
        if "bearing_temperatures" in tb_ic:
            for k in range(4):
                fields[("turb.bearing_metal_temp", 0, k)] = np.array([float(tb_ic["bearing_temperatures"][k])])
        self.ignored_initial_conditions += ["turbine." + k for k in tb_ic if k not in ("rotor_temperature", "bearing_temperatures")]
        self.ignored_initial_conditions += ["condenser." + k for k in (sec.get("condenser", {}).get("initial_conditions", {}) or {})]
        self._env.set_fields(fields)

    @property
    def state(self):
        """ReactorState view of the primary columns (sim.state, sim.py:85,151): attribute name -> current value"""
        out = {}
        for sec, f in (SCHEMA.by_name[k] for k in SCHEMA.by_name if k.startswith("prim.")):
            if f.count == 1 and f.path.startswith("primary_physics.state."):
                out[f.path[len("primary_physics.state."):]] = self._env.get_field("prim." + f.name)[0].item()
        return _Namespace(**out)

    def _state_after_actuators(self, a: int, magnitude: float):
        """``self.state`` with this step's actuator movement applied: PrimaryReactorPhysics._apply_control_actions
        (primary/__init__.py:289-359; rates :174-176 and the 50 ppm/s of :333,341; which action moves what: sim.py:260-288) -- what a
        HeatSource plugin is shown, since the reference moves the actuators before it updates the heat source."""
        st = self.state
        p, dt = self._env.params, self.dt
        if a == ControlAction.CONTROL_ROD_INSERT.value:
            st.control_rod_position = max(0, st.control_rod_position - p.max_control_rod_speed * dt * magnitude)
        elif a == ControlAction.CONTROL_ROD_WITHDRAW.value:
            st.control_rod_position = min(100, st.control_rod_position + p.max_control_rod_speed * dt * magnitude)
        elif a == ControlAction.INCREASE_COOLANT_FLOW.value:
            st.coolant_flow_rate = min(50000, st.coolant_flow_rate + p.max_flow_change_rate * dt * magnitude)
        elif a == ControlAction.DECREASE_COOLANT_FLOW.value:
            st.coolant_flow_rate = max(5000, st.coolant_flow_rate - p.max_flow_change_rate * dt * magnitude)
        elif a == ControlAction.DILUTE_BORON.value:
            st.boron_concentration = max(0, st.boron_concentration - 50.0 * dt * magnitude)
        elif a == ControlAction.BORATE_COOLANT.value:
            st.boron_concentration = min(3000, st.boron_concentration + 50.0 * dt * magnitude)
        elif a == ControlAction.OPEN_STEAM_VALVE.value:
            st.steam_valve_position = min(100, st.steam_valve_position + p.max_valve_speed * dt * magnitude)
        elif a == ControlAction.CLOSE_STEAM_VALVE.value:
            st.steam_valve_position = max(0, st.steam_valve_position - p.max_valve_speed * dt * magnitude)
        return st

    def set_power_setpoint(self, power_percent: float) -> None:
        """heat_source.set_power_setpoint  constant_heat_source.py:93-102 (applied at the next step)."""
        if self._plugin is not None:
            self._plugin.set_power_setpoint(float(power_percent))
        else:
            self.primary_physics.heat_source._pending = float(power_percent)

    def step(self, action: Optional[ControlAction] = None, magnitude: float = 1.0, load_demand: float = None,
             cooling_water_temp: float = None) -> Dict:
        a = ControlAction.NO_ACTION.value if action is None else (action.value if isinstance(action, ControlAction) else int(action))
        hs = self.primary_physics.heat_source
        if self._plugin is not None:
            # primary/__init__.py:200-207: _apply_control_actions FIRST, then heat_source.update(dt, reactor_state, control_action) -- the
            # plugin sees the rods, flow, boron and valve where this step's action has moved them.  The launch moves them too (its
            # result is an input column of the launch), so the view handed to the plugin is moved here, on the host, by the same rule
            res = self._plugin.update(dt=self.dt, reactor_state=self._state_after_actuators(a, float(magnitude)), control_action=ControlAction(a))

            def given(v):           # a key the plugin filled in (None, an empty dict, a zero -- of any numeric type -- count as absent)
                if v is None or (isinstance(v, dict) and not v):
                    return False
                return not (np.isscalar(v) and float(v) == 0.0) if not isinstance(v, dict) else True
            unsupported = [k for k in ("neutron_flux", "reactivity_pcm", "reactivity_components") if k in res and given(res[k])]
            if unsupported:
                raise NotImplementedError("a HeatSource plugin's result may carry thermal_power_mw and power_percent; %s are not supported" % unsupported)
            obs, rew, done, info = self._env.step(action=[a], magnitude=[magnitude], thermal_power_mw=[float(res["thermal_power_mw"])],
                                                  power_percent=[float(res["power_percent"])],
                                                  cooling_water_temp=None if cooling_water_temp is None else [cooling_water_temp])
        else:
            sp, hs._pending = getattr(hs, "_pending", None), None
            obs, rew, done, info = self._env.step(action=[a], magnitude=[magnitude],
                                                  power_setpoint=None if sp is None else [sp],
                                                  cooling_water_temp=None if cooling_water_temp is None else [cooling_water_temp])
        o = obs[0].cpu().numpy().copy()
        rho = info.pop("reactivity_components", None)
        inf = {k: (v[0].item()) for k, v in info.items()}
        inf["scram_activated"] = bool(inf["scram_activated"])
        # sim.py:205: the reactor model's ten terms in pcm; ConstantHeatSource has none (primary/__init__.py:225)
        inf["reactivity_components"] = {} if rho is None else {k: float(v[0].item()) for k, v in rho.items()}
        if self._datetime is not None:
            import datetime as _dt
            self._datetime += _dt.timedelta(minutes=self.dt)
        inf["datetime"] = self._datetime.isoformat() if self._datetime is not None else None
        if not self.enable_secondary:   # sim.py:199-206,225: without a secondary side the dict has the primary keys only
            inf = {k: inf[k] for k in ("time", "datetime", "thermal_power", "scram_activated", "reactivity", "reactivity_components", "trip_flags")}
            o = o[:12]
        else:
            inf["secondary_system"] = self._secondary_result()
        return {"observation": o, "reward": float(rew[0].item()), "done": bool(done[0].item()), "info": inf}

    @staticmethod
    def _load_config_file(path: str) -> dict:
        """SecondaryReactorPhysics(config_file=...)  secondary/__init__.py:181-205: a YAML file, its ``secondary_system``
        section when it has one; the comprehensive configuration's ``maintenance_system`` section rides along"""
        import yaml
        with open(path, "r") as fh:
            data = yaml.safe_load(fh)
        cfg = dict(data.get("secondary_system", data))
        if "maintenance_system" in data and "maintenance_system" not in cfg:
            cfg["maintenance_system"] = data["maintenance_system"]
        return cfg

    def _secondary_result(self) -> Dict[str, float]:
        """info["secondary_system"]: every scalar key of the reference's result dict (secondary/__init__.py:922-1010) that is a
        function of what the step produces -- all 56 scalars (turbine_efficiency, turbine_hp_power and turbine_lp_power are the
        step's info columns NPB_INFO_TURBINE_*; condenser_thermal_performance is end-of-step state)."""
        return {k: float(v[0].item()) for k, v in self._env.secondary_result().items()}

    def reset(self, start_at_steady_state: bool = True):
        """sim.py:546-581: the reference's reset (not a re-construction); the configured feedwater initial conditions
        are re-applied as EnhancedFeedwaterPhysics.reset does (feedwater/physics.py:1286-1323)."""
        env = self._env
        obs = env.reset(reference=True, start_at_steady_state=start_at_steady_state)
        if getattr(self, "_feedwater_ic", None):
            from . import scenarios
            eff = torch.stack([env.get_field("pump.lubrication_effectiveness", instance=k) for k in range(scenarios.NUM_PUMPS)], dim=1).cpu().numpy()
            env.set_fields(scenarios.feedwater_reset_fields(self._feedwater_ic, 1, eff, start_at_steady_state))
            obs = env.get_observation()
        if self._plugin is not None:
            self._plugin.reset()        # primary/__init__.py reset_system -> heat_source.reset()
        else:
            self.primary_physics.heat_source._pending = None
        if hasattr(self.primary_physics.heat_source, "power_setpoint_percent"):
            self.primary_physics.heat_source.power_setpoint_percent = 100.0
        self.load_demand = 100.0
        self.cooling_water_temp = 25.0
        return obs[0].cpu().numpy().copy()[: 22 if self.enable_secondary else 12]

    def get_observation(self) -> np.ndarray:
        return self._env.get_observation()[0].cpu().numpy().copy()[: 22 if self.enable_secondary else 12]


class NuclearPlantEnv:
    """Gym-style wrapper  sim.py:911-940."""

    def __init__(self, **kw):
        self.sim = NuclearPlantSimulator(**kw)
        self.action_space_size = len(ControlAction)
        self.observation_space_size = 22 if self.sim.enable_secondary else 12   # sim.py:917-918

    def render(self):
        return None

    def reset(self):
        return self.sim.reset()

    def step(self, action_idx: int, load_demand: float = None, cooling_water_temp: float = None):
        r = self.sim.step(ControlAction(action_idx), load_demand=load_demand, cooling_water_temp=cooling_water_temp)
        return r["observation"], r["reward"], r["done"], r["info"]
