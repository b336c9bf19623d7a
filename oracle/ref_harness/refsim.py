"""Harness that imports the *reference* simulator from /root/reference (this
container only; the reference never travels to the GPU box) so that golden
vectors can be generated and the C oracle can be checked leaf-by-leaf.

Not part of the product. Not imported by anything outside oracle/ and the
fixture generator scripts.

What it does beyond a plain import:
  * puts /root/reference/nuclear_simulator on sys.path (the reference imports
    itself as top-level `systems.*` / `simulator.*`, sim.py:11-13);
  * attaches a `from_dict` classmethod to the reference's config dataclasses,
    because `dataclass_wizard` (requirements.txt:10) is not installed here and
    `SecondaryReactorPhysics.__init__` needs it (secondary/__init__.py:240-243);
  * silences the reference's print() chatter;
  * neutralises the pH controller's *unseeded global* numpy RNG
    (ph_control_system.py:278,288,409-420) so runs are reproducible:
    normal() -> 0, random() -> 1.0.
"""
import contextlib
import dataclasses
import io
import os
import sys
import typing

REF_ROOT = os.environ.get("NPB_REFERENCE_ROOT", "/root/reference")
REF_PKG = os.path.join(REF_ROOT, "nuclear_simulator")


def available() -> bool:
    return os.path.isdir(REF_PKG)


def _build_dataclass(cls, data):
    """Recursive dict -> dataclass (snake_case keys only), the subset of
    dataclass_wizard.from_dict the reference relies on."""
    if not dataclasses.is_dataclass(cls) or not isinstance(data, dict):
        return data
    hints = typing.get_type_hints(cls)
    kwargs = {}
    for f in dataclasses.fields(cls):
        if f.name not in data or not f.init:
            continue
        v = data[f.name]
        t = hints.get(f.name, None)
        origin = typing.get_origin(t)
        if dataclasses.is_dataclass(t) and isinstance(v, dict):
            v = _build_dataclass(t, v)
        elif origin in (list, typing.List) and isinstance(v, list):
            (arg,) = typing.get_args(t) or (None,)
            if arg is not None and dataclasses.is_dataclass(arg):
                v = [_build_dataclass(arg, x) for x in v]
        elif origin is typing.Union:
            for arg in typing.get_args(t):
                if dataclasses.is_dataclass(arg) and isinstance(v, dict):
                    v = _build_dataclass(arg, v)
                    break
        # scalar coercion dataclass_wizard would do (PyYAML reads "3000.0e6" as a str)
        if t is float and isinstance(v, (str, int)) and not isinstance(v, bool):
            v = float(v)
        elif t is int and isinstance(v, str):
            v = int(float(v))
        elif origin in (list, typing.List) and isinstance(v, list):
            (arg,) = typing.get_args(t) or (None,)
            if arg is float:
                v = [float(x) if isinstance(x, (str, int)) and not isinstance(x, bool) else x for x in v]
        kwargs[f.name] = v
    return cls(**kwargs)


_ready = False


_REAL_NORMAL = _FLAT_NORMAL = None


def setup():
    global _ready
    if _ready:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_PKG)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    if REF_PKG not in sys.path:
        sys.path.insert(0, REF_PKG)
    with contextlib.redirect_stdout(io.StringIO()):
        import systems.secondary.config as sc
        import systems.secondary.feedwater.config as fc
        import systems.secondary.condenser.config as cc
        import systems.secondary.turbine.config as tc
        import systems.secondary.steam_generator.config as gc
    for cls in (sc.SecondarySystemConfig, fc.FeedwaterConfig, cc.CondenserConfig,
                tc.TurbineConfig, gc.SteamGeneratorConfig):
        if not hasattr(cls, "from_dict"):
            cls.from_dict = classmethod(_build_dataclass)
    # neutralise the unseeded global RNG used only by the pH controller
    import numpy as np
    def _normal(loc=0.0, scale=1.0, size=None):
        return 0.0 if size is None else np.zeros(size)

    def _random(size=None):
        return 1.0 if size is None else np.ones(size)
    global _REAL_NORMAL, _FLAT_NORMAL
    _REAL_NORMAL, _FLAT_NORMAL = np.random.normal, _normal
    np.random.normal = _normal
    np.random.random = _random
    _ready = True


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def make_sim(dt=1.0, heat_source="constant", noise=False, noise_std_percent=0.1,
             noise_seed=42, secondary=None, state_management=False, enable_secondary=True):
    """Construct a reference NuclearPlantSimulator."""
    setup()
    with quiet():
        from simulator.core.sim import NuclearPlantSimulator
        from systems.primary.reactor.heat_sources import ConstantHeatSource, ReactorHeatSource
        if heat_source == "constant":
            hs = ConstantHeatSource(3000.0, noise_enabled=noise,
                                    noise_std_percent=noise_std_percent,
                                    noise_seed=noise_seed)
        elif callable(heat_source):
            hs = scripted_heat_source(heat_source)
        else:
            hs = ReactorHeatSource(3000.0)
        cfg = {"secondary_system": secondary or {}}
        sim = NuclearPlantSimulator(dt=dt, heat_source=hs, secondary_config=cfg,
                                    enable_state_management=state_management, enable_secondary=enable_secondary)
    return sim


def scripted_heat_source(script, rated_power_mw=3000.0):
    """a user-supplied heat source written against the reference's plugin interface (heat_source_interface.py:23-112):
    script(k) -> (thermal_power_mw, power_percent) of its k-th update"""
    from systems.primary.reactor.heat_sources.heat_source_interface import HeatSource

    class ScriptedHeatSource(HeatSource):
        def __init__(self):
            super().__init__(rated_power_mw)
            self.k = 0
            self.results = []

        def get_thermal_power_mw(self):
            return self.current_power_mw

        def get_power_percent(self):
            return self.current_power_mw / self.rated_power_mw * 100.0

        def set_power_setpoint(self, power_percent):
            self.power_setpoint_percent = power_percent

        def update(self, dt, **kwargs):
            # a script of two arguments is shown the reactor state the reference hands its heat source (primary/__init__.py:203-207)
            tp, pp = script(self.k, kwargs["reactor_state"]) if script.__code__.co_argcount == 2 else script(self.k)
            self.k += 1
            self.current_power_mw = tp
            self.results.append((tp, pp))
            return {"thermal_power_mw": tp, "power_percent": pp}

        def get_state_dict(self):
            return {"thermal_power_mw": self.current_power_mw}

        def reset(self):
            self.current_power_mw = 0.0
    return ScriptedHeatSource()


def make_runner_sim(action="oil_top_off", duration_hours=4.0, feedwater_ic=None, randomization_seed=None):
    """The simulator as data_gen's MaintenanceScenarioRunner builds it for an action-test scenario
    (maintenance_scenario_runner.py:210-244): composed config, dt = 5 min, state management and
    AutoMaintenanceSystem on.  feedwater_ic overrides entries of
    secondary_system.feedwater.initial_conditions (e.g. pump_oil_levels); randomization_seed switches the
    composer's per-seed randomisation on (comprehensive_composer.py:183-252)."""
    setup()
    with quiet():
        from data_gen.config_engine.composers.comprehensive_composer import ComprehensiveComposer
        from data_gen.runners.maintenance_scenario_runner import MaintenanceScenarioRunner
        import numpy as np
        np.random.normal = _REAL_NORMAL    # the composer's randomiser draws its "normal" parameters from the seeded global stream
        try:
            cfg = ComprehensiveComposer().compose_action_test_scenario(
                target_action=action, duration_hours=duration_hours, randomize=randomization_seed is not None,
                randomization_seed=randomization_seed)
        finally:
            np.random.normal = _FLAT_NORMAL
        for k, v in (feedwater_ic or {}).items():
            cfg["secondary_system"]["feedwater"]["initial_conditions"][k] = v
        runner = MaintenanceScenarioRunner(cfg, verbose=False)
    return runner, runner.simulator
