"""GPU: the HIP stepper (through the C ABI) against the golden vectors and against the CPU oracle
on identical seeded inputs.  Bar: integer / flag columns bit-exact, fp64 columns within 1e-6
relative (BASELINE.json north_star)."""
import numpy as np
import pytest

from golden_util import Golden, compare_state, fixture_names, RTOL, ATOL_SMALL, EXEMPT_PREFIXES

pytestmark = pytest.mark.gpu


def _env(g=None, n=1, **kw):
    from nuclear_sim_amd.env import BatchedPlantEnv
    if g is not None:
        m = g.meta
        kw.setdefault("dt", m.get("dt", 1.0))
        kw.setdefault("heat_source", m.get("heat_source", "constant"))
        kw.setdefault("noise_enabled", bool(m.get("noise")))
        kw.setdefault("noise_std_percent", m.get("noise_std_percent", 0.1))
        kw.setdefault("maintenance", bool(m.get("runner") or m.get("state_management")))
        if m.get("maint_params"):    # execution delays by priority (the non-aggressive mode of a simulator without a maintenance configuration)
            kw.setdefault("params", dict(m["maint_params"]))
        kw.setdefault("mode", "full" if m.get("enable_secondary", True) else "primary")   # NuclearPlantSimulator(enable_secondary=False)
        kw.setdefault("reactivity_components", g.rc is not None)
        if m.get("maint_thresholds"):    # the run used a maintenance configuration other than the default one
            kw.setdefault("maintenance_thresholds", dict((nm, c) for nm, c in m["maint_thresholds"]))
    env = BatchedPlantEnv(n, **kw)
    if g is not None and g.meta.get("runner"):    # the data-gen runner's plant: the composer's provider names and construction-fixed log values
        from nuclear_sim_amd import scenarios
        env.log_naming = "composed"
        env.log_side_columns = scenarios.log_side_columns(g.meta["runner"]["action"], [0], randomize=False)
    return env


def _host_state(env):
    f, i = env.state_arrays()
    return f.cpu().numpy(), i.cpu().numpy()


def test_native_library_is_loaded():
    import torch
    from nuclear_sim_amd import _lib
    assert torch.cuda.is_available()
    L = _lib.load()
    assert L.npb_version() >= 100
    maps = open("/proc/self/maps").read()
    assert "libnpb.so" in maps


def test_construction_state_matches_reference():
    g = Golden("s1_constant_steady")
    env = _env(g)
    f, i = _host_state(env)
    compare_state(g, f[:, 0], i[:, 0], g.state[0], "construction state")
    # and get_observation() right after construction
    obs = env.get_observation().cpu().numpy()
    assert obs.shape == (1, 22) and np.all(np.isfinite(obs))


# which kernel npb_step must have launched for a forced variant (include/npb.h npb_set_step_kernel) at a batch of <= 32 768
# plants in full mode; 0 = by batch size
KERNEL_OF_VARIANT = {0: "npb_step4_kernel", 1: "npb_step_kernel", 2: "npb_step2_wide_kernel", 3: "npb_step2_kernel", 4: "npb_step_nt_kernel",
                     5: "npb_step4_kernel"}
# fixtures replayed on EVERY shipped step kernel: reactor and constant heat sources, load following, pump trips, the data-gen
# runner with maintenance, handler promotion, fuzzed states (flags flipped, maintenance under fire), reset(), pump start / stop,
# pump trip reasons, turbine trips
EVERY_KERNEL_FIXTURES = ("s2_reactor_actions", "s5_load_following", "s7_pump_trips", "m1_oil_top_off_staggered",
                         "m8_handlers_inspection_overhaul_promotion", "z3_fuzzed_state_constant", "z13_fuzzed_state_running",
                         "z22_fuzzed_maintenance", "r1_reset_steady", "c1_pump_stopping_starting", "c6_pump_trip_reasons", "c7_turbine_trips")


@pytest.mark.parametrize("name", fixture_names())
def test_hip_replays_golden(name):
    """Every reference-generated fixture replayed on the GPU (all 64 lanes of a wave run copies;
    lane 0 and lane 63 are checked) with the kernel npb_step picks for the batch."""
    _replay_golden(name, 0)


@pytest.mark.parametrize("variant", [1, 2, 3, 4])
@pytest.mark.parametrize("name", EVERY_KERNEL_FIXTURES)
def test_hip_replays_golden_on_every_step_kernel(name, variant):
    """The reference's fixtures against each shipped step kernel, not only the one a 64-plant batch selects: the one-wave
    kernel (the headline's), the 256-register two-wave build (32 769 .. 57 344 plants) and the streaming-store build
    (> 90 112 plants) forced on the same replay; which kernel really ran is asked of the library."""
    _replay_golden(name, variant)


def _replay_golden(name, variant):
    import torch
    g = Golden(name)
    n = 64
    env = _env(g, n=n)
    env.set_step_kernel(variant)
    want_kernel = KERNEL_OF_VARIANT[variant] if g.meta.get("enable_secondary", True) else "npb_step_primary_kernel"
    if env.params.maint_enabled and g.meta.get("enable_secondary", True):
        want_kernel = want_kernel.replace("_kernel", "_maint_kernel")       # the build with the automatic maintenance compiled in
    f0, i0 = _host_state(env)
    f, i, fm, im = g.split_state(g.state[0])
    f0[fm, :] = f[fm, None]; i0[im, :] = i[im, None]
    env.load_state_arrays(f0, i0)
    sampled = {int(s): k for k, s in enumerate(g.state_steps)}
    for t in range(g.T):
        for label, v in g.pokes.get(t, []):
            kind, slot = g.label_slot(label)
            col = torch.full((n,), v, dtype=torch.float64 if kind == "f64" else torch.int32, device=env.device)
            from nuclear_sim_amd import _lib
            import ctypes
            _lib.check(env.L.npb_set_field(env._h, 0 if kind == "f64" else 1, slot, ctypes.c_void_p(col.data_ptr()), 1, env._stream()), env._h)
        if t in g.resets:    # NuclearPlantSimulator.reset(start_at_steady_state) in mid-run (sim.py:546-581)
            steady, ref_obs, ref_state = g.resets[t]
            env.reset(reference=True, start_at_steady_state=steady)
            ic = ((g.meta.get("secondary") or {}).get("feedwater") or {}).get("initial_conditions")
            if ic:   # EnhancedFeedwaterPhysics.reset re-applies the configured initial conditions (physics.py:1286-1323)
                from nuclear_sim_amd import scenarios
                eff = torch.stack([env.get_field("pump.lubrication_effectiveness", instance=k) for k in range(4)], dim=1).cpu().numpy()
                env.set_fields(scenarios.feedwater_reset_fields(ic, n, eff, steady))
            robs = env.get_observation().cpu().numpy()
            fs, is_ = _host_state(env)
            for lane in (0, n - 1):
                np.testing.assert_allclose(robs[lane], ref_obs, rtol=RTOL, atol=1e-12, err_msg="%s reset obs before step %d" % (name, t))
                compare_state(g, fs[:, lane], is_[:, lane], ref_state, "after reset before step %d (lane %d)" % (t, lane))
        sp = None if np.isnan(g.setpoint[t]) else g.setpoint[t]
        cw = None if np.isnan(g.cooling[t]) else g.cooling[t]
        obs, rew, done, info = env.step(action=int(g.action[t]), magnitude=float(g.magnitude[t]), power_setpoint=sp,
                                        cooling_water_temp=cw, noise_z=float(g.noise_z[t]))
        assert env.last_step_kernel() == want_kernel, (env.last_step_kernel(), want_kernel)
        obs = obs.cpu().numpy(); rew = rew.cpu().numpy(); done = done.cpu().numpy()
        for lane in (0, n - 1):
            np.testing.assert_allclose(obs[lane], g.obs[t], rtol=RTOL, atol=1e-12, err_msg="%s obs step %d" % (name, t))
            np.testing.assert_allclose(rew[lane], g.reward[t], rtol=RTOL, atol=1e-9, err_msg="%s reward step %d" % (name, t))
            assert int(done[lane]) == int(g.done[t]), "%s done step %d" % (name, t)
        if g.rc is not None:    # info["reactivity_components"] (sim.py:205), the ten terms of the reactor model in the dict's order
            rho = info["reactivity_components"]
            assert list(rho) == g.rc_keys
            for j, k in enumerate(g.rc_keys):
                for lane in (0, n - 1):
                    np.testing.assert_allclose(rho[k][lane].item(), g.rc[t, j], rtol=RTOL, atol=1e-9, err_msg="%s reactivity_components[%s] step %d" % (name, k, t))
        if t + 1 in sampled:
            fs, is_ = _host_state(env)
            compare_state(g, fs[:, 0], is_[:, 0], g.state[sampled[t + 1]], "after step %d" % t)
            compare_state(g, fs[:, n - 1], is_[:, n - 1], g.state[sampled[t + 1]], "after step %d (lane 63)" % t)
        if g.sec is not None and g.sec.shape[1] and t % 7 == 0:
            # the scalar keys of the reference's info["secondary_system"], heat-flow / chemistry-flow tracker outputs included
            res = env.secondary_result()
            checked = 0
            for k, col in res.items():
                if k in ("turbine_efficiency", "turbine_hp_power", "turbine_lp_power") and t in g.pokes:
                    continue   # stale cached stage factors in the reference on a poked step (see test_oracle_replays_golden)
                if k in g.sec_keys:
                    want = g.sec[t, g.sec_keys.index(k)]
                    # energy_balance_error is a ~1e-3 difference of ~3000 MW sums: absolute floor 1e-9 MW
                    np.testing.assert_allclose(col[0].item(), want, rtol=RTOL, atol=1e-9, err_msg="%s secondary_system[%s] step %d" % (name, k, t))
                    checked += 1
            assert checked >= 49, checked   # all 52 recorded scalar keys but the three turbine-internal ones on a poked step


@pytest.mark.parametrize("heat_source,mode", [("constant", "full"), ("reactor", "full"), ("reactor", "primary_sg")])
def test_hip_matches_oracle_on_random_batch(oracle_lib, heat_source, mode):
    """N heterogeneous plants (ragged N, random ICs / actions / setpoints / noise) stepped by both."""
    n, T = 333, 160   # not a multiple of the wave size on purpose
    rng = np.random.default_rng(2024)
    env = _env(n=n, heat_source=heat_source, noise_enabled=True, mode=mode)
    P = oracle_lib.Params()
    P.heat_source = 1 if heat_source == "reactor" else 0
    P.hs_noise_enabled = 1
    P.mode = 1 if mode == "primary_sg" else 0
    ora = oracle_lib.OraclePlants(n, P)
    # heterogeneous initial conditions through the field API on both sides
    flow0 = rng.uniform(15000, 25000, n); flow0[::17] = rng.uniform(3000, 4900, len(flow0[::17]))      # low-flow scrams
    fuel0 = rng.uniform(400, 600, n); fuel0[5::23] = rng.uniform(1210, 1400, len(fuel0[5::23]))         # fuel-temperature scrams
    ics = {"prim.coolant_flow_rate": flow0, "prim.control_rod_position": rng.uniform(80, 100, n),
           "prim.fuel_temperature": fuel0, "prim.coolant_pressure": np.where(rng.random(n) < 0.05, 17.4, 15.5)}
    per_inst = {("pump.oil_level", k): rng.uniform(8.0, 100.0, n) for k in range(4)}
    per_inst.update({("sg.water_level", k): rng.uniform(11.5, 13.5, n) for k in range(3)})
    for name, v in ics.items():
        env.set_field(name, v); ora.set(name, v)
    for (name, k), v in per_inst.items():
        env.set_field(name, v, instance=k); ora.set(name, v, instance=k)
    acts = rng.choice([0, 1, 2, 3, 4, 5, 8, 9, 10], size=(T, n)).astype(np.int32)
    mags = rng.uniform(0, 1, size=(T, n))
    z = rng.standard_normal((T, n))
    sp = 90.0 + 10.0 * np.sin(np.arange(T)[:, None] / 15.0 + np.arange(n)[None, :])
    cw = 25.0 + 3.0 * np.cos(np.arange(T)[:, None] / 30.0 + np.arange(n)[None, :])
    for t in range(T):
        o_obs, o_rew, o_done, o_flags, o_info = ora.step(action=acts[t], magnitude=mags[t], setpoint=sp[t], noise_z=z[t], cw_temp=cw[t])
        obs, rew, done, info = env.step(action=acts[t], magnitude=mags[t], power_setpoint=sp[t], cooling_water_temp=cw[t], noise_z=z[t])
        np.testing.assert_allclose(obs.cpu().numpy(), o_obs, rtol=RTOL, atol=1e-12, err_msg="obs step %d" % t)
        np.testing.assert_allclose(rew.cpu().numpy(), o_rew, rtol=RTOL, atol=1e-9, err_msg="reward step %d" % t)
        assert np.array_equal(done.cpu().numpy(), o_done), "done step %d" % t
        assert np.array_equal(info["trip_flags"].cpu().numpy().astype(np.uint32), o_flags), "trip flags step %d" % t
    f, i = _host_state(env)
    cols = env_cols()
    for pl in range(0, n, 37):
        of, oi = ora.state(pl)
        for kind, slot, label, _p in cols:
            if kind == "i32":
                assert int(i[slot, pl]) == int(oi[slot]), (label, pl)
            else:
                assert abs(f[slot, pl] - of[slot]) <= RTOL * abs(of[slot]) + ATOL_SMALL, (label, pl, f[slot, pl], of[slot])


@pytest.mark.parametrize("n", [1, 63, 65, 129])
def test_ragged_batch_sizes(oracle_lib, n):
    """Batch sizes around the wave width (padding lanes must neither disturb live lanes nor be written to outputs)."""
    import torch
    rng = np.random.default_rng(n)
    env = _env(n=n, noise_enabled=True)
    P = oracle_lib.Params(); P.hs_noise_enabled = 1
    ora = oracle_lib.OraclePlants(n, P)
    guard_obs = torch.full((n + 64, 22), -7.0, dtype=torch.float64, device=env.device)
    env._obs = guard_obs[:n]                       # output buffer with a guard band behind it
    for t in range(12):
        z = rng.standard_normal(n); sp = rng.uniform(60, 100, n)
        o_obs, o_rew, o_done, o_flags, o_info = ora.step(setpoint=sp, noise_z=z)
        obs, rew, done, info = env.step(power_setpoint=sp, noise_z=z)
        np.testing.assert_allclose(obs.cpu().numpy(), o_obs, rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(rew.cpu().numpy(), o_rew, rtol=RTOL, atol=1e-9)
    assert bool((guard_obs[n:] == -7.0).all()), "rows beyond n_plants were written"


@pytest.mark.parametrize("integrator", ["reference", "rk4"])
def test_config2_reactor_and_sg_only_at_4096(oracle_lib, integrator):
    """BASELINE config 2 shape: 4096 plants, point-kinetics heat source + steam generators only, dt = 0.1, random
    actuator actions, in both of its integrator modes: the reference's clipped explicit Euler (the parity path) and "rk4" --
    the point-kinetics equations advanced by RK4 sub-steps inside the step kernel, which the reference does not have
    (tests/test_rk4_cpu.py checks that mode against the closed-form solution; here the kernel must do what the CPU
    restatement does)."""
    n, T = 4096, 40
    rng = np.random.default_rng(4096)
    env = _env(n=n, dt=0.1, heat_source="reactor", mode="primary_sg", integrator=integrator)
    env.set_fields(__import__("nuclear_sim_amd.env", fromlist=["equilibrium_state"]).equilibrium_state())
    P = oracle_lib.Params(); P.dt = 0.1; P.heat_source = 1; P.mode = 1
    P.kinetics_rk4_substeps = env.params.kinetics_rk4_substeps
    assert (P.kinetics_rk4_substeps == 50) == (integrator == "rk4")
    ora = oracle_lib.OraclePlants(n, P)
    from nuclear_sim_amd.env import equilibrium_state
    for key, v in equilibrium_state().items():
        name, inst, k = (key, 0, 0) if not isinstance(key, tuple) else (key[0], key[1], key[2] if len(key) > 2 else 0)
        ora.set(name, v, instance=inst, k=k)
    # rk4 mode: rods down to 40 % (about -3 000 pcm) so that lanes of one wave sit on both sides of the explicit / implicit
    # switch at h a = -2, and a tenth of the plants scrammed by hand at step 10 (rho = -0.5: h a = -101)
    rods = rng.uniform(40 if integrator == "rk4" else 85, 100, n)
    env.set_field("prim.control_rod_position", rods); ora.set("prim.control_rod_position", rods)
    scrammed = (rng.random(n) < 0.1).astype(np.int32)
    for t in range(T):
        if t == 10 and integrator == "rk4":
            env.set_field("prim.scram_status", scrammed); ora.set("prim.scram_status", scrammed)
        acts = rng.choice([0, 1, 2, 3, 4, 5, 8, 9, 10], size=n).astype(np.int32); mags = rng.uniform(0, 1, n)
        o_obs, o_rew, o_done, o_flags, o_info = ora.step(action=acts, magnitude=mags)
        obs, rew, done, info = env.step(action=acts, magnitude=mags)
        np.testing.assert_allclose(obs.cpu().numpy(), o_obs, rtol=RTOL, atol=1e-12, err_msg="obs step %d" % t)
        assert np.array_equal(done.cpu().numpy(), o_done)
        assert np.array_equal(info["trip_flags"].cpu().numpy().astype(np.uint32), o_flags)
        if integrator == "rk4":
            assert not (o_flags & 4).any(), "NaN reset in rk4 mode at step %d" % t     # what an unstable sub-step ends in
    if integrator == "rk4":
        flux = env.get_field("prim.neutron_flux").cpu().numpy()
        assert (flux[scrammed == 1] < 0.02e13).all() and (flux[scrammed == 1] > 1e8).all()      # prompt drop, then delayed decay
    f, i = _host_state(env)
    of, oi = ora.state_all()
    for kind, slot, label, _p in env_cols():
        if kind == "i32":
            assert np.array_equal(i[slot, :n], oi[:, slot]), label
        else:
            np.testing.assert_allclose(f[slot, :n], of[:, slot], rtol=RTOL, atol=ATOL_SMALL, err_msg=label)


def test_config2_as_specified_per_plant_equilibria_2000_steps(oracle_lib):
    """BASELINE config 2 as SURVEY 8(d) C2 writes it: 4 096 plants, ReactorHeatSource, primary + steam generators only, every
    plant started from its own create_equilibrium_state(power ~ U[60, 100] %, rods ~ U[80, 100] %) (default_rng(1234);
    equilibrium_state on arrays, pinned to the reference's constructor by tests/golden/ic_config2_equilibrium.npz), i.i.d. actions
    over {0, 1, 2, 3, 8, 9, 10} with magnitude U[0, 1] (seed 1235), 2 000 steps of the reference's integrator at dt = 1.0 -- the
    parity-gated mode.  The oracle steps 128 sampled plants through all 2 000 steps with the same per-plant inputs; observations,
    done and flags compared every 50th step and at the end every state member."""
    import torch
    from nuclear_sim_amd.env import equilibrium_state, config2_draws
    n, T = 4096, 2000
    env = _env(n=n, dt=1.0, heat_source="reactor", mode="primary_sg")
    power, rods = config2_draws(n)
    ic = equilibrium_state(power, rods)
    env.set_fields(ic)
    rng = np.random.default_rng(77)
    sample = np.unique(np.concatenate([[0, 63, 64, n - 1], rng.choice(n, 124, replace=False)]))
    P = oracle_lib.Params(); P.dt = 1.0; P.heat_source = 1; P.mode = 1
    ora = oracle_lib.OraclePlants(len(sample), P)
    for key, v in ic.items():
        name, inst, k = (key, 0, 0) if not isinstance(key, tuple) else (key[0], key[1], key[2] if len(key) > 2 else 0)
        ora.set(name, np.asarray(v)[sample], instance=inst, k=k)
    arng = np.random.default_rng(1235)
    idx = torch.as_tensor(sample, device=env.device)
    scrams = 0
    for t in range(T):
        acts = arng.choice([0, 1, 2, 3, 8, 9, 10], size=n).astype(np.int32); mags = arng.uniform(0, 1, n)
        obs, rew, done, info = env.step(action=acts, magnitude=mags)
        o_obs, o_rew, o_done, o_flags, _ = ora.step(action=acts[sample], magnitude=mags[sample])
        scrams += int(o_done.sum())
        if t % 50 == 0 or t == T - 1:
            np.testing.assert_allclose(obs[idx].cpu().numpy(), o_obs, rtol=RTOL, atol=1e-12, err_msg="obs step %d" % t)
            assert np.array_equal(done[idx].cpu().numpy(), o_done), t
            assert np.array_equal(info["trip_flags"][idx].cpu().numpy().astype(np.uint32), o_flags), t
    f, i = env.state_arrays()
    f = f[:, idx].cpu().numpy(); i = i[:, idx].cpu().numpy()
    of, oi = ora.state_all()
    for kind, slot, label, _p in env_cols():
        if kind == "i32":
            assert np.array_equal(i[slot], oi[:, slot]), label
        else:
            np.testing.assert_allclose(f[slot], of[:, slot], rtol=RTOL, atol=ATOL_SMALL, err_msg=label)
    flux = env.get_field("prim.neutron_flux").cpu().numpy()
    assert np.isfinite(flux).all() and (flux >= 1e8).all() and (flux <= 1e14).all()      # (after 2 000 steps of random actions every plant sits on a clip: the parity is in the comparisons above)


def test_long_run_stays_on_the_oracle(oracle_lib):
    """1500 steps of one wave of heterogeneous plants: the last-bit differences of the device arithmetic
    (reciprocal multiplication, lean exp/log) must not grow past the parity budget."""
    n, T = 64, 1500
    rng = np.random.default_rng(15)
    env = _env(n=n, noise_enabled=True)
    P = oracle_lib.Params(); P.hs_noise_enabled = 1
    ora = oracle_lib.OraclePlants(n, P)
    lv = rng.uniform(30, 100, n)
    env.set_field("pump.oil_level", lv, instance=1); ora.set("pump.oil_level", lv, instance=1)
    for t in range(T):
        z = rng.standard_normal(n); sp = 85.0 + 15.0 * np.sin(t / 40.0 + np.arange(n))
        ora.step(setpoint=sp, noise_z=z); env.step(power_setpoint=sp, noise_z=z)
    f, i = _host_state(env)
    of, oi = ora.state_all()
    for kind, slot, label, _p in env_cols():
        if kind == "i32":
            assert np.array_equal(i[slot, :n], oi[:, slot]), label
        else:
            np.testing.assert_allclose(f[slot, :n], of[:, slot], rtol=RTOL, atol=ATOL_SMALL, err_msg=label)


def test_hip_maintenance_matches_oracle_on_random_batch(oracle_lib):
    """Automatic maintenance on a ragged batch: oil levels scattered around the 58 % threshold, wear /
    contamination / NPSH scattered around theirs so that every orchestrator branch and most handlers occur somewhere
    in the batch, shortened cooldowns so that orders re-trigger; every maint.*, mpump.* and pump.* column compared with
    the oracle after every step; plus the counting identities of the queue."""
    n, T = 333, 90
    rng = np.random.default_rng(99)
    mp = {"maint_oil_level_cooldown_hours": 1.0, "maint_work_order_cooldown": 24.0, "maint_start_delay_hours": 0.1,
          "maint_top_off_target": 58.5}   # top off just above the threshold so pumps cross it again within the run
    env = _env(n=n, dt=5.0, noise_enabled=True, maintenance=True, params=mp)
    P = oracle_lib.Params(); P.dt = 5.0; P.hs_noise_enabled = 1; P.maint_enabled = 1
    for k, v in mp.items():
        setattr(P, k, v)
    ora = oracle_lib.OraclePlants(n, P)
    for k in range(4):
        for name, lo, hi, frac in (("pump.oil_level", 57.0, 59.5, 1.0), ("pump.oil_contamination", 14.0, 16.0, 0.3),
                                   ("pump.wear_impeller", 7.0, 9.0, 0.2), ("pump.wear_motor_bearings", 7.5, 9.5, 0.2),
                                   ("pump.wear_pump_bearings", 5.5, 7.5, 0.2), ("pump.wear_thrust_bearing", 3.5, 5.5, 0.2),
                                   ("pump.wear_mechanical_seals", 15.0, 17.0, 0.2), ("pump.npsh_available", 16.0, 20.0, 0.3),
                                   ("pump.cavitation_damage", 7.0, 9.0, 0.2), ("pump.antioxidant_level", 2.0, 40.0, 0.2)):
            base = ora.get(name, instance=k)
            v = np.where(rng.random(n) < frac, rng.uniform(lo, hi, n), base)
            env.set_field(name, v, instance=k); ora.set(name, v, instance=k)
    z = rng.standard_normal((T, n))
    cols = [c for c in env_cols() if c[2].startswith(("maint", "mpump", "pump"))]
    for t in range(T):
        ora.step(setpoint=np.full(n, 90.0), noise_z=z[t])
        env.step(power_setpoint=np.full(n, 90.0), noise_z=z[t])
        f, i = _host_state(env)
        of, oi = ora.state_all()
        for kind, slot, label, _p in cols:
            if kind == "i32":
                assert np.array_equal(i[slot, :n], oi[:, slot]), (label, t)
            else:
                np.testing.assert_allclose(f[slot, :n], of[:, slot], rtol=RTOL, atol=ATOL_SMALL, err_msg="%s step %d" % (label, t))
    created = env.get_field("maint.work_orders_created").cpu().numpy()
    done_ = env.get_field("maint.maintenance_actions_performed").cpu().numpy()
    open_ = sum((env.get_field("mpump.wo_order", instance=k, k=a).cpu().numpy() > 0).astype(np.int64) for k in range(4) for a in range(18))
    executed = np.stack([env.get_field("maint.executed", k=a).cpu().numpy() for a in range(18)])
    assert np.array_equal(executed.sum(axis=0), done_)
    assert (executed.sum(axis=1) > 0).sum() >= 8, "the batch reaches most action types: %s" % executed.sum(axis=1)
    assert created.max() > 4, "cooldowns were shortened so that pumps re-trigger"
    assert np.array_equal(created, done_ + open_)          # every order is either executed or still open
    assert done_.max() <= (T * 5.0) / 15.0 + 1             # at most one execution per 15-min check


def test_config4_randomized_oil_top_off_scenario(oracle_lib):
    """BASELINE config 4 at test size: plants with per-seed randomised initial conditions from the scenario
    catalog (nuclear_sim_amd.scenarios), the data-gen runner's settings (dt = 5 min, seeded heat-source noise,
    automatic maintenance), 2 h; every column against the oracle started from the same columns, and the
    event counts bit-exact."""
    from nuclear_sim_amd.env import BatchedPlantEnv
    from nuclear_sim_amd import scenarios
    n, T = 700, 24
    seeds = list(range(1000, 1000 + n))
    env = BatchedPlantEnv.action_test("oil_top_off", seeds)
    P = oracle_lib.Params(); P.dt = 5.0; P.hs_noise_enabled = 1; P.maint_enabled = 1
    ora = oracle_lib.OraclePlants(n, P)
    eff = float(ora.get("pump.lubrication_effectiveness"))
    for key, v in scenarios.action_test_fields("oil_top_off", seeds, eff).items():
        name, inst, k = (key, 0, 0) if not isinstance(key, tuple) else (key[0], key[1], key[2] if len(key) > 2 else 0)
        ora.set(name, v, instance=inst, k=k)
    z = np.random.RandomState(42).standard_normal(T)          # every plant's heat source is seeded 42
    for t in range(T):
        ora.step(setpoint=np.full(n, 90.0), noise_z=np.full(n, z[t]))
        obs, rew, done, info = env.step(power_setpoint=np.full(n, 90.0))   # noise drawn by the env's own seeded streams
    f, i = _host_state(env)
    of, oi = ora.state_all()
    for kind, slot, label, _p in env_cols():
        if kind == "i32":
            assert np.array_equal(i[slot, :n], oi[:, slot]), label
        else:
            np.testing.assert_allclose(f[slot, :n], of[:, slot], rtol=RTOL, atol=ATOL_SMALL, err_msg=label)
    created = env.get_field("maint.work_orders_created").cpu().numpy()
    assert created.max() >= 1 and created.min() == 0, "the scenario mix has plants that trigger within 2 h and plants that do not"


@pytest.mark.parametrize("n,T", [(32768, 36), (65536, 24)])
def test_config4_at_its_size_with_maintenance_against_the_oracle_on_a_sample(oracle_lib, n, T):
    """BASELINE config 4 at its real per-GPU shape (maintenance_scenario_runner.py:349-411, sim.py:208-223): 32 768 plants with
    per-seed randomised oil_top_off initial conditions, dt = 5 min, the automatic maintenance ON -- npb_step4_maint_kernel at
    full occupancy: two groups of four waves per CU polling their LDS progress words beside each other, the rule called with its
    2 KB/lane scratch frame by the waves whose screen fires -- and once more at 65 536 plants (two rounds of groups on the segmented
    arena).  Against the CPU oracle on ~200 sampled plants (first, last, wave and segment boundaries) started from the same
    columns: observations, flags and the event count after every step, every column incl. maint.* / mpump.* at the end; and the
    job-wide histogram of executions (sharding.event_histogram) against the counts column."""
    import torch
    from nuclear_sim_amd.env import BatchedPlantEnv
    from nuclear_sim_amd import scenarios, sharding
    seeds = np.arange(n)
    env = BatchedPlantEnv.action_test("oil_top_off", seeds)
    rng = np.random.default_rng(404)
    edges = [0, 1, 63, 64, 65, 127, 255, 256, 16383, 16384, 16385, n // 2 - 1, n // 2, n - 65, n - 64, n - 1]
    sample = np.unique(np.concatenate([edges, rng.choice(n, 184, replace=False)]))
    P = oracle_lib.Params(); P.dt = 5.0; P.hs_noise_enabled = 1; P.maint_enabled = 1
    ora = oracle_lib.OraclePlants(len(sample), P)
    eff = float(ora.get("pump.lubrication_effectiveness"))
    for key, v in scenarios.action_test_fields("oil_top_off", seeds, eff).items():
        name, inst, k = (key, 0, 0) if not isinstance(key, tuple) else (key[0], key[1], key[2] if len(key) > 2 else 0)
        v = np.asarray(v)
        ora.set(name, v[sample] if v.shape[0] == n else v, instance=inst, k=k)
    z = np.random.RandomState(42).standard_normal(T)          # every plant's heat source is seeded 42, as the runner's
    idx = torch.as_tensor(sample, device=env.device)
    gid = np.arange(n)
    for t in range(T):
        sp = 90.0 + 8.0 * np.sin(2.0 * np.pi * t / (20.0 + gid % 7))      # the runner's profile moves the load; per plant here
        obs, rew, done, info = env.step(power_setpoint=sp)
        assert env.last_step_kernel() == "npb_step4_maint_kernel"
        o_obs, o_rew, o_done, o_flags, _ = ora.step(setpoint=sp[sample], noise_z=np.full(len(sample), z[t]))
        np.testing.assert_allclose(obs[idx].cpu().numpy(), o_obs, rtol=RTOL, atol=1e-12, err_msg="obs step %d" % t)
        assert np.array_equal(done[idx].cpu().numpy(), o_done), t
        assert np.array_equal(info["trip_flags"][idx].cpu().numpy().astype(np.uint32), o_flags), t
        o_count = np.array([ora.get("maint.maintenance_actions_performed", plant=p) for p in range(len(sample))], dtype=np.int64)
        assert np.array_equal(info["maintenance_event_count"][idx].cpu().numpy().astype(np.int64), o_count), t
    f, i = env.state_arrays()
    f = f[:, idx].cpu().numpy(); i = i[:, idx].cpu().numpy()
    of, oi = ora.state_all()
    for kind, slot, label, _p in env_cols():
        if kind == "i32":
            assert np.array_equal(i[slot], oi[:, slot]), label
        else:
            np.testing.assert_allclose(f[slot], of[:, slot], rtol=RTOL, atol=ATOL_SMALL, err_msg=label)
    performed = env.get_field("maint.maintenance_actions_performed")
    assert torch.equal(info["maintenance_event_count"].to(torch.int64), performed.to(torch.int64))
    hist = sharding.event_histogram(performed).cpu().numpy()
    assert hist.sum() == n and np.array_equal(hist, np.bincount(performed.cpu().numpy().astype(np.int64), minlength=16)[:16])
    assert hist[0] > 0 and hist[1:].sum() > n // 4, "the catalog's three scenarios: plants that top off within the run and plants that never do (%s)" % hist[:6]
    from nuclear_sim_amd import _lib
    executed_top_off = env.get_field("maint.executed", k=_lib.MAINT_ACTIONS.index("oil_top_off")).cpu().numpy()
    assert np.array_equal(executed_top_off.astype(np.int64), performed.cpu().numpy().astype(np.int64)), "every execution in this scenario is an oil top-off"


def test_facade_without_a_maintenance_configuration_is_the_references_default():
    """NuclearPlantSimulator(enable_state_management=True) with NO maintenance configuration (the constructor's default): the
    reference gives a feedwater pump one threshold, oil_level < 30 -> oil_top_off, and delays HIGH-priority work by an hour
    (sim.py:97-128, auto_maintenance.py:187-198, state_manager.py _create_default_maintenance_config) -- not the data-gen
    action-test table, which would top a pump off at 58 % at once.  Fixture m14 is that reference run: FWP-1 at 57 % is left
    alone, FWP-2 (25 %) and FWP-3 (29.9 %) get one order each at step 0, carried out one per check after the hour."""
    from nuclear_sim_amd.env import NuclearPlantSimulator, ConstantHeatSource, ControlAction
    g = Golden("m14_default_configuration_maintenance")
    sim = NuclearPlantSimulator(dt=1.0, heat_source=ConstantHeatSource(rated_power_mw=3000.0), secondary_config={"secondary_system": {}},
                                enable_state_management=True)
    for k, lv in ((0, 57.0), (1, 25.0), (2, 29.9)):
        sim._env.set_field("pump.oil_level", np.array([lv]), instance=k)
    cols = {c[2]: j for j, c in enumerate(g.cols)}
    sampled = {int(s): k for k, s in enumerate(g.state_steps)}
    for t in range(g.T):
        r = sim.step(action=ControlAction.NO_ACTION)
        np.testing.assert_allclose(r["observation"], g.obs[t], rtol=RTOL, atol=1e-12, err_msg="obs step %d" % t)
        if t + 1 in sampled:
            row = g.state[sampled[t + 1]]
            assert int(sim._env.get_field("maint.work_orders_created")[0].item()) == int(row[cols["maint.work_orders_created"]]), t
            assert r["info"]["maintenance_event_count"] == int(row[cols["maint.maintenance_actions_performed"]]), t
            for k in range(3):
                np.testing.assert_allclose(sim._env.get_field("pump.oil_level", instance=k)[0].item(), row[cols["pump[%d].oil_level" % k]], rtol=RTOL)
    assert int(sim._env.get_field("maint.maintenance_actions_performed")[0].item()) == 2
    assert sim._env.get_field("pump.oil_level", instance=0)[0].item() < 57.0      # never topped off: 57 % is above the default's 30 %


def test_heat_source_plugin_through_the_facade():
    """The reference's HeatSource plugin interface (heat_sources/heat_source_interface.py:23-112): a user's heat source object
    handed to NuclearPlantSimulator is called once per step on the host and its thermal_power_mw / power_percent reach the
    step as input columns (NPB_HEAT_EXTERNAL).  Fixture h1 is the reference itself running the same scripted source, with
    actuator actions and a cooling-water swing; an object the facade cannot map is refused, not run as the reactor model."""
    from nuclear_sim_amd.env import NuclearPlantSimulator, HeatSource, ControlAction
    g = Golden("h1_heat_source_plugin")

    class Scripted(HeatSource):
        def __init__(self):
            super().__init__(3000.0)
            self.k = 0

        def update(self, dt, **kwargs):
            assert "reactor_state" in kwargs and "control_action" in kwargs          # primary/__init__.py:203-207
            k = self.k; self.k += 1
            return {"thermal_power_mw": 3000.0 * (0.82 + 0.15 * float(np.sin(k / 17.0))), "power_percent": 100.0 * (0.80 + 0.17 * float(np.sin(k / 17.0 + 0.2)))}

    sim = NuclearPlantSimulator(dt=1.0, heat_source=Scripted(), secondary_config={"secondary_system": {}}, enable_state_management=False)
    for t in range(g.T):
        np.testing.assert_allclose([3000.0 * (0.82 + 0.15 * np.sin(t / 17.0)), 100.0 * (0.80 + 0.17 * np.sin(t / 17.0 + 0.2))], [g.noise_z[t], g.setpoint[t]], rtol=1e-15)
        r = sim.step(action=ControlAction(int(g.action[t])), magnitude=float(g.magnitude[t]), cooling_water_temp=float(g.cooling[t]))
        np.testing.assert_allclose(r["observation"], g.obs[t], rtol=RTOL, atol=1e-12, err_msg="obs step %d" % t)
        np.testing.assert_allclose(r["reward"], g.reward[t], rtol=RTOL, atol=1e-9)
        assert r["info"]["reactivity"] == 0.0 and r["info"]["reactivity_components"] == {}      # primary/__init__.py:218-225
    # fixture h2: a plugin that READS the state it is handed -- its power follows the rod position, which the reference moves before
    # it updates the heat source (primary/__init__.py:200-207); numpy scalars in the result's optional keys are "absent" too
    g2 = Golden("h2_heat_source_plugin_reads_state")

    class FollowsRods(HeatSource):
        def __init__(self):
            super().__init__(3000.0)

        def update(self, dt, **kwargs):
            st = kwargs["reactor_state"]
            return {"thermal_power_mw": 3000.0 * (0.5 + 0.005 * st.control_rod_position),
                    "power_percent": 100.0 * (0.5 + 0.005 * st.control_rod_position) - 0.001 * st.steam_valve_position,
                    "reactivity_pcm": np.float64(0.0), "reactivity_components": {}}

    sim2 = NuclearPlantSimulator(dt=1.0, heat_source=FollowsRods(), secondary_config={"secondary_system": {}}, enable_state_management=False)
    for t in range(g2.T):
        r = sim2.step(action=ControlAction(int(g2.action[t])), magnitude=float(g2.magnitude[t]))
        np.testing.assert_allclose(r["observation"], g2.obs[t], rtol=RTOL, atol=1e-12, err_msg="h2 obs step %d" % t)
        np.testing.assert_allclose(r["info"]["thermal_power"], g2.noise_z[t], rtol=1e-12)      # what the reference's plugin returned at this step
    with pytest.raises(TypeError):
        NuclearPlantSimulator(heat_source=object())
    with pytest.raises(ValueError):
        _env(n=4, heat_source="external").step()                      # the plugin's columns are this mode's required inputs
    with pytest.raises(ValueError):
        _env(n=4).step(thermal_power_mw=np.full(4, 3000.0))


def test_config4_counts_held_by_the_reference():
    """BASELINE config 4's headline quantity against the REFERENCE, not the oracle: BatchedPlantEnv.action_test("oil_top_off",
    seeds) for the 64 seeds of tests/golden/counts_c4_64seeds.npz (runner-built reference simulators, all three catalog
    scenarios), 48 steps of 5 min under the recorded set-points -- work_orders_created and maintenance_actions_performed after
    every step bit-exact, and at the end every column: executions by action, final oil levels, open orders, cooldown stamps."""
    from golden_util import Config4Counts
    from nuclear_sim_amd.env import BatchedPlantEnv
    c4 = Config4Counts()
    n = len(c4.seeds)
    env = BatchedPlantEnv.action_test("oil_top_off", c4.seeds)
    f, i = _host_state(env)
    for j in range(n):
        compare_state(c4, f[:, j], i[:, j], c4.initial_state[j], "seed %d initial state" % c4.seeds[j])
    for t in range(c4.T):
        obs, rew, done, info = env.step(power_setpoint=c4.setpoint[:, t])       # noise from the env's own seeded streams (seed 42, as the runner's)
        assert np.array_equal(env.get_field("maint.work_orders_created").cpu().numpy(), c4.created[:, t]), "work_orders_created after step %d" % t
        assert np.array_equal(env.get_field("maint.maintenance_actions_performed").cpu().numpy(), c4.performed[:, t]), "maintenance_actions_performed after step %d" % t
    np.testing.assert_allclose(obs.cpu().numpy(), c4.final_obs, rtol=RTOL, atol=1e-12)
    f, i = _host_state(env)
    for j in range(n):
        compare_state(c4, f[:, j], i[:, j], c4.final_state[j], "seed %d after %d steps" % (c4.seeds[j], c4.T))


def test_single_plant_facade_runs_the_data_gen_loop():
    """Drop-in check: the reference-style loop of MaintenanceScenarioRunner.run_scenario
    (maintenance_scenario_runner.py:383-411: ramped set_power_setpoint on the heat source, then
    sim.step(action=NO_ACTION)) on nuclear_sim_amd's NuclearPlantSimulator, constructed the way the runner
    constructs the reference's (:210-244) with the initial conditions of fixture m1; observations, rewards and the
    maintenance event count must be the reference's."""
    from nuclear_sim_amd.env import NuclearPlantSimulator, ConstantHeatSource, ControlAction
    from nuclear_sim_amd import scenarios
    g = Golden("m1_oil_top_off_staggered")
    ic = dict(scenarios.ACTION_TEST_TEMPLATE["feedwater"]); ic.update(scenarios.OIL_TOP_OFF_CONDITIONS)
    ic["pump_oil_levels"] = [58.3, 58.1, 98.0, 57.0]
    cfg = {"secondary_system": {"feedwater": {"initial_conditions": ic},
                                "steam_generator": {"initial_conditions": scenarios.ACTION_TEST_TEMPLATE["steam_generator"]},
                                "turbine": {"initial_conditions": scenarios.ACTION_TEST_TEMPLATE["turbine"]}},
           "maintenance_system": {"maintenance_mode": "aggressive"}}     # as the composer's action-test configuration has it
    # ... thresholds included (data_gen/config_engine/templates/nuclear_plant_comprehensive_config.yaml:682-1010; without them a
    # simulator gets the state manager's factory default, fixture m14): the live dict of the reference's run, tests/golden/maint_table.json
    import json, os
    from golden_util import GOLDEN_DIR
    rows = json.load(open(os.path.join(GOLDEN_DIR, "maint_table.json")))["thresholds"]
    cfg["maintenance_system"]["component_configs"] = {"feedwater": {"thresholds": {
        r["name"]: {k: r[k] for k in ("threshold", "comparison", "action", "cooldown_hours", "priority", "component_id")} for r in rows}}}
    hs = ConstantHeatSource(rated_power_mw=3000.0, noise_enabled=True, noise_std_percent=0.1, noise_seed=42)
    sim = NuclearPlantSimulator(heat_source=hs, dt=5.0, enable_secondary=True, enable_state_management=True, secondary_config=cfg)
    assert not sim.ignored_initial_conditions
    events = 0
    for t in range(g.T):
        sim.primary_physics.heat_source.set_power_setpoint(float(g.setpoint[t]))
        r = sim.step(action=ControlAction.NO_ACTION)
        np.testing.assert_allclose(r["observation"], g.obs[t], rtol=RTOL, atol=1e-12, err_msg="obs step %d" % t)
        np.testing.assert_allclose(r["reward"], g.reward[t], rtol=RTOL, atol=1e-9)
        assert r["done"] == bool(g.done[t])
        events = r["info"]["maintenance_event_count"]
    lab = [c[2] for c in g.cols]
    assert events == int(g.state[-1, lab.index("maint.maintenance_actions_performed")]) == 3
    assert abs(sim.state.power_level - g.state[-1, lab.index("prim.power_level")]) <= RTOL * 100


@pytest.mark.parametrize("name", ["r1_reset_steady", "r4_reset_feedwater_ic"])
def test_facade_reset_follows_the_reference(name):
    """Drop-in check of reset(): an RL loop that calls NuclearPlantSimulator.reset() in mid-run (sim.py:546-581) gets the
    reference's observation back and the reference's trajectory afterwards -- default configuration (r1) and
    configured feedwater initial conditions, which the reference's reset re-applies (r4)."""
    from nuclear_sim_amd.env import NuclearPlantSimulator, ConstantHeatSource, ControlAction
    g = Golden(name)
    hs = ConstantHeatSource(rated_power_mw=3000.0, noise_enabled=True, noise_std_percent=0.1, noise_seed=g.meta["noise_seed"])
    cfg = {"secondary_system": g.meta["secondary"]} if g.meta.get("secondary") else None
    sim = NuclearPlantSimulator(heat_source=hs, dt=1.0, secondary_config=cfg, enable_state_management=False)   # as the fixture's run
    assert not sim.ignored_initial_conditions
    for t in range(g.T):
        if t in g.resets:
            steady, ref_obs, _ = g.resets[t]
            np.testing.assert_allclose(sim.reset(start_at_steady_state=steady), ref_obs, rtol=RTOL, atol=1e-12, err_msg="reset obs")
        r = sim.step(action=ControlAction.NO_ACTION)
        np.testing.assert_allclose(r["observation"], g.obs[t], rtol=RTOL, atol=1e-12, err_msg="%s obs step %d" % (name, t))
        np.testing.assert_allclose(r["reward"], g.reward[t], rtol=RTOL, atol=1e-9)
        assert r["done"] == bool(g.done[t])


@pytest.mark.parametrize("heat_source,storage", [("constant", "f64"), ("reactor", "f64"), ("constant", "f32")])
def test_the_two_step_kernels_agree(heat_source, storage):
    """npb_step has two kernels (one wavefront per 64 plants with an LDS staging pipeline; two wavefronts that own different
    subsystems, npd_step2.h) and picks one by batch size.  Both call the same device functions in the same per-plant
    order, so every int32 column, flag and counter must be identical and every real within rounding noise of the other
    (the build's -freciprocal-math lets the compiler share reciprocals differently in the two inlining contexts: last-bit
    differences, 1e-12 allowed) -- on a heterogeneous ragged batch with trips, scrams, maintenance and the pump-demand
    gate in its serial mode."""
    n, T = 333, 70
    rng = np.random.default_rng(5)

    def run(variant):
        env = _env(n=n, dt=5.0 if heat_source == "constant" else 1.0, heat_source=heat_source, noise_enabled=True,
                   maintenance=heat_source == "constant", storage=storage)
        env.set_step_kernel(variant)
        launched = set()
        r = np.random.default_rng(11)
        if heat_source == "reactor":
            from nuclear_sim_amd.env import equilibrium_state
            env.set_fields(equilibrium_state())
        env.set_field("prim.coolant_flow_rate", np.where(r.random(n) < 0.1, 4500.0, 20000.0))
        for k in range(4):
            env.set_field("pump.oil_level", r.uniform(8.0, 100.0, n), instance=k)
            env.set_field("pump.npsh_available", np.where(r.random(n) < 0.1, 11.0, 20.0), instance=k)
        env.set_field("pump.status", np.where(r.random(n) < 0.3, 0, 1).astype(np.int32), instance=3)   # spare RUNNING: the demand gate can close
        env.set_field("fw.running_mask", r.choice([0, 3, 7, 15], n).astype(np.int32))
        z = r.standard_normal((T, n)); sp = r.uniform(60, 100, (T, n)); acts = r.choice([0, 1, 3, 8, 8], size=(T, n)).astype(np.int32)
        outs = []
        for t in range(T):
            obs, rew, done, info = env.step(action=acts[t], magnitude=np.ones(n), power_setpoint=sp[t], noise_z=z[t])
            launched.add(env.last_step_kernel())
            outs.append([x.cpu().numpy().copy() for x in (obs, rew, done, info["trip_flags"], info["electrical_power"], info["condenser_pressure"])])
        want = KERNEL_OF_VARIANT[variant].replace("_kernel", "_maint_kernel") if env.params.maint_enabled else KERNEL_OF_VARIANT[variant]
        assert launched == {want}, (variant, launched)      # the kernel that was meant is the kernel that ran
        f, i = _host_state(env)
        return outs, f, i

    o1, f1, i1 = run(1)
    tol = 1e-12 if storage == "f64" else 3e-7     # fp32 storage: a last-bit difference before the rounding can move the float
    for variant in (2, 3, 5):    # the two builds of the two-wave kernel (whole register file / room for two waves per SIMD), the four-wave kernel
        o2, f2, i2 = run(variant)
        assert np.array_equal(i1, i2)
        np.testing.assert_allclose(f1, f2, rtol=tol, atol=1e-300, equal_nan=True)
        for a, b in zip(o1, o2):
            for x, y in zip(a, b):
                if x.dtype.kind == "f":
                    np.testing.assert_allclose(x, y, rtol=tol, atol=1e-300, equal_nan=True)
                else:
                    assert np.array_equal(x, y)
    assert (i1 != 0).any()
    # the streaming build of the one-wave kernel (what batches far past the Infinity Cache take) differs from it in the cache
    # policy of its state stores only: every column and every output to the bit
    o4, f4, i4 = run(4)
    assert np.array_equal(i1, i4) and np.array_equal(f1.view(np.int64), f4.view(np.int64))
    for a, b in zip(o1, o4):
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)


def test_results_do_not_depend_on_how_the_batch_is_split_into_handles():
    """contiguous shards by global plant id give results independent of the number of GPUs.  One 200-plant
    episode as ONE handle and as TWO handles (100 + 100: neither a multiple of the wave size, so plants sit in
    different lanes and waves): every column and every output bit-identical for the same global plant."""
    import torch
    n, T = 200, 30
    rng = np.random.default_rng(21)
    oil = rng.uniform(56.0, 100.0, (4, n)); rods = rng.uniform(80, 100, n)
    z = rng.standard_normal((T, n)); sp = rng.uniform(60, 100, (T, n)); acts = rng.choice([0, 1, 3, 8, 8], size=(T, n)).astype(np.int32)

    def episode(lo, hi):
        env = _env(n=hi - lo, dt=5.0, noise_enabled=True, maintenance=True)
        env.set_step_kernel(2)
        for k in range(4):
            env.set_field("pump.oil_level", oil[k, lo:hi], instance=k)
        env.set_field("prim.control_rod_position", rods[lo:hi])
        for t in range(T):
            obs, rew, done, info = env.step(action=acts[t, lo:hi], magnitude=np.ones(hi - lo), power_setpoint=sp[t, lo:hi], noise_z=z[t, lo:hi])
        f, i = _host_state(env)
        return f, i, obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), info["trip_flags"].cpu().numpy().copy()

    whole = episode(0, n)
    parts = [episode(0, 100), episode(100, n)]
    for j in range(5):
        joined = np.concatenate([p_[j] for p_ in parts], axis=1 if j < 2 else 0)
        assert np.array_equal(whole[j], joined, equal_nan=True), j


def test_lane_independence_and_determinism():
    """Plants are independent: perturbing one plant's state and inputs must leave every other plant's state and
    outputs bit-identical (also across the wave-level decisions: store elision ballots, the turbine stage pass's
    fast-path ballot), and two identical runs must agree bit for bit."""
    n, T = 200, 25
    rng = np.random.default_rng(77)
    z = rng.standard_normal((T, n)); sp = rng.uniform(60, 100, (T, n))
    acts = rng.choice([0, 1, 3, 8, 8, 8], size=(T, n)).astype(np.int32)

    def run(perturb):
        env = _env(n=n, heat_source="reactor", noise_enabled=False)
        from nuclear_sim_amd.env import equilibrium_state
        env.set_fields(equilibrium_state())
        zz, ss, aa = z.copy(), sp.copy(), acts.copy()
        if perturb:
            for victim in (5, 70, 199):   # one per wave, incl. the ragged last wave
                lv = np.full(n, 100.0); lv[victim] = 9.0                      # trips that plant's pump 0
                env.set_field("pump.oil_level", lv, instance=0)
                fl = env.get_field("prim.coolant_flow_rate").cpu().numpy(); fl[victim] = 4000.0   # scrams it
                env.set_field("prim.coolant_flow_rate", fl)
                aa[:, victim] = 3
        outs = []
        for t in range(T):
            obs, rew, done, info = env.step(action=aa[t], magnitude=np.ones(n), power_setpoint=ss[t], noise_z=zz[t])
            outs.append((obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), info["trip_flags"].cpu().numpy().copy()))
        f, i = _host_state(env)
        return outs, f, i

    o1, f1, i1 = run(False)
    o2, f2, i2 = run(False)
    assert np.array_equal(f1, f2, equal_nan=True) and np.array_equal(i1, i2)
    o3, f3, i3 = run(True)
    keep = np.ones(n, dtype=bool); keep[[5, 70, 199]] = False
    assert np.array_equal(f1[:, :n][:, keep], f3[:, :n][:, keep], equal_nan=True)
    assert np.array_equal(i1[:, :n][:, keep], i3[:, :n][:, keep])
    for (a, b, c), (d, e, g_) in zip(o1, o3):
        assert np.array_equal(a[keep], d[keep]) and np.array_equal(b[keep], e[keep]) and np.array_equal(c[keep], g_[keep])
    assert not np.array_equal(f1[:, 5], f3[:, 5])


def env_cols():
    from nuclear_sim_amd.schema import SCHEMA
    return SCHEMA.columns()


def test_reset_mask_and_field_roundtrip():
    import torch
    env = _env(n=130)
    env.step(); env.step()
    before_f, before_i = _host_state(env)
    mask = np.zeros(130, dtype=np.uint8); mask[::2] = 1
    env.reset(mask=mask)
    f, i = _host_state(env)
    fresh = _env(n=130)
    ff, fi = _host_state(fresh)
    assert np.array_equal(f[:, ::2], ff[:, ::2]) and np.array_equal(i[:, ::2], fi[:, ::2])       # reset lanes
    assert np.array_equal(f[:, 1::2], before_f[:, 1::2]) and np.array_equal(i[:, 1::2], before_i[:, 1::2])  # untouched lanes
    v = torch.arange(130, dtype=torch.float64, device=env.device)
    env.set_field("pump.oil_level", v, instance=2)
    assert torch.equal(env.get_field("pump.oil_level", instance=2), v)


def test_full_size_properties():
    """BASELINE size (65 536 plants): size-independent properties instead of an oracle run --
    identical plants stay identical, a permutation of the plants permutes the outputs, and the
    obs block is finite and consistent with the state columns."""
    import torch
    n = 65536
    env = _env(n=n, noise_enabled=True)
    rng = np.random.default_rng(7)
    z = rng.standard_normal(n)
    sp = rng.uniform(70, 100, n)
    perm = rng.permutation(n)
    env2 = _env(n=n, noise_enabled=True)
    for _ in range(5):
        o1, r1, d1, _i1 = env.step(power_setpoint=sp, noise_z=z)
        o2, r2, d2, _i2 = env2.step(power_setpoint=sp[perm], noise_z=z[perm])
    o1 = o1.cpu().numpy(); o2 = o2.cpu().numpy()
    assert np.all(np.isfinite(o1))
    assert np.array_equal(o1[perm], o2)
    assert np.array_equal(r1.cpu().numpy()[perm], r2.cpu().numpy())
    sf = env.get_field("sec.total_steam_flow").cpu().numpy()
    np.testing.assert_allclose(o1[:, 14], sf / 1665, rtol=2e-7, atol=0)   # an output member: kept as float in the arena


@pytest.mark.parametrize("storage,n,T", [("f64", 65536, 60), ("f32", 65536, 60), ("f64", 131072, 30), ("f64", 40960, 60), ("f64", 32768, 40), ("f64", 81900, 30)])
def test_full_size_against_the_oracle_on_a_sample(oracle_lib, storage, n, T):
    """BASELINE config 3 (and, with fp32 state storage, config 5's size) at its full size -- 65 536 plants, the bench's workload (per-plant load-following setpoints, per-plant
    noise), the kernel and the arena placement the bench runs with -- against the CPU oracle on 192 of the plants spread over
    the whole batch (first, last, and the wave boundaries included): plants are independent, so the oracle steps just
    those with their own inputs.  Every state member, observation, reward and flag of the sampled plants.  (fp32 storage: against
    the oracle with its state rounded to float after every step, the same algorithm.)  At 131 072 plants npb_step takes the
    streaming build of the one-wave kernel (two rounds of waves, state stores past the caches), at 40 960 the 256-register
    build of the two-wave kernel (32 769 .. 57 344 plants), at 32 768 -- BASELINE config 4's share per GPU -- the
    four-wave kernel, at 65 536 and at 81 900 (ragged: the last group of 64 is partly padding, the last segment of the arena partly
    empty) the four-wave kernel on the handle's segmented arena; each run asserts that kernel."""
    import torch
    import bench
    want_kernel = {65536: "npb_step4_kernel", 131072: "npb_step_nt_kernel", 40960: "npb_step2_kernel", 32768: "npb_step4_kernel", 81900: "npb_step4_kernel"}[n]
    assert bench.step_kernel_name(n, storage, forced="0") == want_kernel
    rng = np.random.default_rng(2024)
    sample = np.unique(np.concatenate([[0, 1, 63, 64, 65, 127, n - 65, n - 64, n - 1], rng.choice(n, 183, replace=False)]))
    env = _env(n=n, noise_enabled=True, storage=storage)
    narrow = storage == "f32"
    tol = F32_EMU_RTOL if narrow else RTOL
    P = oracle_lib.Params(); P.hs_noise_enabled = 1
    ora = oracle_lib.OraclePlants(len(sample), P)
    gid = np.arange(n); period = 600.0 + 60.0 * (gid % 16)
    for k in range(4):      # heterogeneous pumps, as everywhere else
        lv = rng.uniform(30.0, 100.0, n)
        env.set_field("pump.oil_level", lv, instance=k); ora.set("pump.oil_level", lv[sample], instance=k)
    if narrow:
        ora.round_state_f32()
    for t in range(T):
        sp = 90.0 + 10.0 * np.sin(2.0 * np.pi * t / period)
        z = rng.standard_normal(n)
        obs, rew, done, info = env.step(power_setpoint=sp, noise_z=z)
        assert env.last_step_kernel() == want_kernel
        o_obs, o_rew, o_done, o_flags, _ = ora.step(setpoint=sp[sample], noise_z=z[sample])
        if narrow:
            ora.round_state_f32()
        idx = torch.as_tensor(sample, device=env.device)
        np.testing.assert_allclose(obs[idx].cpu().numpy(), o_obs, rtol=tol, atol=1e-12)
        np.testing.assert_allclose(rew[idx].cpu().numpy(), o_rew, rtol=tol, atol=1e-6 if narrow else 1e-9)
        assert np.array_equal(done[idx].cpu().numpy(), o_done)
        assert np.array_equal(info["trip_flags"][idx].cpu().numpy().astype(np.uint32), o_flags)
    f, i = env.state_arrays()
    f = f[:, idx].cpu().numpy(); i = i[:, idx].cpu().numpy()
    of, oi = ora.state_all()        # [sampled plants, members]
    assert np.array_equal(i, oi.T)
    np.testing.assert_allclose(f, of.T, rtol=tol, atol=1e-9 if narrow else ATOL_SMALL)


# ---------------------------------------------------------------------------------------------------
# fp32 state storage (BASELINE config 5): fp64 arithmetic, columns kept as float in HBM
F32_EMU_RTOL = 2e-5     # against the oracle with its state rounded to float every step (same algorithm)
F32_OBS_RTOL = 1e-4     # against the plain fp64 oracle (BASELINE config 5's bar, on observations)


def _f32_pair(oracle_lib, n, **kw):
    env = _env(n=n, storage="f32", **kw)
    P = oracle_lib.Params()
    P.dt = kw.get("dt", 1.0)
    P.heat_source = 1 if kw.get("heat_source") == "reactor" else 0
    P.hs_noise_enabled = 1 if kw.get("noise_enabled") else 0
    P.maint_enabled = 1 if kw.get("maintenance") else 0
    return env, P


@pytest.mark.parametrize("heat_source", ["constant", "reactor"])
def test_fp32_storage_matches_rounding_oracle(oracle_lib, heat_source):
    """The fp32-storage build computes in fp64 and rounds the carried state to float once per step.  The oracle
    with its state rounded to float between steps is that same algorithm, so the two must agree far inside the
    mode's 1e-4 budget: integer columns and trip flags exactly, real columns to a few float ulps; and the
    observations must stay within 1e-4 of the plain fp64 oracle."""
    n, T = 200, 150
    rng = np.random.default_rng(515)
    env, P = _f32_pair(oracle_lib, n, heat_source=heat_source, noise_enabled=True)
    assert env.handle_step_bytes_per_plant() < env.step_bytes_per_plant()
    emu = oracle_lib.OraclePlants(n, P)
    ref = oracle_lib.OraclePlants(n, P)
    ics = {"prim.coolant_flow_rate": rng.uniform(15000, 25000, n), "prim.control_rod_position": rng.uniform(85, 100, n),
           "prim.fuel_temperature": rng.uniform(450, 600, n)}
    for name, v in ics.items():
        env.set_field(name, v); emu.set(name, v); ref.set(name, v)
    for k in range(4):
        v = rng.uniform(40.0, 100.0, n)
        env.set_field("pump.oil_level", v, instance=k); emu.set("pump.oil_level", v, instance=k); ref.set("pump.oil_level", v, instance=k)
    emu.round_state_f32()
    acts = rng.choice([0, 1, 4, 5, 8, 9, 10], size=(T, n)).astype(np.int32)
    mags = rng.uniform(0, 1, size=(T, n))
    z = rng.standard_normal((T, n))
    sp = 90.0 + 10.0 * np.sin(np.arange(T)[:, None] / 15.0 + np.arange(n)[None, :])
    worst = 0.0
    for t in range(T):
        e_obs, e_rew, e_done, e_flags, _ = emu.step(action=acts[t], magnitude=mags[t], setpoint=sp[t], noise_z=z[t])
        emu.round_state_f32()
        r_obs, _r, _d, _f, _ = ref.step(action=acts[t], magnitude=mags[t], setpoint=sp[t], noise_z=z[t])
        obs, rew, done, info = env.step(action=acts[t], magnitude=mags[t], power_setpoint=sp[t], noise_z=z[t])
        obs = obs.cpu().numpy()
        np.testing.assert_allclose(obs, e_obs, rtol=F32_EMU_RTOL, atol=1e-9, err_msg="obs vs rounding oracle, step %d" % t)
        np.testing.assert_allclose(rew.cpu().numpy(), e_rew, rtol=F32_EMU_RTOL, atol=1e-7, err_msg="reward step %d" % t)
        assert np.array_equal(done.cpu().numpy(), e_done), "done step %d" % t
        assert np.array_equal(info["trip_flags"].cpu().numpy().astype(np.uint32), e_flags), "trip flags step %d" % t
        np.testing.assert_allclose(obs, r_obs, rtol=F32_OBS_RTOL, atol=1e-7, err_msg="obs vs fp64 oracle, step %d" % t)
        worst = max(worst, float(np.max(np.abs(obs - r_obs) / np.maximum(np.abs(r_obs), 1e-3))))
    f, i = _host_state(env)
    of, oi = emu.state_all()
    for kind, slot, label, _p in env_cols():
        if kind == "i32":
            assert np.array_equal(i[slot, :n], oi[:, slot]), label
        else:
            np.testing.assert_allclose(f[slot, :n], of[:, slot], rtol=F32_EMU_RTOL, atol=1e-7, err_msg=label)
    print("fp32 storage: worst observation deviation from the fp64 oracle %.2e" % worst)


def test_fp32_storage_field_roundtrip_and_maintenance(oracle_lib):
    """get/set_field convert between the fp64 ABI and the float columns; the maintenance rule (its own kernel,
    same storage type) still matches the rounding oracle, counters bit-exact."""
    import torch
    n, T = 130, 40
    rng = np.random.default_rng(61)
    env, P = _f32_pair(oracle_lib, n, dt=5.0, noise_enabled=True, maintenance=True)
    v = torch.tensor(rng.uniform(20, 90, n), dtype=torch.float64, device=env.device)
    env.set_field("pump.oil_level", v, instance=2)
    back = env.get_field("pump.oil_level", instance=2)
    assert torch.equal(back, v.float().double())          # rounded to float exactly once
    host = env.get_field("pump.oil_level", instance=2).cpu().numpy()
    env.set_field("pump.oil_level", host, instance=2)     # host buffers go through the conversion column
    assert torch.equal(env.get_field("pump.oil_level", instance=2), back)
    emu = oracle_lib.OraclePlants(n, P)
    for k in range(4):
        lv = rng.uniform(57.0, 59.5, n)
        env.set_field("pump.oil_level", lv, instance=k); emu.set("pump.oil_level", lv, instance=k)
    emu.round_state_f32()
    z = rng.standard_normal((T, n))
    cols = [c for c in env_cols() if c[2].startswith(("maint", "pump"))]
    for t in range(T):
        emu.step(setpoint=np.full(n, 90.0), noise_z=z[t]); emu.round_state_f32()
        env.step(power_setpoint=np.full(n, 90.0), noise_z=z[t])
    f, i = _host_state(env)
    of, oi = emu.state_all()
    for kind, slot, label, _p in cols:
        if kind == "i32":
            assert np.array_equal(i[slot, :n], oi[:, slot]), label
        else:
            np.testing.assert_allclose(f[slot, :n], of[:, slot], rtol=F32_EMU_RTOL, atol=1e-7, err_msg=label)
    assert env.get_field("maint.maintenance_actions_performed").max().item() >= 1


def test_fp32_storage_replays_golden_within_1e4():
    """The reference's own trajectories (fixtures) replayed with fp32 storage: observations within 1e-4."""
    for name in ("s1_constant_steady", "s2_reactor_actions", "s5_load_following", "m1_oil_top_off_staggered"):
        g = Golden(name)
        n = 64
        env = _env(g, n=n, storage="f32")
        f0, i0 = _host_state(env)
        f, i, fm, im = g.split_state(g.state[0])
        f0[fm, :] = f[fm, None]; i0[im, :] = i[im, None]
        env.load_state_arrays(f0, i0)
        if g.pokes:
            continue
        for t in range(g.T):
            sp = None if np.isnan(g.setpoint[t]) else g.setpoint[t]
            cw = None if np.isnan(g.cooling[t]) else g.cooling[t]
            obs, rew, done, info = env.step(action=int(g.action[t]), magnitude=float(g.magnitude[t]), power_setpoint=sp,
                                            cooling_water_temp=cw, noise_z=float(g.noise_z[t]))
            np.testing.assert_allclose(obs[0].cpu().numpy(), g.obs[t], rtol=F32_OBS_RTOL, atol=1e-7, err_msg="%s step %d" % (name, t))
            assert int(done[0].item()) == int(g.done[t])


def _config5_scripts(n, T):
    """transient script per plant chosen by i mod 4, interleaved so that every wave diverges:
    (0) none, (1) DECREASE_COOLANT_FLOW from step 100 on, (2) CONTROL_ROD_WITHDRAW alternating with DILUTE_BORON,
    (3) state pokes: fuel_temperature = 1300 at step 150, oil level of pump 0 = 9 % at step 200.
    In the reference's clipped physics (1) ends at the 5 000 kg/s floor, just above the low-flow scram, and (2)
    saturates at rods 100 % / boron 0 without an excursion, so neither scrams by itself; (2) therefore also gets
    the S4a poke (neutron_flux = 1.3e13, overpower scram) at step 250."""
    kind = np.arange(n) % 4
    acts = np.full((T, n), 8, dtype=np.int32)
    acts[100:, kind == 1] = 3
    acts[50::2, kind == 2] = 1
    acts[51::2, kind == 2] = 9
    return kind, acts


@pytest.mark.parametrize("storage", ["f64", "f32"])
def test_config5_transients(oracle_lib, storage):
    """BASELINE config 5 / SURVEY 8d C5 on 2048 interleaved plants, ReactorHeatSource, 600 steps, against the fp64
    oracle: fp64 storage to 1e-6, fp32 storage to 1e-4 on the observations; the step at which each plant scrams
    and the step at which each trip flag first rises must be the oracle's exactly in both."""
    from nuclear_sim_amd.env import equilibrium_state
    n, T = 2048, 600
    env = _env(n=n, heat_source="reactor", storage=storage)
    P = oracle_lib.Params(); P.heat_source = 1
    ora = oracle_lib.OraclePlants(n, P)
    env.set_fields(equilibrium_state())
    for key, v in equilibrium_state().items():
        name, inst, k = (key, 0, 0) if not isinstance(key, tuple) else (key[0], key[1], key[2] if len(key) > 2 else 0)
        ora.set(name, v, instance=inst, k=k)
    kind, acts = _config5_scripts(n, T)
    rtol = RTOL if storage == "f64" else F32_OBS_RTOL
    first_done = np.full(n, -1); o_first_done = np.full(n, -1)
    first_flag = {}; o_first_flag = {}
    worst = 0.0
    for t in range(T):
        if t == 150:
            v = np.where(kind == 3, 1300.0, env.get_field("prim.fuel_temperature").cpu().numpy())
            ov = np.where(kind == 3, 1300.0, np.array([ora.get("prim.fuel_temperature", plant=p) for p in range(n)]))
            env.set_field("prim.fuel_temperature", v); ora.set("prim.fuel_temperature", ov)
        if t == 250:
            v = np.where(kind == 2, 1.3e13, env.get_field("prim.neutron_flux").cpu().numpy())
            ov = np.where(kind == 2, 1.3e13, np.array([ora.get("prim.neutron_flux", plant=p) for p in range(n)]))
            env.set_field("prim.neutron_flux", v); ora.set("prim.neutron_flux", ov)
        if t == 200:
            v = np.where(kind == 3, 9.0, env.get_field("pump.oil_level").cpu().numpy())
            ov = np.where(kind == 3, 9.0, np.array([ora.get("pump.oil_level", plant=p) for p in range(n)]))
            env.set_field("pump.oil_level", v); ora.set("pump.oil_level", ov)
        o_obs, _r, o_done, o_flags, _i = ora.step(action=acts[t])
        obs, _rew, done, info = env.step(action=acts[t])
        obs = obs.cpu().numpy(); done = done.cpu().numpy(); flags = info["trip_flags"].cpu().numpy().astype(np.uint32)
        np.testing.assert_allclose(obs, o_obs, rtol=rtol, atol=1e-7 if storage == "f32" else 1e-12, err_msg="obs step %d" % t)
        worst = max(worst, float(np.max(np.abs(obs - o_obs) / np.maximum(np.abs(o_obs), 1e-3))))
        first_done = np.where((first_done < 0) & (done != 0), t, first_done)
        o_first_done = np.where((o_first_done < 0) & (o_done != 0), t, o_first_done)
        for bit in range(12):
            m = (flags >> bit) & 1; om = (o_flags >> bit) & 1
            a = first_flag.setdefault(bit, np.full(n, -1)); b = o_first_flag.setdefault(bit, np.full(n, -1))
            first_flag[bit] = np.where((a < 0) & (m != 0), t, a); o_first_flag[bit] = np.where((b < 0) & (om != 0), t, b)
    assert np.array_equal(first_done, o_first_done), "scram step indices"
    for bit in range(12):
        assert np.array_equal(first_flag[bit], o_first_flag[bit]), "first step of trip flag bit %d" % bit
    assert (first_done[kind == 3] == 150).all() and (first_done[kind == 2] >= 250).all() and (first_done[kind < 2] < 0).all()
    assert (first_flag[8][kind == 3] >= 200).all(), "pump 0 trips on low oil level after the poke at step 200"
    print("config 5 (%s storage): worst observation deviation %.2e, scrams %d of %d" % (storage, worst, int((first_done >= 0).sum()), n))


@pytest.mark.parametrize("storage", ["f64", "f32"])
def test_config5_transients_at_full_size(oracle_lib, storage):
    """The same at BASELINE config 5's own size, 65 536 interleaved plants (every wave holds all four transient scripts),
    300 steps (all three pokes inside), the oracle on 256 plants spread over the batch: observations to the mode's
    tolerance, scram steps and the first step of every trip flag exact."""
    import torch
    from nuclear_sim_amd.env import equilibrium_state
    n, T = 65536, 300
    rng = np.random.default_rng(55)
    sample = np.unique(np.concatenate([np.arange(8), [n - 4, n - 3, n - 2, n - 1], rng.choice(n, 244, replace=False)]))
    m = len(sample)
    env = _env(n=n, heat_source="reactor", storage=storage)
    P = oracle_lib.Params(); P.heat_source = 1
    ora = oracle_lib.OraclePlants(m, P)
    env.set_fields(equilibrium_state())
    for key, v in equilibrium_state().items():
        name, inst, k = (key, 0, 0) if not isinstance(key, tuple) else (key[0], key[1], key[2] if len(key) > 2 else 0)
        ora.set(name, v, instance=inst, k=k)
    kind, acts = _config5_scripts(n, T)
    ks = kind[sample]
    assert all((ks == j).sum() >= 30 for j in range(4))
    idx = torch.as_tensor(sample, device=env.device)
    rtol = RTOL if storage == "f64" else F32_OBS_RTOL
    first_done = np.full(m, -1); o_first_done = np.full(m, -1)
    first_flag = {}; o_first_flag = {}

    def poke(name, which, value):
        v = np.where(kind == which, value, env.get_field(name).cpu().numpy())
        ov = np.where(ks == which, value, np.array([ora.get(name, plant=q) for q in range(m)]))
        env.set_field(name, v); ora.set(name, ov)
    for t in range(T):
        if t == 150:
            poke("prim.fuel_temperature", 3, 1300.0)
        if t == 200:
            poke("pump.oil_level", 3, 9.0)
        if t == 250:
            poke("prim.neutron_flux", 2, 1.3e13)
        o_obs, _r, o_done, o_flags, _i = ora.step(action=acts[t][sample])
        obs, _rew, done, info = env.step(action=acts[t])
        obs = obs[idx].cpu().numpy(); done = done[idx].cpu().numpy(); flags = info["trip_flags"][idx].cpu().numpy().astype(np.uint32)
        np.testing.assert_allclose(obs, o_obs, rtol=rtol, atol=1e-7 if storage == "f32" else 1e-12, err_msg="obs step %d" % t)
        first_done = np.where((first_done < 0) & (done != 0), t, first_done)
        o_first_done = np.where((o_first_done < 0) & (o_done != 0), t, o_first_done)
        for bit in range(12):
            a = first_flag.setdefault(bit, np.full(m, -1)); b = o_first_flag.setdefault(bit, np.full(m, -1))
            first_flag[bit] = np.where((a < 0) & (((flags >> bit) & 1) != 0), t, a)
            o_first_flag[bit] = np.where((b < 0) & (((o_flags >> bit) & 1) != 0), t, b)
    assert np.array_equal(first_done, o_first_done), "scram step indices"
    for bit in range(12):
        assert np.array_equal(first_flag[bit], o_first_flag[bit]), "first step of trip flag bit %d" % bit
    assert (first_done[ks == 3] == 150).all() and (first_done[ks == 2] >= 250).all() and (first_done[ks < 2] < 0).all()
    # and over the whole batch: every plant of script 3 scrammed, none of scripts 0 and 1
    scram = (env.get_field("prim.scram_status").cpu().numpy() != 0)
    assert scram[kind == 3].all() and scram[kind == 2].all() and not scram[kind < 2].any()


def test_nan_state_propagates_like_the_reference(oracle_lib):
    """np.clip and Python's max / min pass a NaN first operand through; the device code clips with the hardware
    min / max (which drop NaN) plus a term that restores exactly that (npd_common.h).  Poke NaN into secondary-side
    columns of some plants and require the same NaN pattern and the same finite values as the oracle, which clips
    by compare-and-select; the primary's check_for_nan_values reset (primary/__init__.py:247-270) is part of it."""
    n, T = 192, 6
    rng = np.random.default_rng(77)
    env = _env(n=n, noise_enabled=True)
    P = oracle_lib.Params(); P.hs_noise_enabled = 1
    ora = oracle_lib.OraclePlants(n, P)
    pokes = [("sg.water_level", 1), ("pump.oil_level", 0), ("turb.rotor_speed", 0), ("cond.condenser_pressure", 0),
             ("prim.fuel_temperature", 0), ("fw.total_flow_rate", 0), ("pump.motor_temperature", 2), ("sg.secondary_pressure", 2)]
    for j, (name, inst) in enumerate(pokes):
        v = env.get_field(name, instance=inst).cpu().numpy().copy()
        v[j::len(pokes) * 2] = np.nan            # every 16th plant, a different column each; plants 8..15 mod 16 stay clean
        env.set_field(name, v, instance=inst); ora.set(name, v, instance=inst)
    for t in range(T):
        z = rng.standard_normal(n)
        o_obs, o_rew, o_done, o_flags, _ = ora.step(noise_z=z)
        obs, rew, done, info = env.step(noise_z=z)
        obs = obs.cpu().numpy()
        assert np.array_equal(np.isnan(obs), np.isnan(o_obs)), "NaN pattern of the observations, step %d" % t
        np.testing.assert_allclose(obs, o_obs, rtol=RTOL, atol=1e-12, equal_nan=True, err_msg="obs step %d" % t)
        assert np.array_equal(done.cpu().numpy(), o_done)
        assert np.array_equal(info["trip_flags"].cpu().numpy().astype(np.uint32), o_flags), "trip flags step %d" % t
    f, i = _host_state(env)
    of, oi = ora.state_all()
    for kind, slot, label, _p in env_cols():
        if kind == "i32":
            assert np.array_equal(i[slot, :n], oi[:, slot]), label
        else:
            assert np.array_equal(np.isnan(f[slot, :n]), np.isnan(of[:, slot])), "NaN pattern of " + label
            np.testing.assert_allclose(f[slot, :n], of[:, slot], rtol=RTOL, atol=ATOL_SMALL, equal_nan=True, err_msg=label)
    clean = (np.arange(n) % 16) >= 8
    assert np.isnan(of[~clean]).any() and not np.isnan(of[clean]).any()


def test_state_log_matches_the_references_log(tmp_path):
    """the columnar state log sampled every step of the m1 data-gen run against the reference's OWN log of the
    same run (tests/golden/log_m1_oil_top_off_staggered.npz = `sim.state_manager.data`): every log column the map claims
    (265 of the reference's 784 numeric columns, several per member, unit factors applied; 81 plain functions of end-of-step
    state; 15 keys of the step's secondary result) must hold the reference's values at every step, under the reference's
    column names."""
    import os
    import pyarrow.parquet as pq
    from golden_util import GOLDEN_DIR
    from nuclear_sim_amd.statelog import StateLog, reference_log_columns, derived_log_columns, result_log_columns
    g = Golden("m1_oil_top_off_staggered")
    z = np.load(os.path.join(GOLDEN_DIR, "log_m1_oil_top_off_staggered.npz"))
    ref_names = [str(x) for x in z["names"]]; ref = z["log"]
    assert ref.shape == (g.T, len(ref_names))
    n = 64
    env = _env(g, n=n)
    f0, i0 = _host_state(env)
    f, i, fm, im = g.split_state(g.state[0])
    f0[fm, :] = f[fm, None]; i0[im, :] = i[im, None]
    env.load_state_arrays(f0, i0)
    log = StateLog(env, every=1, capacity=g.T)
    assert len(log.columns) >= 150
    for t in range(g.T):
        sp = None if np.isnan(g.setpoint[t]) else g.setpoint[t]
        env.step(action=int(g.action[t]), magnitude=float(g.magnitude[t]), power_setpoint=sp, noise_z=float(g.noise_z[t]))
        assert log.maybe_record(t + 1, (t + 1) * env.dt)
    tab = log.table(plants=[0, n - 1])
    assert tab.num_rows == g.T * 2 and tab.column_names[:3] == ["step", "time", "plant"]
    lc = reference_log_columns()
    assert len(lc) >= 250 and set(lc) <= set(tab.column_names) and set(lc) <= set(ref_names)
    derived = derived_log_columns()      # plain functions of the end-of-step state (pump factors, wear sums, SG system averages ...)
    assert len(derived) >= 80 and set(derived) <= set(tab.column_names) and set(derived) <= set(ref_names) and not set(derived) & set(lc)
    results = result_log_columns()       # keys of the step's secondary result dict (heat-flow tracker, stage-system efficiency ...)
    assert len(results) >= 15 and set(results) <= set(tab.column_names) and set(results) <= set(ref_names) and not set(results) & (set(lc) | set(derived))
    checked = 0
    for name in list(lc) + list(derived) + list(results):
        mine = tab[name].to_numpy().reshape(g.T, 2)
        want = ref[:, ref_names.index(name)]
        # 2.5 MW x (TSP pressure-drop ratio - 1): the ratio is an output member, kept as float in the arena (6e-8 of 1), so the
        # penalty is good to 2e-7 MW of a 5 MW pump -- an absolute floor instead of 1e-6 of a value that is itself ~1e-7
        floor = 1e-6 if name.endswith("fouling_energy_penalty_mw") else 1e-9
        for lane in (0, 1):
            ok = np.abs(mine[:, lane] - want) <= RTOL * np.abs(want) + floor
            assert ok.all(), (name, int(np.argmin(ok)), mine[~ok, lane][:3], want[~ok][:3])
        checked += g.T
    assert checked > 10000
    path = str(tmp_path / "m1.parquet")
    log.write_parquet(path, plants=[0])
    back = pq.read_table(path)
    assert back.num_rows == g.T and np.allclose(back["secondary.feedwater_FWP-1.oil_level"].to_numpy(), ref[:, ref_names.index("secondary.feedwater_FWP-1.oil_level")], rtol=1e-9)


def test_facade_reads_and_pokes_through_reference_attribute_paths():
    """The single-plant facade answers the reference's own attribute chains (they are the quoted paths of the schema), so
    a loop that reads or pokes the reference's object tree runs unchanged: here the S7 poke -- oil level of FWP-1 to
    9 % -- written exactly as the reference harness writes it, trips that pump on the next step."""
    from nuclear_sim_amd.env import NuclearPlantSimulator, ConstantHeatSource
    sim = NuclearPlantSimulator(dt=1.0, heat_source=ConstantHeatSource(noise_enabled=False))
    pump = sim.secondary_physics.feedwater_system.pump_system.pumps['FWP-1']
    assert pump.lubrication_system.oil_level == sim._env.get_field("pump.oil_level", instance=0)[0].item() == 100.0
    sg1 = sim.secondary_physics.steam_generator_system.steam_generators[1]
    assert sg1.secondary_pressure == sim._env.get_field("sg.secondary_pressure", instance=1)[0].item()
    assert sim.primary_physics.state.control_rod_position == sim.state.control_rod_position
    assert isinstance(pump.state.trip_active, int) and pump.state.trip_active == 0
    sim.step()
    pump.lubrication_system.oil_level = 9.0
    assert pump.lubrication_system.oil_level == 9.0
    r = sim.step()
    assert pump.state.trip_active == 1 and (int(r["info"]["trip_flags"]) >> 8) & 1
    sec = r["info"]["secondary_system"]     # the reference's result keys that are plant state, under the reference's names
    assert sec["electrical_power_mw"] == sim._env.get_field("sec.electrical_power_output")[0].item() == r["info"]["electrical_power"]
    # (the result carries the fp64 value the step computed; the fw.total_flow_rate column keeps it as a float output)
    assert sec["feedwater_total_flow"] == pytest.approx(sim._env.get_field("fw.total_flow_rate")[0].item(), rel=1e-6) and len(sec) >= 15
    sim.secondary_physics._previous_sg_conditions['levels'][0] = 16.2       # the other S7 poke, dict-and-index syntax
    assert sim.secondary_physics._previous_sg_conditions['levels'][0] == 16.2
    with pytest.raises(AttributeError):
        sim.secondary_physics.turbine.no_such_attribute
    with pytest.raises(AttributeError):
        sim.secondary_physics.feedwater_system.pump_system.pumps['FWP-9']


@pytest.mark.parametrize("action", ["seal_replacement", "oil_change", "motor_bearing_replacement", "pump_inspection",
                                    "tsp_chemical_cleaning", "scale_removal", "vacuum_leak_detection"])
def test_other_feedwater_action_scenarios_run_on_the_oracle_track(oracle_lib, action):
    """Other randomised action-test scenarios -- feedwater (pre-degraded seals / oil / bearings, reduced NPSH), steam
    generator (TSP and tube-scale deposits), condenser (air in-leakage): plants
    built from nuclear_sim_amd.scenarios on the GPU and on the oracle from the same columns, 2 h of the runner's loop,
    every column compared, the maintenance columns of the feedwater scenarios included (the whole threshold table and
    all thirteen handlers run on the device; work orders of steam generators and the condenser are outside the path)."""
    from nuclear_sim_amd.env import BatchedPlantEnv
    from nuclear_sim_amd import scenarios
    n, T = 192, 24
    seeds = list(range(500, 500 + n))
    env = BatchedPlantEnv.action_test(action, seeds)
    P = oracle_lib.Params(); P.dt = 5.0; P.hs_noise_enabled = 1; P.maint_enabled = 1
    ora = oracle_lib.OraclePlants(n, P)
    eff = float(ora.get("pump.lubrication_effectiveness"))
    for key, v in scenarios.action_test_fields(action, seeds, eff).items():
        name, inst, k = (key, 0, 0) if not isinstance(key, tuple) else (key[0], key[1], key[2] if len(key) > 2 else 0)
        ora.set(name, v, instance=inst, k=k)
    z = np.random.RandomState(42).standard_normal(T)
    for t in range(T):
        o_obs, _r, o_done, o_flags, _ = ora.step(setpoint=np.full(n, 90.0), noise_z=np.full(n, z[t]))
        obs, rew, done, info = env.step(power_setpoint=np.full(n, 90.0))
        np.testing.assert_allclose(obs.cpu().numpy(), o_obs, rtol=RTOL, atol=1e-12, err_msg="obs step %d" % t)
        assert np.array_equal(info["trip_flags"].cpu().numpy().astype(np.uint32), o_flags), "trip flags step %d" % t
    f, i = _host_state(env)
    of, oi = ora.state_all()
    for kind, slot, label, _p in env_cols():
        if kind == "i32":
            assert np.array_equal(i[slot, :n], oi[:, slot]), label
        else:
            np.testing.assert_allclose(f[slot, :n], of[:, slot], rtol=RTOL, atol=ATOL_SMALL, err_msg=label)


def test_state_log_diagnostics_match_the_references_log():
    """The step-internal columns of the state log (per turbine stage: inlet / outlet pressure and temperature, power output,
    loading factor -- TurbineStage.get_state_dict, left over from inside the expansion; per steam generator: the primary
    temperatures it was given, the overall heat-transfer coefficient, the feedwater flow the fouled TSPs let through): written by the diagnostics build of
    the step kernel (npb_set_diagnostics), against the reference's own log of the m1 run at every step; and that build
    leaves every state column and output exactly as the plain one-wave kernel does."""
    import os
    import torch
    from golden_util import GOLDEN_DIR
    from nuclear_sim_amd.statelog import StateLog, diagnostic_log_columns, reference_log_columns, derived_log_columns, result_log_columns
    g = Golden("m1_oil_top_off_staggered")
    z = np.load(os.path.join(GOLDEN_DIR, "log_m1_oil_top_off_staggered.npz"))
    ref_names = [str(x) for x in z["names"]]; ref = z["log"]
    n = 64
    envs = []
    for diag in (True, False):
        env = _env(g, n=n)
        f0, i0 = _host_state(env)
        f, i, fm, im = g.split_state(g.state[0])
        f0[fm, :] = f[fm, None]; i0[im, :] = i[im, None]
        env.load_state_arrays(f0, i0)
        env.set_step_kernel(1)
        envs.append(env)
    log = StateLog(envs[0], every=1, capacity=g.T, diagnostics=True)
    assert envs[0].diagnostics is not None and envs[1].__dict__.get("diagnostics") is None
    for t in range(g.T):
        sp = None if np.isnan(g.setpoint[t]) else g.setpoint[t]
        outs = [env.step(action=int(g.action[t]), magnitude=float(g.magnitude[t]), power_setpoint=sp, noise_z=float(g.noise_z[t])) for env in envs]
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), t
        log.record(t + 1, (t + 1) * envs[0].dt)
    (fa, ia), (fb, ib) = _host_state(envs[0]), _host_state(envs[1])
    assert np.array_equal(fa.view(np.int64), fb.view(np.int64)) and np.array_equal(ia, ib)
    tab = log.table(plants=[0, n - 1])
    dc = diagnostic_log_columns()
    others = set(reference_log_columns()) | set(derived_log_columns()) | set(result_log_columns())
    assert len(dc) >= 95 and set(dc) <= set(tab.column_names) and set(dc) <= set(ref_names) and not set(dc) & others
    for name in dc:
        mine = tab[name].to_numpy().reshape(g.T, 2)
        want = ref[:, ref_names.index(name)]
        for lane in (0, 1):
            ok = np.abs(mine[:, lane] - want) <= RTOL * np.abs(want) + 1e-9
            assert ok.all(), (name, int(np.argmin(ok)), mine[~ok, lane][:3], want[~ok][:3])


@pytest.mark.parametrize("fixture", ["m1_oil_top_off_staggered", "e1_eventful_log", "l1_reactor_log", "l2_feedwater_events_log", "l3_turbine_sg_events_log"])
def test_state_log_reproduces_the_references_log_column_by_column(fixture):
    """SURVEY 8f-3 as a whole: the state log with diagnostics on, sampled every step, against the reference's OWN log
    (sim.state_manager.data) of five runs -- the quiet m1 run; the eventful e1 run (pump trip on low oil, NPSH collapse, load
    and cooling-water swings, worn components, a fouled steam generator, a hot turbine bearing); and the three runs of round 4 that
    move what those leave at rest: l1, a ReactorHeatSource plant under rod / boron / flow / valve actions into a scram (default
    configuration: the reference's other provider naming); l2, four kinds of maintenance on four pumps, the pH controller out of
    ammonia, an NPSH collapse, then every pump tripped on SG level; l3, TSP deposits into the fouling model's shutdown, a rotor
    overspeed excursion, a vibration trip, the ejectors' rotation -- under the reference's column names: state members,
    functions of end-of-step state, keys of the step's secondary result, step counters, the step's own outputs, step-internal
    diagnostics from the diagnostics build, windows over logged history, and the 86 columns the reference itself never moves.
    All 784 columns at every step."""
    import os
    from golden_util import GOLDEN_DIR
    from nuclear_sim_amd import statelog
    g = Golden(fixture)
    z = np.load(os.path.join(GOLDEN_DIR, "log_%s.npz" % fixture))
    ref_names = [str(x) for x in z["names"]]; ref = z["log"]
    assert ref.shape == (g.T, len(ref_names)) and len(ref_names) == 784
    n = 64
    env = _env(g, n=n)
    f0, i0 = _host_state(env)
    f, i, fm, im = g.split_state(g.state[0])
    f0[fm, :] = f[fm, None]; i0[im, :] = i[im, None]
    env.load_state_arrays(f0, i0)
    log = statelog.StateLog(env, every=1, capacity=g.T, diagnostics=True)
    for t in range(g.T):
        for label, v in g.pokes.get(t, []):
            kind, slot = g.label_slot(label)
            env._set_slot(kind, slot, np.full(n, v))
        sp = None if np.isnan(g.setpoint[t]) else g.setpoint[t]
        cw = None if np.isnan(g.cooling[t]) else g.cooling[t]
        env.step(action=int(g.action[t]), magnitude=float(g.magnitude[t]), power_setpoint=sp, cooling_water_temp=cw, noise_z=float(g.noise_z[t]))
        log.record(t + 1, (t + 1) * env.dt)
    tab = log.table(plants=[0, n - 1])
    produced = [c for c in tab.column_names if c not in ("step", "time", "plant")]
    assert set(produced) <= set(ref_names)
    not_produced = sorted(set(ref_names) - set(produced))
    assert not_produced == [] and len(produced) == 784
    poked = set(g.pokes)
    for name in produced:
        mine = tab[name].to_numpy().reshape(g.T, 2)
        want = ref[:, ref_names.index(name)]
        floor = 1e-6 if name.endswith("fouling_energy_penalty_mw") else 1e-9          # (an output member kept as float: see the m1 test above)
        for lane in (0, 1):
            ok = np.abs(mine[:, lane] - want) <= RTOL * np.abs(want) + floor
            # the stage system's own efficiency on a step whose state was poked: the reference's stages expand with factors cached
            # the step before (see test_hip_replays_golden)
            if "turbine" in name:
                for t in poked:
                    if t < g.T:
                        ok[t] = True
            assert ok.all(), (name, int(np.argmin(ok)), mine[~ok, lane][:3], want[~ok][:3])


@pytest.mark.parametrize("variant", [0, 1])
def test_segmented_arena_is_only_a_layout(variant, monkeypatch):
    """A handle of 45 057 .. 114 688 plants keeps its arena in segments of 16 384 plants (include/npb.h, npb_state_arena): every
    kernel that addresses the arena -- init, field get / set, the gather, reset, observe, the step kernels, the maintenance rule --
    moves its base pointer by its plant's segment and otherwise runs the same code, so a ragged batch stepped on a segmented arena
    and on a one-block arena (NPB_ARENA_SEGMENT=0) must agree in every column and every output to the bit, with the four-wave
    kernel npb_step picks there and with the one-wave kernel forced."""
    import torch
    n, T = 50000, 6          # ragged: the last group is partly padding, the last (fourth) segment 848 plants

    def run(segment):
        if segment is None:
            monkeypatch.delenv("NPB_ARENA_SEGMENT", raising=False)
        else:
            monkeypatch.setenv("NPB_ARENA_SEGMENT", segment)
        env = _env(n=n, dt=5.0, noise_enabled=True, maintenance=True)
        assert int(env.L.npb_state_arena_segment(env._h)) == (16384 if segment is None else 0)
        # the raw-arena entry points: a segmented arena's pointer is only handed out together with its segment size
        import ctypes
        ptr = ctypes.c_void_p(); pitch = ctypes.c_size_t(); seg = ctypes.c_size_t(); ncol = ctypes.c_int(); stor = ctypes.c_int()
        rc_old = env.L.npb_state_arena(env._h, ctypes.byref(ptr), ctypes.byref(pitch), ctypes.byref(stor))
        assert (rc_old != 0) == (segment is None), "npb_state_arena refuses a segmented arena"
        assert env.L.npb_state_arena_layout(env._h, ctypes.byref(ptr), ctypes.byref(pitch), ctypes.byref(seg), ctypes.byref(ncol), ctypes.byref(stor)) == 0
        assert ptr.value and int(seg.value) == (16384 if segment is None else 0) and int(pitch.value) == (16384 if segment is None else (n + 63) // 64 * 64) and ncol.value == 796
        env.set_step_kernel(variant)
        r = np.random.default_rng(3)
        for k in range(4):
            env.set_field("pump.oil_level", r.uniform(8.0, 100.0, n), instance=k)
        env.set_field("prim.coolant_flow_rate", np.where(r.random(n) < 0.1, 4500.0, 20000.0))
        outs = []
        for t in range(T):
            obs, rew, done, info = env.step(power_setpoint=r.uniform(60, 100, n), noise_z=r.standard_normal(n))
            outs.append([x.cpu().numpy().copy() for x in (obs, rew, done, info["trip_flags"], info["electrical_power"], info["maintenance_event_count"])])
        assert env.last_step_kernel() == ("npb_step4_maint_kernel" if variant == 0 else "npb_step_maint_kernel")
        outs.append([env.reset(mask=torch.as_tensor(r.random(n) < 0.01, device=env.device), reference=True).cpu().numpy().copy()])
        f, i = _host_state(env)
        return outs, f, i

    o_seg, f_seg, i_seg = run(None)
    o_one, f_one, i_one = run("0")
    assert np.array_equal(i_seg, i_one) and np.array_equal(f_seg.view(np.int64), f_one.view(np.int64))
    for a, b in zip(o_seg, o_one):
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
    assert (i_seg != 0).any()


def test_largest_handle_uses_the_whole_32bit_offset_range():
    """One handle at the top of what npb_create accepts (1 000 000 plants: the last arena column starts 4.07 GB into
    the arena, 95 % of the kernel's 32-bit byte offsets): the first and the last wave must do exactly what the same
    plants do in a small batch -- an offset that wrapped would read or write somebody else's column."""
    import torch
    n = 1_000_000
    big = _env(n=n, noise_enabled=True)
    rng = np.random.default_rng(123)
    pick = np.r_[0:64, n - 64:n]
    small = _env(n=len(pick), noise_enabled=True)
    big.set_step_kernel(0); small.set_step_kernel(1)      # by batch size: the one-wave kernel's streaming build; same arithmetic, so bits can be compared
    lv = rng.uniform(20, 100, n)
    big.set_field("pump.oil_level", lv, instance=3); small.set_field("pump.oil_level", lv[pick], instance=3)
    tr = rng.uniform(300, 360, n)
    big.set_field("turb.rotor_temperature", tr); small.set_field("turb.rotor_temperature", tr[pick])
    for t in range(4):
        z = rng.standard_normal(n); sp = rng.uniform(70, 100, n)
        ob, rb, db, ib = big.step(power_setpoint=sp, noise_z=z)
        os_, rs, ds, is_ = small.step(power_setpoint=sp[pick], noise_z=z[pick])
    idx = torch.as_tensor(pick, device=big.device)
    assert torch.equal(ob[idx], os_) and torch.equal(rb[idx], rs) and torch.equal(ib["trip_flags"][idx], is_["trip_flags"])
    assert bool(torch.isfinite(ob).all())
    for name, inst in (("sec.operating_hours", 0), ("cond.rotation_timer", 0), ("pump.oil_level", 3), ("maint.last_check_time", 0)):
        assert torch.equal(big.get_field(name, instance=inst)[idx], small.get_field(name, instance=inst)), name
    big.close()


def test_c_abi_accepts_null_inputs_and_outputs():
    """npb_step with every optional pointer NULL (include/npb.h: defaults NO_ACTION, magnitude 1, setpoint / cooling
    unchanged, noise 0; no outputs wanted) advances the state exactly like a call that passes the defaults explicitly."""
    import ctypes
    import torch
    from nuclear_sim_amd import _lib
    n = 100
    a = _env(n=n); b = _env(n=n)
    none = ctypes.c_void_p(None)
    for _ in range(3):
        _lib.check(a.L.npb_step(a._h, none, none, none, none, none, none, none, none, none, none, a._stream()), a._h)
        b.step(action=np.full(n, 8, dtype=np.int32), magnitude=np.ones(n), noise_z=np.zeros(n))
    torch.cuda.synchronize()
    fa, ia = _host_state(a); fb, ib = _host_state(b)
    assert np.array_equal(fa, fb, equal_nan=True) and np.array_equal(ia, ib)
    obs = a.get_observation()
    assert torch.equal(obs, b.get_observation())


def test_two_handles_on_two_streams_are_independent():
    """Distinct handles are independent (include/npb.h): two batches stepped on two HIP streams at the same time end
    where the same batches end when stepped one after the other."""
    import torch
    n, T = 4096, 25
    rng = np.random.default_rng(8)
    z = rng.standard_normal((T, 2, n)); sp = rng.uniform(70, 100, (2, n))
    seq = [_env(n=n, noise_enabled=True) for _ in range(2)]
    for t in range(T):
        for k in range(2):
            seq[k].step(power_setpoint=sp[k], noise_z=z[t, k])
    torch.cuda.synchronize()
    par = [_env(n=n, noise_enabled=True) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    zs = [torch.as_tensor(z[:, k], device=par[k].device) for k in range(2)]
    sps = [torch.as_tensor(sp[k], device=par[k].device) for k in range(2)]
    torch.cuda.synchronize()
    for t in range(T):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                par[k].step(power_setpoint=sps[k], noise_z=zs[k][t])
    torch.cuda.synchronize()
    for k in range(2):
        f1, i1 = _host_state(seq[k]); f2, i2 = _host_state(par[k])
        assert np.array_equal(f1, f2, equal_nan=True) and np.array_equal(i1, i2), "handle %d" % k


def test_set_params_takes_effect_at_the_next_step(oracle_lib):
    """npb_set_params: a new time step mid-run (the reference's sim.dt) applies from the next step on, like the oracle's."""
    import ctypes
    from nuclear_sim_amd import _lib
    n = 96
    env = _env(n=n, noise_enabled=True)
    P = oracle_lib.Params(); P.hs_noise_enabled = 1
    ora = oracle_lib.OraclePlants(n, P)
    rng = np.random.default_rng(4)
    for t in range(12):
        if t == 6:
            env.params.dt = 0.25; P.dt = 0.25
            _lib.check(env.L.npb_set_params(env._h, ctypes.byref(env.params)), env._h)
        z = rng.standard_normal(n)
        o_obs = ora.step(noise_z=z)[0]
        obs = env.step(noise_z=z)[0].cpu().numpy()
        np.testing.assert_allclose(obs, o_obs, rtol=RTOL, atol=1e-12, err_msg="step %d" % t)
    assert abs(env.get_field("prim.sim_time")[0].item() - (6 * 1.0 + 6 * 0.25)) < 1e-12


def test_facade_without_a_secondary_side_and_with_the_reactor_models_terms(tmp_path):
    """NuclearPlantSimulator(enable_secondary=False) (sim.py:155,309,333): twelve observations, the primary keys of info
    only; with the reactor heat source info["reactivity_components"] carries the model's ten terms and they add up to
    info["reactivity"]; info["datetime"] advances by dt minutes per step from the state manager's random start date and
    is None without state management; secondary_config_file is read as the reference reads it (YAML, its
    secondary_system section)."""
    import datetime
    from nuclear_sim_amd.env import NuclearPlantSimulator, NuclearPlantEnv, ConstantHeatSource, ReactorHeatSource, ControlAction
    g = Golden("p1_primary_only_reactor")
    sim = NuclearPlantSimulator(dt=1.0, heat_source=ReactorHeatSource(), enable_secondary=False, enable_state_management=False)
    assert sim.secondary_physics is None
    f0, i0 = _host_state(sim._env)
    f, i, fm, im = g.split_state(g.state[0])
    f0[fm, :] = f[fm, None]; i0[im, :] = i[im, None]
    sim._env.load_state_arrays(f0, i0)
    for t in range(40):
        r = sim.step(ControlAction(int(g.action[t])), magnitude=float(g.magnitude[t]))
        assert r["observation"].shape == (12,)
        np.testing.assert_allclose(r["observation"], g.obs[t, :12], rtol=RTOL, atol=1e-12)
        np.testing.assert_allclose(r["reward"], g.reward[t], rtol=RTOL, atol=1e-9)
        inf = r["info"]
        assert "secondary_system" not in inf and "electrical_power" not in inf and inf["datetime"] is None
        np.testing.assert_allclose([inf["thermal_power"], inf["reactivity"], inf["time"]], g.info[t, [0, 1, 8]], rtol=RTOL, atol=1e-9)
        assert list(inf["reactivity_components"]) == g.rc_keys
        np.testing.assert_allclose(list(inf["reactivity_components"].values()), g.rc[t], rtol=RTOL, atol=1e-9)
        assert sum(inf["reactivity_components"].values()) == pytest.approx(inf["reactivity"], rel=1e-12, abs=1e-9)
    assert sim.get_observation().shape == (12,) and NuclearPlantEnv(enable_secondary=False, heat_source="constant").observation_space_size == 12
    # the state manager's clock
    sim = NuclearPlantSimulator(dt=5.0, heat_source=ConstantHeatSource(noise_enabled=False))
    a = datetime.datetime.fromisoformat(sim.step()["info"]["datetime"])
    sim.reset()
    r = sim.step()
    b = datetime.datetime.fromisoformat(r["info"]["datetime"])
    assert b - a == datetime.timedelta(minutes=5.0) and 2020 <= a.year <= 2030 and r["info"]["reactivity_components"] == {}
    # a configuration file
    import yaml
    cfg = {"secondary_system": {"feedwater": {"initial_conditions": {"pump_oil_levels": [61.0, 62.0, 63.0, 64.0]}}},
           "maintenance_system": {"maintenance_mode": "conservative"}}
    path = tmp_path / "plant.yaml"
    path.write_text(yaml.safe_dump(cfg))
    sim = NuclearPlantSimulator(dt=1.0, heat_source=ConstantHeatSource(noise_enabled=False), secondary_config_file=str(path))
    assert [sim._env.get_field("pump.oil_level", instance=k)[0].item() for k in range(4)] == [61.0, 62.0, 63.0, 64.0]
    assert sim._env.params.maint_start_delay_hours == 1.0


def test_placement_probing_leaves_the_construction_state(monkeypatch):
    """npb_create times the step kernel on candidate arenas when the step's working set is about the size of the Infinity Cache
    (65 536 plants) and keeps the fastest: the handle it returns must be in the construction-time state all the same, and
    behave exactly like one created with the probe turned off."""
    import torch
    n = 65536
    z = np.random.default_rng(3).standard_normal((3, n))
    sp = np.linspace(70.0, 100.0, n)

    def run():
        env = _env(n=n, noise_enabled=True)
        f0, i0 = env.state_arrays()
        outs = []
        for t in range(3):
            obs, rew, done, info = env.step(power_setpoint=sp, noise_z=z[t])
            outs.append((obs.clone(), rew.clone(), info["trip_flags"].clone()))
        f1, i1 = env.state_arrays()
        return f0, i0, f1, i1, outs

    monkeypatch.setenv("NPB_PLACEMENT_PROBE", "0")
    a = run()
    monkeypatch.setenv("NPB_PLACEMENT_PROBE", "1")
    b = run()
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)
    for (o1, r1, f1), (o2, r2, f2) in zip(a[4], b[4]):
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(f1, f2)
