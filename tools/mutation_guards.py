"""Why a surviving mutant of the restatement cannot show (companion of tools/mutate_oracle.py --annotate; CPU, test infrastructure).

A survivor is either a hole in the fixtures or a change no state of the reference can observe.  Three classes are recognised
mechanically (mutate_oracle.annotate):
    equality   `<` <-> `<=`, `>` <-> `>=`: the two programs differ only when both operands are equal to the last bit;
    sliver     the threshold of a comparison scaled by 1.001 while no dropped / forced branch and no reversed comparison of the same line
               survives (the comparison is exercised: where its branch was never taken, the mutants INSIDE the branch survive and show up as
               unexplained): the mutant moves the threshold by 0.1 % and it takes a value inside that 0.1 % to tell (tools/mutant_fuzz.py found
               such values for some: those have fixtures c12 ... c15 with pokes computed to land there);
    guarded    the entries below: (file, text the mutated line contains, operators or None for all, the guard[, mutated tokens or all]).
A guard names the line of the restatement (or of the reference) that keeps the mutated token from mattering at the reference's constants.
Everything listed here was also stepped against the original on 27 000 random plant-steps with every member scaled by 0.3 ... 3
(tools/mutant_fuzz.py) without a single differing bit, unless the entry says what reaches it.
"""

RATED = "sim.reset() hands initialize_to_steady_state the rated power (sim.py:558-563: primary_physics.thermal_power_mw was zeroed the line before), so thermal_power_mw = 3000 in every call: load_demand = 100, delta_t = 33.74 K against a 'realistic' 34 K, inlet 326.7 C, outlet 293 C, the efficiency band is the first one, feedwater 1665 kg/s = 3 pumps at 100 %"
TSAT35 = "the condenser pressure it is called with is steam_partial + air_partial = max(0.005, 0.007 - p_air) + p_air with p_air in [0.0001, 0.005] (npo_condenser.h:177-180), i.e. 0.007 ... 0.010 MPa; the Antoine constants are the mmHg ones applied to bar and give a negative temperature there, which the [35, 45] clip of the same line returns as 35.0: every token of the function but that 35.0 is unobservable"
EJECTOR_REQUEST = "an ejector delivers min(available, request) (npo_condenser.h:160); request <= air leakage (capped at 0.15 kg/s) + 50 x pressure error (<= 0.003 MPa) = 0.3 kg/s, available >= 25 x sqrt(1.1) x 0.5 x 0.6 x 0.7 x 0.59 = 3.2 kg/s: the available capacity never binds, and the motive pressure is the literal 1.2 MPa the caller passes (npo_secondary.h:142): motive_p = 1.1"
COUPLING5 = "the coupling clips every loop's hot leg to at least its cold leg + 5 K (npo_primary.h:346, sim.py:418-420): primary_temp_in - primary_temp_out >= 5 K in every call"
FLOW03 = "primary flow per loop = 5700 x max(0.3, power fraction) (npo_primary.h:324-325): never below 1710 kg/s, tube velocity never below 1.76 m/s"
GUARDS = [
    # ---- npo_primary.h
    ("npo_primary.h", "const double PC[5]", None, "the L-stable branch of the rk4 mode (no reference counterpart): held to the matrix exponential of the same system at 1e-9 by tests/test_rk4_cpu.py; 0.1 % on the z^4 coefficient of the stability polynomial is below that on the slow modes (|z| < 0.08) and the prompt mode is damped to nothing either way"),
    ("npo_primary.h", "if (st < 3) { yn = n + w * kn[st]", None, "st == 3 would compute a trial point that nothing reads: the same program"),
    ("npo_primary.h", "s->neutron_flux = npo_clip(n, 1e8, 1e14);", ["const"], "rk4 mode: n was held below FLUX_CEILING = 1e14 by the line before the loop's end; the upper bound here never binds (the lower one is pinned by tests/test_rk4_cpu.py::test_the_flux_floor_and_the_power_it_reports)", ['1e14']),
    ("npo_primary.h", "flux_dot = (eff - BETA) / LAMBDA_PROMPT", ["sign"], "inside |rho| >= 0.01: |(rho -+ beta) / Lambda| >= 350 / s with either sign of beta and the sign of rho, so the +-10 % per second clip two lines below returns the same bound"),
    ("npo_primary.h", "max_change = s->neutron_flux * 0.0001", None, "dead: the enclosing branch is |rho| >= 0.01 (point_kinetics.py:47-50 as the reference wrote it), so the three narrower bands are never selected"),
    ("npo_primary.h", "max_change = s->neutron_flux * 0.001;", None, "dead: see the band above it"),
    ("npo_primary.h", "max_change = s->neutron_flux * 0.01;", None, "dead: see the bands above it"),
    ("npo_primary.h", "if (P->hs_noise_enabled) {", ["branch"], "with the noise off z = 0 and the filter state starts at 0: alpha * 0 + (1 - alpha) * 0 stays 0, the forced branch adds 0"),
    ("npo_primary.h", "reynolds = npo_pymax(reynolds, 1000.0);", None, "swallowed: at Re = 1000 the coefficient is 1.7e5 W/K, far below the 10e6 lower clip four lines on, which binds up to Re = 1.6e5"),
    ("npo_primary.h", "COOLANT_HEAT_CAPACITY = 5200.0", None, "a constant the reference declares and never uses (thermal_hydraulics.py:30)"),
    ("npo_primary.h", "double delta_t_core = (total_primary_flow > 0)", None, "total_primary_flow = 17100 x max(0.3, power fraction) > 0"),
    ("npo_primary.h", "cold_leg_temp = npo_clip(cold_leg_temp, 285.0, 300.0);", None, "cold leg = 293 + 2 (pf - 1) + 3 (steam flow / 1665 - 1): 288 ... 299 C over power fractions 0 ... 3.5 and steam flows 0 ... 1.2 x design; the clip never binds"),
    ("npo_primary.h", "loop_cold = npo_clip(loop_cold, 285.0, 300.0);", None, "the same cold leg +- 0.46 K of loop variation: never outside [285, 300]"),
    ("npo_primary.h", "hot_leg_temp = npo_clip(hot_leg_temp, cold_leg_temp + 5.0, 350.0);", ["const", "sign"], "the first of the two identical lines (npo_primary.h:331): its lower bound binds only below 4.45 % power (delta_t_core < 5 K), where the next line assigns the same cold_leg + 5 on a plant's first step and line 338 recomputes the clip on every later one", ["5.0", "+"]),
    ("npo_primary.h", "hot_leg_temp = npo_clip(hot_leg_temp, cold_leg_temp + 5.0, 350.0);", ["const"], "the 350 C bound: delta_t_core = 3e6 pf / (17100 max(0.3, pf) 5.2) <= 33.74 K over a cold leg <= 300 C", ['350.0']),
    ("npo_primary.h", "loop_hot = npo_clip(loop_hot, loop_cold + 5.0, 350.0);", ["const"], "the 350 C bound: see the hot leg", ['350.0']),
    ("npo_primary.h", "double pressure_dot = npo_clip(-0.01 * pressure_error, -0.05, 0.05);", ["const"], "the bound that survives is the lower one (the upper is reached from 10.05 MPa in fixture c11): -0.05 needs a pressure more than 5 MPa above its set-point, and the pressure is clipped to 20 against a set-point of at least 15.3", ["0.05"]),
    # ---- npo_chem.h
    ("npo_chem.h", "c->water_aggressiveness = npo_clip(1.0 + iron_effect", ["const"], "the lower bound: 1.0 + 0.05 + three non-negative terms >= 1.05", ['0.5']),
    ("npo_chem.h", "double blend_factor = npo_pymin(0.05 * dt_hours * 0.1, 0.5);", ["const"], "dt_hours <= 100 / 60 by the unit guess above it (a dt above 100 is divided by 3600): the blend is at most 0.0083, the 0.5 cap never binds", ['0.5']),
    ("npo_chem.h", "double concentration_factor = 1.0 / (0.02 + 0.01);", None, "a constant expression, 33.3, cut to 5.0 by the cap on the next line whatever 0.1 % does to it"),
    ("npo_chem.h", "if (concentration_factor > 1.1) {", None, "concentration_factor is the constant 5.0"),
    # ---- npo_reset.h
    ("npo_reset.h", "const double load_demand = npo_pymin(100.0,", None, RATED),
    ("npo_reset.h", "const double power_fraction = load_demand / 100.0;", None, RATED + "; power_fraction only builds the 34-K figure the next guard is about"),
    ("npo_reset.h", "double delta_t = (primary_flow_per_sg > 0)", None, RATED),
    ("npo_reset.h", "const double hot_leg_temp = cold_leg_temp + (34.0 * power_fraction);", None, RATED + ": |33.74 - 34.0x| stays below 10"),
    ("npo_reset.h", "if (fabs(delta_t - realistic_delta_t) > 10.0) {", None, RATED),
    ("npo_reset.h", "if (thermal_power_per_sg > 0) delta_t = realistic_delta_t", None, RATED + ": inside the branch that is never taken"),
    ("npo_reset.h", "inlet_temp = npo_clip(inlet_temp, 293.0, 350.0);", None, RATED),
    ("npo_reset.h", "outlet_temp = npo_clip(outlet_temp, 280.0, 300.0);", None, RATED),
    ("npo_reset.h", "if (inlet_temp <= outlet_temp) inlet_temp = outlet_temp + 5.0;", None, RATED),
    ("npo_reset.h", "if (load_demand >= 100.0) eq->thermal_efficiency = 0.34;", None, RATED),
    ("npo_reset.h", "else if (load_demand >= 75.0) eq->thermal_efficiency", None, RATED),
    ("npo_reset.h", "else if (load_demand >= 50.0) eq->thermal_efficiency", None, RATED),
    ("npo_reset.h", "else eq->thermal_efficiency = 0.20 + 0.08", None, RATED),
    ("npo_reset.h", "int needed = (int)ceil(eq->feedwater_flow / 555.0);", None, RATED + ": ceil(1665 / 555) = ceil(1665 / 555.555) = 3"),
    ("npo_reset.h", "needed = needed < 3 ? 3 : needed; needed = needed > 4 ? 4 : needed;", None, RATED + ": needed = 3"),
    ("npo_reset.h", "eq->pump_speed = npo_pymin(100.0,", None, RATED + ": (1665 / 3) / 555 x 100 = 100.0 exactly, the cap's own value"),
    # ---- npo_condenser.h
    ("npo_condenser.h", "if (pressure_mpa <= 0.001) return 10.0;", None, TSAT35),
    ("npo_condenser.h", "const double A = 8.07131, B = 1730.63, C = 233.426;", None, TSAT35),
    ("npo_condenser.h", "double pressure_bar = npo_clip(pressure_mpa * 10.0, 0.01, 100.0);", None, TSAT35),
    ("npo_condenser.h", "double temp_c = B / (A - log10(pressure_bar)) - C;", None, TSAT35),
    ("npo_condenser.h", "if (pressure_mpa >= 0.005 && pressure_mpa <= 0.01) temp_c = npo_clip(temp_c, 35.0, 45.0);", None, TSAT35 + " (a pressure poked above 0.01 MPa for one step, fixture c14, leaves the band and is clipped to 10.0 by the next line: also independent of these tokens)"),
    ("npo_condenser.h", "return npo_clip(temp_c, 10.0, 374.0);", None, TSAT35),
    ("npo_condenser.h", "if (d1 > 0 && d2 > 0) return (d1 - d2) / log(d1 / d2);", None, "both differences were raised to at least 0.1 two lines above: always true"),
    ("npo_condenser.h", "return (d1 + d2) / 2.0;", None, "dead: the line above it always returns (d1, d2 >= 0.1)"),
    ("npo_condenser.h", "tubes_failed = npo_pymin(tubes_failed, cd->active_tube_count * 0.01);", ["const"], "the failure rate is 1e-6 x (1 + 10 vd) (1 + 5 cd / 0.00159) (1 + aggressiveness) per hour with dt <= 1.7 h: three orders below 1 % of the tubes per step even with the damage terms at the values fixture c8 pokes", ['0.01']),
    ("npo_condenser.h", "cd->fouling_distribution_factor = npo_pymin(1.5,", ["const"], "needs 4 380 h since cleaning; reached by fixture c8 (6 000 h) -- listed only if that fixture is absent", ['1.5']),
    ("npo_condenser.h", "double motive_p = motive_steam_pressure - 0.1;", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "int motive_available = motive_p > 0.9;", None, "a flag the reference computes and never reads (vacuum_system.py:449-452; the ejectors keep their own)"),
    ("npo_condenser.h", "required_capacity = npo_clip(required_capacity, 0.0, (0.0 + 25.0 + 25.0) * 1.2);", ["const"], "the upper bound, 60 kg/s, against a requirement of at most 0.3 kg/s", ['25.0', '1.2']),
    ("npo_condenser.h", "if (cd->lead_ejector < 0) {", None, "lead_ejector is -1 only in the constructed state and 0 / 1 from the first step on"),
    ("npo_condenser.h", "if (!((cd->ej_operating_mask >> cd->lead_ejector) & 1)) cmd[cd->lead_ejector] = 1;", ["branch"], "a start command to an ejector that is already operating sets a bit that is set"),
    ("npo_condenser.h", "if (cd->lag_ejector >= 0) {", None, "lag_ejector is 1 - lead_ejector from the first step on (two ejectors, both available)"),
    ("npo_condenser.h", "else if (cd->condenser_pressure < 0.006 && lag_operating) cmd[cd->lag_ejector] = 0;", None, "condenser_pressure >= 0.007 (steam partial pressure >= 0.005 plus air, or 0.007 - p_air + p_air): the lag ejector, once started, is only ever stopped by the weekly rotation -- in the reference as here; a pressure poked below 0.006 for one step is in fixture c14"),
    ("npo_condenser.h", "int new_lag_index = (lag_index >= 0) ? (lag_index + 1) % 2 : -1;", None, "lag_index is 0 or 1; (x + 1) % 2 and (x - 1) % 2 pick the same other ejector, and the next line repairs a collision with the lead"),
    ("npo_condenser.h", "if (cmd[e] == 1) { if (!(motive_p < 0.8))", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "else if (motive_p < 0.8) capacity = 0.0;", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "double pressure_capacity_factor = pow(motive_p / 1.0, 0.5);", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "double temp_ratio = (motive_steam_temperature + 273.15) / (180.0 + 273.15);", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "double temp_capacity_factor = pow(temp_ratio, 0.25);", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "double suction_pressure_ratio = suction / 0.007;", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "double suction_capacity_factor = 1.0 / (1.0 + 0.5 * (suction_pressure_ratio - 1.0));", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "double available_capacity = (25.0 * pressure_capacity_factor", None, EJECTOR_REQUEST),
    ("npo_condenser.h", "if (n_running > 0) {", None, "the lead ejector is commanded on every step it is found off and the motive steam is always there: n_running >= 1 from the first step on"),
    # ---- npo_sg.h
    ("npo_sg.h", "if (pressure_mpa <= 0.001) return 10.0;", None, "the secondary pressure is clipped to [1, 8] MPa (npo_sg.h:276)"),
    ("npo_sg.h", "return npo_clip(temp_c, 10.0, 374.0);", None, "the fit gives 180 ... 303 C over 1 ... 9 MPa"),
    ("npo_sg.h", "return npo_pymin(total, 0.9);", None, "total = 0.6 (ff^1.5 + 0.3 ff) <= 0.78 for a fouling fraction <= 1"),
    ("npo_sg.h", "double ph_factor = 1.0 + 0.5 * fabs(P->sgchem_ph - 9.2);", None, "sgchem_ph is the 9.2 the reference's TSP model is constructed with and never changes (tsp_fouling_model.py:60): the tokens multiply |0|"),
    ("npo_sg.h", "velocity_factor = npo_clip(velocity_factor, 0.5, 2.0);", ["const"], "the lower bound needs a tube velocity below 0.75 m/s: " + FLOW03 + " (the upper one is reached at 350 % power, fixture c12)", ['0.5']),
    ("npo_sg.h", "double bio_temp_factor = (temperature < 60) ? 1.0", None, "temperature is the secondary side's saturation temperature, >= 180 C at >= 1 MPa"),
    ("npo_sg.h", "var += (levels[i] - mean) * (levels[i] - mean);", ["sign"], "sum (l + m)(l - m) = sum l^2 - 7 m^2 = sum (l - m)^2 when m is the mean: the same number up to rounding, and it only feeds the >= 0.30 comparison"),
    ("npo_sg.h", "double maldistribution = npo_pymin(stdv / (mean + 0.01), 1.0);", None, "maldistribution is only compared with 0.30: the cap at 1.0 cannot matter, and the 0.01 moves the ratio by < 0.1 % of itself unless the mean blockage is below 1 %, where nothing is near the threshold"),
    ("npo_sg.h", "if (g->tsp_fouling_fraction >= 0.85) shutdown = 1;", None, "implied: the pressure-drop criterion two lines below fires from a fouling fraction of 0.553 on (1 / (1 - ff)^2 >= 5)"),
    ("npo_sg.h", "if (g->tsp_ht_degradation >= (1.0 - 0.60)) shutdown = 1;", None, "implied: 0.6 (ff^1.5 + 0.3 ff) >= 0.4 needs ff >= 0.63, where the pressure-drop criterion (ff >= 0.553) has already fired"),
    ("npo_sg.h", "double lithium_factor = npo_pymax(0.5, 1.0 + (lithium - 2.0) * 0.1);", None, "lithium is the literal 2.0 two lines above (the primary chemistry SteamGenerator.update_state passes, steam_generator.py:721-730): the factor is 1.0 whatever multiplies the zero"),
    ("npo_sg.h", "double ph_factor = 1.0 + 0.5 * fabs(ph - 7.2);", None, "ph is the literal 7.2 of the same dict"),
    ("npo_sg.h", "double velocity_factor = npo_clip(pow(flow_velocity / 5.0, -0.6), 0.5, 2.0);", ["const"], "the upper bound needs a tube velocity below 1.57 m/s: " + FLOW03 + " (the lower one is reached at 350 % power, fixture c12)", ['2.0']),
    ("npo_sg.h", "formation_rate = npo_clip(formation_rate, 0.0, 0.1);", None, "0.001 x factors of order one: three orders below the 0.1 bound, and every factor is positive"),
    ("npo_sg.h", "if (fabs(delta_t1 - delta_t2) < 1.0) lmtd = (delta_t1 + delta_t2) / 2.0;", None, COUPLING5 + ", so |dT1 - dT2| >= 5 and the arithmetic mean is never taken (with the sign flipped the condition is |dT1 + dT2| < 1, true only when the saturation temperature lies between the two legs, where the logarithm of the original is already NaN in the reference)"),
    ("npo_sg.h", "double temp_difference = primary_temp_in - primary_temp_out;", None, COUPLING5 + ": with the sum instead of the difference the two low-difference branches are not taken either"),
    ("npo_sg.h", "if (temp_difference < 1.0) heat_transfer = 0.0;", None, COUPLING5),
    ("npo_sg.h", "else if (temp_difference < 5.0) heat_transfer = npo_pymin(heat_transfer, max_heat_from_primary * 0.1);", None, COUPLING5 + " (a difference that rounds to just below 5 is visited by fixture c9's 0.5 % load and agrees)"),
    ("npo_sg.h", "if (primary_flow < 100.0) heat_transfer = 0.0;", None, FLOW03),
    ("npo_sg.h", "double heat_input_factor = (design_heat_input > 0)", None, "a positive parameter"),
    ("npo_sg.h", "double steam_demand_factor = (P->sg_secondary_design_flow > 0)", None, "a positive parameter"),
    ("npo_sg.h", "double steam_supply_factor = (P->sg_secondary_design_flow > 0)", None, "a positive parameter"),
    ("npo_sg.h", "equilibrium_pressure = npo_clip(equilibrium_pressure, 3.0, 8.5);", None, "6.9 x (0.7 + 0.3 x heat input / design) - 0.5 x steam demand / design: 4.1 ... 7.9 MPa for heat inputs 0 ... 1.5 x design and demands 0 ... 1.5 x design; neither bound binds"),
    ("npo_sg.h", "double new_pressure = npo_clip(base_new_pressure + pressure_corrections, 1.0, 8.0);", ["const"], "the lower bound: the pressure relaxes towards >= 4.1 MPa and a step's corrections are clipped to +-0.2", ['1.0']),
    ("npo_sg.h", "if (q_flow_factor > 1.1) quality_degradation += npo_pymin((q_flow_factor - 1.1) * 0.01, 0.03);", None, "never taken: a generator's steam demand is 1500 x min(1, load) / 3 <= 500 kg/s = its secondary_design_flow (npo_secondary.h:59 caps the load fraction at 1.0, enhanced_physics.py:549-584 splits it evenly), so q_flow_factor <= 1.0; the reversed comparison, which would take it always, is killed"),
    ("npo_sg.h", "if (heat_flux_ratio > 1.2) quality_degradation += npo_pymin((heat_flux_ratio - 1.2) * 0.005, 0.02);", ["const"], "the 0.02 cap needs a heat flux of 5.2 x design", ['0.02']),
    ("npo_sg.h", "double target_quality = npo_clip(0.995 - quality_degradation, 0.90, 1.0);", None, "the degradation terms add up to at most 0.02 + 0.03 + 0.02: the target stays in 0.925 ... 0.995"),
    ("npo_sg.h", "if (total_primary_flow > 0) demands[i]", None, FLOW03),
]
