/* wfstep.h — C ABI of libwfstep.so: the MI355X-native batched wind-farm step.
 *
 * Drop-in boundary: these entry points are what a binding for the reference's
 * simulator-interface layer would call in place of the FLORIS calls made by
 *   reference wfcrl/interface.py:444-671  (class FlorisInterface).
 * One handle evaluates, for a batch of B independent farm instances that share one layout, the
 * FLORIS-3.5 Gauss-Curl-Hybrid steady-state solve configured by
 *   reference wfcrl/simulators/floris/inputs/template/case.yaml:14-89
 * and the measurement extraction of interface.py:565-577, 622-648.
 *
 * Plain C types only: no C++ or torch types cross this boundary.  Device pointers are plain
 * `float*` / `double*` in the caller's HIP context (e.g. torch.Tensor.data_ptr()).
 *
 * Units (same as the reference): yaw absolute degrees (FLORIS sign convention); wind speed m/s;
 * wind direction meteorological degrees (270 = wind along +x); power W; loads = (TI [-], std u,
 * std v, std w [m/s]) — i.e. local_load_proxies() WITHOUT the reference's x1e7 / /1e7 round trip
 * (interface.py:575-577, mdp.py:281-283, net identity).
 *
 * Threading: a handle is not re-entrant; distinct handles may be used from distinct threads.
 * All work of a handle is queued on the handle's HIP stream (own stream unless wf_set_stream).
 */
#ifndef WFSTEP_H
#define WFSTEP_H

#ifdef __cplusplus
extern "C" {
#endif

#define WF_ABI_VERSION 7

/* status codes (0 = ok, negative = error; text via wf_last_error) */
#define WF_OK 0
#define WF_E_INVALID -1     /* bad argument / call order */
#define WF_E_UNSUPPORTED -2 /* model option outside what the kernels implement */
#define WF_E_NODEVICE -3    /* no HIP device / device id out of range */
#define WF_E_HIP -4         /* a HIP runtime call failed */
#define WF_E_NOMEM -5

#define WF_MAX_TURBINES 256
#define WF_MAX_TABLE 64
#define WF_MAX_TURBINE_TYPES 4

typedef struct wf_handle wf_handle;

/* Model constants.  Defaults (wf_default_model) are the values of the reference's case.yaml and of
 * FLORIS 3.5's `nrel_5MW` turbine.  The power/thrust table is
 * DATA: pass any FLORIS `power_thrust_table` (wind_speed, thrust=Ct, power=Cp). */
typedef struct wf_model_params {
  /* flow field — case.yaml:30-39 */
  double air_density, ambient_ti, shear, veer; /* veer != 0: FLORIS' rotated Gaussian; served by the VEER instantiations of
                                                   both step kernels (about 3/4 of the veer-free rate) */
  /* turbine — FLORIS turbine_library/nrel_5MW */
  double rotor_diameter, hub_height, tsr, pP, pT, gen_eff, ref_density;
  /* gauss velocity model — case.yaml:76-80 (alpha, beta, ka, kb); gauss deflection model — case.yaml:52-59 (ad, bd,
   * dm and its OWN alpha, beta, ka, kb: the defl_* fields further down; the reference template gives both the same) */
  double alpha, beta, ka, kb, ad, bd, dm;
  /* crespo_hernandez — case.yaml:84-89 */
  double ch_initial, ch_constant, ch_ai, ch_downstream;
  /* GCH internals of FLORIS 3.5 */
  double eps_gain, num_eps, kappa, gch_gain, overlap_thresh, near_wake_c;
  /* gauss deflection model's own wake-expansion set — case.yaml:55-59 */
  double defl_alpha, defl_beta, defl_ka, defl_kb;
  /* power_thrust_table, n_table <= WF_MAX_TABLE, wind speeds strictly ascending */
  int n_table;
  const double* table_ws;
  const double* table_ct;
  const double* table_cp;
  /* switches of FLORIS' sequential solver — case.yaml:46-50 (nonzero = enabled; all enabled in the reference template):
   * secondary steering adds wake_added_yaw to the yaw the deflection model sees; yaw-added recovery adds the mixing term
   * to the source's TI; transverse velocities are the V / W fields (off: they stay zero). */
  int enable_secondary_steering, enable_yaw_added_recovery, enable_transverse_velocities;
} wf_model_params;

int wf_version(void);

/* Fill *p with the reference defaults; table pointers reference static storage inside the library. */
int wf_default_model(wf_model_params* p);

/* Named power / thrust tables shipped as data (pointers into static storage; *n entries each):
 *   "nrel_5MW_floris3" (alias "nrel_5MW", the default of wf_default_model): the turbine the reference's case.yaml:27-28
 *       selects, with the six-decimal Cp column of FLORIS 3.x' turbine_library/nrel_5MW.yaml as recollected — a 5.000 MW
 *       plateau from 11.5 m/s up.  Not reference-held (FLORIS is not vendored): DESIGN.md §2 gives the evidence;
 *   "nrel_5MW_survey_a5": the 8-decimal Cp column of SURVEY.md Appendix A.5 (FLORIS v2's example input: 4.969 MW at
 *       12 m/s, 5.116 MW at 25 m/s), the default of rounds 1-2.  Thrust is the same in both.
 * Unknown name: WF_E_INVALID. */
int wf_turbine_table(const char* name, int* n, const double** ws, const double** ct, const double** cp);

/* Replaces `tools.FlorisInterface(simul_file)` (interface.py:479): create a handle bound to HIP
 * device `device_id`, with the default model.  Fails with WF_E_NODEVICE when no GPU is visible —
 * there is no CPU fallback. */
int wf_create(int device_id, wf_handle** out);
int wf_destroy(wf_handle* h);

/* external != 0: adopt the caller's hipStream_t (e.g. torch's current stream; NULL = the HIP null stream);
 * external == 0: back to the handle's own non-blocking stream.  Drains the previous stream first. */
int wf_set_stream(wf_handle* h, void* hip_stream, int external);
void* wf_get_stream(wf_handle* h);

/* Replaces the model section of case.yaml (simul_utils.py:34-48 writes it, interface.py:479 reads it). */
int wf_set_model(wf_handle* h, const wf_model_params* p);

/* Replaces farm.layout_x / layout_y of case.yaml (simul_utils.py:39-40). n <= WF_MAX_TURBINES. */
int wf_set_layout(wf_handle* h, int n_turbines, const double* x, const double* y);

/* Several turbine definitions per farm — farm.turbine_type of case.yaml is a LIST (reference
 * wfcrl/simulators/floris/inputs/template/case.yaml:27-28; the template writes one entry, FLORIS 3.5 takes one per turbine):
 * `defs[0 .. n_types-1]` (n_types <= WF_MAX_TURBINE_TYPES) and, per turbine of the layout in the caller's order, the index of
 * its definition.  A definition carries what FLORIS reads per turbine type on this path: the power_thrust_table (thrust
 * coefficient of the source's rotor speed, power of the yaw-corrected speed: Turbine.fCt / power_interp), TSR (wake
 * rotation circulation), pP (yaw exponent of the power), generator_efficiency and ref_density_cp_ct.  The definitions
 * share the ROTOR — rotor_diameter and hub_height stay those of the model (wf_set_model): the rotor grid, the vortex
 * geometry and the shear profile are per handle; a caller whose definitions differ there must refuse (the Python host does).
 * While definitions are set every farm is solved by the float64 kernels (wf_resolve_mt.hip) on every step — mode 2 of
 * wf_set_risk_resolve, which then refuses mode 0 (the float32 kernels know one table), keeps mode 1 for when the
 * definitions are cleared, and wf_get_risk_resolve reports 2.  About 1.2e6 farm-steps/s on
 * HornsRev1 instead of 7.6e7: the feature is for the rare mixed farm, not for the benchmark.  With layouts of different
 * turbine counts (wf_set_layouts_counts) the index is per turbine SLOT, shared by all layouts.
 * n_types == 0 clears the definitions (back to the model's single table; the resolve mode set before is in force again).
 * Call after wf_set_layout; a later layout with another turbine count invalidates them (the next step fails until they are
 * set again or cleared). */
typedef struct wf_turbine_def {
  int n_table; /* 2 .. WF_MAX_TABLE - 1, wind speeds strictly ascending */
  const double* table_ws;
  const double* table_ct;
  const double* table_cp;
  double tsr, pP, gen_eff, ref_density;
} wf_turbine_def;
int wf_set_turbine_types(wf_handle* h, int n_types, const wf_turbine_def* defs, const int* type_of);
int wf_get_turbine_types(wf_handle* h, int* n_types);

/* Number of independent farm instances evaluated per step (not in the reference: it holds one). */
int wf_set_batch(wf_handle* h, int env_batch);

/* Several layouts in one batch (not in the reference: one env holds one layout, envs.make builds one per case — here a
 * batch may mix them, e.g. to train one policy across layouts).  n_layouts layouts of the n_turbines of wf_set_layout,
 * x / y: [n_layouts][n_turbines]; layout_of: [env_batch] layout of each farm, or NULL with n_layouts == env_batch (farm
 * b has layout b).  Call after wf_set_batch (which returns the handle to the first layout for every farm); the wind has
 * to be set again.  Every farm is rotated and sorted on its own; under ONE wind direction the farms of a layout share
 * geometry and pair table (a grouped launch, as for direction groups), with a direction per farm the on-the-fly kernels
 * serve the batch; series playback and binned sampling fall back to a geometry per farm.  The layouts may lie anywhere
 * (every lateral offset is taken from the float64 coordinates).  n_layouts == 1 equals wf_set_layout. */
int wf_set_layouts(wf_handle* h, int n_layouts, const double* x, const double* y, const int* layout_of);
/* ... of DIFFERENT turbine counts (the reference's registry builds `Turb<N>_Row1` for any N, registration.py:43-68; a
 * learner trained across it needs several N in one batch): counts[l] in 1..n_turbines is the number of turbines layout l
 * really has — its row of x / y holds them first, the rest is ignored.  The missing turbines become placeholders that
 * the geometry kernel puts 10 000 km and more DOWNSTREAM of the farm for whatever direction it is rotated to: last in
 * the sorted order, outside every reach and gate — they receive wakes and give none, so the real turbines' results are
 * those of the unpadded farm.  Their outputs (columns counts[l] .. n_turbines-1 of a farm's rows, caller's order) are
 * written as 0, and the fused reward averages over the real turbines.  counts == NULL: wf_set_layouts. */
int wf_set_layouts_counts(wf_handle* h, int n_layouts, const double* x, const double* y, const int* counts, const int* layout_of);

/* Replaces FlorisInterface.update_wind -> fi.reinitialize (interface.py:663-671).
 * count == 1: one (ws, wd) shared by the whole batch; count == env_batch: one per instance.
 * Host arrays are validated (ws > 0, wd finite); device arrays are NOT (no readback on the asynchronous path): a
 * non-positive or non-finite wind there yields NaN outputs for that farm.
 * Performs wd % 360, the layout rotation and the upstream->downstream sort on the device (float64).
 * Host pointers unless on_device != 0.  Host arrays whose directions are all equal (a speed per instance under one
 * direction) share the rotation, the sort and the pair-coefficient table like count == 1. */
int wf_set_wind(wf_handle* h, const double* ws, const double* wd, int count, int on_device);

/* The same with separate counts: n_ws speeds and n_wd directions, each 1 or env_batch (n_ws == 1 requires n_wd == 1).
 * n_wd == 1 with n_ws == env_batch states "one direction, a speed per farm" explicitly — the only way to say so for
 * device arrays, which are never read back — and keeps the shared geometry and the pair-coefficient table path (e.g.
 * BASELINE configs[4]'s direction sweep with a speed per farm). */
int wf_set_wind_counts(wf_handle* h, const double* ws, int n_ws, const double* wd, int n_wd, int on_device);

/* Replaces fi.calculate_wake(yaw_angles) + fi.get_turbine_powers() + local_wind_measurements() +
 * local_load_proxies() (interface.py:564, 623, 629-648).
 *   yaw        [B*N]    absolute yaw, degrees, caller's turbine order
 *   power      [B*N]    W                         (may be NULL to skip the copy-out)
 *   wind_speed [B*N]    m/s   cbrt(mean u^3)      (may be NULL)
 *   wind_dir   [B*N]    deg   mean(wd - atan2(v,u)) (may be NULL)
 *   load       [B*N*4]  (TI, std u, std v, std w) (may be NULL)
 * on_device == 0: host pointers; inputs/outputs are staged through pinned buffers and the call
 *                 returns after the results are in the caller's arrays.
 * on_device != 0: device pointers; the call only enqueues work on the handle's stream. */
int wf_step(wf_handle* h, const float* yaw, float* power, float* wind_speed, float* wind_dir, float* load,
            int on_device);

int wf_sync(wf_handle* h);

/* ---- Risk flags: where float32 cannot reproduce a float64 decision of the reference -------------------------
 * The model has one state-dependent discontinuity: the overlap count of the wake-added turbulence, "deficit * Uinit >
 * 0.05" per rotor-grid point (FLORIS 3.5 solver: `area_overlap = sum(velocity_deficit * u_initial > 0.05) / n`,
 * SURVEY.md Appendix A.3-8).  The kernels evaluate the deficit in float32 (relative error ~1e-6 after the recurrence);
 * when a deficit lies within a relative band `rel_band` of the threshold, a float64 evaluation may count the point
 * the other way and move the turbine's TI by 1/9 of the added term.  Every step therefore records, per farm instance,
 *   WF_RISK_OVERLAP     a (source, target, grid point) deficit within the guard band of the threshold, at a pair where
 *                       the count matters (inside the 15 D reach and the 2 D lateral gate);
 *   WF_RISK_POWER_KNEE  a turbine at a point of the power curve whose relative condition number v |P'| / max(P, 1 kW)
 *                       exceeds 30, where a float32-sized wind-speed error (~3e-6) is amplified past 1e-4 of
 *                       max(P, 1 kW): just above cut-in (P < ~7 kW) and on the cut-out drop of nrel_5MW;
 *   WF_RISK_THRUST_RAMP a turbine on a segment of the thrust table with v |dCt/dv| > 5: the cut-in ramp of nrel_5MW
 *                       (Ct 0 -> 0.99 between 2.5 and 3 m/s, 20x steeper than anywhere in the operating range) and its
 *                       cut-out drop.  With a row of turbines on the ramp the float64 result itself moves by 7e-5 in
 *                       power for 1e-5 deg of wind direction (tests/golden/README: case bad_512_56); float32 wind
 *                       speeds (3e-7 relative each) are amplified the same way;
 *   WF_RISK_THRUST_UNITY a turbine whose thrust coefficient exceeds 0.995 (user tables only: nrel_5MW peaks at 0.99): 1 - Ct
 *                       cancels in float32 and the velocity behind such a turbine is a small difference of O(1) numbers —
 *                       there is no float32 bound (TI off by 0.7 seen on a dense farm behind a table clipped at 0.9999).
 *                       Such farms are ALWAYS solved again in float64 and the flag cleared, in every wf_set_risk_resolve
 *                       mode (round 5; round 4 exempted them from every bound on the float32-only path): a caller never
 *                       sees this flag after a step, only in wf_get_resolve_stats' raw flags.
 *   WF_RISK_NEGATIVE_SPEED a rotor-grid speed that is not positive: summed deficits beyond 1 in an unphysically tight farm
 *                       (the reference keeps computing there, and so does this path).  The cube mean of mixed-sign speeds
 *                       cancels: float32 keeps about 1e-5 of the rotor wind speed of such a turbine (round-5 fuzz).
 *                       What the FLOAT64 kernels guarantee there (modes 1 / 2): the combined deficit of every rotor-grid point
 *                       to 6e-7 relative (profiles/r05_f64_deviation_probe.txt: with Ct at 0.9999 the deficit amplitude sits at
 *                       the clip of its square root, where two float64 orders of summation differ that much) — i.e. a wind
 *                       speed error of 6e-7 of the FREE STREAM on such a turbine (5e-5 of |u| where u is the small difference
 *                       of two free-stream-sized numbers), and the contract's tolerances everywhere else.
 * Farms with flag 0 match the float64 path within the parity tolerances (power 1e-4 of max(P, 1 kW), wind speed 5e-5,
 * direction 3e-4 deg, TI 5e-6).  By default (wf_set_risk_resolve mode 1) the flagged farms are solved again in float64 behind
 * every step and no flag is left.  A FLAGGED farm left in float32 (mode 0, the opt-out) may differ by the bounded signature of
 * its event, per flag combination
 * (tests/parity.py, measured maxima in brackets):
 *   POWER_KNEE alone        power 5e-2 of max(P, 1 kW), or — a turbine ON the cut-out drop, where the power is next to
 *                           nothing on one side — 2e-2 of the rated power [0.4e-2]; wind field as an unflagged farm's
 *   THRUST_RAMP, no OVERLAP power 1e-2, wind speed 1e-3, direction 1e-2 deg, TI 2e-4 — on the cut-in ramp / cut-out drop
 *   OVERLAP                 power 1e-1 [5.7e-2], wind speed 2e-2, direction 0.1 deg, TI 2e-2   (one overlap count flipped)
 *   OVERLAP | THRUST_RAMP   power 4e-1 [2.7e-1], wind speed 4e-2, direction 0.2 deg [0.13]     (a flip below ~4 m/s, where
 *                           the thrust ramp and the power curve both amplify it: 1.5 x the one measured case)
 * With wf_set_risk_resolve on, exactly those farms are solved again in float64 and their flags cleared (below): no
 * exemption is left.  All geometric discontinuities (upstream/downstream order, dx > 0.1, 15 D reach, 2 D
 * lateral gate) are decided in float64 on the device and need no flag.
 * wf_get_risk_flags copies the flags of the last wf_step / wf_env_step (env_batch ints). */
#define WF_RISK_OVERLAP 1
#define WF_RISK_POWER_KNEE 2
#define WF_RISK_THRUST_RAMP 4
#define WF_RISK_THRUST_UNITY 8
#define WF_RISK_NEGATIVE_SPEED 16
int wf_set_risk_guard(wf_handle* h, double rel_band); /* default (chosen by wf_set_layout until this is called): 1e-5 for farms of up
                                                         to 128 turbines — 5 x the narrowest band at which no unflagged farm of 4.3 M
                                                         left the tolerances, profiles/r06_band_study.txt — and 2e-5 beyond (rounds
                                                         4-5: 2e-5 throughout); 0 disables WF_RISK_OVERLAP */
int wf_get_risk_flags(wf_handle* h, int* flags, int on_device);

/* ---- Float64 re-solve: the 1e-4 contract without exemptions ------------------------------------------------
 * The reference evaluates this path in float64 (interface.py:564 -> FLORIS `calculate_wake`).  With mode 1, every
 * wf_step / wf_env_step is followed, on the same stream and without a host round trip, by a compaction of the farms
 * whose risk flags are nonzero and a float64 solve of exactly those farms (csrc/wf_resolve.hip: the same recurrence with
 * double arithmetic and FLORIS' own comparisons), which overwrites their outputs (and reward) and clears their flags:
 * afterwards EVERY farm of the batch matches the float64 path within the parity tolerances.  Cost: nothing measurable
 * when no farm is flagged (three tiny launches); otherwise the latency of one farm's float64 chain — 0.7 to 1.4 ms for up
 * to ~1400 flagged 80-turbine farms.
 * mode 1 is the DEFAULT of a new handle (ABI 6; ABI <= 5 started in mode 0): a binding that calls nothing but wf_create ...
 * wf_step gets what the reference computes — every farm within the tolerances of the float64 path, every flag 0.
 * mode 2 solves every farm in float64 (validation: 1.2e6 farm-steps/s on HornsRev1).
 * mode 0 is the opt-out: float32 results with flags, flagged farms within the per-flag bounds listed above (only farms with
 * WF_RISK_THRUST_UNITY are still re-solved — a thrust table that reaches 0.995; costs nothing with nrel_5MW).
 * wf_get_risk_resolve: the handle's current mode.
 * wf_get_resolve_stats: number of farms the last step solved in float64, and (raw_flags != NULL, env_batch ints) the
 * flags as the float32 kernels raised them before they were cleared. */
int wf_set_risk_resolve(wf_handle* h, int mode);
int wf_get_risk_resolve(wf_handle* h, int* mode);
int wf_get_resolve_stats(wf_handle* h, int* n_resolved, int* raw_flags, int on_device);

/* ---- Fused env step (not in the reference, which does this in Python per farm) ----
 * Device-resident env state per farm instance: absolute yaw [B*N], actuation accumulator [B*N], move
 * counter [B].  One wf_env_step launch performs, per farm,
 *   the actuation-budget gate         (reference wfcrl/simple_env.py:64-72),
 *   the clipped yaw transition        (reference wfcrl/mdp.py:291-319),
 *   the farm solve + measurements     (wf_step),
 *   the reward  mean_j(P_j[MW]*1e3/ws^3) - load_coef*mean|loads|   (reference wfcrl/simple_env.py:78-84)
 * with the reference's float32 arithmetic for the MDP part. */
typedef struct wf_env_params {
  float yaw_lo, yaw_hi, yaw_step; /* controls["yaw"] = (lo, hi, step): data_cases.py:19-23 default (-40, 40, 5) */
  float actuator_rate;            /* WindFarmMDP.ACTUATORS_RATE["yaw"] = 0.3 deg/s (mdp.py:52) */
  float dt;                       /* FarmCase.dt, 60 s for FLORIS cases */
  float budget;                   /* 0.1: an actuator may move at most 10 % of the time */
  float load_coef;                /* env kwarg load_coef, default 0.1 */
  int discrete;                   /* 0: action = dyaw clipped to +-step; 1: action in {0,1,2} = down/hold/up */
} wf_env_params;

int wf_env_config(wf_handle* h, const wf_env_params* p);

/* Unit of the `power` output of wf_env_step (ABI 7): 0 watts (default, as wf_step), 1 megawatts — what the reference's env hands
 * out (info["power"] = powers / 1e6, wfcrl/mdp.py:284; wfcrl/simple_env.py:91): the float32 watts times 1e-6f, written by the
 * step kernel itself instead of by a scaling pass over [B*N] behind every step.  wf_step and the reward are not affected. */
int wf_env_set_power_unit(wf_handle* h, int megawatts);

/* Zero yaw, accumulators and move counters (WindFarmMDP.reset, mdp.py:267-270). Wind: wf_set_wind. */
int wf_env_reset(wf_handle* h);

/* action [B*N] (NULL: no transition, solve at the current yaw — the warm-up solve of reset, mdp.py:261-262).
 * Outputs (each may be NULL = not written): reward [B]; yaw [B*N] new absolute yaw; then as wf_step. */
int wf_env_step(wf_handle* h, const float* action, float* reward, float* yaw, float* power, float* wind_speed,
                float* wind_dir, float* load, int on_device);

/* The reward of wf_env_step is normalised by the free-stream speed of the state BEFORE the step (reference
 * wfcrl/simple_env.py:78-80: `self.mdp.state["freewind_measurements"][0]` is read before `step_interface`).  It equals
 * the current wind except right after a series tick (handled inside wf_wind_series_step) and when the caller's start
 * state holds a clipped wind (reference mdp.py:263-266 clips the whole start state to the observation bounds): pass
 * that speed per farm here.  Used by the next wf_env_step that computes a reward, once. */
int wf_env_set_prev_wind(wf_handle* h, const double* wind_speed /* env_batch */, int on_device);

/* Checkpoint / resume of the device-resident env state: yaw [B*N], acc [B*N] (accumulated |dyaw|),
 * moves [B].  set == 0 copies the state out, set != 0 overwrites it.  NULL pointers are skipped. */
int wf_env_state(wf_handle* h, float* yaw, float* acc, int* moves, int set, int on_device);

/* ---- On-device wind process ------------------------------------------------------
 * wf_wind_sample: per-farm reset sampling with the reference's distributions (wfcrl/mdp.py:237-258):
 *   ws = clip(ws_scale * Weibull(ws_shape), ws_lo, ws_hi);  wd = clip(Normal(wd_mean, wd_std) mod 360, wd_lo, wd_hi)
 * from a counter-based generator keyed by (seed, farm index); dist == NULL uses 8, 8, 3, 28, 270, 20, 0, 360.
 * Then the geometry (rotation + sort) of every farm is rebuilt, as wf_set_wind does. */
typedef struct wf_wind_dist {
  double ws_scale, ws_shape, ws_lo, ws_hi, wd_mean, wd_std, wd_lo, wd_hi;
} wf_wind_dist;
int wf_wind_sample(wf_handle* h, unsigned long long seed, const wf_wind_dist* dist);

/* Build-defined variant (not in the reference): the sampled direction of every farm is rounded to the nearest multiple
 * of `step_deg` (which must divide 360), so the batch holds at most K = 360 / step_deg distinct directions.  Farms are
 * grouped by direction; the sorted geometry and the pair-coefficient table of each grid direction are built once per
 * layout / model and kept across resets, and every step takes the table path (about twice the throughput of a
 * continuous direction per farm).  It is a DIFFERENT wind process, not an approximation of the continuous one within the
 * parity tolerances: a 2-degree grid moves HornsRev1's farm power by 0.8 % in the median and 7.7 % at the 99th percentile
 * (profiles/archive/r03_binning_error.txt).  Falls back to wf_wind_sample when K is too large for the batch (padding each group
 * to whole blocks would cost more than it saves). */
int wf_wind_sample_binned(wf_handle* h, unsigned long long seed, const wf_wind_dist* dist, double step_deg);

/* Time-series mode (wfcrl/interface.py:512-524): a shared series of T (ws, wd) rows; farm b plays it from
 * start[b] (host array of env_batch ints, or NULL: drawn uniformly in [0, T) from `seed`).  The call positions
 * every farm on its first row; each wf_wind_series_step advances all farms by one row.  Like the reference's
 * finite generator, stepping past T rows fails (WF_E_INVALID, "wind series exhausted").
 * A shared series holds only T distinct winds: when T is small against env_batch the farms are grouped by start row,
 * one sorted geometry + pair-coefficient table is built per ROW at this call, and the whole playback runs on the table
 * path (wf_kernel_info.direction_groups = T); otherwise each farm's geometry is rebuilt every tick (on-the-fly path). */
int wf_wind_series(wf_handle* h, int T, const double* ws, const double* wd, const int* start, unsigned long long seed);
int wf_wind_series_step(wf_handle* h);

/* Current free-stream wind of every farm (env_batch entries each; shared wind is broadcast). */
int wf_get_wind(wf_handle* h, double* ws, double* wd, int on_device);

/* HIP-event timing of the step kernel on the handle's stream (used by bench.py for the roofline
 * object): wf_timing_begin records an event, wf_timing_end records another, synchronises, and
 * returns the elapsed milliseconds between them. */
int wf_timing_begin(wf_handle* h);
int wf_timing_end(wf_handle* h, float* elapsed_ms);

/* Introspection: which kernel variant serves the current layout (lanes per farm, target slots per
 * lane), its register/LDS footprint and the launch geometry. */
typedef struct wf_kernel_info {
  int lanes_per_env, slots_per_lane, envs_per_block, threads_per_block, grid_blocks;
  int vgprs, lds_bytes, scratch_bytes; /* from hipFuncGetAttributes */
  int pair_table; /* 1: shared-wind pair-coefficient table path, 0: per-farm on-the-fly path */
  int direction_groups; /* > 0: farms grouped by that many distinct wind directions, one pair table each */
  int mixed_main_farms; /* > 0: mixed launch (wf_kernel_choice::mixed) — the figures describe the kernel of the farms [0,
                           mixed_main_farms); wf_step_kernel serves the rest */
  int one_block_kernel; /* 1: the figures describe wf_step_ll_kernel (one target block of lanes_per_env x slots_per_lane turbines in
                           registers at a time, csrc/wf_kernels_ll.hip), which serves every wind direction without an
                           x' tie across a block boundary; the register-slot kernel is enqueued behind it for the rest */
} wf_kernel_info;
int wf_get_kernel_info(wf_handle* h, wf_kernel_info* info);

/* Which kernels may serve a handle.  Per handle (two handles in two threads may be steered differently); the
 * environment variables of earlier builds (WF_KERNEL_GS, WF_LL, WF_LL_G, WF_NO_PAIR_TABLE, WF_LL_FLY) only seed this
 * structure once, at wf_create.  Setting a choice drops the current wind (geometry, groups and tables were laid out
 * for the previous kernels): call wf_set_wind / wf_wind_* again before the next step.
 * wf_get_kernel_choice returns the request as set; what actually runs is reported by wf_get_kernel_info. */
typedef struct wf_kernel_choice {
  int slot_G, slot_S;   /* wf_step_kernel<G,S> (register-slot kernel): lanes per farm x target slots per lane; 0, 0 = by N
                           and batch.  Ignored when G x S cannot hold the layout's turbines. */
  int one_block;        /* wf_step_ll_kernel: -1 by the rounds model, 0 never, 1 always with (ll_G, ll_S) */
  int ll_G, ll_S;       /* one of 4x1, 8x1, 16x1, 4x2, 2x2 (used when one_block == 1) */
  int pair_table;       /* -1 whenever the batch shares a wind direction, 0 never (everything on the fly) */
  int fly_one_block;    /* -1: a wind per farm runs wf_step_ll_kernel on the fly where it pays; 0: stays on wf_step_kernel */
  int far_skip;         /* -1 / 1: wf_step_ll_kernel leaves out the deficit / turbulence work of (source, target block) pairs
                           more than 6.12 sigma_y + D/4 off the wake's centre line (the nearest rotor-grid column would get
                           exp2(-27) of the amplitude: no effect on any float32 result); 0: every pair is evaluated (A/B
                           and the bit-identity test, tests/test_hip_parity.py) */
  int calibrate;        /* -1 / 1: with one_block == -1, the FIRST step after a (re)configuration (a fused env step is probed
                           without its action: a solve at the current yaw state, no transition) first times three launches of
                           every kernel family the rounds model prices within 60 % of its best guess, on the handle's own
                           batch / layout / wind (a few ms, once; that one call synchronises — or call wf_calibrate at a point
                           of your choosing), and the fastest serves the handle from its first real launch on — unless the
                           guess is within 4 % of it: near-ties are not left to noise.  Every step of a handle therefore comes
                           from ONE family.  The result is cached per process under (device, turbines, batch, layout, model):
                           a re-created handle neither times again nor changes family; wf_get_calibration +
                           wf_set_calibration carry it across processes (kernel families agree within the parity tolerances,
                           not bit for bit: another summation order).  0: the rounds model's guess stands (measured on one
                           MI355X: wf_dispatch.hip) */
  int mixed;            /* -1 / 1: a batch a little beyond a whole number of ROUNDS of its kernel family (a launch costs whole
                           rounds: 69 632 HornsRev1 farms are one round of the 2x2 kernel plus 4 096 farms that would take a
                           second one) is served by two launches on disjoint farm ranges — the whole rounds on the family, the
                           remainder on wf_step_kernel behind it, same stream — where that is faster (rounds model; measured
                           by the calibration); 0: always one launch.  Results of a farm do not depend on which range it is in
                           beyond the family difference stated under `calibrate`. */
} wf_kernel_choice;
int wf_set_kernel_choice(wf_handle* h, const wf_kernel_choice* c);
/* What the calibration (wf_kernel_choice::calibrate) found: *code = (G << 4) | S of the wf_step_ll_kernel shape it chose, 0
 * = wf_step_kernel, -1 = it has not run for the current configuration; family_ms[6] = ms per launch of the families it
 * timed, in the order {wf_step_kernel, 8x1, 4x2, 4x1, 2x2, 16x1}, 0 = not timed.  Either pointer may be NULL. */
int wf_get_calibration(wf_handle* h, int* code, float* family_ms);
/* The same for the on-the-fly path (a wind per farm: the first step there times wf_step_ll_kernel of the table
 * path's family against wf_step_kernel, which has to win by 4 %): *choice = 0 not timed yet, 1 wf_step_ll_kernel, 2
 * wf_step_kernel; ms[2] = ms per launch of the two.  Either pointer may be NULL. */
int wf_get_fly_calibration(wf_handle* h, int* choice, float* ms);
/* The mixed launch of the current configuration: *main_farms = farms the one-block family keeps (0: one launch), *mixed_ms = ms
 * per step the calibration measured for it (0: not timed — the rounds model decided).  Either pointer may be NULL. */
int wf_get_mixed_launch(wf_handle* h, int* main_farms, float* mixed_ms);
int wf_get_kernel_choice(wf_handle* h, wf_kernel_choice* c);
/* Time the kernel families NOW for the handle's current layout / batch / wind (after wf_set_wind*), on scratch buffers with
 * zero yaw, ignoring and then refreshing the process cache; synchronises.  Lets a caller keep the timing out of its first
 * step (stream capture, asynchronous pipelines).  A configuration that has nothing to calibrate (forced kernel choice,
 * N <= 16, the latency regime, grouped launches) returns WF_OK and does nothing. */
int wf_calibrate(wf_handle* h);
/* Take a saved calibration as is — nothing is timed for this configuration afterwards: code as wf_get_calibration returned
 * it (-1: leave the table path to the timing), fly_choice as wf_get_fly_calibration returned it (0: leave it to the timing).
 * After wf_set_batch; a later wf_set_batch / wf_set_layout / wf_set_kernel_choice starts over.  With it a run can be
 * replayed on exactly the kernels of the run that saved it. */
int wf_set_calibration(wf_handle* h, int code, int fly_choice);

const char* wf_last_error(wf_handle* h); /* h may be NULL: last error of a failed wf_create */

#ifdef __cplusplus
}
#endif
#endif /* WFSTEP_H */
