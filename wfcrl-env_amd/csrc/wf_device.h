// wf_device.h — per-farm constants handed to the HIP kernels by value (kernarg segment -> SGPRs).
//
// Everything here is derived on the host in float64 from wf_model_params (include/wfstep.h) and
// rounded once to float32.  Nothing depends on the wind speed: every quantity that FLORIS 3.5
// scales by the inflow (Uinit, Uinf, dU/dz) is stored per unit wind speed.
#pragma once

#define WF_TABLE_PAD 64
#define WF_BUCKETS 2048

struct WfConsts {
  int N;             // turbines
  float D, invD;     // rotor diameter
  float off[3];      // rotor-grid offsets  -D/4, 0, +D/4            [A.1-3]
  float yoff[3];     // off[j] + NUM_EPS                              [A.3-4]
  float shearf[3];   // (z_k/HH)^shear : Uinit_k = ws*shearf[k]       [A.2]
  float uinf_f;      // mean shearf    : Uinf    = ws*uinf_f
  float decay_a[3];  // 4 nu_k / (Uinf eps^2) : decay_k = 1/(decay_a[k]*dx + 1)   [A.3-4]
  float exp_c;       // log2(e)/eps^2  : exp(-r/eps^2) = exp2(-r*exp_c)
  // the six vortices of the transverse-velocity model [A.3-4]: v = top (h = HH+R), bottom (HH-R), wake
  // rotation (HH), each with a ground mirror.  On the 3x3 grid (rows z_k = HH + k'q, q = D/4, k' = -1,0,1)
  // the 9 real offsets z_k - h + eps take only 7 distinct values m*q + eps, m = -3..3, and the 9 mirror
  // offsets z_k + h + eps the 7 values 2HH + m*q + eps: class index = m + 3.
  //   real:   top m = k'-2, bottom m = k'+2, rotation m = k'      mirror: top k'+2, bottom k'-2, rotation k'
  float zc[7], zc2[7], ez[7];    // zc = m q + eps       ; ez  = exp(-zc^2/eps^2)
  float inv_eps2, m_half_inv_eps4, yl2_small;  // 1 / eps^2, -1 / (2 eps^4); yL^2 below which class 0 takes the series of its core factor (wf_model.hip)
  float zm[7], zm2[7], ezm[7];   // zm = 2HH + m q + eps ; ezm = exp(-zm^2/eps^2)
  int mirror_core_n;             // mirror classes [0, n) need the core factor; for the others 1 - Ey*ezm == 1.0f exactly
  float gam_top, gam_bot;  // (1/2pi)(pi/8) D vel_{top,bot} uinf_f : Gamma/(2pi) = gam*ws*ct
  float gam_wr;            // (1/2pi) 0.25*2pi*D/TSR               : Gamma_wr/(2pi) = gam_wr*(a-a^2)*ubar
  // secondary steering: mean_9( z/(r) * core ) on the source's own grid (dx = 0, dy = 0)   [A.3-2]
  float ks_top, ks_bot, ks_core;
  // gauss deflection / deficit                                                      [A.3-3, A.3-6]
  float alpha4, beta2, ka, kb, ad, bd, dm03;   // alpha4 .. kb: the velocity model's set
  float alpha4_d, beta2_d, ka_d, kb_d;          // the deflection model's own set (case.yaml:55-59)
  // solver switches of case.yaml:46-50 as multipliers (no branches in the kernel): secondary steering scales the added
  // yaw's argument (2, or 0 when off), transverse velocities scale the applied circulations (1 / 0); yaw-added
  // recovery off is gch_gain = 0
  float sw_steer, sw_tv;
  float e0c1, e0c2;        // 3 e^(1/12), 3 e^(1/3)
  float sz0v;              // D/2 * sqrt(uR/(Uinf+u0)) of the deficit model == D/(2 sqrt 2)
  float near_c;            // 0.501 * D
  float kdef;              // D^2/8
  float kdef_sy0v;         // kdef / (sz0v [cos veer]): ct cos(yaw) kdef = sM^2 sy0v kdef_sy0v (wf_step_ll_kernel's replay)
  // far-source / far-pair skip of wf_step_ll_kernel (wf_kernels_ll.hip): far_k = 6.12 sigma_y widths (1e30 and
  // far_on = 0 with the skip disabled — wf_kernel_choice::far_skip = 0, or a model with a negative wake growth rate)
  float far_k;
  int far_on;
  // crespo-hernandez + overlap gating                                               [A.3-8]
  float ch_c, ch_ai, ch_down;  // ch_c = constant * ambient^initial
  float amb, amb2, gch_gain, overlap_thr, twoD;
  // outputs                                                                         [A.4]
  float rho, pw, dens_f;   // ref density, pP/3, (air_density/ref_density)^(1/3)
  // power / thrust table
  int n_table;
  float bucket_h_inv, bucket_x0;
  int max_probe;
  // 15 D in float64: the reach of the wake-added TI is tested as FLORIS does, x_t <= x_i + 15 D on the float64
  // coordinates (on regular grids whole multiples of D sit on this threshold and the rounding of the rotation decides)
  double fifteenD_d;
  // lateral gate of the wake-added TI, |y_i - (y_t + off_j)| < 2 D, evaluated in float64 as FLORIS does [A.3-8]
  double q_d;  // D/4 (grid offsets -q, 0, +q; 2 D = 8 q, all exact)
  // risk flags (include/wfstep.h WF_RISK_*): relative half-width of the guard band around the overlap threshold
  // "deficit * Uinit > overlap_thr", and the relative condition number of the power curve v |P'| / max(P, 1 kW) above
  // which a turbine counts as sitting on a knee of the curve
  float ct_kappa;  // v |dCt/dv| above which a turbine counts as sitting on a ramp of the thrust table
  float guard_inv, inv_overlap_thr, knee_kappa;  // 1 / guard band (2^50 when the band is 0), 1 / overlap_thr
  // wind_veer (case.yaml:36) [FLORIS gauss.py rCalt]: the Gaussian of the deficit rotated by the veer angle phi.  Only the
  // VEER instantiation of wf_step_kernel reads these; veer_on selects it at launch.
  int veer_on;
  float cos_veer;             // sigma_y0 = sigma_z0 cos(yaw) cos(phi)
  float veer_c2, veer_s2;     // cos^2 phi, sin^2 phi
  float veer_bq;              // sin(2 phi) * D/4 : the cross term's factor at the outer grid rows (z - HH = -+ D/4)
};

// Power/thrust table in global memory, staged to LDS by each block.
struct WfTables {
  float knot[WF_TABLE_PAD];      // wind speeds, padded with +huge
  float ct[WF_TABLE_PAD];        // Ct at knot
  float ct_slope[WF_TABLE_PAD];  // (ct[j+1]-ct[j])/(knot[j+1]-knot[j])
  float pw[WF_TABLE_PAD];        // 1/2 A Cp eta ws^3 at knot
  float pw_slope[WF_TABLE_PAD];
  unsigned char bucket[WF_BUCKETS];  // index of the last knot <= start of the previous bucket
};

// Device-resident env state + fused MDP transition / reward.  All pointers may be null:
// then the kernel is the plain farm step on `yaw_in`.
//   reference semantics: simple_env.py:64-72 (actuation budget), mdp.py:291-319 (transition),
//   simple_env.py:78-85 (reward); float32 arithmetic exactly as NumPy performs it there.
struct WfEnvArgs {
  float* yaw_state;     // [B][N] absolute yaw, in/out (caller's turbine order)
  float* acc;           // [B][N] accumulated |dyaw|, in/out
  int* moves;           // [B] number of env steps taken, in/out
  const float* action;  // [B][N] dyaw (continuous) or {0,1,2} (discrete); null = no transition
  float* reward;        // [B] out (null = skip)
  const double* ws_prev; // [B] free-stream speed of the state BEFORE the step (null: the current one)
  float yaw_step, yaw_lo, yaw_hi;  // controls["yaw"] = (lo, hi, step)
  float rate, dt, budget;          // ACTUATORS_RATE["yaw"] = 0.3 deg/s, case.dt, 0.1
  float load_coef;
  int discrete;
  // round 6 (the two device-side copies a learner's step paid for): the new absolute yaw also goes to the caller's array as it is
  // written to the state (null: not wanted / no transition — wf_env_step copies the state instead), and the power output may
  // leave the kernel in MW, the unit the reference's env hands out (info["power"] = powers / 1e6, mdp.py:284)
  float* yaw_out;
  int power_mw;
};

// Grouped launch (one pair table + sorted geometry per distinct wind direction) and the per-farm risk
// flags.  All pointers may be null.
struct WfGroupArgs {
  const int* perm;       // [n_blocks * farms per block] farm index per launch slot, -1 = padding; null = identity
  const int* blk_group;  // [n_blocks] direction group of the block's farms, -1 = unused block; null = ungrouped
  int shift, mod;        // table / geometry index of group g = (g + shift) % mod  (series playback: shift = tick)
  int n_blocks;          // blocks of the grouped launch at blk_unit farms per block
  int blk_unit;          // farms per entry of blk_group (a kernel with F farms per block reads entry block * F / blk_unit)
  int n_slots;           // launch slots of the grouped launch (= entries of perm)
  const int* pred;       // [groups] when non-null, wf_step_kernel serves only the groups with pred[g] != 0 (the others
                         // are served by wf_step_ll_kernel, launched with the opposite predicate)
  const int* farm_pred;  // [B] the same per farm (a wind per farm): wf_step_kernel serves the farms with farm_pred[b] != 0
  int* risk_flags;       // [B] out: WF_RISK_* bits of each farm; null = not written
  // Compaction for the float64 re-solve (wf_resolve.hip), folded into the step kernel's epilogue (round 5: one memset and one
  // launch fewer behind every step): a farm whose flags meet res_mask appends itself to res_list; the raw flags are kept in
  // flags_raw; block 0 zeroes the counter the NEXT step will use (two counters used alternately).  All null: no re-solve.
  int* res_list;         // [B]
  int* res_count;        // this step's counter
  int* res_zero;         // the other counter
  int* flags_raw;        // [B]
  int res_mask;          // WF_RISK_* bits that select a farm
  // Mixed launch (round 5, wf_dispatch.hip): the launch serves the farms [env_base, env_end) of the batch (launch slot s holds
  // farm env_base + s); env_end == 0: all B farms from 0.  Lets one step be served by two kernels on disjoint farm ranges —
  // whole rounds of the throughput family, the remainder on the register-slot kernel — instead of a nearly empty last round.
  int env_base, env_end;
  const int* n_real;     // [B] turbines the farm really has (wf_set_layouts_counts: layouts of fewer than N turbines are
                         // padded with placeholders far downstream, which nothing real can see); null = N.  Outputs of
                         // the placeholders are written as 0 and stay out of the reward.
};
#define WF_RISK_OVERLAP 1
#define WF_RISK_POWER_KNEE 2
#define WF_RISK_THRUST_RAMP 4
#define WF_RISK_THRUST_UNITY 8
#define WF_RISK_NEGATIVE_SPEED 16

// Pair-coefficient table (shared wind only): for source i and target t (sorted indices) the
// transverse-velocity contribution is linear in the source's circulations.  The tip vortices' circulations share
// the farm-dependent factor Gy = sin(yaw) cos(yaw) Ct ws  (Gt = gam_top*Gy, Gb = -gam_bot*Gy), so they fold into ONE
// coefficient; the wake-rotation circulation Gwr keeps its own:
//   V_wake(j,k) = Gy*aV + Gwr*bV        W_wake(j,k) = max(0, Gy*aW + Gwr*bW)
// with coefficients that depend on geometry and model constants only (decay, vortex cores, mirrors folded in),
// evaluated in float64 and rounded once.
// Layout per pair (WF_PAIR_STRIDE = 44 floats = 11 float4; 44 words is a conflict-free LDS stride for 16-byte reads
// of consecutive lanes): grid point (j,k) is the float4 {aV, bV, aW, bW} at [(3j+k)*4 .. +3]; then
//   [36] dx = x'_t - x'_i (float64 difference, rounded once; < 0: target upstream, -1 for padding targets)
//   [37] dy = y'_t - y'_i        [38] (dx'/D)^ch_downstream of the Crespo-Hernandez term [A.3-8], 0 where
//   dx > 15 D in float64 (out of reach of the wake-added TI)
//   [39] int bits: bit j (0..2) = grid column j of the target inside the lateral gate |y_i - (y_t + off_j)| < 2 D,
//        bit 3 = x_t > x_i + 0.1 (velocity deficit on) — both decided in float64 as FLORIS does     [40..43] 0
// A source's row (all targets) is padded to a multiple of 1 KiB: the step kernel stages it into LDS with 1-KiB
// global_load_lds wave-instructions.
#define WF_PAIR_STRIDE 44
#define WF_PAIR_DX 36
#define WF_PAIR_DY 37
#define WF_PAIR_TIPOW 38
#define WF_PAIR_BITS 39
#define WF_PAIR_MAX_N 128
#define WF_PAIR_ROW_FLOATS(n) ((((n) * WF_PAIR_STRIDE * 4 + 1023) / 1024) * 256)

// Target-block order of the same records for wf_step_ll_kernel (wf_kernels_ll.hip): block J (targets J G .. J G + G - 1)
// holds the records of its sources i = 0 .. min(N, (J + 1) G) - 1, source-major, padded to whole chunks of 64 records
// (64 / G sources = 11 KiB, the unit of the LDS staging); blocks follow each other.  Offset of block J in floats:
#ifdef __HIPCC__
__host__ __device__
#endif
inline size_t wf_ll_block_offset(int J, int N, int G) {
  const int CH = 64 / G;
  size_t off = 0;
  for (int b = 0; b < J; ++b) {
    const int n_src = (b + 1) * G < N ? (b + 1) * G : N;
    off += (size_t)((n_src + CH - 1) / CH) * 64 * WF_PAIR_STRIDE;
  }
  return off;
}
// Source log of wf_step_ll_kernel per (farm slot, source): a HOT part of 2 floats (the circulations: read by every later
// block), a COLD part of 12 floats (deflection / deficit / turbulence constants: read by the blocks the wake can reach)
// and a 16-byte side record that only split-TI sources write and read (wf_kernels_ll.hip: SrcLog, ColdRec, WfLogSide).
// One allocation of R = (farm slots) x (padded turbines) records: [R x 2 floats][R x 12 floats][R x 4 floats].
#define WF_LOG_HOT_FLOATS 2
#define WF_LOG_COLD_FLOATS 12
#define WF_LOG_FLOATS (WF_LOG_HOT_FLOATS + WF_LOG_COLD_FLOATS)
#define WF_LOG_SIDE_FLOATS 4

struct WfPairConsts {
  int N;   // turbines
  int NP;  // targets per row = capacity G*S of the step-kernel variant (entries t >= N carry dx = -1: never active)
  double D, HH, eps2, num_eps;
  double off[3];
  double decay_a[3];
  double ch_down;
  double gam_top, gam_bot;  // tip-vortex circulations per unit (sin cos Ct ws), over 2 pi  [A.3-1]
  double fifteenD;          // reach of the wake-added TI [A.3-8]: beyond it the record's TI power is stored as 0
  double twoD;              // lateral gate of the wake-added TI [A.3-8]
};
