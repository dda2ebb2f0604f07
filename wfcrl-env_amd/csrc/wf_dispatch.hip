// wf_dispatch.hip — which kernel serves a handle and how it is launched: the per-handle kernel choice
// (wf_set_kernel_choice), the rounds model, the pair tables, the step launch with the float64 re-solve behind it,
// kernel introspection.
#include "wf_handle.h"

#include <cstdlib>
#include <cstddef>
#include <map>
#include <tuple>

namespace wfi {

// Kernel variant for N turbines and B farms: G lanes per farm, S target slots per lane, G*S >= N.
// Throughput regime (the grid fills the chip): smaller G wastes fewer lanes on the triangular
// (upstream->downstream) structure and amortises the per-source work over more farms per wave; S is bounded by
// the 256-VGPR budget that keeps two waves per SIMD resident.
// Latency regime (few farms, e.g. the reference's single-farm env): the chip is not full anyway, so G is widened
// as long as all waves still fit in one residency round — fewer slot passes per source step.
int find_variant(int G, int S) {
  for (int i = 0; i < wfk_num_variants(); ++i) {
    int g, s; const void* fn;
    wfk_variant(i, &g, &s, &fn);
    if (g == G && s == S) return i;
  }
  return -1;
}

int pick_variant(const wf_handle* h, int N, int B) {
  if (h->choice.slot_G > 0 && h->choice.slot_G * h->choice.slot_S >= N) {  // forced (wf_set_kernel_choice), e.g. 16 x 5
    const int v = find_variant(h->choice.slot_G, h->choice.slot_S);
    if (v >= 0) return v;
  }
  static const int pref[][3] = {  // {max N, G, S}
      {4, 4, 1}, {8, 4, 2}, {12, 4, 3}, {16, 4, 4}, {24, 8, 3}, {32, 8, 4}, {48, 16, 3}, {64, 16, 4},
      {80, 16, 5}, {96, 16, 6}, {128, 32, 4}, {192, 64, 3}, {256, 64, 4}};
  int G = 0, S = 0;
  for (auto& r : pref)
    if (N <= r[0]) { G = r[1]; S = r[2]; break; }
  if (!G) return -1;
  if (B > 0) {
    const long resident = (long)h->n_cu * 4 * 2;  // waves the chip holds at two per SIMD
    while (G < 64) {
      const int g2 = G * 2, s2 = (N + g2 - 1) / g2;
      if ((long)B * g2 / 64 > resident / 2 || find_variant(g2, s2) < 0) break;
      G = g2; S = s2;
    }
  }
  return find_variant(G, S);
}

// Rounds model.  A wave solves its 64 / G farms start to finish, so a launch runs in ROUNDS of (blocks the chip holds) x
// (farms per block), and within a round the time depends on how many blocks share a CU (one wave per SIMD each).
// Measured per family at 1, 2, 3 blocks per CU on one MI355X (256 CUs) for farms of 32, 48, 64, 80, 91 turbines
// (tools/rounds_table.py -> profiles/archive/r03_rounds_table.txt), ms; between the measured N the times are interpolated linearly in N (N + 1) / 2, the
// number of (source, target) pairs, beyond them extrapolated the same way.  The number of CUs comes from the device
// (a partitioned or smaller part has shorter rounds, the same time per round).
//   code = (G << 4) | S of wf_step_ll_kernel, 0 = the register-slot kernel wf_step_kernel (its variant for N: pick_variant)
struct LlFamily { int code, farms_per_block, per_cu; };
const LlFamily kLlFamilies[] = {{0, 16, 2}, {(8 << 4) | 1, 32, 3}, {(4 << 4) | 2, 64, 2}, {(4 << 4) | 1, 64, 3}, {(2 << 4) | 2, 128, 2},
                                {(16 << 4) | 1, 16, 3}};
constexpr int kNumFamilies = 6, kNumRoundsN = 5;
const int kRoundsN[kNumRoundsN] = {32, 48, 64, 80, 91};
// [family][N index][blocks per CU - 1]
// (N = 32, 48, 64: the first N turbines of HornsRev2 at 263 deg; N = 80, 91: HornsRev1 / HornsRev2 at 270 deg, the
// BASELINE configs — profiles/r05_rounds_table.txt (re-measured on round 5's kernels: one-slot families in their two-wave build
// up to two blocks per CU; rounds 3-4: r03_v34_rounds_table.txt), r03_v34_batch_sweep_fine.txt, measured with the far-pair skip of pass 2
// (wf_kernels_ll.hip: WF_LL_FAR_SKIP), whose yield depends on layout and direction; layouts move these by up to 10 %)
const double kRoundsMs[kNumFamilies][kNumRoundsN][3] = {
    {{0.088, 0.112, 0.0}, {0.115, 0.145, 0.187}, {0.156, 0.204, 0.0}, {0.210, 0.274, 0.0}, {0.279, 0.345, 0.0}},            // slot (8x4, 16x3, 16x4, 16x5, 16x6)
    {{0.089, 0.117, 0.157}, {0.162, 0.212, 0.282}, {0.234, 0.301, 0.393}, {0.308, 0.390, 0.517}, {0.391, 0.490, 0.648}},  // 8x1
    {{0.124, 0.163, 0.0}, {0.216, 0.281, 0.0}, {0.314, 0.398, 0.0}, {0.428, 0.535, 0.0}, {0.537, 0.666, 0.0}},            // 4x2
    {{0.129, 0.172, 0.250}, {0.215, 0.283, 0.389}, {0.310, 0.405, 0.565}, {0.421, 0.535, 0.767}, {0.514, 0.650, 0.946}},  // 4x1
    {{0.179, 0.250, 0.0}, {0.307, 0.432, 0.0}, {0.454, 0.648, 0.0}, {0.622, 0.856, 0.0}, {0.798, 1.026, 0.0}},            // 2x2
    {{0.068, 0.090, 0.122}, {0.116, 0.151, 0.198}, {0.171, 0.224, 0.292}, {0.228, 0.296, 0.386}, {0.287, 0.361, 0.474}}};  // 16x1
// A partial round behind full ones overlaps with their tail: its cost relative to the same round on an idle chip, by
// (blocks per CU it reaches, resident blocks per CU of the family) — fitted on profiles/archive/r03_batch_sweep_fine.txt
const double kTailFactor[2][3] = {{0.80, 0.95, 0.0}, {0.62, 0.79, 0.86}};  // [per_cu - 2][tail blocks per CU - 1]

// farms per block and resident blocks per CU of family fi for N turbines (the register-slot kernel's follow from its
// variant for N: three waves per SIMD where S <= 3, two otherwise — wf_kernels.hip)
void family_shape(const wf_handle* h, int fi, int N, int* fpb, int* per_cu) {
  *fpb = kLlFamilies[fi].farms_per_block;
  *per_cu = kLlFamilies[fi].per_cu;
  if (kLlFamilies[fi].code == 0) {
    const int v = pick_variant(h, N, 1 << 30);
    if (v >= 0) {
      int G, S; const void* fn;
      wfk_variant(v, &G, &S, &fn);
      *fpb = wfk_tab_waves() * (64 / G);
      *per_cu = S <= 3 ? 3 : 2;
    }
  }
}

// ms of one round of family `fi` at `per_cu` blocks per CU for N turbines
double round_ms(int fi, int N, int per_cu) {
  auto pairs = [](int n) { return 0.5 * n * (n + 1); };
  auto at = [&](int k) { return kRoundsMs[fi][k][per_cu - 1]; };
  int lo = -1, hi = -1;  // measured neighbours (entries that are 0 were not measured: skipped)
  for (int k = 0; k < kNumRoundsN; ++k) {
    if (at(k) <= 0.0) continue;
    if (kRoundsN[k] <= N) lo = k;
    if (kRoundsN[k] >= N && hi < 0) hi = k;
  }
  if (lo < 0 && hi < 0) return 1e300;
  if (lo < 0) return at(hi) * pairs(N) / pairs(kRoundsN[hi]);
  if (hi < 0) return at(lo) * pairs(N) / pairs(kRoundsN[lo]);
  if (lo == hi) return at(lo);
  const double w = (pairs(N) - pairs(kRoundsN[lo])) / (pairs(kRoundsN[hi]) - pairs(kRoundsN[lo]));
  return at(lo) + w * (at(hi) - at(lo));
}

// ms for `farms` farm slots on family fi: whole rounds at full occupancy, then the partial round at the occupancy it
// reaches; a partial round behind full ones overlaps with their tail (kTailFactor)
double ll_estimate(const wf_handle* h, int fi, int N, long farms) {
  int fpb, per_cu;
  family_shape(h, fi, N, &fpb, &per_cu);
  const long blocks = (farms + fpb - 1) / fpb, per_round = (long)h->n_cu * per_cu;
  const long full = blocks / per_round, rem = blocks % per_round;
  double t = full * round_ms(fi, N, per_cu);
  if (rem) {
    const int p = (int)((rem + h->n_cu - 1) / h->n_cu);
    t += (full ? kTailFactor[per_cu - 2][p - 1] : 1.0) * round_ms(fi, N, p);
  }
  return t;
}

// Mixed launch (round 5; priced in round 3, asked for twice since).  A launch costs whole ROUNDS, so a batch a little beyond a
// whole number of rounds of its family pays a nearly empty last round — 69 632 HornsRev1 farms are one round of the 2x2 kernel
// (65 536) plus 4 096 farms that take another full round's time on a sixteenth of the chip: 0.66 of the envelope.  Instead the
// whole rounds run on the family and the remainder on the register-slot kernel, the lowest-latency one, enqueued behind it on
// the same stream for the farms [main, B) (WfGroupArgs::env_base / env_end): its blocks move in as the last round drains.
// Returns the number of farms the family keeps (0: no mixing): by the rounds model here, by measurement in calibrate_families.
int mix_candidate(const wf_handle* h, int fi, int N, long B) {
  if (kLlFamilies[fi].code == 0 || h->choice.mixed == 0) return 0;
  int fpb, per_cu;
  family_shape(h, fi, N, &fpb, &per_cu);
  const long per_round = (long)h->n_cu * per_cu * fpb;
  const long full = B / per_round, rem = B % per_round;
  if (full < 1 || rem == 0) return 0;
  int fpb0, per_cu0;
  family_shape(h, 0, N, &fpb0, &per_cu0);
  if (rem > 2l * h->n_cu * per_cu0 * fpb0) return 0;  // (more than two rounds of the slot kernel: the partial round is well filled)
  return (int)(full * per_round);
}
double mix_estimate(const wf_handle* h, int fi, int N, long B, int main_farms) {
  int fpb, per_cu;
  family_shape(h, fi, N, &fpb, &per_cu);
  const long per_round = (long)h->n_cu * per_cu * fpb;
  return (main_farms / per_round) * round_ms(fi, N, per_cu) + 0.9 * ll_estimate(h, 0, N, B - main_farms);
}

// ms of family fi at B farms in its better form — one launch, or whole rounds + remainder; *mix = the farms the family keeps
double family_estimate(const wf_handle* h, int fi, int N, long B, int* mix) {
  double t = ll_estimate(h, fi, N, B);
  *mix = 0;
  const int m = mix_candidate(h, fi, N, B);
  if (m) {
    const double tm = mix_estimate(h, fi, N, B, m);
    if (tm < 0.93 * t) { t = tm; *mix = m; }
  }
  return t;
}

// Lane-group width of the one-block-at-a-time kernel for N turbines and B farms, 0 = keep wf_step_kernel.  It pays once
// the farm spans several blocks (the register-slot kernel is then pinned at two waves per SIMD by its 27 S state
// registers) and the batch fills the chip.  wf_set_kernel_choice: one_block = 0 disables it, 1 forces (ll_G, ll_S).
int pick_ll(const wf_handle* h, int N, int B) {  // returns (G << 4) | S, 0 = keep wf_step_kernel
  if (h->choice.one_block == 0) return 0;
  const bool veer = h->model.veer != 0.0;  // wind veer: 9 sums per slot; the G <= 4 shapes are instantiated (wfk_ll_has_veer)
  if (N > WF_PAIR_MAX_N) return 0;
  if (h->choice.one_block == 1) {
    const int g = h->choice.ll_G, sl = h->choice.ll_S > 0 ? h->choice.ll_S : 1;
    const bool ok = ((g == 4 || g == 8 || g == 16) && sl == 1) || ((g == 4 || g == 2) && sl == 2);
    return (ok && N > g * sl && (!veer || wfk_ll_has_veer(g, sl, 1))) ? ((g << 4) | sl) : 0;
  }
  // the cheapest estimate wins: at N = 80 the register-slot kernel up to ~8192 farms, G = 8 up to ~24576, then the two
  // G = 4 kernels depending on how the batch divides into rounds of 32768 / 49152, G = 2 x 2 on whole rounds of 65536
  if (N <= 16) return 0;
  {  // less than one block per CU of the register-slot kernel: the latency regime, where pick_variant widens that kernel's
     // lane group (fewer slot passes per source) — the table's throughput variants do not describe it
    int fpb, per_cu;
    family_shape(h, 0, N, &fpb, &per_cu);
    if ((long)B <= (long)h->n_cu * fpb) return 0;
  }
  int best = 0;
  double t_best = 1e300;
  for (int fi = 0; fi < kNumFamilies; ++fi) {
    const LlFamily& f = kLlFamilies[fi];
    if (f.code && N <= (f.code >> 4) * (f.code & 15)) continue;  // needs more than one block
    if (f.code == ((8 << 4) | 1) && N <= 32) continue;           // (not instantiated to pay below that)
    if (veer && f.code && !wfk_ll_has_veer(f.code >> 4, f.code & 15, 1)) continue;
    int mix_unused;
    double t = family_estimate(h, fi, N, B, &mix_unused);
    // G = 16 runs neck and neck with the register-slot kernel up to two blocks per CU (0.290 against 0.294 ms at
    // HornsRev1 x 8192, either way round from layout to layout): there it has to win by 4 % — its case is the third block
    if (f.code == ((16 << 4) | 1) && (long)B <= 2l * h->n_cu * f.farms_per_block) t *= 1.04;
    if (t < t_best) { t_best = t; best = f.code; }
  }
  return best;
}

// A grouped launch (series rows / binned directions) pads every group to whole blocks: more farm slots than farms.  The
// two G = 4 kernels have the same block size, so the choice between them can follow the padded count without touching
// the group lists (HornsRev1 x 65536 in 104 groups = 1072 blocks: three rounds of the two-slot kernel, 1.65 ms, against
// two of the one-slot kernel).
int repick_ll_slots(const wf_handle* h, int N, int ll_G, int ll_S, long farm_slots) {
  if (ll_G != 4 || h->choice.one_block == 1 || N <= 8) return ll_S;
  return ll_estimate(h, 3, N, farm_slots) < ll_estimate(h, 2, N, farm_slots) ? 1 : 2;
}

// (Re)pick the kernels of a handle for N turbines and B farms under its choice: the register-slot variant and the
// one-block kernel's shape; frees what was laid out for another shape.  The caller has drained the stream.
void reset_calibration(wf_handle* h) {
  h->calib_done = false; h->calib_code = -1; h->tab_slot = false; h->mix_main = 0;
  for (float& m : h->calib_ms) m = 0.0f;
  h->fly_calib = 0; h->fly_calib_ms[0] = h->fly_calib_ms[1] = 0.0f;
}

// the rounds model's answer to "mix?" for family `code` at B farms: farms the family keeps, 0 = one launch
int model_mix(const wf_handle* h, int code, int N, int B) {
  if (!code || h->choice.one_block == 0) return 0;
  for (int fi = 0; fi < kNumFamilies; ++fi)
    if (kLlFamilies[fi].code == code) {
      int m = 0;
      family_estimate(h, fi, N, B, &m);
      return m;
    }
  return 0;
}

void apply_kernel_pick(wf_handle* h, int N, int B, bool* variant_changed) {
  reset_calibration(h);
  const int v = pick_variant(h, N, B);
  const int llg = pick_ll(h, N, B);
  set_ll_shape(h, llg >> 4, llg ? (llg & 15) : 1);
  h->mix_main = model_mix(h, llg, N, B);
  if (variant_changed) *variant_changed = v != h->variant;
  if (v != h->variant) {  // the pair table is laid out for the variant's capacity
    hipFree(h->d_pair_tab); hipFree(h->d_pair_first);
    h->d_pair_tab = nullptr; h->d_pair_first = nullptr; h->pair_dirty = true; h->pair_groups_cap = 0;
    h->variant = v;
  }
}

// The one-block kernel's (G, S) of a handle: its pair table and source log are laid out for (N, G, S).  The caller has
// made sure no launch is in flight.
void set_ll_shape(wf_handle* h, int G, int S) {
  if (G == h->ll_G && S == h->ll_S) return;
  hipFree(h->d_ll_tab); hipFree(h->d_ll_flag); hipFree(h->d_src_log);
  h->d_ll_tab = h->d_src_log = nullptr; h->d_ll_flag = nullptr; h->ll_groups_cap = h->log_records_cap = 0;
  h->ll_G = G; h->ll_S = S; h->pair_dirty = true;
}
// Shared wind: (re)build the geometry-only pair table after the geometry kernel (same stream).  Returns the table
// pointer to hand to the step kernel, or nullptr when the on-the-fly path applies (per-farm wind, N too large,
// or WF_NO_PAIR_TABLE set for A/B runs).
int pair_table(wf_handle* h, const float** out) {
  *out = nullptr;
  if ((h->wind_count != 1 && !h->shared_dir && h->n_groups == 0) || h->N > WF_PAIR_MAX_N || !wfk_variant_has_table(h->variant) || h->choice.pair_table == 0)
    return WF_OK;
  int vG, vS; const void* vfn;
  wfk_variant(h->variant, &vG, &vS, &vfn);
  const int NP = vG * vS;
  const size_t ng = h->n_groups > 0 ? (size_t)h->n_groups : 1;
  if (!h->d_pair_tab || h->pair_groups_cap < ng) {
    hipFree(h->d_pair_tab); hipFree(h->d_pair_first);
    h->d_pair_tab = nullptr; h->d_pair_first = nullptr; h->pair_groups_cap = 0;
    WF_HIP(h, hipMalloc(&h->d_pair_tab, sizeof(float) * ng * h->N * WF_PAIR_ROW_FLOATS(NP)));
    WF_HIP(h, hipMalloc(&h->d_pair_first, sizeof(int) * ng * h->N));
    h->pair_groups_cap = ng;
    h->pair_dirty = true;
  }
  if (h->pair_dirty) {
    const wf_model_params& m = h->model;
    WfPairConsts pc{};
    const double D = m.rotor_diameter, HH = m.hub_height, eps = m.eps_gain * D;
    pc.N = h->N; pc.NP = NP; pc.D = D; pc.HH = HH; pc.eps2 = eps * eps; pc.num_eps = m.num_eps; pc.ch_down = m.ch_downstream;
    const double off[3] = {-D / 4, 0.0, D / 4};
    double uinf = 0;
    for (int k = 0; k < 3; ++k) uinf += std::pow((HH + off[k]) / HH, m.shear) / 3.0;
    pc.fifteenD = 15.0 * D;
    pc.twoD = 2.0 * D;
    pc.gam_top = (1.0 / 16.0) * D * std::pow((HH + D / 2) / HH, m.shear) * uinf;  // (1/2pi)(pi/8) D vel_top uinf
    pc.gam_bot = (1.0 / 16.0) * D * std::pow((HH - D / 2) / HH, m.shear) * uinf;
    for (int k = 0; k < 3; ++k) {
      pc.off[k] = off[k];
      const double z = HH + off[k];
      const double dudz = m.shear * std::pow(1.0 / HH, m.shear) * std::pow(z, m.shear - 1.0);
      const double lm = m.kappa * z / (1.0 + m.kappa * z / (D / 8.0));
      pc.decay_a[k] = 4.0 * lm * lm * std::fabs(dudz) / uinf / pc.eps2;
    }
    WF_HIP(h, wfk_launch_pair_table(&pc, (int)ng, h->d_gx, h->d_gy, h->d_pair_tab, h->d_pair_first, h->stream));
    if (h->ll_G) {  // the same records in target-block order, and the per-direction cross-block-tie flag
      if (!h->d_ll_tab || h->ll_groups_cap < ng) {
        hipFree(h->d_ll_tab); hipFree(h->d_ll_flag);
        h->d_ll_tab = nullptr; h->d_ll_flag = nullptr; h->ll_groups_cap = 0;
        WF_HIP(h, hipMalloc(&h->d_ll_tab, sizeof(float) * ng * wfk_ll_table_floats(h->N, h->ll_G * h->ll_S)));
        WF_HIP(h, hipMalloc(&h->d_ll_flag, sizeof(int) * ng));
        h->ll_groups_cap = ng;
      }
      WF_HIP(h, wfk_launch_pair_table_ll(&pc, h->ll_G * h->ll_S, (int)ng, h->d_gx, h->d_gy, h->d_ll_tab, h->d_ll_flag, h->stream));
      // which kernel serves which direction is decided on the device (no host round trip on the asynchronous path);
      // where the wind came through a synchronising call anyway, the flags are read back once so that a launch nobody
      // needs is not enqueued at all
      h->ll_ties = 2;
      if (h->wind_sync) {
        std::vector<int> f(ng);
        WF_HIP(h, hipMemcpyAsync(f.data(), h->d_ll_flag, sizeof(int) * ng, hipMemcpyDeviceToHost, h->stream));
        WF_HIP(h, hipStreamSynchronize(h->stream));
        size_t tied = 0;
        for (int v : f) tied += v != 0;
        h->ll_ties = tied == 0 ? 0 : (tied == ng ? 1 : 2);
      }
    }
    h->pair_dirty = false;
  }
  *out = h->d_pair_tab;
  return WF_OK;
}

// Rotation + sort of `n_env` wind conditions on the handle's stream.  A geometry per farm (n_env == B) also yields the
// per-farm cross-block-tie flags for the on-the-fly one-block kernel; sync_ok: the caller synchronises anyway, so the
// "any farm tied" flag is read back and a launch nobody needs is never enqueued.
int ll_fly_S(const wf_handle* h);
int ll_fly_G(const wf_handle* h);
int run_geometry(wf_handle* h, int n_env, const double* d_wd, bool sync_ok) {
  const bool per_farm = n_env == h->B && h->B > 1 && h->ll_G && wfk_ll_has_fly(ll_fly_G(h), ll_fly_S(h));
  // several layouts in the batch (wf_set_layouts): a geometry per farm whatever the wind (the callers pass n_env == B)
  WF_HIP(h, wfk_launch_geometry(n_env, h->N, h->d_lx, h->d_ly, h->d_centre, h->n_layouts == 1 ? 0 : (h->d_layout_of ? 2 : 1),
                                h->d_layout_of, h->d_layout_n, d_wd, 1, h->d_gx, h->d_gy, h->d_gidx,
                                per_farm ? ll_fly_G(h) * ll_fly_S(h) : 0, h->d_farm_tie, h->d_farm_tie ? h->d_farm_tie + h->B : nullptr,
                                h->stream));
  h->farm_ties = 2;
  h->dir_slots = 0;
  h->geo_tie_block = per_farm ? ll_fly_G(h) * ll_fly_S(h) : 0;
  if (per_farm && h->n_layouts == 1 && h->choice.fly_one_block != 0 && h->choice.far_skip != 0) {
    // launch order of the on-the-fly one-block kernel: ascending wind direction (wf_sort.hip), padded to its whole blocks
    const int fpb = wfk_ll_farms_per_block(ll_fly_G(h));
    const size_t slots = (size_t)((h->B + fpb - 1) / fpb) * fpb;
    if (slots > h->dir_perm_cap) {
      WF_HIP(h, hipStreamSynchronize(h->stream));
      hipFree(h->d_dir_perm); hipFree(h->d_sort_keys); hipFree(h->d_sort_vals); hipFree(h->d_sort_tmp);
      h->d_dir_perm = h->d_sort_vals = nullptr; h->d_sort_keys = nullptr; h->d_sort_tmp = nullptr; h->dir_perm_cap = 0;
      WF_HIP(h, wfk_sort_tmp_bytes(h->B, &h->sort_tmp_bytes));
      WF_HIP(h, hipMalloc(&h->d_dir_perm, sizeof(int) * slots));
      WF_HIP(h, hipMalloc(&h->d_sort_keys, sizeof(float) * 2 * (size_t)h->B));
      WF_HIP(h, hipMalloc(&h->d_sort_vals, sizeof(int) * (size_t)h->B));
      WF_HIP(h, hipMalloc(&h->d_sort_tmp, h->sort_tmp_bytes > 0 ? h->sort_tmp_bytes : 16));
      h->dir_perm_cap = slots;
    }
    WF_HIP(h, wfk_sort_by_direction(h->B, (int)slots, d_wd, h->d_sort_keys, h->d_sort_vals, h->d_sort_tmp, h->sort_tmp_bytes,
                                    h->d_dir_perm, h->stream));
    h->dir_slots = (int)slots;
  }
  if (per_farm && sync_ok) {
    int any = 0;
    WF_HIP(h, hipMemcpyAsync(&any, h->d_farm_tie + h->B, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    WF_HIP(h, hipStreamSynchronize(h->stream));
    h->farm_ties = any ? 1 : 0;
  }
  return WF_OK;
}

// Target slots per lane of the one-block kernel ON THE FLY (a wind per farm): two at G = 4 whatever the table path
// uses — there the second slot halves the per-source geometry work as well (HornsRev1 x 65536: 3.48 ms against 4.14).
int ll_fly_S(const wf_handle* h) { return h->ll_G <= 4 ? 2 : h->ll_S; }
// ... and its lane-group width: the table path's where an on-the-fly instantiation exists
int ll_fly_G(const wf_handle* h) {
  // G = 2 x 2 on the fly too where the table path runs it (round 4: with the farms of a wave sorted by direction its
  // 4-turbine blocks make the wave-level far test take — 3.31 against 3.49 ms at HornsRev1 x 65536; round 3, before the
  // skips: 3.43 against 3.40); a veer model has no such instantiation, and neither has G = 16: G = 4 x 2 / G = 8 serve
  if (h->ll_G == 2) return h->model.veer != 0.0 ? 4 : 2;
  return h->ll_G == 16 ? 8 : h->ll_G;
}

// turbines per farm in the source log of the one-block kernel: whole lane-group blocks (of the larger of the two
// block sizes: the table path and the on-the-fly path share the buffer)
size_t ll_npad(const wf_handle* h) {
  const int a = h->ll_G * h->ll_S, b = ll_fly_G(h) * ll_fly_S(h), gs = a > b ? a : b;  // (powers of two)
  return (size_t)((h->N + gs - 1) / gs) * gs;
}

int ll_log_fpb(const wf_handle* h) {
  const int a = wfk_ll_farms_per_block(h->ll_G), b = wfk_ll_farms_per_block(ll_fly_G(h));
  return a > b ? a : b;
}
// The source log of the one-block kernel: `slots` farm slots x ll_npad(h) records of WF_LOG_FLOATS + WF_LOG_SIDE_FLOATS
// floats.  Both factors can grow under an unchanged (G, S) — the batch, and the block size of the on-the-fly kernel with a
// veer model (ll_fly_G) — so the capacity is kept in records and checked at every launch.
static int ensure_log(wf_handle* h, size_t slots, size_t* records) {
  *records = slots * ll_npad(h);
  if (*records > h->log_records_cap) {
    WF_HIP(h, hipStreamSynchronize(h->stream));
    hipFree(h->d_src_log); h->d_src_log = nullptr; h->log_records_cap = 0;
    WF_HIP(h, hipMalloc(&h->d_src_log, sizeof(float) * *records * (WF_LOG_FLOATS + WF_LOG_SIDE_FLOATS)));
    h->log_records_cap = *records;
  }
  return WF_OK;
}
// One launch of the step kernel on the handle's stream with the handle's current geometry / wind / table state.
int launch_step_f32(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* ea) {
  const int gstride = (h->wind_count == 1 || h->shared_dir) ? 0 : h->N;
  const int wstride = (h->wind_count == 1) ? 0 : 1;
  const float* ptab = nullptr;
  int rc = pair_table(h, &ptab);
  if (rc != WF_OK) return rc;
  WfGroupArgs ga{};
  ga.mod = 1;
  ga.risk_flags = h->d_flags;
  ga.n_real = h->d_nreal;
  if (h->res_mask) {  // the real launch of a step with the re-solve behind it (not a calibration probe): launch_step
    ga.res_list = h->d_res_list; ga.flags_raw = h->d_flags_raw; ga.res_mask = h->res_mask;
    ga.res_count = h->d_res_count + h->res_parity; ga.res_zero = h->d_res_count + (h->res_parity ^ 1);
  }
  ga.blk_unit = 1;
  if (h->n_groups > 0) {
    ga.perm = h->d_perm; ga.blk_group = h->d_blk_group; ga.n_blocks = h->n_blocks;
    ga.shift = h->group_shift; ga.mod = h->n_groups;
    ga.blk_unit = group_unit(h); ga.n_slots = h->n_slots;
  }
  if (ptab && h->ll_G && !(h->tab_slot && h->n_groups == 0)) {
    // the one-block-at-a-time kernel serves every direction without a cross-block tie; wf_step_kernel, enqueued right
    // behind it, serves the others (device-side predicate, no host round trip)
    const int fpb = ll_log_fpb(h);  // (farm slots of the log: whole blocks of the wider of the two paths' blocks)
    const size_t slots = h->n_groups > 0 ? (size_t)h->n_slots : (size_t)((h->B + fpb - 1) / fpb) * fpb;
    size_t log_records = 0;
    if ((rc = ensure_log(h, slots, &log_records)) != WF_OK) return rc;
    // mixed launch (mix_candidate above): the family keeps its whole rounds, the farms [M, B) go to wf_step_kernel behind it
    const int M = (h->choice.mixed != 0 && h->mix_main > 0 && h->mix_main < h->B && h->n_groups == 0) ? h->mix_main : 0;  // (mixed == 0: always ONE launch, include/wfstep.h)
    if (M) { ga.env_base = 0; ga.env_end = M; }
    if (h->ll_ties != 1)
      WF_HIP(h, wfk_launch_step_ll(h->ll_G, h->ll_S, &h->consts, h->d_tab, h->d_gidx, h->d_ws, h->d_wd, wstride, yaw, power, wspd, wdir,
                                   load, h->B, ea, h->d_ll_tab, h->d_ll_flag, h->d_src_log,
                                   log_records, &ga, h->stream));
    if (M) {
      if (h->ll_ties != 0) {  // (a direction with a cross-block tie: the family's share is wf_step_kernel's too)
        WfGroupArgs gp = ga;
        gp.pred = h->d_ll_flag;
        WF_HIP(h, wfk_launch_step(h->variant, &h->consts, h->d_tab, h->d_gx, h->d_gy, h->d_gidx, gstride, h->d_ws, h->d_wd,
                                  wstride, yaw, power, wspd, wdir, load, h->B, ea, ptab, h->d_pair_first, &gp, h->stream, &h->grid));
      }
      // the remainder, behind the family on the same stream.  (On a side stream beside it — tried: 1.139 against 1.116 ms at
      // 69 632 farms — it gains nothing: a block of the family lives for the whole round, so the half of them that has to
      // wait for the remainder's blocks to leave finishes that much later.)
      ga.env_base = M; ga.env_end = h->B;
      WF_HIP(h, wfk_launch_step(h->variant, &h->consts, h->d_tab, h->d_gx, h->d_gy, h->d_gidx, gstride, h->d_ws, h->d_wd,
                                wstride, yaw, power, wspd, wdir, load, h->B, ea, ptab, h->d_pair_first, &ga, h->stream, &h->grid));
      return WF_OK;
    }
    if (h->ll_ties == 0) return WF_OK;
    ga.pred = h->d_ll_flag;
  }
  if (!ptab && gstride != 0 && h->B > 1 && h->ll_G && wfk_ll_has_fly(ll_fly_G(h), ll_fly_S(h)) && h->choice.fly_one_block != 0 &&
      h->fly_calib != 2) {
    // a wind per farm: the one-block kernel on the fly; wf_step_kernel behind it for the farms whose own geometry has
    // an x' tie across a block boundary (per-farm device flags from the geometry kernel)
    const int fpb = ll_log_fpb(h);
    const size_t slots = (size_t)((h->B + fpb - 1) / fpb) * fpb;
    size_t log_records = 0;
    if ((rc = ensure_log(h, slots, &log_records)) != WF_OK) return rc;
    // The tie flags and the launch order were laid out by the geometry pass for ONE block shape; a change of shape
    // since (leaving a grouped launch, a veer model switched on or off) means another pass over the same wind first.
    if (h->geo_tie_block != ll_fly_G(h) * ll_fly_S(h)) {
      if ((rc = run_geometry(h, h->B, h->d_wd, false)) != WF_OK) return rc;
    }
    WfGroupArgs gf = ga;
    if (h->dir_slots > 0 && h->n_groups == 0) gf.perm = h->d_dir_perm;  // farms of like direction share a wave (run_geometry)
    WF_HIP(h, wfk_launch_step_ll_fly(ll_fly_G(h), ll_fly_S(h), &h->consts, h->d_tab, h->d_gidx, h->d_gx, h->d_gy, h->d_ws, h->d_wd, yaw,
                                     power, wspd, wdir, load, h->B, ea, h->d_farm_tie, h->d_src_log,
                                     log_records, &gf, h->stream));
    if (h->farm_ties == 0) return WF_OK;
    ga.farm_pred = h->d_farm_tie;
  }
  WF_HIP(h, wfk_launch_step(h->variant, &h->consts, h->d_tab, h->d_gx, h->d_gy, h->d_gidx, gstride, h->d_ws, h->d_wd,
                            wstride, yaw, power, wspd, wdir, load, h->B, ea, ptab, h->d_pair_first, &ga, h->stream, &h->grid));
  return WF_OK;
}

// Per-handle calibration of the kernel family.  The rounds model above is a table of milliseconds measured on ONE box
// for two layouts; layouts, directions, clocks and partitioned devices move the families against each other by up to
// 10 %.  So the handle measures — BEFORE its first real launch of a configuration (round 5; rounds 3-4 waited for the third
// step, so that steps 1-2 and 3+ could come from two families with different summation orders): every family the model
// prices within 60 % of its best guess is launched on the caller's own buffers (the step is stateless: the real launch
// follows and overwrites them; each probe carries its own warm launch), three launches of each are timed one by one with
// HIP events, and the fastest is kept — the guess itself when it is within 4 % of it.  A few ms, once; that one call
// synchronises (wf_calibrate does the same at a point of the caller's choosing).  The result is cached per process under
// (device, turbines, batch, layout, model, skip switch): a re-created handle of the same configuration does not time
// again and is served by the same family; wf_get_calibration / wf_set_calibration save and restore it across processes.
// The rounds model remains the cold-start guess and the price list of grouped launches.
static bool calibration_due(const wf_handle* h) {
  if (h->calib_done || h->choice.calibrate == 0 || h->choice.one_block != -1 || h->choice.pair_table == 0) return false;
  if (h->N <= 16 || h->N > WF_PAIR_MAX_N || h->n_groups > 0 || h->n_layouts != 1) return false;
  if (!(h->wind_count == 1 || h->shared_dir) || !wfk_variant_has_table(h->variant)) return false;
  int fpb, per_cu;
  family_shape(h, 0, h->N, &fpb, &per_cu);
  return (long)h->B > (long)h->n_cu * fpb;  // below that: the latency regime, pick_variant's widened kernel
}

// ---- process-wide cache of calibration results ----
struct CalibKey {
  int device, N, B, n_cu, far_skip;
  int mixed, slot_G, slot_S;  // (a handle created with mixed = 0, or with a forced register-slot shape, must not inherit the split / the timing of one without)
  unsigned long long layout_hash, model_hash;
  bool operator<(const CalibKey& o) const {
    return std::tie(device, N, B, n_cu, far_skip, mixed, slot_G, slot_S, layout_hash, model_hash) <
           std::tie(o.device, o.N, o.B, o.n_cu, o.far_skip, o.mixed, o.slot_G, o.slot_S, o.layout_hash, o.model_hash);
  }
};
struct CalibVal {
  bool have_tab = false, have_fly = false;
  int code = -1, fly = 0, mix_main = 0;
  float ms[8] = {}, fly_ms[2] = {};
};
static std::mutex g_calib_mu;
static std::map<CalibKey, CalibVal> g_calib;

static unsigned long long fnv(const void* p, size_t n, unsigned long long hsh = 1469598103934665603ull) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; ++i) hsh = (hsh ^ b[i]) * 1099511628211ull;
  return hsh;
}
static CalibKey calib_key(const wf_handle* h) {
  CalibKey k{};
  k.device = h->device; k.N = h->N; k.B = h->B; k.n_cu = h->n_cu; k.far_skip = h->choice.far_skip;
  k.mixed = h->choice.mixed != 0; k.slot_G = h->choice.slot_G; k.slot_S = h->choice.slot_S;
  k.layout_hash = fnv(h->ly.data(), sizeof(double) * h->ly.size(), fnv(h->lx.data(), sizeof(double) * h->lx.size()));
  const wf_model_params& m = h->model;  // every scalar up to the table (the pointers behind it are this handle's own copies)
  unsigned long long mh = fnv(&m, offsetof(wf_model_params, n_table));
  const int sw[3] = {m.enable_secondary_steering, m.enable_yaw_added_recovery, m.enable_transverse_velocities};
  mh = fnv(sw, sizeof(sw), mh);
  mh = fnv(h->tws.data(), sizeof(double) * h->tws.size(), mh);
  mh = fnv(h->tct.data(), sizeof(double) * h->tct.size(), mh);
  k.model_hash = fnv(h->tcp.data(), sizeof(double) * h->tcp.size(), mh);
  return k;
}

// the family `code` ((G << 4) | S, 0 = the register-slot kernel) serves the handle's ungrouped table path from now on.
// The caller has drained the stream.
static void apply_family(wf_handle* h, int code, int guess_code) {
  if (code == 0) {
    // ll_G / ll_S keep the rounds model's shape: grouped launches and the on-the-fly path (a wind per farm) go on using it
    set_ll_shape(h, guess_code >> 4, guess_code ? (guess_code & 15) : 1);
    h->tab_slot = guess_code != 0;
  } else {
    set_ll_shape(h, code >> 4, code & 15);
    h->tab_slot = false;
  }
}

static bool family_valid(const wf_handle* h, int fi) {
  const LlFamily& f = kLlFamilies[fi];
  const bool veer = h->model.veer != 0.0;
  if (f.code && h->N <= (f.code >> 4) * (f.code & 15)) return false;
  if (f.code == ((8 << 4) | 1) && h->N <= 32) return false;
  if (veer && f.code && !wfk_ll_has_veer(f.code >> 4, f.code & 15, 1)) return false;
  return true;
}

static int calibrate_families(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* ea) {
  const int N = h->N, B = h->B;
  double est[kNumFamilies], best_est = 1e300;
  int mix_f[kNumFamilies];  // every family is timed in the form the rounds model prefers for it: one launch, or whole rounds + remainder
  for (int fi = 0; fi < kNumFamilies; ++fi) {
    est[fi] = 1e300;
    mix_f[fi] = 0;
    if (!family_valid(h, fi)) continue;
    est[fi] = family_estimate(h, fi, N, B, &mix_f[fi]);
    if (est[fi] < best_est) best_est = est[fi];
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  WF_HIP(h, hipEventCreate(&e0));
  WF_HIP(h, hipEventCreate(&e1));
  int best = -1, guess = -1;
  float best_ms = 1e30f;
  int rc = WF_OK;
  const int code0 = h->ll_G ? ((h->ll_G << 4) | h->ll_S) : 0;  // the rounds model's guess
  for (int fi = 0; fi < kNumFamilies && rc == WF_OK; ++fi) {
    if (kLlFamilies[fi].code == code0 && est[fi] < 1e300) guess = fi;
    if (!(est[fi] <= 1.6 * best_est) && fi != guess) continue;
    hipStreamSynchronize(h->stream);
    apply_family(h, kLlFamilies[fi].code, code0);
    h->mix_main = mix_f[fi];
    if ((rc = launch_step_f32(h, yaw, power, wspd, wdir, load, ea)) != WF_OK) break;  // tables, log, first-launch costs
    float fam_ms = 1e30f;  // the fastest of three launches timed one by one (two handles on one kernel differ by 3-5 %: noise counts)
    for (int r = 0; r < 3 && rc == WF_OK; ++r) {
      hipError_t e = hipEventRecord(e0, h->stream);
      rc = launch_step_f32(h, yaw, power, wspd, wdir, load, ea);
      if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.0f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e != hipSuccess) { rc = fail(h, WF_E_HIP, std::string("calibration: ") + hipGetErrorString(e)); break; }
      if (ms < fam_ms) fam_ms = ms;
    }
    if (rc != WF_OK) break;
    h->calib_ms[fi] = fam_ms;
    if (fam_ms < best_ms) { best_ms = fam_ms; best = fi; }
  }
  hipStreamSynchronize(h->stream);
  if (rc != WF_OK) {  // a failed probe leaves the handle on the rounds model's guess, not on a candidate
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    apply_family(h, code0, code0);
    h->mix_main = guess >= 0 ? mix_f[guess] : 0;
    return rc;
  }
  // The guess stands unless another family beats it by 4 %: near-ties would otherwise be decided by timing noise, and two
  // runs of one program would be served by different families (which agree within the parity tolerances, not bit for bit).
  if (guess >= 0 && h->calib_ms[guess] > 0.0f && h->calib_ms[guess] <= 1.04f * best_ms) best = guess;
  const int code = best >= 0 ? kLlFamilies[best].code : code0;
  apply_family(h, code, code0);
  h->calib_code = code;
  // ... and the winner's OTHER form, where the batch has a mixed launch for it (whole rounds on the family, the remainder on
  // wf_step_kernel): measured like a family; the model's preference stands unless the other form beats it by 4 %
  h->mix_main = best >= 0 ? mix_f[best] : 0;
  h->calib_ms[6] = 0.0f;
  const int m = (best >= 0 && code) ? mix_candidate(h, best, N, B) : 0;
  if (m) {
    const int model_form = h->mix_main;
    h->mix_main = model_form ? 0 : m;
    float alt_ms = 1e30f;
    rc = launch_step_f32(h, yaw, power, wspd, wdir, load, ea);
    for (int r = 0; r < 3 && rc == WF_OK; ++r) {
      hipError_t e = hipEventRecord(e0, h->stream);
      rc = launch_step_f32(h, yaw, power, wspd, wdir, load, ea);
      if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.0f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e != hipSuccess) { rc = fail(h, WF_E_HIP, std::string("calibration: ") + hipGetErrorString(e)); break; }
      if (ms < alt_ms) alt_ms = ms;
    }
    hipStreamSynchronize(h->stream);
    if (rc != WF_OK) { h->mix_main = model_form; hipEventDestroy(e0); hipEventDestroy(e1); return rc; }
    h->calib_ms[6] = model_form ? h->calib_ms[best] : alt_ms;  // (what the mixed form measured)
    if (!(alt_ms < 0.96f * h->calib_ms[best])) h->mix_main = model_form;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  h->calib_done = true;
  return WF_OK;
}

// The same question on the on-the-fly path (a wind per farm).  The family comes from the table path's pick; what the rounds
// model cannot know is how the register-slot kernel compares there — on the fly it evaluates every pair once, the
// one-block kernel re-reads its log per block, and between one and three rounds of the slot kernel (HornsRev1 x 16384,
// Ormonde x 16384 ... 24576) the slot kernel is 10-15 % faster (profiles/r04_fly_pick_sweep.txt).  Before the first step
// with a wind per farm: both timed on the caller's buffers, the slot kernel has to win by 4 % (near-ties stay where they are).
static bool fly_calibration_due(const wf_handle* h) {
  if (h->fly_calib != 0 || h->choice.calibrate == 0 || h->choice.one_block != -1 || h->choice.fly_one_block != -1) return false;
  if (h->wind_count == 1 || h->shared_dir || h->n_groups > 0 || h->B <= 1 || !h->ll_G) return false;
  return wfk_ll_has_fly(ll_fly_G(h), ll_fly_S(h));
}

static int calibrate_fly(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* ea) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  WF_HIP(h, hipEventCreate(&e0));
  WF_HIP(h, hipEventCreate(&e1));
  int rc = WF_OK;
  for (int opt = 1; opt <= 2 && rc == WF_OK; ++opt) {
    h->fly_calib = opt;
    if ((rc = launch_step_f32(h, yaw, power, wspd, wdir, load, ea)) != WF_OK) break;  // log, first-launch costs
    float best = 1e30f;
    for (int r = 0; r < 3 && rc == WF_OK; ++r) {
      hipError_t e = hipEventRecord(e0, h->stream);
      rc = launch_step_f32(h, yaw, power, wspd, wdir, load, ea);
      if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.0f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e != hipSuccess) { rc = fail(h, WF_E_HIP, std::string("calibration: ") + hipGetErrorString(e)); break; }
      if (ms < best) best = ms;
    }
    h->fly_calib_ms[opt - 1] = best;
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  h->fly_calib = (rc == WF_OK && h->fly_calib_ms[1] < 0.96f * h->fly_calib_ms[0]) ? 2 : 1;
  return rc;
}

// Whatever calibration the handle's current configuration and wind regime still owe, now (the stream is drained on the
// way).  use_cache: take / leave the process-wide result for this configuration instead of timing again.
int calibrate_now(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* probe, bool use_cache) {
  const bool tab_due = calibration_due(h), fly_due = fly_calibration_due(h);
  if (!tab_due && !fly_due) return WF_OK;
  const CalibKey key = calib_key(h);
  CalibVal cv;
  {
    std::lock_guard<std::mutex> lk(g_calib_mu);
    auto it = g_calib.find(key);
    if (it != g_calib.end()) cv = it->second;
  }
  int rc = WF_OK;
  if (tab_due) {
    const int code0 = h->ll_G ? ((h->ll_G << 4) | h->ll_S) : 0;
    if (use_cache && cv.have_tab) {
      const int cur = h->tab_slot ? 0 : code0;
      if (cv.code != cur) {
        WF_HIP(h, hipStreamSynchronize(h->stream));
        apply_family(h, cv.code, code0);
      }
      h->calib_code = cv.code; h->calib_done = true; h->mix_main = h->choice.mixed == 0 ? 0 : cv.mix_main;
      for (int k = 0; k < 8; ++k) h->calib_ms[k] = cv.ms[k];
    } else {
      if ((rc = calibrate_families(h, yaw, power, wspd, wdir, load, probe)) != WF_OK) return rc;
      cv.have_tab = true; cv.code = h->calib_code; cv.mix_main = h->mix_main;
      for (int k = 0; k < 8; ++k) cv.ms[k] = h->calib_ms[k];
    }
  }
  if (fly_due) {
    if (use_cache && cv.have_fly) {
      h->fly_calib = cv.fly; h->fly_calib_ms[0] = cv.fly_ms[0]; h->fly_calib_ms[1] = cv.fly_ms[1];
    } else {
      if ((rc = calibrate_fly(h, yaw, power, wspd, wdir, load, probe)) != WF_OK) return rc;
      cv.have_fly = true; cv.fly = h->fly_calib; cv.fly_ms[0] = h->fly_calib_ms[0]; cv.fly_ms[1] = h->fly_calib_ms[1];
    }
  }
  std::lock_guard<std::mutex> lk(g_calib_mu);
  g_calib[key] = cv;
  return WF_OK;
}

// A calibration saved from another handle / process (wf_get_calibration, wf_get_fly_calibration): taken as is, nothing is timed.
int apply_saved_calibration(wf_handle* h, int code, int fly_choice) {
  if (code >= 0) {
    int fi = -1;
    for (int k = 0; k < kNumFamilies; ++k)
      if (kLlFamilies[k].code == code) fi = k;
    if (fi < 0 || !family_valid(h, fi)) return fail(h, WF_E_INVALID, "wf_set_calibration: not a kernel family of this turbine count / model");
    if (h->choice.one_block != -1 || h->choice.pair_table == 0) return fail(h, WF_E_INVALID, "wf_set_calibration: the kernel choice of this handle is forced");
    WF_HIP(h, hipStreamSynchronize(h->stream));
    apply_family(h, code, pick_ll(h, h->N, h->B));  // (tables and log are laid out per shape: rebuilt by the next launch)
    h->calib_code = code; h->calib_done = true;
    h->mix_main = model_mix(h, code, h->N, h->B);  // (deterministic: the rounds model's answer for this family)
  }
  if (fly_choice == 1 || fly_choice == 2) h->fly_calib = fly_choice;
  return WF_OK;
}

// The step as the ABI sees it: the float32 kernels, then — unless switched off (wf_set_risk_resolve) — the float64 solve of
// the flagged (or all) farms on the same stream, overwriting their outputs.  (wind_veer != 0 no longer forces it: the
// VEER instantiations of the float32 kernels serve such models, with the same flags.)
int launch_step(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, const WfEnvArgs* ea) {
  // The timed launches must leave no trace: a plain step is stateless, and a fused env step is timed WITHOUT its action —
  // a solve at the yaw the env state holds now (no transition, no reward), whose outputs the real launch overwrites.
  if (calibration_due(h) || fly_calibration_due(h)) {
    WfEnvArgs probe_args;
    const WfEnvArgs* probe = ea;
    if (ea && ea->action) {
      probe_args = *ea;
      probe_args.action = nullptr; probe_args.reward = nullptr;
      probe = &probe_args;
    }
    int rc = calibrate_now(h, yaw, power, wspd, wdir, load, probe, true);
    if (rc != WF_OK) return rc;
  }
  // Which farms are solved again in float64 behind this launch: mode 1 every flagged farm, mode 2 all of them, mode 0 only
  // the farms with WF_RISK_THRUST_UNITY — a thrust coefficient above 0.995, where float32 carries no bound at all
  // (wf_kernel_common.h: table_ct) — and that only when the handle's thrust table gets there (nrel_5MW does not: nothing
  // is enqueued for it).
  const int mode = h->types.empty() ? h->resolve_mode : 2;  // (several turbine definitions: the float64 kernels solve every farm)
  int mask = mode == 1 ? 0x1F : 0;
  if (mode == 0) {
    double ct_max = 0.0;
    for (double v : h->tct) ct_max = v > ct_max ? v : ct_max;
    if (ct_max > 0.994) mask = WF_RISK_THRUST_UNITY;
  }
  if (mode != 0 || mask) {
    if (!h->d_res_list) {
      WF_HIP(h, hipMalloc(&h->d_res_list, sizeof(int) * h->cap_env));
      WF_HIP(h, hipMalloc(&h->d_res_count, sizeof(int) * 3));
      WF_HIP(h, hipMalloc(&h->d_flags_raw, sizeof(int) * h->cap_env));
      WF_HIP(h, hipMemsetAsync(h->d_res_count, 0, sizeof(int) * 3, h->stream));
      // the flagged list's length as the float64 kernel last found it (wf_resolve.h: seen_host): unknown yet -> "long"
      if (!h->h_res_seen) WF_HIP(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_res_seen), sizeof(int), hipHostMallocDefault));
      *h->h_res_seen = 0x7fffffff;
      h->res_parity = 0;
    }
  }
  h->res_last = mode != 0 || mask != 0;
  if (mask) h->res_parity ^= 1;  // this step's counter; the step kernels zero the other one for the next step
  h->res_mask = mask;
  int rc = launch_step_f32(h, yaw, power, wspd, wdir, load, ea);
  h->res_mask = 0;
  if (rc != WF_OK) {
    //  a launch that failed before any kernel was enqueued has not zeroed the other counter: both are cleared, so
    // that no later step appends behind a stale count
    if (mask && h->d_res_count) hipMemsetAsync(h->d_res_count, 0, sizeof(int) * 2, h->stream);
    return rc;
  }
  if (mode == 0 && !mask) return WF_OK;
  WfResolveArgs ra{};
  ra.tab64 = h->d_tab64; ra.list = h->d_res_list; ra.count = h->d_res_count + h->res_parity; ra.flags = h->d_flags;
  // a launch with helper waves where the previous one found a farm per CU or so (a hint: the results are the same bits either way)
  ra.seen_host = h->h_res_seen; ra.seen_dev = h->d_res_count + 2;
  ra.wide_hint = h->h_res_seen && *const_cast<volatile int*>(h->h_res_seen) <= (h->n_cu * 5) / 4 ? 1 : 0;
  ra.gx = h->d_gx; ra.gy = h->d_gy; ra.gidx = h->d_gidx;
  ra.geom_stride = (h->wind_count == 1 || h->shared_dir) ? 0 : (size_t)h->N;
  ra.mod = 1;
  if (h->n_groups > 0) {
    ra.farm_group = h->n_layouts > 1 ? h->d_layout_of : (h->series_T > 0 ? h->d_series_start : h->d_bins);  // what build_groups partitioned by
    ra.shift = h->group_shift; ra.mod = h->n_groups;
  }
  ra.ws = h->d_ws; ra.wd = h->d_wd; ra.wind_stride = h->wind_count == 1 ? 0 : 1;
  ra.n_real = h->d_nreal;
  ra.yaw_in = yaw;
  ra.o_power = power; ra.o_ws = wspd; ra.o_wd = wdir; ra.o_load = load;
  if (ea) {
    ra.yaw_state = ea->yaw_state; ra.reward = ea->reward; ra.ws_prev = ea->ws_prev; ra.load_coef = ea->load_coef;
    ra.power_mw = ea->power_mw;
  }
  h->rconsts.N = h->N;
  if (!h->types.empty()) {
    ra.tab64 = h->d_tab64_mt; ra.n_types = (int)h->types.size(); ra.type_of = h->d_type_of; ra.type_consts = h->d_type_consts;
    WF_HIP(h, wfk_launch_resolve_mt(&h->rconsts, &ra, h->B, 1, h->d_flags_raw, h->n_cu, h->stream));
    return WF_OK;
  }
  WF_HIP(h, wfk_launch_resolve(&h->rconsts, &ra, h->B, mode == 2 ? 1 : 0, h->d_flags_raw, h->n_cu, h->stream));
  return WF_OK;
}

}  // namespace wfi

using namespace wfi;

extern "C" {

int wf_get_kernel_info(wf_handle* h, wf_kernel_info* info) {
  if (!h || !info) return WF_E_INVALID;
  if (h->variant < 0) return fail(h, WF_E_INVALID, "wf_set_layout must be called first");
  int G, S; const void* fn;
  wfk_variant(h->variant, &G, &S, &fn);
  // the instantiation the next step would launch: pair table (shared wind), general mirror cores, or default
  if (h->model_dirty && h->N > 0) { int rc = build_consts(h); if (rc != WF_OK) return rc; }
  const bool tab = (h->wind_count == 1 || h->shared_dir || h->n_groups > 0) && h->N <= WF_PAIR_MAX_N && wfk_variant_has_table(h->variant) && h->choice.pair_table != 0;
  const bool veer = h->model.veer != 0.0;
  fn = wfk_variant_fn(h->variant, tab ? (h->wind_count == 1 ? (veer ? 5 : 2) : (veer ? 6 : 3)) : (veer ? 4 : (h->consts.mirror_core_n <= 1 ? 0 : 1)));
  info->pair_table = tab ? 1 : 0;
  info->direction_groups = h->n_groups;
  hipFuncAttributes a;
  WF_ON_DEVICE(h);
  WF_HIP(h, hipFuncGetAttributes(&a, fn));
  info->lanes_per_env = G; info->slots_per_lane = S;
  const int wpb = tab ? wfk_tab_waves() : 4;
  info->envs_per_block = wpb * (64 / G); info->threads_per_block = 64 * wpb;
  info->grid_blocks = h->n_groups > 0 ? (h->n_slots + info->envs_per_block - 1) / info->envs_per_block
                                      : (h->B > 0 ? (h->B + info->envs_per_block - 1) / info->envs_per_block : 0);
  const bool ll_fly = !tab && h->wind_count == h->B && h->B > 1 && h->n_groups == 0 && h->ll_G && wfk_ll_has_fly(ll_fly_G(h), ll_fly_S(h)) && h->choice.fly_one_block != 0 &&
                      h->fly_calib != 2;
  info->one_block_kernel = ((tab && h->ll_G && !(h->tab_slot && h->n_groups == 0)) || ll_fly) ? 1 : 0;
  info->mixed_main_farms = (tab && info->one_block_kernel && !ll_fly && h->n_groups == 0 && h->choice.mixed != 0 && h->mix_main > 0 && h->mix_main < h->B) ? h->mix_main : 0;
  if (info->one_block_kernel) {
    // what serves every wind direction without an x' tie across a block boundary; wf_step_kernel (the variant the
    // fields above would describe) is enqueued behind it for the directions that have one
    const int ll_s = tab ? h->ll_S : ll_fly_S(h), ll_g = tab ? h->ll_G : ll_fly_G(h);
    info->lanes_per_env = ll_g; info->slots_per_lane = ll_s;
    info->envs_per_block = wfk_ll_farms_per_block(ll_g); info->threads_per_block = 256;
    info->grid_blocks = (int)(((h->n_groups > 0 ? (size_t)h->n_slots : (size_t)h->B) + info->envs_per_block - 1) / info->envs_per_block);
    const int ll_blocks = info->mixed_main_farms ? info->mixed_main_farms / info->envs_per_block : info->grid_blocks;
    WF_HIP(h, wfk_ll_func_attributes(ll_g, ll_s, h->wind_count == 1 ? 1 : 0, tab ? 1 : 0, veer ? 1 : 0,
                                     (ll_s == 1 && ll_blocks <= 2 * h->n_cu) ? 1 : 0, &a));
  }
  info->vgprs = a.numRegs;
  info->lds_bytes = (int)a.sharedSizeBytes; info->scratch_bytes = (int)a.localSizeBytes;
  return WF_OK;
}


int wf_set_kernel_choice(wf_handle* h, const wf_kernel_choice* c) {
  if (!h || !c) return WF_E_INVALID;
  if ((c->slot_G > 0) != (c->slot_S > 0) || (c->slot_G > 0 && find_variant(c->slot_G, c->slot_S) < 0))
    return fail(h, WF_E_INVALID, "no wf_step_kernel variant with these lanes per farm x slots per lane");
  if (c->one_block < -1 || c->one_block > 1 || c->pair_table < -1 || c->pair_table > 1 || c->fly_one_block < -1 || c->fly_one_block > 1 ||
      c->far_skip < -1 || c->far_skip > 1 || c->calibrate < -1 || c->calibrate > 1 || c->mixed < -1 || c->mixed > 1)
    return fail(h, WF_E_INVALID, "kernel choice switches must be -1 (automatic), 0 or 1");
  if (c->one_block == 1) {
    const int g = c->ll_G, sl = c->ll_S > 0 ? c->ll_S : 1;
    if (!(((g == 4 || g == 8 || g == 16) && sl == 1) || ((g == 4 || g == 2) && sl == 2)))
      return fail(h, WF_E_INVALID, "wf_step_ll_kernel is instantiated for G x S in {4x1, 8x1, 16x1, 4x2, 2x2}");
  }
  WF_ON_DEVICE(h);
  WF_HIP(h, hipStreamSynchronize(h->stream));
  if (c->far_skip != h->choice.far_skip) h->model_dirty = true;  // WfConsts::far_on / far_k follow the choice (build_consts)
  h->choice = *c;
  if (h->N > 0) {
    const int v = pick_variant(h, h->N, h->B);
    if (v < 0) return fail(h, WF_E_UNSUPPORTED, "no kernel variant for this turbine count");
    apply_kernel_pick(h, h->N, h->B, nullptr);
  }
  // geometry, groups and tables of the current wind were laid out for the previous choice: the wind has to be set again
  h->wind_count = 0; h->shared_dir = false; h->n_groups = 0; h->grid_step = 0.0; h->series_T = 0; h->pair_dirty = true;
  return WF_OK;
}

int wf_calibrate(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  if (h->wind_count == 0) return fail(h, WF_E_INVALID, "wf_set_wind must be called before wf_calibrate");
  WF_ON_DEVICE(h);
  if (h->model_dirty) {
    int rc = build_consts(h);
    if (rc != WF_OK) return rc;
  }
  WF_HIP(h, hipStreamSynchronize(h->stream));
  apply_kernel_pick(h, h->N, h->B, nullptr);  // start over from the rounds model's guess
  if (h->variant < 0) return fail(h, WF_E_UNSUPPORTED, "no kernel variant for this turbine count");
  // probes on scratch buffers: zero yaw in, outputs dropped
  const size_t bn = (size_t)h->B * h->N;
  float* d = nullptr;
  WF_HIP(h, hipMalloc(&d, sizeof(float) * bn * 8));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(float) * bn, h->stream);
  int rc = e == hipSuccess ? calibrate_now(h, d, d + bn, d + 2 * bn, d + 3 * bn, d + 4 * bn, nullptr, false)
                           : fail(h, WF_E_HIP, std::string("wf_calibrate: ") + hipGetErrorString(e));
  hipStreamSynchronize(h->stream);
  hipFree(d);
  return rc;
}

int wf_set_calibration(wf_handle* h, int code, int fly_choice) {
  if (!h) return WF_E_INVALID;
  if (h->N <= 0 || h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_layout and wf_set_batch must be called before wf_set_calibration");
  if (code < -1 || fly_choice < 0 || fly_choice > 2) return fail(h, WF_E_INVALID, "wf_set_calibration: code is -1 or a family code, fly_choice 0, 1 or 2");
  WF_ON_DEVICE(h);
  return apply_saved_calibration(h, code, fly_choice);
}

int wf_get_calibration(wf_handle* h, int* code, float* family_ms) {
  if (!h) return WF_E_INVALID;
  if (code) *code = h->calib_done ? h->calib_code : -1;
  if (family_ms)
    for (int fi = 0; fi < 6; ++fi) family_ms[fi] = fi < kNumFamilies ? h->calib_ms[fi] : 0.0f;
  return WF_OK;
}

int wf_get_mixed_launch(wf_handle* h, int* main_farms, float* mixed_ms) {
  if (!h) return WF_E_INVALID;
  if (main_farms) *main_farms = (h->choice.mixed != 0 && h->mix_main > 0 && h->mix_main < h->B) ? h->mix_main : 0;
  if (mixed_ms) *mixed_ms = h->calib_ms[6];
  return WF_OK;
}

int wf_get_fly_calibration(wf_handle* h, int* choice, float* ms) {
  if (!h) return WF_E_INVALID;
  if (choice) *choice = h->fly_calib;
  if (ms) { ms[0] = h->fly_calib_ms[0]; ms[1] = h->fly_calib_ms[1]; }
  return WF_OK;
}

int wf_get_kernel_choice(wf_handle* h, wf_kernel_choice* c) {
  if (!h || !c) return WF_E_INVALID;
  *c = h->choice;
  return WF_OK;
}

}  // extern "C"
