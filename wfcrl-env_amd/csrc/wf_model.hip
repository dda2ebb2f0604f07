// wf_model.hip — turbine tables and model constants of a handle (include/wfstep.h: wf_default_model, wf_turbine_table,
// wf_set_model); build_consts derives what the kernels take by value (wf_device.h: WfConsts; wf_resolve.h).
#include "wf_handle.h"

namespace {

// ---- nrel_5MW power/thrust table (DATA, replaceable via wf_set_model) ----
const double kCtFrom3[45] = {
    0.99,       0.99,       0.97373036, 0.92826162, 0.89210543, 0.86100905, 0.835423,   0.81237673, 0.79225789,
    0.77584769, 0.7629228,  0.76156073, 0.76261984, 0.76169723, 0.75232027, 0.74026851, 0.72987175, 0.70701647,
    0.54054532, 0.45509459, 0.39343381, 0.34250785, 0.30487242, 0.27164979, 0.24361964, 0.21973831, 0.19918151,
    0.18131868, 0.16537679, 0.15103727, 0.13998636, 0.1289037,  0.11970413, 0.11087113, 0.10339901, 0.09617888,
    0.09009926, 0.08395078, 0.0791188,  0.07448356, 0.07050731, 0.06684119, 0.06345518, 0.06032267, 0.05741999};
const double kCpFrom3SurveyA5[45] = {  // "nrel_5MW_survey_a5": the 8-decimal column (FLORIS v2 example input; 4.969 MW at 12 m/s, 5.116 MW at 25 m/s)
    0.1780851,  0.28907459, 0.34902166, 0.3847278,  0.40605878, 0.4202279,  0.42882274, 0.43387274, 0.43622267,
    0.43684468, 0.43657497, 0.43651053, 0.4365612,  0.43651728, 0.43590309, 0.43467276, 0.43322955, 0.43003137,
    0.37655587, 0.33328466, 0.29700574, 0.26420779, 0.23839379, 0.21459275, 0.19382354, 0.1756635,  0.15970926,
    0.14561785, 0.13287856, 0.12130194, 0.11219941, 0.10311631, 0.09545392, 0.08813781, 0.08186763, 0.07585005,
    0.07071926, 0.06557558, 0.06148104, 0.05755207, 0.05413366, 0.05097969, 0.04806545, 0.04536883, 0.04287006};
// "nrel_5MW_floris3", the default: the six-decimal column of FLORIS 3.x' turbine_library/nrel_5MW.yaml as recollected
// (not reference-held): below rated the values above rounded, from 11.5 m/s the rated-power plateau
// Cp = 5 MW / (1/2 rho A v^3)
const double kCpFrom3Floris3[45] = {
    0.178085, 0.289075, 0.349022, 0.384728, 0.406059, 0.420228, 0.428823, 0.433873, 0.436223,
    0.436845, 0.436575, 0.436511, 0.436561, 0.436517, 0.435903, 0.434673, 0.433230, 0.430466,
    0.378869, 0.335199, 0.297991, 0.266092, 0.238588, 0.214748, 0.193981, 0.175808, 0.159835,
    0.145741, 0.133256, 0.122157, 0.112257, 0.103399, 0.095449, 0.088294, 0.081836, 0.075993,
    0.070692, 0.065875, 0.061484, 0.057476, 0.053809, 0.050447, 0.047358, 0.044518, 0.041900,
};
double g_tab_ws[51], g_tab_ct[51], g_tab_cp[51], g_tab_cp_a5[51];
std::once_flag g_tab_once;
void fill_default_table() {
  int n = 0;
  const double head[3] = {0.0, 2.0, 2.5};
  for (int i = 0; i < 3; ++i) { g_tab_ws[n] = head[i]; g_tab_ct[n] = 0.0; g_tab_cp[n] = g_tab_cp_a5[n] = 0.0; ++n; }
  for (int i = 0; i < 45; ++i) { g_tab_ws[n] = 3.0 + 0.5 * i; g_tab_ct[n] = kCtFrom3[i]; g_tab_cp[n] = kCpFrom3Floris3[i]; g_tab_cp_a5[n] = kCpFrom3SurveyA5[i]; ++n; }
  const double tail[3] = {25.01, 25.02, 50.0};
  for (int i = 0; i < 3; ++i) { g_tab_ws[n] = tail[i]; g_tab_ct[n] = 0.0; g_tab_cp[n] = g_tab_cp_a5[n] = 0.0; ++n; }
}

}  // namespace

namespace wfi {

void init_default_table() { std::call_once(g_tab_once, fill_default_table); }  // concurrent wf_create calls

int build_consts(wf_handle* h) {
  const wf_model_params& m = h->model;
  const int n = (int)h->tws.size();
  if (n < 2 || n > WF_MAX_TABLE - 1) return fail(h, WF_E_INVALID, "power_thrust_table needs 2..63 entries");
  for (int i = 1; i < n; ++i)
    if (!(h->tws[i] > h->tws[i - 1])) return fail(h, WF_E_INVALID, "table wind speeds must be strictly ascending");
  if (!(m.rotor_diameter > 0) || !(m.hub_height > m.rotor_diameter / 2))
    return fail(h, WF_E_INVALID, "need rotor_diameter > 0 and hub_height > rotor radius");

  WfConsts& c = h->consts;
  const double D = m.rotor_diameter, HH = m.hub_height, R = D / 2, eps = m.eps_gain * D, eps2 = eps * eps;
  c.N = h->N;
  c.D = (float)D; c.invD = (float)(1.0 / D);
  const double off[3] = {-D / 4, 0.0, D / 4};
  double shearf[3], uinf = 0;
  for (int k = 0; k < 3; ++k) {
    c.off[k] = (float)off[k];
    c.yoff[k] = (float)(off[k] + m.num_eps);
    shearf[k] = std::pow((HH + off[k]) / HH, m.shear);
    c.shearf[k] = (float)shearf[k];
    uinf += shearf[k] / 3.0;
  }
  c.uinf_f = (float)uinf;
  for (int k = 0; k < 3; ++k) {
    const double z = HH + off[k];
    const double dudz = m.shear * std::pow(1.0 / HH, m.shear) * std::pow(z, m.shear - 1.0);  // per unit ws
    const double lm = m.kappa * z / (1.0 + m.kappa * z / (D / 8.0));
    const double nu = lm * lm * std::fabs(dudz);
    c.decay_a[k] = (float)(4.0 * nu / uinf / eps2);
  }
  c.exp_c = (float)(1.4426950408889634 / eps2);
  const double m_eps = m.num_eps;
  const double q = D / 4.0;
  for (int m = -3; m <= 3; ++m) {
    const double zc = m * q + m_eps, zm = 2.0 * HH + m * q + m_eps;
    c.zc[m + 3] = (float)zc; c.zc2[m + 3] = (float)(zc * zc); c.ez[m + 3] = (float)std::exp(-zc * zc / eps2);
    c.zm[m + 3] = (float)zm; c.zm2[m + 3] = (float)(zm * zm); c.ezm[m + 3] = (float)std::exp(-zm * zm / eps2);
  }
  // class 0 (the rotation vortex seen from the hub row: zc = num_eps): with the target column within a couple of metres
  // of the vortex line, r^2 / eps^2 is so small that 1 - exp(-r^2 / eps^2) cancels in float32 (r^2 = 2e-6 m^2 exactly
  // behind the source: the plain form returns 0) — below yl2_small the kernels take the series of (1 - exp(-s)) / r^2
  c.inv_eps2 = (float)(1.0 / eps2);
  c.m_half_inv_eps4 = (float)(-0.5 / (eps2 * eps2));
  c.yl2_small = (float)(0.005 * eps2);
  // 1 - Ey*ezm with Ey <= 1 rounds to exactly 1.0f once ezm < 2^-25: those classes skip the core factor
  c.mirror_core_n = 0;
  for (int m = 0; m < 7; ++m)
    if (c.ezm[m] >= 2.9e-8f) c.mirror_core_n = m + 1;
  const double hs[3] = {HH + R, HH - R, HH};
  double ks[3] = {0, 0, 0};
  for (int v = 0; v < 3; ++v)
    for (int k = 0; k < 3; ++k) {
      const double zc = HH + off[k] - hs[v] + m_eps;
      for (int j = 0; j < 3; ++j) {  // secondary-steering means on the source's own grid [A.3-2]
        const double yL = off[j] + m_eps;
        const double r = yL * yL + zc * zc;
        ks[v] += zc / r * (1.0 - std::exp(-r / eps2)) / 9.0;
      }
    }
  c.ks_top = (float)ks[0]; c.ks_bot = (float)ks[1]; c.ks_core = (float)ks[2];
  const double vel_top = std::pow((HH + R) / HH, m.shear), vel_bot = std::pow((HH - R) / HH, m.shear);
  const double inv2pi = 1.0 / (2.0 * M_PI);
  c.gam_top = (float)(inv2pi * (M_PI / 8.0) * D * vel_top * uinf);
  c.gam_bot = (float)(inv2pi * (M_PI / 8.0) * D * vel_bot * uinf);
  c.gam_wr = (float)(inv2pi * 0.25 * 2.0 * M_PI * D / m.tsr);
  c.alpha4 = (float)(4.0 * m.alpha); c.beta2 = (float)(2.0 * m.beta);
  c.ka = (float)m.ka; c.kb = (float)m.kb; c.ad = (float)m.ad; c.bd = (float)m.bd; c.dm03 = (float)(0.3 * m.dm);
  c.alpha4_d = (float)(4.0 * m.defl_alpha); c.beta2_d = (float)(2.0 * m.defl_beta);
  c.ka_d = (float)m.defl_ka; c.kb_d = (float)m.defl_kb;
  c.sw_steer = m.enable_secondary_steering ? 2.0f : 0.0f;
  c.sw_tv = m.enable_transverse_velocities ? 1.0f : 0.0f;
  c.e0c1 = (float)(3.0 * std::exp(1.0 / 12.0)); c.e0c2 = (float)(3.0 * std::exp(1.0 / 3.0));
  c.sz0v = (float)(D / (2.0 * std::sqrt(2.0)));
  c.near_c = (float)(m.near_wake_c * D);
  c.kdef = (float)(D * D / 8.0);
  c.ch_c = (float)(m.ch_constant * std::pow(m.ambient_ti, m.ch_initial));
  c.ch_ai = (float)m.ch_ai; c.ch_down = (float)m.ch_downstream;
  c.amb = (float)m.ambient_ti; c.amb2 = (float)(m.ambient_ti * m.ambient_ti);
  c.gch_gain = m.enable_yaw_added_recovery ? (float)m.gch_gain : 0.0f; c.overlap_thr = (float)m.overlap_thresh;
  c.twoD = (float)(2.0 * D); c.fifteenD_d = 15.0 * D;
  c.q_d = D / 4.0;
  c.guard_inv = h->guard_rel > 0.0 ? (float)(1.0 / h->guard_rel) : 1125899906842624.0f;
  c.inv_overlap_thr = (float)(1.0 / m.overlap_thresh);
  // far-source / far-pair skip of the one-block kernel: its bounds assume wakes that only widen downstream and near-wake
  // lengths that fall with TI — every physical parameter set; anything else runs without the skip
  c.far_on = (h->choice.far_skip != 0 && m.ka >= 0.0 && m.kb >= 0.0 && m.defl_ka >= 0.0 && m.defl_kb >= 0.0 && m.alpha >= 0.0 &&
              m.beta >= 0.0 && m.defl_alpha >= 0.0 && m.defl_beta >= 0.0) ? 1 : 0;
  c.far_k = c.far_on ? 6.12f : 1.0e30f;
  c.ct_kappa = 5.0f;     // nrel_5MW: 5.9 on the cut-in ramp (2.5-3 m/s), 143 on the cut-out drop, <= 4.0 everywhere else
  c.knee_kappa = 30.0f;  // 30 x (wind-speed error ~3e-6) ~ 1e-4 of max(P, 1 kW)
  {  // wind veer: the rotated Gaussian of the deficit [FLORIS gauss.py rCalt]
    const double vr = m.veer * M_PI / 180.0;
    c.veer_on = m.veer != 0.0 ? 1 : 0;
    c.cos_veer = (float)std::cos(vr);
    c.veer_c2 = (float)(std::cos(vr) * std::cos(vr)); c.veer_s2 = (float)(std::sin(vr) * std::sin(vr));
    c.veer_bq = (float)(std::sin(2.0 * vr) * D / 4.0);
    c.kdef_sy0v = (float)((D * D / 8.0) / ((D / (2.0 * std::sqrt(2.0))) * std::cos(vr)));
  }
  c.rho = (float)m.ref_density; c.pw = (float)(m.pP / 3.0);
  c.dens_f = (float)std::cbrt(m.air_density / m.ref_density);

  // table + bucket index
  WfTables t;
  const double area = M_PI * R * R;
  std::vector<double> pwv(n);
  for (int i = 0; i < n; ++i) pwv[i] = 0.5 * area * h->tcp[i] * m.gen_eff * h->tws[i] * h->tws[i] * h->tws[i];
  for (int i = 0; i < WF_TABLE_PAD; ++i) {
    const bool in = i < n;
    t.knot[i] = in ? (float)h->tws[i] : 3.0e38f;
    t.ct[i] = in ? (float)h->tct[i] : 0.f;
    t.pw[i] = in ? (float)pwv[i] : 0.f;
    const bool seg = i + 1 < n;
    t.ct_slope[i] = seg ? (float)((h->tct[i + 1] - h->tct[i]) / (h->tws[i + 1] - h->tws[i])) : 0.f;
    t.pw_slope[i] = seg ? (float)((pwv[i + 1] - pwv[i]) / (h->tws[i + 1] - h->tws[i])) : 0.f;
  }
  const double x0 = h->tws[0], x1 = h->tws[n - 1];
  const double bh = (x1 - x0) / WF_BUCKETS;
  c.n_table = n; c.bucket_x0 = (float)x0; c.bucket_h_inv = (float)(1.0 / bh);
  // bucket[b] = last knot <= start of bucket b-1; the kernel probes forward from there.  One bucket of
  // slack on either side absorbs float rounding of the bucket index computed on the device.
  int max_probe = 1;
  for (int b = 0; b < WF_BUCKETS; ++b) {
    const double lo = x0 + (b - 1) * bh, hi = x0 + (b + 2) * bh;
    int j = 0;
    while (j + 1 < n && h->tws[j + 1] <= lo) ++j;
    t.bucket[b] = (unsigned char)j;
    int k = j;
    while (k + 1 < n && h->tws[k + 1] <= hi) ++k;
    if (k - j > max_probe) max_probe = k - j;
  }
  c.max_probe = max_probe;
  // the float64 solve (wf_resolve.hip): the same model in double
  {
    WfResolveConsts& r = h->rconsts;
    r.N = h->N; r.n_table = n;
    r.sw_steer = m.enable_secondary_steering ? 1 : 0; r.sw_yar = m.enable_yaw_added_recovery ? 1 : 0;
    r.sw_tv = m.enable_transverse_velocities ? 1 : 0;
    r.D = D; r.HH = HH; r.TSR = m.tsr; r.amb = m.ambient_ti; r.eps2 = eps2; r.num_eps = m.num_eps; r.sqrt2 = std::sqrt(2.0);
    r.inv_D = 1.0 / D; r.inv_TSR = 1.0 / m.tsr; r.inv_eps2 = 1.0 / eps2;
    r.uinf1 = 0.0;
    for (int k = 0; k < 3; ++k) {
      r.off[k] = off[k];
      r.shearf[k] = shearf[k];
      r.uinf1 += shearf[k];
      const double z = HH + off[k];
      const double dudz1 = m.shear * std::pow(1.0 / HH, m.shear) * std::pow(z, m.shear - 1.0);
      const double lm = m.kappa * z / (1.0 + m.kappa * z / (D / 8.0));
      r.nu1[k] = lm * lm * std::fabs(dudz1);
    }
    r.uinf1 /= 3.0;
    r.vel_top = vel_top; r.vel_bot = vel_bot;
    double kk[3] = {0, 0, 0};
    for (int v = 0; v < 3; ++v)
      for (int k = 0; k < 3; ++k) {
        const double zc = HH + off[k] - hs[v] + m_eps;
        for (int j = 0; j < 3; ++j) {
          const double yL = off[j] + m_eps;
          const double rr = yL * yL + zc * zc;
          kk[v] += zc / (2.0 * M_PI * rr) * (1.0 - std::exp(-rr / eps2)) / 9.0;
        }
      }
    r.mirror_core = 0;
    for (int mm = -3; mm <= 3; ++mm) {
      const double zc = mm * q + m_eps, zm = 2.0 * HH + mm * q + m_eps;
      r.zc[mm + 3] = zc; r.zc2[mm + 3] = zc * zc; r.ezc[mm + 3] = std::exp(-zc * zc / eps2);
      r.zm7[mm + 3] = zm; r.zm2[mm + 3] = zm * zm; r.ezm7[mm + 3] = std::exp(-zm * zm / eps2);
      if (r.ezm7[mm + 3] >= 1.0e-17) r.mirror_core = 1;
    }
    r.k_top = kk[0]; r.k_bot = kk[1]; r.k_core = kk[2];
    r.alpha = m.alpha; r.beta = m.beta; r.ka = m.ka; r.kb = m.kb; r.ad = m.ad; r.bd = m.bd; r.dm = m.dm;
    r.defl_alpha = m.defl_alpha; r.defl_beta = m.defl_beta; r.defl_ka = m.defl_ka; r.defl_kb = m.defl_kb;
    r.e0c1 = 3.0 * std::exp(1.0 / 12.0); r.e0c2 = 3.0 * std::exp(1.0 / 3.0);
    r.near_c = m.near_wake_c * D;
    r.ch_constant = m.ch_constant; r.ch_ai = m.ch_ai; r.ch_amb_pow = std::pow(m.ambient_ti, m.ch_initial); r.ch_down = m.ch_downstream;
    r.gch_gain = m.gch_gain; r.overlap_thr = m.overlap_thresh;
    const double vr = m.veer * M_PI / 180.0;
    r.cos_veer = std::cos(vr); r.cos2_veer = std::cos(vr) * std::cos(vr); r.sin2_veer = std::sin(vr) * std::sin(vr);
    r.sin_2veer = std::sin(2.0 * vr);
    r.rho_ref = m.ref_density; r.dens_cbrt = std::pow(m.air_density / m.ref_density, 1.0 / 3.0); r.pP3 = m.pP / 3.0;
    std::vector<double> t64(3 * WF_TABLE_PAD, 0.0);
    for (int i = 0; i < n; ++i) { t64[i] = h->tws[i]; t64[WF_TABLE_PAD + i] = h->tct[i]; t64[2 * WF_TABLE_PAD + i] = pwv[i]; }
    if (!h->d_tab64) {
      hipError_t e64 = hipMalloc(&h->d_tab64, sizeof(double) * 3 * WF_TABLE_PAD);
      if (e64 != hipSuccess) return fail(h, WF_E_HIP, std::string("table64 alloc: ") + hipGetErrorString(e64));
    }
    // on the handle's stream, like d_tab below: a blocking copy on the null stream is NOT ordered behind a float64 solve
    // still running on h->stream (non-blocking or adopted from torch), which stages this very table into LDS; the
    // source is pageable host memory, so the call returns once the bytes are staged and the sync below covers the rest
    hipError_t e64 = hipStreamSynchronize(h->stream);
    if (e64 == hipSuccess) e64 = hipMemcpyAsync(h->d_tab64, t64.data(), sizeof(double) * t64.size(), hipMemcpyHostToDevice, h->stream);
    if (e64 == hipSuccess) e64 = hipStreamSynchronize(h->stream);
    if (e64 != hipSuccess) return fail(h, WF_E_HIP, std::string("table64 upload: ") + hipGetErrorString(e64));
  }
  hipError_t e = hipMemcpyAsync(h->d_tab, &t, sizeof(WfTables), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) return fail(h, WF_E_HIP, std::string("table upload: ") + hipGetErrorString(e));
  if (!h->types.empty()) {  // several turbine definitions (wf_set_turbine_types): their float64 tables and scalars
    const int nt = (int)h->types.size();
    if ((int)h->type_of.size() != h->N)
      return fail(h, WF_E_INVALID, "wf_set_turbine_types was given for another turbine count than the layout's: set the definitions again");
    std::vector<double> tabs((size_t)nt * 3 * WF_TABLE_PAD, 0.0), tc((size_t)nt * (1 + WF_TYPE_CONSTS));
    for (int k = 0; k < nt; ++k) {
      const wf_handle::TurbineType& ty = h->types[k];
      double* tb = tabs.data() + (size_t)k * 3 * WF_TABLE_PAD;
      for (size_t i = 0; i < ty.ws.size(); ++i) {
        tb[i] = ty.ws[i]; tb[WF_TABLE_PAD + i] = ty.ct[i];
        tb[2 * WF_TABLE_PAD + i] = 0.5 * area * ty.cp[i] * ty.gen_eff * ty.ws[i] * ty.ws[i] * ty.ws[i];  // [A.4], as pwv above
      }
      double* c1 = tc.data() + (size_t)k * (1 + WF_TYPE_CONSTS);
      c1[0] = (double)ty.ws.size(); c1[1] = 1.0 / ty.tsr; c1[2] = ty.pP / 3.0;
      c1[3] = std::pow(m.air_density / ty.ref_density, 1.0 / 3.0); c1[4] = ty.ref_density;
    }
    hipError_t et = hipSuccess;
    if (!h->d_tab64_mt) et = hipMalloc(&h->d_tab64_mt, sizeof(double) * WF_MAX_TYPES * 3 * WF_TABLE_PAD);
    if (et == hipSuccess && !h->d_type_consts) et = hipMalloc(&h->d_type_consts, sizeof(double) * WF_MAX_TYPES * (1 + WF_TYPE_CONSTS));
    if (et == hipSuccess) { hipFree(h->d_type_of); h->d_type_of = nullptr; et = hipMalloc(&h->d_type_of, sizeof(int) * (size_t)h->N); }
    if (et == hipSuccess) et = hipMemcpyAsync(h->d_tab64_mt, tabs.data(), sizeof(double) * tabs.size(), hipMemcpyHostToDevice, h->stream);
    if (et == hipSuccess) et = hipMemcpyAsync(h->d_type_consts, tc.data(), sizeof(double) * tc.size(), hipMemcpyHostToDevice, h->stream);
    if (et == hipSuccess) et = hipMemcpyAsync(h->d_type_of, h->type_of.data(), sizeof(int) * (size_t)h->N, hipMemcpyHostToDevice, h->stream);
    if (et == hipSuccess) et = hipStreamSynchronize(h->stream);
    if (et != hipSuccess) return fail(h, WF_E_HIP, std::string("turbine definitions upload: ") + hipGetErrorString(et));
  }
  h->model_dirty = false;
  return WF_OK;
}

}  // namespace wfi

using namespace wfi;

extern "C" {

int wf_default_model(wf_model_params* p) {
  if (!p) return WF_E_INVALID;
  init_default_table();
  p->air_density = 1.225; p->ambient_ti = 0.06; p->shear = 0.12; p->veer = 0.0;
  p->rotor_diameter = 126.0; p->hub_height = 90.0; p->tsr = 8.0; p->pP = 1.88; p->pT = 1.88;
  p->gen_eff = 1.0; p->ref_density = 1.225;
  p->alpha = 0.58; p->beta = 0.077; p->ka = 0.38; p->kb = 0.004; p->ad = 0.0; p->bd = 0.0; p->dm = 1.0;
  p->ch_initial = 0.1; p->ch_constant = 0.5; p->ch_ai = 0.8; p->ch_downstream = -0.32;
  p->eps_gain = 0.2; p->num_eps = 0.001; p->kappa = 0.41; p->gch_gain = 2.0; p->overlap_thresh = 0.05;
  p->near_wake_c = 0.501;
  p->defl_alpha = p->alpha; p->defl_beta = p->beta; p->defl_ka = p->ka; p->defl_kb = p->kb;
  p->enable_secondary_steering = p->enable_yaw_added_recovery = p->enable_transverse_velocities = 1;
  p->n_table = 51; p->table_ws = g_tab_ws; p->table_ct = g_tab_ct; p->table_cp = g_tab_cp;
  return WF_OK;
}

int wf_turbine_table(const char* name, int* n, const double** ws, const double** ct, const double** cp) {
  if (!name || !n || !ws || !ct || !cp) return WF_E_INVALID;
  init_default_table();
  const bool f3 = std::strcmp(name, "nrel_5MW_floris3") == 0 || std::strcmp(name, "nrel_5MW") == 0;
  if (!f3 && std::strcmp(name, "nrel_5MW_survey_a5") != 0) return WF_E_INVALID;
  *n = 51; *ws = g_tab_ws; *ct = g_tab_ct; *cp = f3 ? g_tab_cp : g_tab_cp_a5;
  return WF_OK;
}
int wf_set_model(wf_handle* h, const wf_model_params* p) {
  if (!h || !p) return WF_E_INVALID;
  if (p->n_table < 2 || p->n_table > WF_MAX_TABLE - 1 || !p->table_ws || !p->table_ct || !p->table_cp)
    return fail(h, WF_E_INVALID, "power_thrust_table needs 2..63 entries");
  if (!std::isfinite(p->veer)) return fail(h, WF_E_INVALID, "wind_veer must be finite");
  {
    const struct { double v; const char* name; } positive[] = {
        {p->air_density, "air_density"}, {p->ambient_ti, "turbulence_intensity"}, {p->rotor_diameter, "rotor_diameter"},
        {p->hub_height, "hub_height"}, {p->tsr, "TSR"}, {p->pP, "pP"}, {p->gen_eff, "generator_efficiency"},
        {p->ref_density, "ref_density_cp_ct"}, {p->ka * p->ambient_ti + p->kb, "ka*TI + kb"},
        {p->defl_ka * p->ambient_ti + p->defl_kb, "deflection ka*TI + kb"}, {p->defl_alpha, "deflection alpha"},
        {p->alpha, "alpha"}, {p->eps_gain, "eps_gain"}, {p->num_eps, "num_eps"}, {p->kappa, "kappa"},
        {p->ch_constant, "crespo_hernandez.constant"}, {p->overlap_thresh, "overlap_thresh"}};
    for (const auto& q : positive)
      if (!(q.v > 0.0) || !std::isfinite(q.v))
        return fail(h, WF_E_INVALID, std::string("model parameter must be finite and > 0: ") + q.name);
    const double finite[] = {p->defl_beta, p->shear, p->beta, p->ad, p->bd, p->dm, p->ch_initial, p->ch_ai, p->ch_downstream, p->gch_gain,
                             p->overlap_thresh, p->near_wake_c, p->pT};
    for (double v : finite)
      if (!std::isfinite(v)) return fail(h, WF_E_INVALID, "model parameters must be finite");
    if (!(p->hub_height > 0.5 * p->rotor_diameter))
      return fail(h, WF_E_INVALID, "hub_height must exceed the rotor radius (blade tip above ground)");
    for (int i = 0; i < p->n_table; ++i) {
      if (!std::isfinite(p->table_ws[i]) || !std::isfinite(p->table_ct[i]) || !std::isfinite(p->table_cp[i]) ||
          p->table_ct[i] < 0.0 || p->table_cp[i] < 0.0)
        return fail(h, WF_E_INVALID, "power_thrust_table entries must be finite and non-negative");
      if (i && !(p->table_ws[i] > p->table_ws[i - 1]))
        return fail(h, WF_E_INVALID, "table wind speeds must be strictly ascending");
    }
  }
  const bool veer_toggled = (h->model.veer != 0.0) != (p->veer != 0.0);
  h->model = *p;
  h->tws.assign(p->table_ws, p->table_ws + p->n_table);
  h->tct.assign(p->table_ct, p->table_ct + p->n_table);
  h->tcp.assign(p->table_cp, p->table_cp + p->n_table);
  h->model.table_ws = h->model.table_ct = h->model.table_cp = nullptr;
  h->model_dirty = true;
  h->pair_dirty = true;
  if (veer_toggled && h->N > 0) {  // another kernel family serves the handle (wf_dispatch.hip: veer models run wf_step_kernel's
    WF_ON_DEVICE(h);               // VEER instantiation, on the fly): tables, groups and the wind were laid out for the old one
    WF_HIP(h, hipStreamSynchronize(h->stream));
    apply_kernel_pick(h, h->N, h->B, nullptr);
    h->wind_count = 0; h->shared_dir = false; h->n_groups = 0; h->grid_step = 0.0; h->series_T = 0;
  }
  return WF_OK;
}


static_assert(WF_MAX_TYPES == WF_MAX_TURBINE_TYPES, "wf_resolve.h and include/wfstep.h disagree");

int wf_set_turbine_types(wf_handle* h, int n_types, const wf_turbine_def* defs, const int* type_of) {
  if (!h) return WF_E_INVALID;
  if (n_types == 0) {  // back to the one definition of the model
    h->types.clear(); h->type_of.clear();
    return WF_OK;
  }
  if (h->N <= 0) return fail(h, WF_E_INVALID, "wf_set_layout must be called first (the definitions are given per turbine)");
  if (n_types < 1 || n_types > WF_MAX_TURBINE_TYPES || !defs || !type_of)
    return fail(h, WF_E_INVALID, "turbine definitions: 1..WF_MAX_TURBINE_TYPES of them and a definition index per turbine");
  std::vector<wf_handle::TurbineType> types((size_t)n_types);
  for (int k = 0; k < n_types; ++k) {
    const wf_turbine_def& d = defs[k];
    if (d.n_table < 2 || d.n_table > WF_MAX_TABLE - 1 || !d.table_ws || !d.table_ct || !d.table_cp)
      return fail(h, WF_E_INVALID, "turbine definition: power_thrust_table needs 2..63 entries");
    const double positive[] = {d.tsr, d.pP, d.gen_eff, d.ref_density};
    for (double v : positive)
      if (!(v > 0.0) || !std::isfinite(v)) return fail(h, WF_E_INVALID, "turbine definition: TSR, pP, generator_efficiency and ref_density_cp_ct must be finite and > 0");
    for (int i = 0; i < d.n_table; ++i) {
      if (!std::isfinite(d.table_ws[i]) || !std::isfinite(d.table_ct[i]) || !std::isfinite(d.table_cp[i]) || d.table_ct[i] < 0.0 ||
          d.table_cp[i] < 0.0)
        return fail(h, WF_E_INVALID, "turbine definition: power_thrust_table entries must be finite and non-negative");
      if (i && !(d.table_ws[i] > d.table_ws[i - 1])) return fail(h, WF_E_INVALID, "turbine definition: table wind speeds must be strictly ascending");
    }
    wf_handle::TurbineType& ty = types[(size_t)k];
    ty.ws.assign(d.table_ws, d.table_ws + d.n_table);
    ty.ct.assign(d.table_ct, d.table_ct + d.n_table);
    ty.cp.assign(d.table_cp, d.table_cp + d.n_table);
    ty.tsr = d.tsr; ty.pP = d.pP; ty.gen_eff = d.gen_eff; ty.ref_density = d.ref_density;
  }
  for (int t = 0; t < h->N; ++t)
    if (type_of[t] < 0 || type_of[t] >= n_types) return fail(h, WF_E_INVALID, "turbine definitions: a turbine's index is outside 0..n_types-1");
  h->types.swap(types);
  h->type_of.assign(type_of, type_of + h->N);
  h->model_dirty = true;  // (uploaded with the model constants before the next step)
  return WF_OK;
}

int wf_get_turbine_types(wf_handle* h, int* n_types) {
  if (!h || !n_types) return WF_E_INVALID;
  *n_types = (int)h->types.size();
  return WF_OK;
}

}  // extern "C"
