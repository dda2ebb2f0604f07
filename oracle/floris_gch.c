/* TEST INFRASTRUCTURE — CPU oracle (plain C, float64) for the wind-farm step hot path.
 *
 * Checker only: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this; the product (wfcrl-env_amd/) never links or calls it.
 *
 * Restates one farm step of the reference's FLORIS backend
 *   reference wfcrl/interface.py:557-586 (update_command), 622-623 (avg_powers),
 *   629-637 (local_load_proxies), 639-648 (local_wind_measurements), 663-671 (update_wind)
 * whose arithmetic is the third-party FLORIS==3.5 (reference requirements.txt:8, not vendored)
 * sequential Gauss-Curl-Hybrid solver as configured by
 *   reference wfcrl/simulators/floris/inputs/template/case.yaml:14-89.
 * The algorithm follows SURVEY.md Appendix A; section tags [A.x] are cited per block.
 *
 * Parity pin: reproduces the reference's only known-answer vector
 * (reference examples/demo.ipynb:137-139, yaw = 0) and matches oracle/floris_gch_numpy.py to
 * ~1e-12.  PARITY UNPINNED for yaw != 0, powers and loads (the reference holds no such data).
 *
 * Formula order deliberately mirrors the NumPy restatement; the only shortcut is skipping
 * (source, target) pairs with dx < 0, for which every contribution is masked to zero [A.3-4/6/8].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  double air_density, ambient_ti, shear, veer;
  double D, HH, TSR, pP, pT, gen_eff, ref_density;
  double alpha, beta, ka, kb, ad, bd, dm;
  double ch_initial, ch_constant, ch_ai, ch_downstream;
  double eps_gain, num_eps, kappa, gch_gain, overlap_thresh, near_wake_c;
  double defl_alpha, defl_beta, defl_ka, defl_kb; /* gauss deflection model's own set (case.yaml:52-59) */
  int n_table;
  const double* table_ws;
  const double* table_ct;
  const double* table_pow; /* 1/2 A Cp eta ws^3 (W per unit density) */
  int enable_secondary_steering, enable_yaw_added_recovery, enable_transverse_velocities; /* case.yaml:46-50 */
  /* several turbine definitions per farm (farm.turbine_type of case.yaml:27-28 is a list; FLORIS 3.5 evaluates the
   * thrust / power tables, TSR, pP and ref_density_cp_ct per turbine through turbine_type_map): n_types > 0 -> turbine o
   * (caller's order) evaluates with types[type_of[o]] instead of the fields above; the definitions share the rotor (D, HH).
   * As floris_gch_numpy.py: ModelParams.turbine_defs. */
  int n_types;
  const struct wfo_type* types;
  const int* type_of;
} wfo_params;

typedef struct wfo_type {
  double TSR, pP, ref_density;
  int n_table;
  const double* table_ws;
  const double* table_ct;
  const double* table_pow;
} wfo_type;

#define DEG2RAD (M_PI / 180.0)
static inline double cosd(double a) { return cos(a * DEG2RAD); }
static inline double sind(double a) { return sin(a * DEG2RAD); }

/* scipy interp1d(linear, bounds_error=False, fill_value=(lo,hi)) via the np.interp formula */
static double interp_fill(double xq, int n, const double* xs, const double* ys, double lo, double hi) {
  if (xq < xs[0]) return lo;
  if (xq > xs[n - 1]) return hi;
  if (xq == xs[n - 1]) return ys[n - 1];
  int j = 0;
  while (j < n - 2 && xq >= xs[j + 1]) ++j;
  double slope = (ys[j + 1] - ys[j]) / (xs[j + 1] - xs[j]);
  return slope * (xq - xs[j]) + ys[j];
}

static double pymod(double a, double m) { /* Python's % for positive m */
  double r = fmod(a, m);
  if (r != 0.0 && ((r < 0.0) != (m < 0.0))) r += m;
  return r;
}

typedef struct {
  double U[9], V[9], W[9], wake[9], TI[9];
} tstate;

/* one REAL vortex at one point: adds (v,w) scaled by sign (+1 real, -1 ground mirror) */
static inline void vortex(double G, double yL, double zc, double eps, double decay, double sign, double* v, double* w) {
  double r = yL * yL + zc * zc;
  double core = 1.0 - exp(-r / (eps * eps));
  double k = G / (2.0 * M_PI * r) * core * decay;
  *v += sign * (k * zc);
  *w += sign * (-k * yL);
}

/* margin (may be NULL): out, the smallest relative distance |deficit*Uinit - overlap_thresh| / overlap_thresh over all
 * (source, target, grid point) triples whose overlap count can matter (target within the 15 D reach and, for at least
 * one grid column, inside the 2 D lateral gate) — how close this farm comes to the one state-dependent discontinuity
 * of the model [A.3-8].  tie_reverse: order of turbines with EQUAL x' (FLORIS' np.argsort default is not a stable
 * sort, so the order of exact ties is implementation-defined there): 0 = by ascending original index (stable),
 * 1 = by descending original index. */
static void farm_step_one(const wfo_params* p, int N, const double* x, const double* y, double ws, double wd,
                          const double* yaw, double* power, double* wind_speed, double* wind_dir, double* load,
                          double* work /* 3N doubles */, double* vwbuf /* 18N doubles */, int* order, tstate* st,
                          double* margin, int tie_reverse) {
  double min_margin = 1.0e300;
  const double D = p->D, HH = p->HH, R = p->D / 2.0;
  wd = pymod(wd, 360.0); /* reference interface.py:664 */

  /* ---- geometry [A.1] */
  double* xs = work;
  double* ys = work + N;
  double* yaws = work + 2 * N;
  {
    double dev = pymod(pymod(wd - 270.0, 360.0) + 360.0, 360.0);
    double xmin = x[0], xmax = x[0], ymin = y[0], ymax = y[0];
    for (int t = 1; t < N; ++t) {
      if (x[t] < xmin) xmin = x[t];
      if (x[t] > xmax) xmax = x[t];
      if (y[t] < ymin) ymin = y[t];
      if (y[t] > ymax) ymax = y[t];
    }
    double xc = (xmin + xmax) / 2.0, yc = (ymin + ymax) / 2.0;
    double c = cosd(dev), s = sind(dev);
    for (int t = 0; t < N; ++t) {
      double xo = x[t] - xc, yo = y[t] - yc;
      xs[t] = xo * c - yo * s + xc; /* unsorted for now */
      ys[t] = xo * s + yo * c + yc;
      order[t] = t;
    }
    /* stable insertion sort of indices by x' [A.1-2]; tie_reverse: start from descending indices, so that equal x'
     * keep the descending order */
    if (tie_reverse)
      for (int t = 0; t < N; ++t) order[t] = N - 1 - t;
    for (int a = 1; a < N; ++a) {
      int k = order[a];
      int b = a - 1;
      while (b >= 0 && xs[order[b]] > xs[k]) {
        order[b + 1] = order[b];
        --b;
      }
      order[b + 1] = k;
    }
    /* gather into sorted order (use yaws as scratch for x) */
    double* tmp = yaws;
    for (int t = 0; t < N; ++t) tmp[t] = xs[order[t]];
    memcpy(xs, tmp, sizeof(double) * N);
    for (int t = 0; t < N; ++t) tmp[t] = ys[order[t]];
    memcpy(ys, tmp, sizeof(double) * N);
    for (int t = 0; t < N; ++t) yaws[t] = yaw[order[t]];
  }
  /* np.linspace(-D/4, D/4, 3) */
  const double off[3] = {-D / 4.0, 0.0, D / 4.0};

  /* ---- inflow [A.2] */
  double Uinit[3], dUdz[3], Z[3];
  double Uinf = 0.0;
  for (int k = 0; k < 3; ++k) {
    Z[k] = HH + off[k];
    Uinit[k] = ws * pow(Z[k] / HH, p->shear);
    dUdz[k] = ws * p->shear * pow(1.0 / HH, p->shear) * pow(Z[k], p->shear - 1.0);
  }
  { /* mean over all turbines and grid points, summed the way numpy does not matter at 1e-16 */
    double s = 0.0;
    for (int k = 0; k < 3; ++k) s += Uinit[k];
    Uinf = s / 3.0;
  }
  for (int t = 0; t < N; ++t)
    for (int q = 0; q < 9; ++q) {
      st[t].U[q] = Uinit[q % 3];
      st[t].V[q] = 0.0;
      st[t].W[q] = 0.0;
      st[t].wake[q] = 0.0;
      st[t].TI[q] = p->ambient_ti;
    }

  const double eps = p->eps_gain * D;
  const double vel_top = pow((HH + R) / HH, p->shear);
  const double vel_bot = pow((HH - R) / HH, p->shear);
  const double sqrt2 = sqrt(2.0);
  const double hs[3] = {HH + R, HH - R, HH};

  for (int i = 0; i < N; ++i) {
    const double x_i = xs[i], y_i = ys[i], g = yaws[i], cg = cosd(g);
    tstate* S = &st[i];

    /* 1. Ct / induction [A.3-1] */
    double m3 = 0.0;
    for (int q = 0; q < 9; ++q) m3 += S->U[q] * S->U[q] * S->U[q];
    const double ubar = cbrt(m3 / 9.0);
    wfo_type ty_i = {p->TSR, p->pP, p->ref_density, p->n_table, p->table_ws, p->table_ct, p->table_pow};
    if (p->n_types > 0) ty_i = p->types[p->type_of[order[i]]]; /* the source's own definition */
    double ct_tab = interp_fill(ubar, ty_i.n_table, ty_i.table_ws, ty_i.table_ct, 0.0001, 0.9999);
    if (ct_tab < 0.0001) ct_tab = 0.0001;
    if (ct_tab > 0.9999) ct_tab = 0.9999;
    const double ct = ct_tab * cg;
    const double a = 0.5 / cg * (1.0 - sqrt(1.0 - ct * cg));
    const double G_wr = 0.25 * 2.0 * M_PI * D * (a - a * a) * ubar / ty_i.TSR;
    const double gam_top = (M_PI / 8.0) * D * vel_top * Uinf * ct;
    const double gam_bot = (M_PI / 8.0) * D * vel_bot * Uinf * ct;

    /* 2. secondary steering [A.3-2] */
    double v_top = 0.0, v_bot = 0.0, v_core = 0.0, Vmean = 0.0, dummy = 0.0;
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) {
        double yL = off[j] + p->num_eps; /* (Y - y_i) on the source's own grid */
        vortex(gam_top, yL, Z[k] - (HH + R) + p->num_eps, eps, 1.0, 1.0, &v_top, &dummy);
        vortex(-gam_bot, yL, Z[k] - (HH - R) + p->num_eps, eps, 1.0, 1.0, &v_bot, &dummy);
        vortex(G_wr, yL, Z[k] - HH + p->num_eps, eps, 1.0, 1.0, &v_core, &dummy);
        Vmean += S->V[j * 3 + k];
      }
    v_top /= 9.0; v_bot /= 9.0; v_core /= 9.0; Vmean /= 9.0;
    double val = 2.0 * (Vmean - v_core) / (v_top + v_bot);
    if (val < -1.0) val = -1.0;
    if (val > 1.0) val = 1.0;
    const double g_eff = p->enable_secondary_steering ? g + (0.5 * asin(val)) / DEG2RAD : g;

    /* source-side constants of the deflection model [A.3-3] (TI BEFORE mixing), per grid point q */
    const double gd = -g_eff, cgd = cosd(gd);
    const double s_cc = sqrt(1.0 - ct * cgd), s_c = sqrt(1.0 - ct);
    double TIpre[9];
    memcpy(TIpre, S->TI, sizeof(TIpre));
    const double th0 = p->dm * (0.3 * (gd * DEG2RAD) / cgd) * (1.0 - s_cc);

    /* transverse-velocity circulations [A.3-4] (commanded yaw) */
    const double sc = sind(g) * cg;
    const double G_t = sc * gam_top, G_b = -sc * gam_bot;
    const double Gs[3] = {G_t, G_b, G_wr};

    /* 4. the source's own transverse contribution is needed by step 5 before the deficit of step 6,
     * so the transverse pass over all targets runs first and is buffered in vw/ww. */
    double (*vw)[9] = (double (*)[9])vwbuf;
    double (*ww)[9] = vw + N;
    for (int t = 0; t < N; ++t) {
      const double dx = xs[t] - x_i;
      for (int q = 0; q < 9; ++q) vw[t][q] = ww[t][q] = 0.0;
      if (dx < 0.0 || !p->enable_transverse_velocities) continue;
      for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k) {
          const double z = Z[k];
          const double yL = (ys[t] + off[j] - y_i) + p->num_eps;
          const double lm = p->kappa * z / (1.0 + p->kappa * z / (D / 8.0));
          const double nu = lm * lm * fabs(dUdz[k]);
          const double decay = eps * eps / (4.0 * nu * dx / Uinf + eps * eps);
          double v = 0.0, w = 0.0;
          for (int c = 0; c < 3; ++c) {
            vortex(Gs[c], yL, z - hs[c] + p->num_eps, eps, decay, 1.0, &v, &w);
            vortex(Gs[c], yL, z + hs[c] + p->num_eps, eps, decay, -1.0, &v, &w);
          }
          if (w < 0.0) w = 0.0; /* quirk (5) [A.6] */
          vw[t][j * 3 + k] = v;
          ww[t][j * 3 + k] = w;
        }
    }

    /* 5. yaw-added recovery [A.3-5] */
    {
      const double I = S->TI[0];
      const double k_tke = (ubar * I) * (ubar * I) / (2.0 / 3.0);
      double vbar = 0.0, wbar = 0.0;
      for (int q = 0; q < 9; ++q) {
        vbar += S->V[q] + vw[i][q];
        wbar += S->W[q] + ww[i][q];
      }
      vbar /= 9.0; wbar /= 9.0;
      const double I_tot = sqrt((2.0 / 3.0) * 0.5 * (2.0 * k_tke + vbar * vbar + wbar * wbar)) / ubar;
      const double I_mix = I_tot - I;
      if (p->enable_yaw_added_recovery)
        for (int q = 0; q < 9; ++q) S->TI[q] += p->gch_gain * I_mix;
    }
    double TIpost[9];
    memcpy(TIpost, S->TI, sizeof(TIpost));

    const double gv = -g, cgv = cosd(gv);
    const double ch_pref = p->ch_constant * pow(a, p->ch_ai) * pow(p->ambient_ti, p->ch_initial);

    for (int t = 0; t < N; ++t) {
      const double X = xs[t];
      const double dx = X - x_i;
      tstate* T = &st[t];
      if (dx >= 0.0) {
        double defU[9];
        int cnt = 0;
        for (int j = 0; j < 3; ++j)
          for (int k = 0; k < 3; ++k) {
            const int q = j * 3 + k;
            const double Ui = Uinit[k];
            const double Y = ys[t] + off[j], Zk = Z[k];
            /* 3. deflection [A.3-3] */
            double delta;
            {
              const double TIq = TIpre[q];
              const double uR = Ui * ct * cgd / (2.0 * (1.0 - s_cc));
              const double u0 = Ui * s_c;
              const double x0 = D * cgd * (1.0 + s_cc) / (sqrt2 * (4.0 * p->defl_alpha * TIq + 2.0 * p->defl_beta * (1.0 - s_c))) + x_i;
              const double ky = p->defl_ka * TIq + p->defl_kb, kz = ky;
              const double C0 = 1.0 - u0 / Ui;
              const double M0 = C0 * (2.0 - C0);
              const double E0 = C0 * C0 - 3.0 * exp(1.0 / 12.0) * C0 + 3.0 * exp(1.0 / 3.0);
              const double sz0 = D * 0.5 * sqrt(uR / (Ui + u0));
              const double sy0 = sz0 * cgd * cosd(p->veer);
              const double d0 = tan(th0) * (x0 - x_i);
              const double lin = p->ad + p->bd * (X - x_i);
              double d_near = ((X - x_i) / (x0 - x_i)) * d0 + lin;
              if (!(X >= x_i && X <= x0)) d_near = 0.0;
              double d_far = 0.0;
              if (X > x0) {
                const double sy = ky * (X - x0) + sy0, sz = kz * (X - x0) + sz0;
                const double s = sqrt(sy * sz / (sy0 * sz0));
                const double sM = sqrt(M0);
                const double ln_arg = ((1.6 + sM) * (1.6 * s - sM)) / ((1.6 - sM) * (1.6 * s + sM));
                d_far = d0 + th0 * E0 / 5.2 * sqrt(sy0 * sz0 / (ky * kz * M0)) * log(ln_arg) + lin;
              }
              delta = d_near + d_far;
            }
            /* 6. velocity deficit [A.3-6] (TI AFTER mixing, commanded yaw) */
            double deficit = 0.0;
            {
              const double TIq = TIpost[q];
              const double uR = Ui * ct / (2.0 * (1.0 - s_c));
              const double u0 = Ui * s_c;
              const double sz0 = D * 0.5 * sqrt(uR / (Ui + u0));
              const double sy0 = sz0 * cgv * cosd(p->veer);
              const double x0 = D * cgv * (1.0 + s_c) / (sqrt2 * (4.0 * p->alpha * TIq + 2.0 * p->beta * (1.0 - s_c))) + x_i;
              double sy, sz;
              int on = 0;
              if (X > x_i + 0.1 && X < x0) {
                const double up = (X - x_i) / (x0 - x_i), dn = (x0 - X) / (x0 - x_i);
                sy = dn * p->near_wake_c * D * sqrt(ct / 2.0) + up * sy0;
                sz = dn * p->near_wake_c * D * sqrt(ct / 2.0) + up * sz0;
                on = 1;
              } else if (X >= x0) {
                const double ky = p->ka * TIq + p->kb;
                sy = ky * (X - x0) + sy0;
                sz = ky * (X - x0) + sz0;
                on = 1;
              }
              if (on) {
                const double yy = Y - y_i - delta, zz = Zk - HH;
                double r;
                if (p->veer == 0.0) {
                  r = yy * yy / (2.0 * sy * sy) + zz * zz / (2.0 * sz * sz);
                } else { /* FLORIS 3.5 wake_velocity/gauss.py rCalt: the Gaussian rotated by the veer angle */
                  const double vr = p->veer * DEG2RAD;
                  const double ca = cos(vr) * cos(vr) / (2.0 * sy * sy) + sin(vr) * sin(vr) / (2.0 * sz * sz);
                  const double cb = -sin(2.0 * vr) / (4.0 * sy * sy) + sin(2.0 * vr) / (4.0 * sz * sz);
                  const double cc = sin(vr) * sin(vr) / (2.0 * sy * sy) + cos(vr) * cos(vr) / (2.0 * sz * sz);
                  r = ca * yy * yy - 2.0 * cb * yy * zz + cc * zz * zz;
                }
                double dd = 1.0 - ct * cgv / (8.0 * sy * sz / (D * D));
                if (dd < 0.0) dd = 0.0;
                if (dd > 1.0) dd = 1.0;
                deficit = (1.0 - sqrt(dd)) * exp(-r);
              }
            }
            defU[q] = deficit * Ui;
            if (defU[q] > p->overlap_thresh) ++cnt;
          }
        /* 7. SOSFS [A.3-7] */
        for (int q = 0; q < 9; ++q) T->wake[q] = hypot(T->wake[q], defU[q]);
        /* 8. Crespo-Hernandez + overlap gating [A.3-8] */
        {
          const double upm = (dx <= 0.1) ? 1.0 : 0.0, dnm = (dx > -0.1) ? 1.0 : 0.0;
          const double dxp = dx * dnm + upm;
          double ti = ch_pref * pow(dxp / D, p->ch_downstream) * dnm;
          if (isnan(ti)) ti = 0.0;
          if (isinf(ti) && ti > 0) ti = 0.0;
          const double overlap = (double)cnt / 9.0;
          if (margin && X > x_i && X <= x_i + 15.0 * D) {
            int gated = 0;
            for (int j = 0; j < 3; ++j) gated |= fabs(y_i - (ys[t] + off[j])) < 2.0 * D;
            if (gated)
              for (int q = 0; q < 9; ++q) {
                const double mg = fabs(defU[q] - p->overlap_thresh) / p->overlap_thresh;
                if (mg < min_margin) min_margin = mg;
              }
          }
          for (int j = 0; j < 3; ++j) {
            const double Y = ys[t] + off[j];
            double m = (X > x_i ? 1.0 : 0.0) * (fabs(y_i - Y) < 2.0 * D ? 1.0 : 0.0) * (X <= x_i + 15.0 * D ? 1.0 : 0.0);
            const double ti_added = overlap * ti * m;
            const double cand = sqrt(ti_added * ti_added + p->ambient_ti * p->ambient_ti);
            for (int k = 0; k < 3; ++k)
              if (cand > T->TI[j * 3 + k]) T->TI[j * 3 + k] = cand;
          }
        }
      } else {
        /* dx < 0: deficit 0 -> hypot(wake, 0) = wake; ti_added = 0 -> TI = max(ambient, TI) = TI */
      }
    }
    /* 9. field update [A.3-9] */
    for (int t = 0; t < N; ++t)
      for (int q = 0; q < 9; ++q) {
        st[t].U[q] = Uinit[q % 3] - st[t].wake[q];
        st[t].V[q] += vw[t][q];
        st[t].W[q] += ww[t][q];
      }
  }

  /* ---- outputs [A.4] in the caller's turbine order */
  for (int t = 0; t < N; ++t) {
    const int o = order[t];
    const tstate* T = &st[t];
    double m3 = 0.0, mu = 0.0, mv = 0.0, mw = 0.0, mti = 0.0, dir = 0.0;
    for (int q = 0; q < 9; ++q) {
      m3 += T->U[q] * T->U[q] * T->U[q];
      mu += T->U[q]; mv += T->V[q]; mw += T->W[q]; mti += T->TI[q];
      dir += wd - atan2(T->V[q], T->U[q]) / DEG2RAD;
    }
    mu /= 9.0; mv /= 9.0; mw /= 9.0; mti /= 9.0;
    double su = 0.0, sv = 0.0, sw = 0.0;
    for (int q = 0; q < 9; ++q) {
      su += (T->U[q] - mu) * (T->U[q] - mu);
      sv += (T->V[q] - mv) * (T->V[q] - mv);
      sw += (T->W[q] - mw) * (T->W[q] - mw);
    }
    const double wsp = cbrt(m3 / 9.0);
    wind_speed[o] = wsp;
    wind_dir[o] = dir / 9.0;
    wfo_type ty = {p->TSR, p->pP, p->ref_density, p->n_table, p->table_ws, p->table_ct, p->table_pow};
    if (p->n_types > 0) ty = p->types[p->type_of[o]];
    double veff = wsp * pow(cosd(yaw[o]), ty.pP / 3.0);
    veff = pow(p->air_density / ty.ref_density, 1.0 / 3.0) * veff;
    power[o] = ty.ref_density * interp_fill(veff, ty.n_table, ty.table_ws, ty.table_pow, 0.0, 0.0);
    load[o * 4 + 0] = mti;
    load[o * 4 + 1] = sqrt(su / 9.0);
    load[o * 4 + 2] = sqrt(sv / 9.0);
    load[o * 4 + 3] = sqrt(sw / 9.0);
  }
  if (margin) *margin = min_margin;
}

/* Batch entry: ws/wd have B entries when wind_stride = 1, one entry when 0. yaw is B x N row-major.
 * Returns 0 on success. nthreads <= 0 -> OpenMP default. */
int wfo_step_batch_ex(const wfo_params* p, int N, const double* x, const double* y, int B, const double* ws,
                      const double* wd, int wind_stride, const double* yaw, double* power, double* wind_speed,
                      double* wind_dir, double* load, int nthreads, double* margin /* B or NULL */, int tie_reverse) {
  if (!p || N <= 0 || B < 0) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  int err = 0;
#pragma omp parallel
  {
    double* work = (double*)malloc(sizeof(double) * 3 * N);
    double* vwbuf = (double*)malloc(sizeof(double) * 18 * N);
    int* order = (int*)malloc(sizeof(int) * N);
    tstate* st = (tstate*)malloc(sizeof(tstate) * N);
    if (!work || !vwbuf || !order || !st) {
#pragma omp atomic write
      err = -2;
    } else {
#pragma omp for schedule(dynamic, 4)
      for (int b = 0; b < B; ++b) {
        const size_t w = (size_t)b * (wind_stride ? 1 : 0);
        farm_step_one(p, N, x, y, ws[w], wd[w], yaw + (size_t)b * N, power + (size_t)b * N,
                      wind_speed + (size_t)b * N, wind_dir + (size_t)b * N, load + (size_t)b * N * 4, work, vwbuf, order, st,
                      margin ? margin + b : NULL, tie_reverse);
      }
    }
    free(work); free(vwbuf); free(order); free(st);
  }
  return err;
}

int wfo_step_batch(const wfo_params* p, int N, const double* x, const double* y, int B, const double* ws,
                   const double* wd, int wind_stride, const double* yaw, double* power, double* wind_speed,
                   double* wind_dir, double* load, int nthreads) {
  return wfo_step_batch_ex(p, N, x, y, B, ws, wd, wind_stride, yaw, power, wind_speed, wind_dir, load, nthreads, NULL, 0);
}

int wfo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
