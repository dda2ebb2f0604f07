// wf_wind_abi.hip — wf_set_wind* (replaces FlorisInterface.update_wind -> fi.reinitialize, reference
// wfcrl/interface.py:663-671) and the on-device wind process (reset sampling, series playback).
#include "wf_handle.h"

using namespace wfi;

// Several layouts in the batch (wf_set_layouts): every farm has its own geometry, so the wind is always held per farm.
// Under ONE direction the farms of a layout share a geometry: the layouts become the groups of a grouped launch (pair
// table path) when that pays off; otherwise, and with a direction per farm, the on-the-fly path.
static int set_wind_layouts(wf_handle* h, int n_ws, bool one_dir, int on_device) {
  if (n_ws == 1) {  // (d_wd was filled by the caller when a speed per farm came with one direction)
    WF_HIP(h, wfk_launch_fill(h->B, h->d_ws, h->stream));
    WF_HIP(h, wfk_launch_fill(h->B, h->d_wd, h->stream));
  }
  h->series_T = 0; h->grid_step = 0.0;
  ungroup(h);
  const int K = h->n_layouts;
  if (one_dir && !h->layout_of.empty() && groups_fit(h, K)) {
    hipFree(h->d_group_wd); h->d_group_wd = nullptr;
    WF_HIP(h, hipMalloc(&h->d_group_wd, sizeof(double) * K));
    WF_HIP(h, hipMemcpyAsync(h->d_group_wd, h->d_wd, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    WF_HIP(h, wfk_launch_fill(K, h->d_group_wd, h->stream));
    int rc = build_groups(h, h->layout_of.data(), K, h->d_group_wd, true);
    if (rc != WF_OK) return rc;
    h->wind_sync = true;
  } else {
    int rc = run_geometry(h, h->B, h->d_wd, !on_device);
    if (rc != WF_OK) return rc;
    h->wind_sync = !on_device;
  }
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));  // caller's host arrays may go away
  h->shared_dir = false;
  h->wind_count = h->B;
  h->ws_prev_valid = false;
  h->pair_dirty = true;
  return WF_OK;
}

extern "C" {

int wf_set_wind_counts(wf_handle* h, const double* ws, int n_ws, const double* wd, int n_wd, int on_device) {
  if (!h || !ws || !wd) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called before wf_set_wind");
  if ((n_ws != 1 && n_ws != h->B) || (n_wd != 1 && n_wd != h->B)) return fail(h, WF_E_INVALID, "wind count must be 1 or env_batch");
  if (n_ws == 1 && n_wd != 1) return fail(h, WF_E_INVALID, "a direction per farm needs a speed per farm");
  WF_ON_DEVICE(h);
  if (!on_device) {
    for (int i = 0; i < n_ws; ++i)
      if (!(ws[i] > 0.0)) return fail(h, WF_E_INVALID, "wind speed must be > 0 and direction finite");
    for (int i = 0; i < n_wd; ++i)
      if (!std::isfinite(wd[i])) return fail(h, WF_E_INVALID, "wind speed must be > 0 and direction finite");
  }
  // One direction for every farm (explicitly: n_wd == 1; or host arrays whose directions are all equal, e.g. sampled
  // speeds under a fixed direction): the rotation, the sort and the pair table depend on the direction only, so this
  // is the shared-wind path with a speed per farm.
  bool same_dir = n_ws > 1 && (n_wd == 1 || !on_device);
  for (int i = 1; same_dir && i < n_wd; ++i) same_dir = wd[i] == wd[0];
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  WF_HIP(h, hipMemcpyAsync(h->d_ws, ws, sizeof(double) * n_ws, kind, h->stream));
  WF_HIP(h, hipMemcpyAsync(h->d_wd, wd, sizeof(double) * n_wd, kind, h->stream));
  if (n_wd == 1 && n_ws > 1)  // the step kernel reads a direction per farm next to the speed per farm
    WF_HIP(h, wfk_launch_fill(h->B, h->d_wd, h->stream));
  if (h->n_layouts > 1) return set_wind_layouts(h, n_ws, n_ws == 1 || same_dir, on_device);
  const int count = n_ws;
  ungroup(h);  // (first: the geometry pass lays out tie flags and launch order for the kernel shape of the plain batch)
  {
    int rc = run_geometry(h, same_dir ? 1 : count, h->d_wd, !on_device);
    if (rc != WF_OK) return rc;
  }
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));  // caller's host arrays may go away
  h->shared_dir = same_dir;
  h->wind_sync = !on_device;
  h->wind_count = count;
  h->series_T = 0;
  h->ws_prev_valid = false;
  h->pair_dirty = true;
  return WF_OK;
}

int wf_set_wind(wf_handle* h, const double* ws, const double* wd, int count, int on_device) {
  return wf_set_wind_counts(h, ws, count, wd, count, on_device);
}
int wf_wind_sample(wf_handle* h, unsigned long long seed, const wf_wind_dist* dist) {
  if (!h) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called before wf_wind_sample");
  WF_ON_DEVICE(h);
  const wf_wind_dist def{8.0, 8.0, 3.0, 28.0, 270.0, 20.0, 0.0, 360.0};
  const wf_wind_dist d = dist ? *dist : def;
  if (!(d.ws_scale > 0) || !(d.ws_shape > 0) || !(d.ws_lo > 0) || !(d.ws_lo <= d.ws_hi) || !(d.wd_std >= 0))
    return fail(h, WF_E_INVALID, "invalid wind distribution parameters");
  const double dv[8] = {d.ws_scale, d.ws_shape, d.ws_lo, d.ws_hi, d.wd_mean, d.wd_std, d.wd_lo, d.wd_hi};
  WF_HIP(h, wfk_launch_wind_sample(h->B, seed, dv, h->d_ws, h->d_wd, h->stream));
  ungroup(h);  // (before the geometry pass: see wf_set_wind_counts)
  {
    int rc = run_geometry(h, h->B, h->d_wd, false);
    if (rc != WF_OK) return rc;
  }
  h->wind_count = h->B;
  h->shared_dir = false;
  h->wind_sync = false;
  h->series_T = 0;
  h->ws_prev_valid = false;
  h->pair_dirty = true;  // env_batch 1: "one wind per farm" is also "one wind for the batch" (table path)
  return WF_OK;
}

int wf_wind_sample_binned(wf_handle* h, unsigned long long seed, const wf_wind_dist* dist, double step_deg) {
  if (!h) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called before wf_wind_sample_binned");
  if (!(step_deg > 0.0) || !(step_deg <= 90.0)) return fail(h, WF_E_INVALID, "direction step must be in (0, 90] degrees");
  const int K = (int)std::llround(360.0 / step_deg);
  if (std::fabs(K * step_deg - 360.0) > 1e-9) return fail(h, WF_E_INVALID, "direction step must divide 360 degrees");
  WF_ON_DEVICE(h);
  const wf_wind_dist def{8.0, 8.0, 3.0, 28.0, 270.0, 20.0, 0.0, 360.0};
  const wf_wind_dist d = dist ? *dist : def;
  if (!(d.ws_scale > 0) || !(d.ws_shape > 0) || !(d.ws_lo > 0) || !(d.ws_lo <= d.ws_hi) || !(d.wd_std >= 0))
    return fail(h, WF_E_INVALID, "invalid wind distribution parameters");
  if (!groups_pay_off(h, K)) {  // too many bins for this batch (or no table path): sample un-binned directions
    int rc = wf_wind_sample(h, seed, dist);
    return rc;
  }
  const double dv[8] = {d.ws_scale, d.ws_shape, d.ws_lo, d.ws_hi, d.wd_mean, d.wd_std, d.wd_lo, d.wd_hi};
  if (!h->d_bins) WF_HIP(h, hipMalloc(&h->d_bins, sizeof(int) * h->B));
  WF_HIP(h, wfk_launch_wind_sample_binned(h->B, seed, dv, step_deg, h->d_ws, h->d_wd, h->d_bins, h->stream));
  std::vector<int> bins(h->B);
  WF_HIP(h, hipMemcpyAsync(bins.data(), h->d_bins, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipStreamSynchronize(h->stream));
  // geometry and pair tables of the K grid directions depend on layout and model only: built once, kept across resets
  const bool cached = h->grid_step == step_deg && h->n_groups == K && h->series_T == 0;
  if (!cached) {
    hipFree(h->d_group_wd); h->d_group_wd = nullptr;
    WF_HIP(h, hipMalloc(&h->d_group_wd, sizeof(double) * K));
    WF_HIP(h, wfk_launch_bin_centres(K, step_deg, h->d_group_wd, h->stream));
  }
  int rc = build_groups(h, bins.data(), K, h->d_group_wd, !cached);
  if (rc != WF_OK) return rc;
  h->grid_step = step_deg;
  h->wind_sync = true;
  h->wind_count = h->B;
  h->shared_dir = false;
  h->series_T = 0;
  h->ws_prev_valid = false;
  return WF_OK;
}

int wf_wind_series(wf_handle* h, int T, const double* ws, const double* wd, const int* start, unsigned long long seed) {
  if (!h || !ws || !wd) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called before wf_wind_series");
  if (T < 1) return fail(h, WF_E_INVALID, "the wind series needs at least one row");
  for (int i = 0; i < T; ++i)
    if (!(ws[i] > 0.0) || !std::isfinite(wd[i])) return fail(h, WF_E_INVALID, "wind speed must be > 0 and direction finite");
  WF_ON_DEVICE(h);
  WF_HIP(h, hipStreamSynchronize(h->stream));
  hipFree(h->d_series_ws); hipFree(h->d_series_wd); hipFree(h->d_series_start);
  h->d_series_ws = h->d_series_wd = nullptr; h->d_series_start = nullptr;
  WF_HIP(h, hipMalloc(&h->d_series_ws, sizeof(double) * T));
  WF_HIP(h, hipMalloc(&h->d_series_wd, sizeof(double) * T));
  WF_HIP(h, hipMalloc(&h->d_series_start, sizeof(int) * h->B));
  WF_HIP(h, hipMemcpy(h->d_series_ws, ws, sizeof(double) * T, hipMemcpyHostToDevice));
  WF_HIP(h, hipMemcpy(h->d_series_wd, wd, sizeof(double) * T, hipMemcpyHostToDevice));
  if (start) {
    for (int b = 0; b < h->B; ++b)
      if (start[b] < 0 || start[b] >= T) return fail(h, WF_E_INVALID, "series start out of range");
    WF_HIP(h, hipMemcpy(h->d_series_start, start, sizeof(int) * h->B, hipMemcpyHostToDevice));
  } else {
    WF_HIP(h, wfk_launch_series_start(h->B, T, seed, h->d_series_start, h->stream));
  }
  h->series_T = T;
  h->series_t = -1;
  h->ws_prev_valid = false;
  ungroup(h);
  h->grid_step = 0.0;
  // A shared series has only T distinct winds: farms are grouped by their start row (farms with the same start see the
  // same row at every tick), one sorted geometry + pair table per ROW, and the table path serves the whole playback.
  if (groups_pay_off(h, T)) {
    std::vector<int> st(h->B);
    if (start) std::memcpy(st.data(), start, sizeof(int) * h->B);
    else {
      WF_HIP(h, hipMemcpyAsync(st.data(), h->d_series_start, sizeof(int) * h->B, hipMemcpyDeviceToHost, h->stream));
      WF_HIP(h, hipStreamSynchronize(h->stream));
    }
    int rc = build_groups(h, st.data(), T, h->d_series_wd, true);
    if (rc != WF_OK) return rc;
  }
  h->wind_sync = true;
  return wf_wind_series_step(h);
}

int wf_wind_series_step(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  if (h->series_T <= 0) return fail(h, WF_E_INVALID, "wf_wind_series must be called first");
  if (h->series_t + 1 >= h->series_T) return fail(h, WF_E_INVALID, "wind series exhausted");
  WF_ON_DEVICE(h);
  h->series_t += 1;
  if (h->series_t >= 1) {  // keep the wind of the state before this tick for the reward normalisation
    if (!h->d_ws_prev) WF_HIP(h, hipMalloc(&h->d_ws_prev, sizeof(double) * h->B));
    WF_HIP(h, hipMemcpyAsync(h->d_ws_prev, h->d_ws, sizeof(double) * h->B, hipMemcpyDeviceToDevice, h->stream));
    h->ws_prev_valid = true;  // consumed by the next wf_env_step that computes a reward
  }
  WF_HIP(h, wfk_launch_series_gather(h->B, h->series_T, h->series_t, h->d_series_start, h->d_series_ws, h->d_series_wd,
                                     h->d_ws, h->d_wd, h->stream));
  h->wind_count = h->B;
  h->shared_dir = false;
  if (h->n_groups > 0) {
    h->group_shift = h->series_t;  // group g (= start row g) is on row (g + t) % T now: geometry and tables are per row
  } else {
    int rc = run_geometry(h, h->B, h->d_wd, false);
    if (rc != WF_OK) return rc;
    h->pair_dirty = true;  // see wf_wind_sample
  }
  return WF_OK;
}

int wf_get_wind(wf_handle* h, double* ws, double* wd, int on_device) {
  if (!h || !ws || !wd) return WF_E_INVALID;
  if (h->wind_count == 0) return fail(h, WF_E_INVALID, "no wind has been set");
  WF_ON_DEVICE(h);
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (h->wind_count == h->B) {
    WF_HIP(h, hipMemcpyAsync(ws, h->d_ws, sizeof(double) * h->B, kind, h->stream));
    WF_HIP(h, hipMemcpyAsync(wd, h->d_wd, sizeof(double) * h->B, kind, h->stream));
    if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));
    return WF_OK;
  }
  double v[2];
  WF_HIP(h, hipMemcpyAsync(&v[0], h->d_ws, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipMemcpyAsync(&v[1], h->d_wd, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipStreamSynchronize(h->stream));
  std::vector<double> a(h->B, v[0]), b(h->B, v[1]);
  const hipMemcpyKind k2 = on_device ? hipMemcpyHostToDevice : hipMemcpyHostToHost;
  WF_HIP(h, hipMemcpy(ws, a.data(), sizeof(double) * h->B, k2));
  WF_HIP(h, hipMemcpy(wd, b.data(), sizeof(double) * h->B, k2));
  return WF_OK;
}

}  // extern "C"
