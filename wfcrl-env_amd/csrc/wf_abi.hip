// wf_abi.hip — core of the C ABI declared in include/wfstep.h: handle life cycle, layout / batch, wf_step, risk flags and
// the float64 re-solve switch, timing.  (Model: wf_model.hip; kernel choice and launch: wf_dispatch.hip; wind:
// wf_wind_abi.hip; fused env step: wf_env_abi.hip; direction groups: wf_groups.hip.)
#include "wf_handle.h"

namespace wfi {

thread_local std::string g_create_error;

void free_batch(wf_handle* h) {
  hipFree(h->d_ws); hipFree(h->d_wd); hipFree(h->d_gx); hipFree(h->d_gy); hipFree(h->d_gidx); hipFree(h->d_flags);
  hipFree(h->d_farm_tie);
  hipFree(h->d_dir_perm); hipFree(h->d_sort_keys); hipFree(h->d_sort_vals); hipFree(h->d_sort_tmp);
  h->d_dir_perm = h->d_sort_vals = nullptr; h->d_sort_keys = nullptr; h->d_sort_tmp = nullptr;
  h->sort_tmp_bytes = h->dir_perm_cap = 0; h->dir_slots = 0;
  hipFree(h->d_res_list); hipFree(h->d_res_count); hipFree(h->d_flags_raw);
  h->d_res_list = h->d_res_count = h->d_flags_raw = nullptr;
  if (h->h_res_seen) (void)hipHostFree(h->h_res_seen);
  h->h_res_seen = nullptr;
  h->d_flags = h->d_farm_tie = nullptr;
  hipFree(h->d_yaw); hipFree(h->d_out);
  hipFree(h->d_env_yaw); hipFree(h->d_env_acc); hipFree(h->d_env_act); hipFree(h->d_env_out); hipFree(h->d_env_moves);
  if (h->h_env_act) hipHostFree(h->h_env_act);
  if (h->h_env_out) hipHostFree(h->h_env_out);
  h->d_env_yaw = h->d_env_acc = h->d_env_act = h->d_env_out = h->h_env_act = h->h_env_out = nullptr;
  h->d_env_moves = nullptr;
  hipFree(h->d_series_ws); hipFree(h->d_series_wd); hipFree(h->d_series_start); hipFree(h->d_ws_prev);
  hipFree(h->d_pair_tab); hipFree(h->d_pair_first); h->d_pair_tab = nullptr; h->d_pair_first = nullptr; h->pair_dirty = true;
  hipFree(h->d_ll_tab); hipFree(h->d_ll_flag); hipFree(h->d_src_log);
  h->d_ll_tab = h->d_src_log = nullptr; h->d_ll_flag = nullptr; h->ll_groups_cap = h->log_records_cap = 0;
  hipFree(h->d_perm); hipFree(h->d_blk_group); hipFree(h->d_group_wd); hipFree(h->d_bins);
  h->d_perm = h->d_blk_group = h->d_bins = nullptr; h->d_group_wd = nullptr;
  h->perm_cap = h->blk_cap = h->pair_groups_cap = 0; h->n_groups = 0; h->grid_step = 0.0;
  h->d_series_ws = h->d_series_wd = h->d_ws_prev = nullptr; h->d_series_start = nullptr; h->series_T = 0;
  if (h->h_yaw) hipHostFree(h->h_yaw);
  if (h->h_out) hipHostFree(h->h_out);
  h->d_ws = h->d_wd = h->d_gx = h->d_gy = nullptr; h->d_gidx = nullptr;
  h->d_yaw = h->d_out = h->h_yaw = h->h_out = nullptr;
  h->cap_env = h->cap_bn = 0;
}

}  // namespace wfi

using namespace wfi;

extern "C" {

int wf_version(void) { return WF_ABI_VERSION; }
int wf_create(int device_id, wf_handle** out) {
  if (!out) return fail(nullptr, WF_E_INVALID, "out == NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, WF_E_NODEVICE, std::string("no HIP device visible (") + hipGetErrorString(e) +
                                            "); libwfstep has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, WF_E_NODEVICE, "device id out of range");
  wf_handle* h = new (std::nothrow) wf_handle();
  if (!h) return fail(nullptr, WF_E_NOMEM, "out of host memory");
  h->device = device_id;
  {  // A/B runs: the environment seeds the handle's kernel choice once, here (wf_set_kernel_choice is the interface)
    wf_kernel_choice& c = h->choice;
    if (getenv("WF_NO_PAIR_TABLE")) c.pair_table = 0;
    const char* f = getenv("WF_LL_FLY");
    if (f && f[0] == '0') c.fly_one_block = 0;
    const char* gs = getenv("WF_KERNEL_GS");  // e.g. "16x5"
    int og = 0, os = 0;
    if (gs && sscanf(gs, "%dx%d", &og, &os) == 2 && og > 0 && os > 0) { c.slot_G = og; c.slot_S = os; }
    const char* off = getenv("WF_LL");
    if (off && off[0] == '0') c.one_block = 0;
    const char* fs = getenv("WF_LL_FAR_SKIP");
    if (fs && fs[0] == '0') c.far_skip = 0;
    const char* rr = getenv("WF_RISK_RESOLVE");  // "0": float32 only (the tests that hold the float32 kernels to their flag contract)
    if (rr && rr[0] >= '0' && rr[0] <= '2' && rr[1] == 0) h->resolve_mode = rr[0] - '0';
    const char* cal = getenv("WF_CALIBRATE");
    if (cal && cal[0] == '0') c.calibrate = 0;
    const char* force = getenv("WF_LL_G");  // "8" or "4x2"
    if (force && c.one_block != 0) {
      int g = 0, sl = 1;
      if (sscanf(force, "%dx%d", &g, &sl) >= 1) { c.one_block = 1; c.ll_G = g; c.ll_S = sl; }
      else c.one_block = 0;
    }
  }
  DeviceGuard guard(device_id);
  if ((e = guard.err) != hipSuccess || (e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess ||
      (e = hipMalloc(&h->d_tab, sizeof(WfTables))) != hipSuccess) {
    std::string msg = std::string("wf_create: ") + hipGetErrorString(e);
    delete h;
    return fail(nullptr, WF_E_HIP, msg);
  }
  h->stream = h->own_stream;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
  }
  wf_model_params p;
  wf_default_model(&p);
  h->model = p;
  h->tws.assign(p.table_ws, p.table_ws + p.n_table);
  h->tct.assign(p.table_ct, p.table_ct + p.n_table);
  h->tcp.assign(p.table_cp, p.table_cp + p.n_table);
  *out = h;
  return WF_OK;
}

int wf_destroy(wf_handle* h) {
  if (!h) return WF_OK;
  DeviceGuard guard(h->device);
  hipStreamSynchronize(h->stream);
  free_batch(h);
  hipFree(h->d_tab64_mt); hipFree(h->d_type_consts); hipFree(h->d_type_of);
  hipFree(h->d_tab); hipFree(h->d_tab64); hipFree(h->d_lx); hipFree(h->d_ly); hipFree(h->d_centre); hipFree(h->d_layout_of);
  hipEventDestroy(h->ev0); hipEventDestroy(h->ev1);
  hipStreamDestroy(h->own_stream);
  delete h;
  return WF_OK;
}

int wf_set_stream(wf_handle* h, void* s, int external) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  hipStream_t next = external ? (hipStream_t)s : h->own_stream;
  if (next != h->stream) {
    WF_HIP(h, hipStreamSynchronize(h->stream));
    h->stream = next;
  }
  return WF_OK;
}
void* wf_get_stream(wf_handle* h) { return h ? (void*)h->stream : nullptr; }

// Layout data of the handle: K layouts of n turbines, their centres of rotation [A.1-1] and the farm -> layout map.
// counts (or null): turbines each layout really has, 1..n — the rest of its row are placeholders (wf_set_layouts_counts)
static int upload_layouts(wf_handle* h, int n, int K, const double* x, const double* y, const int* layout_of, const int* counts = nullptr) {
  std::vector<double> centre(2 * (size_t)K);
  for (int l = 0; l < K; ++l) {
    const double *xl = x + (size_t)l * n, *yl = y + (size_t)l * n;
    double xmin = xl[0], xmax = xl[0], ymin = yl[0], ymax = yl[0];
    const int n_l = counts ? counts[l] : n;  // the centre of rotation is that of the REAL turbines' bounding box [A.1-1]
    for (int i = 0; i < n_l; ++i) {
      if (!std::isfinite(xl[i]) || !std::isfinite(yl[i])) return fail(h, WF_E_INVALID, "turbine coordinates must be finite");
      xmin = std::fmin(xmin, xl[i]); xmax = std::fmax(xmax, xl[i]);
      ymin = std::fmin(ymin, yl[i]); ymax = std::fmax(ymax, yl[i]);
    }
    centre[2 * l] = (xmin + xmax) / 2.0; centre[2 * l + 1] = (ymin + ymax) / 2.0;  // centre of rotation [A.1-1]
  }
  WF_HIP(h, hipStreamSynchronize(h->stream));
  hipFree(h->d_lx); hipFree(h->d_ly); hipFree(h->d_centre); hipFree(h->d_layout_of); hipFree(h->d_layout_n); hipFree(h->d_nreal);
  h->d_lx = h->d_ly = h->d_centre = nullptr; h->d_layout_of = h->d_layout_n = h->d_nreal = nullptr;
  h->layout_n.clear();
  const size_t kn = (size_t)K * n;
  WF_HIP(h, hipMalloc(&h->d_lx, sizeof(double) * kn));
  WF_HIP(h, hipMalloc(&h->d_ly, sizeof(double) * kn));
  WF_HIP(h, hipMalloc(&h->d_centre, sizeof(double) * 2 * K));
  WF_HIP(h, hipMemcpy(h->d_lx, x, sizeof(double) * kn, hipMemcpyHostToDevice));
  WF_HIP(h, hipMemcpy(h->d_ly, y, sizeof(double) * kn, hipMemcpyHostToDevice));
  WF_HIP(h, hipMemcpy(h->d_centre, centre.data(), sizeof(double) * 2 * K, hipMemcpyHostToDevice));
  h->layout_of.clear();
  if (layout_of) {
    h->layout_of.assign(layout_of, layout_of + h->B);
    WF_HIP(h, hipMalloc(&h->d_layout_of, sizeof(int) * h->B));
    WF_HIP(h, hipMemcpy(h->d_layout_of, layout_of, sizeof(int) * h->B, hipMemcpyHostToDevice));
  }
  if (counts) {
    bool ragged = false;
    for (int l = 0; l < K; ++l) ragged = ragged || counts[l] != n;
    if (ragged) {
      h->layout_n.assign(counts, counts + K);
      std::vector<int> nreal((size_t)h->B);
      for (int b = 0; b < h->B; ++b) nreal[b] = counts[layout_of ? layout_of[b] : (K == 1 ? 0 : b)];
      WF_HIP(h, hipMalloc(&h->d_layout_n, sizeof(int) * K));
      WF_HIP(h, hipMalloc(&h->d_nreal, sizeof(int) * h->B));
      WF_HIP(h, hipMemcpy(h->d_layout_n, counts, sizeof(int) * K, hipMemcpyHostToDevice));
      WF_HIP(h, hipMemcpy(h->d_nreal, nreal.data(), sizeof(int) * h->B, hipMemcpyHostToDevice));
    }
  }
  h->lx.assign(x, x + kn); h->ly.assign(y, y + kn);
  h->n_layouts = K;
  h->xc = centre[0]; h->yc = centre[1];
  return WF_OK;
}

int wf_set_layout(wf_handle* h, int n, const double* x, const double* y) {
  if (!h || !x || !y) return WF_E_INVALID;
  if (n < 1 || n > WF_MAX_TURBINES) return fail(h, WF_E_INVALID, "n_turbines must be in 1..256");
  WF_ON_DEVICE(h);
  const int v = pick_variant(h, n, h->B);
  if (v < 0) return fail(h, WF_E_UNSUPPORTED, "no kernel variant for this turbine count");
  {
    const int llg = pick_ll(h, n, h->B);
    if (n != h->N) set_ll_shape(h, 0, 1);  // table and source log are laid out for (N, G, S)
    set_ll_shape(h, llg >> 4, llg ? (llg & 15) : 1);
  }
  {
    int rc = upload_layouts(h, n, 1, x, y, nullptr);
    if (rc != WF_OK) return rc;
  }
  if (n != h->N) { free_batch(h); h->B = 0; }
  h->N = n; h->variant = v; h->wind_count = 0; h->shared_dir = false; h->model_dirty = true;
  if (!h->guard_user) h->guard_rel = n > 128 ? 2.0e-5 : 1.0e-5;  // (wf_handle.h: guard_rel)
  reset_calibration(h);
  h->n_groups = 0; h->grid_step = 0.0;
  return WF_OK;
}

int wf_set_batch(wf_handle* h, int B) {
  if (!h) return WF_E_INVALID;
  if (h->N <= 0) return fail(h, WF_E_INVALID, "wf_set_layout must be called before wf_set_batch");
  if (B < 1) return fail(h, WF_E_INVALID, "env_batch must be >= 1");
  WF_ON_DEVICE(h);
  WF_HIP(h, hipStreamSynchronize(h->stream));
  if (pick_variant(h, h->N, B) < 0) return fail(h, WF_E_UNSUPPORTED, "no kernel variant for this turbine count");
  apply_kernel_pick(h, h->N, B, nullptr);
  if ((size_t)B != h->cap_env) {
    free_batch(h);
    // nothing is committed until every allocation has succeeded: a failure leaves the handle without a batch
    // (wf_set_batch must be called again) instead of with half-allocated buffers under the old sizes
    h->B = 0; h->wind_count = 0; h->shared_dir = false; h->ws_prev_valid = false;
    const size_t bn = (size_t)B * h->N;
    WF_HIP(h, hipMalloc(&h->d_ws, sizeof(double) * B));
    WF_HIP(h, hipMalloc(&h->d_wd, sizeof(double) * B));
    WF_HIP(h, hipMalloc(&h->d_gx, sizeof(double) * bn));
    WF_HIP(h, hipMalloc(&h->d_gy, sizeof(double) * bn));
    WF_HIP(h, hipMalloc(&h->d_flags, sizeof(int) * B));
    WF_HIP(h, hipMalloc(&h->d_farm_tie, sizeof(int) * ((size_t)B + 1)));
    WF_HIP(h, hipMalloc(&h->d_gidx, sizeof(int) * bn));
    h->cap_env = B; h->cap_bn = bn;
  }
  h->B = B; h->wind_count = 0; h->shared_dir = false; h->ws_prev_valid = false;
  h->n_groups = 0; h->grid_step = 0.0;
  if (h->n_layouts > 1 || !h->layout_n.empty()) {
    // The farm -> layout map and the per-farm turbine counts were laid out for the old batch: every farm goes back to the
    // first layout — with the turbine count that layout really has (wf_set_layouts_counts), so that its placeholders stay
    // placeholders and d_nreal has one entry per farm of the NEW batch.
    std::vector<double> x0(h->lx.begin(), h->lx.begin() + h->N), y0(h->ly.begin(), h->ly.begin() + h->N);
    const int n0 = h->layout_n.empty() ? h->N : h->layout_n[0];
    int rc = upload_layouts(h, h->N, 1, x0.data(), y0.data(), nullptr, n0 < h->N ? &n0 : nullptr);
    if (rc != WF_OK) return rc;
    h->model_dirty = true;
  }
  return WF_OK;
}
int wf_set_layouts(wf_handle* h, int n_layouts, const double* x, const double* y, const int* layout_of) {
  return wf_set_layouts_counts(h, n_layouts, x, y, nullptr, layout_of);
}

int wf_set_layouts_counts(wf_handle* h, int n_layouts, const double* x, const double* y, const int* counts, const int* layout_of) {
  if (!h || !x || !y) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_layout and wf_set_batch must be called before wf_set_layouts");
  if (n_layouts < 1 || n_layouts > h->B) return fail(h, WF_E_INVALID, "n_layouts must be in 1..env_batch");
  if (!layout_of && n_layouts != 1 && n_layouts != h->B)
    return fail(h, WF_E_INVALID, "without layout_of, n_layouts must be 1 or env_batch (farm b has layout b)");
  if (layout_of)
    for (int b = 0; b < h->B; ++b)
      if (layout_of[b] < 0 || layout_of[b] >= n_layouts) return fail(h, WF_E_INVALID, "layout_of entry out of range");
  if (counts)
    for (int l = 0; l < n_layouts; ++l)
      if (counts[l] < 1 || counts[l] > h->N) return fail(h, WF_E_INVALID, "turbine count of a layout must be in 1..n_turbines of the handle");
  WF_ON_DEVICE(h);
  {
    int rc = upload_layouts(h, h->N, n_layouts, x, y, n_layouts > 1 ? layout_of : nullptr, counts);
    if (rc != WF_OK) return rc;
  }
  h->wind_count = 0; h->shared_dir = false; h->pair_dirty = true;
  h->series_T = 0; h->grid_step = 0.0;
  ungroup(h);
  return WF_OK;
}

int wf_step(wf_handle* h, const float* yaw, float* power, float* wspd, float* wdir, float* load, int on_device) {
  if (!h || !yaw) return WF_E_INVALID;
  if (h->wind_count == 0) return fail(h, WF_E_INVALID, "wf_set_wind must be called before wf_step");
  WF_ON_DEVICE(h);
  if (h->model_dirty) {
    int rc = build_consts(h);
    if (rc != WF_OK) return rc;
  }
  const size_t bn = (size_t)h->B * h->N;
  if (on_device) return launch_step(h, yaw, power, wspd, wdir, load, nullptr);
  if (!h->d_yaw) WF_HIP(h, hipMalloc(&h->d_yaw, sizeof(float) * bn));
  if (!h->h_yaw) WF_HIP(h, hipHostMalloc(&h->h_yaw, sizeof(float) * bn, hipHostMallocDefault));
  if (!h->d_out) WF_HIP(h, hipMalloc(&h->d_out, sizeof(float) * bn * 7));
  if (!h->h_out) WF_HIP(h, hipHostMalloc(&h->h_out, sizeof(float) * bn * 7, hipHostMallocDefault));
  std::memcpy(h->h_yaw, yaw, sizeof(float) * bn);
  WF_HIP(h, hipMemcpyAsync(h->d_yaw, h->h_yaw, sizeof(float) * bn, hipMemcpyHostToDevice, h->stream));
  {
    int rc = launch_step(h, h->d_yaw, h->d_out, h->d_out + bn, h->d_out + 2 * bn, h->d_out + 3 * bn, nullptr);
    if (rc != WF_OK) return rc;
  }
  WF_HIP(h, hipMemcpyAsync(h->h_out, h->d_out, sizeof(float) * bn * 7, hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipStreamSynchronize(h->stream));
  if (power) std::memcpy(power, h->h_out, sizeof(float) * bn);
  if (wspd) std::memcpy(wspd, h->h_out + bn, sizeof(float) * bn);
  if (wdir) std::memcpy(wdir, h->h_out + 2 * bn, sizeof(float) * bn);
  if (load) std::memcpy(load, h->h_out + 3 * bn, sizeof(float) * bn * 4);
  return WF_OK;
}
int wf_set_risk_guard(wf_handle* h, double rel_band) {
  if (!h) return WF_E_INVALID;
  if (!(rel_band >= 0.0) || !(rel_band < 0.5)) return fail(h, WF_E_INVALID, "risk guard band must be in [0, 0.5)");
  h->guard_rel = rel_band;
  h->guard_user = true;
  h->consts.guard_inv = rel_band > 0.0 ? (float)(1.0 / rel_band) : 1125899906842624.0f;
  return WF_OK;
}

int wf_set_risk_resolve(wf_handle* h, int mode) {
  if (!h) return WF_E_INVALID;
  if (mode < 0 || mode > 2) return fail(h, WF_E_INVALID, "risk resolve mode must be 0 (off), 1 (flagged farms) or 2 (every farm)");
  if (!h->types.empty() && mode == 0)  // (1 is kept for when the definitions are cleared; 2 is what runs meanwhile)
    return fail(h, WF_E_INVALID, "several turbine definitions (wf_set_turbine_types): the float32 kernels know one table, mode 0 cannot be served");
  h->resolve_mode = mode;
  return WF_OK;
}

int wf_get_risk_resolve(wf_handle* h, int* mode) {
  if (!h || !mode) return WF_E_INVALID;
  *mode = h->types.empty() ? h->resolve_mode : 2;
  return WF_OK;
}

int wf_get_resolve_stats(wf_handle* h, int* n_resolved, int* raw_flags, int on_device) {
  if (!h) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called first");
  WF_ON_DEVICE(h);
  if (!h->d_res_list || !h->res_last) {  // the last step had no re-solve behind it
    if (n_resolved) *n_resolved = 0;
    if (raw_flags) return wf_get_risk_flags(h, raw_flags, on_device);
    return WF_OK;
  }
  if (raw_flags)
    WF_HIP(h, hipMemcpyAsync(raw_flags, h->d_flags_raw, sizeof(int) * h->B, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             h->stream));
  if (n_resolved) WF_HIP(h, hipMemcpyAsync(n_resolved, h->d_res_count + h->res_parity, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (n_resolved || (raw_flags && !on_device)) WF_HIP(h, hipStreamSynchronize(h->stream));
  return WF_OK;
}

int wf_get_risk_flags(wf_handle* h, int* flags, int on_device) {
  if (!h || !flags) return WF_E_INVALID;
  if (h->B <= 0 || !h->d_flags) return fail(h, WF_E_INVALID, "wf_set_batch must be called first");
  WF_ON_DEVICE(h);
  WF_HIP(h, hipMemcpyAsync(flags, h->d_flags, sizeof(int) * h->B, on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           h->stream));
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));
  return WF_OK;
}

int wf_sync(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  WF_HIP(h, hipStreamSynchronize(h->stream));
  return WF_OK;
}

int wf_timing_begin(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  WF_HIP(h, hipEventRecord(h->ev0, h->stream));
  return WF_OK;
}

int wf_timing_end(wf_handle* h, float* ms) {
  if (!h || !ms) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  WF_HIP(h, hipEventRecord(h->ev1, h->stream));
  WF_HIP(h, hipEventSynchronize(h->ev1));
  WF_HIP(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return WF_OK;
}
const char* wf_last_error(wf_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

}  // extern "C"
