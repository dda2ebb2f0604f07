// wf_env_abi.hip — the fused env step: device-resident yaw / actuation state, transition + budget gate +
// reward inside the step launch (reference wfcrl/simple_env.py:64-85, wfcrl/mdp.py:291-319).
#include "wf_handle.h"

using namespace wfi;

extern "C" {

int wf_env_config(wf_handle* h, const wf_env_params* p) {
  if (!h || !p) return WF_E_INVALID;
  if (!(p->yaw_lo < p->yaw_hi) || !(p->yaw_step > 0) || !(p->actuator_rate > 0) || !(p->dt > 0))
    return fail(h, WF_E_INVALID, "need yaw_lo < yaw_hi, yaw_step > 0, actuator_rate > 0, dt > 0");
  h->env = *p;
  return WF_OK;
}

int wf_env_set_power_unit(wf_handle* h, int megawatts) {
  if (!h) return WF_E_INVALID;
  h->env_power_mw = megawatts ? 1 : 0;
  return WF_OK;
}

static int env_alloc(wf_handle* h) {
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called first");
  if (h->d_env_yaw) return WF_OK;
  const size_t bn = (size_t)h->B * h->N;
  WF_HIP(h, hipMalloc(&h->d_env_yaw, sizeof(float) * bn));
  WF_HIP(h, hipMalloc(&h->d_env_acc, sizeof(float) * bn));
  WF_HIP(h, hipMalloc(&h->d_env_moves, sizeof(int) * h->B));
  WF_HIP(h, hipMemsetAsync(h->d_env_yaw, 0, sizeof(float) * bn, h->stream));
  WF_HIP(h, hipMemsetAsync(h->d_env_acc, 0, sizeof(float) * bn, h->stream));
  WF_HIP(h, hipMemsetAsync(h->d_env_moves, 0, sizeof(int) * h->B, h->stream));
  return WF_OK;
}

int wf_env_reset(wf_handle* h) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  const bool fresh = h->d_env_yaw == nullptr;
  int rc = env_alloc(h);
  if (rc != WF_OK) return rc;
  if (!fresh) {
    const size_t bn = (size_t)h->B * h->N;
    WF_HIP(h, hipMemsetAsync(h->d_env_yaw, 0, sizeof(float) * bn, h->stream));
    WF_HIP(h, hipMemsetAsync(h->d_env_acc, 0, sizeof(float) * bn, h->stream));
    WF_HIP(h, hipMemsetAsync(h->d_env_moves, 0, sizeof(int) * h->B, h->stream));
  }
  return WF_OK;
}

int wf_env_set_prev_wind(wf_handle* h, const double* ws, int on_device) {
  if (!h || !ws) return WF_E_INVALID;
  if (h->B <= 0) return fail(h, WF_E_INVALID, "wf_set_batch must be called first");
  WF_ON_DEVICE(h);
  if (!h->d_ws_prev) WF_HIP(h, hipMalloc(&h->d_ws_prev, sizeof(double) * h->B));
  WF_HIP(h, hipMemcpyAsync(h->d_ws_prev, ws, sizeof(double) * h->B, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           h->stream));
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));
  h->ws_prev_valid = true;
  return WF_OK;
}

int wf_env_state(wf_handle* h, float* yaw, float* acc, int* moves, int set, int on_device) {
  if (!h) return WF_E_INVALID;
  WF_ON_DEVICE(h);
  int rc = env_alloc(h);
  if (rc != WF_OK) return rc;
  const size_t bn = (size_t)h->B * h->N;
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : (set ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
  auto xfer = [&](void* user, void* dev, size_t bytes) -> hipError_t {
    if (!user) return hipSuccess;
    return set ? hipMemcpyAsync(dev, user, bytes, kind, h->stream) : hipMemcpyAsync(user, dev, bytes, kind, h->stream);
  };
  WF_HIP(h, xfer(yaw, h->d_env_yaw, sizeof(float) * bn));
  WF_HIP(h, xfer(acc, h->d_env_acc, sizeof(float) * bn));
  WF_HIP(h, xfer(moves, h->d_env_moves, sizeof(int) * h->B));
  if (!on_device) WF_HIP(h, hipStreamSynchronize(h->stream));
  return WF_OK;
}

int wf_env_step(wf_handle* h, const float* action, float* reward, float* yaw, float* power, float* wspd, float* wdir,
                float* load, int on_device) {
  if (!h) return WF_E_INVALID;
  if (h->wind_count == 0) return fail(h, WF_E_INVALID, "wf_set_wind must be called before wf_env_step");
  WF_ON_DEVICE(h);
  int rc = env_alloc(h);
  if (rc != WF_OK) return rc;
  if (h->model_dirty && (rc = build_consts(h)) != WF_OK) return rc;
  const size_t bn = (size_t)h->B * h->N, B = (size_t)h->B;
  WfEnvArgs ea{};
  ea.yaw_state = h->d_env_yaw; ea.acc = h->d_env_acc; ea.moves = h->d_env_moves;
  ea.yaw_step = h->env.yaw_step; ea.yaw_lo = h->env.yaw_lo; ea.yaw_hi = h->env.yaw_hi;
  ea.rate = h->env.actuator_rate; ea.dt = h->env.dt; ea.budget = h->env.budget;
  ea.load_coef = h->env.load_coef; ea.discrete = h->env.discrete;
  // free wind of the state BEFORE the step, when it differs from the current one (series tick, wf_env_set_prev_wind):
  // valid for one reward only — a second env step without a new tick normalises by the current wind again
  ea.ws_prev = (h->ws_prev_valid && h->d_ws_prev) ? h->d_ws_prev : nullptr;
  if (reward) h->ws_prev_valid = false;
  ea.power_mw = h->env_power_mw;
  if (on_device) {
    ea.action = action; ea.reward = reward;
    ea.yaw_out = action ? yaw : nullptr;  // (a transition writes the new yaw to the caller's array as it writes the state)
    if ((rc = launch_step(h, nullptr, power, wspd, wdir, load, &ea)) != WF_OK) return rc;
    if (yaw && !ea.yaw_out) WF_HIP(h, hipMemcpyAsync(yaw, h->d_env_yaw, sizeof(float) * bn, hipMemcpyDeviceToDevice, h->stream));
    return WF_OK;
  }
  if (!h->d_env_act) {
    WF_HIP(h, hipMalloc(&h->d_env_act, sizeof(float) * bn));
    WF_HIP(h, hipMalloc(&h->d_env_out, sizeof(float) * B));
    WF_HIP(h, hipHostMalloc(&h->h_env_act, sizeof(float) * bn, hipHostMallocDefault));
    WF_HIP(h, hipHostMalloc(&h->h_env_out, sizeof(float) * (B + bn), hipHostMallocDefault));
  }
  if (!h->d_out) WF_HIP(h, hipMalloc(&h->d_out, sizeof(float) * bn * 7));
  if (!h->h_out) WF_HIP(h, hipHostMalloc(&h->h_out, sizeof(float) * bn * 7, hipHostMallocDefault));
  if (action) {
    std::memcpy(h->h_env_act, action, sizeof(float) * bn);
    WF_HIP(h, hipMemcpyAsync(h->d_env_act, h->h_env_act, sizeof(float) * bn, hipMemcpyHostToDevice, h->stream));
    ea.action = h->d_env_act;
  }
  ea.reward = reward ? h->d_env_out : nullptr;
  if ((rc = launch_step(h, nullptr, power ? h->d_out : nullptr, wspd ? h->d_out + bn : nullptr,
                        wdir ? h->d_out + 2 * bn : nullptr, load ? h->d_out + 3 * bn : nullptr, &ea)) != WF_OK) return rc;
  if (reward) WF_HIP(h, hipMemcpyAsync(h->h_env_out, h->d_env_out, sizeof(float) * B, hipMemcpyDeviceToHost, h->stream));
  if (yaw) WF_HIP(h, hipMemcpyAsync(h->h_env_out + B, h->d_env_yaw, sizeof(float) * bn, hipMemcpyDeviceToHost, h->stream));
  if (power || wspd || wdir || load)
    WF_HIP(h, hipMemcpyAsync(h->h_out, h->d_out, sizeof(float) * bn * 7, hipMemcpyDeviceToHost, h->stream));
  WF_HIP(h, hipStreamSynchronize(h->stream));
  if (reward) std::memcpy(reward, h->h_env_out, sizeof(float) * B);
  if (yaw) std::memcpy(yaw, h->h_env_out + B, sizeof(float) * bn);
  if (power) std::memcpy(power, h->h_out, sizeof(float) * bn);
  if (wspd) std::memcpy(wspd, h->h_out + bn, sizeof(float) * bn);
  if (wdir) std::memcpy(wdir, h->h_out + 2 * bn, sizeof(float) * bn);
  if (load) std::memcpy(load, h->h_out + 3 * bn, sizeof(float) * bn * 4);
  return WF_OK;
}

}  // extern "C"
