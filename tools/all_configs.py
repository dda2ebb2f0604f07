"""GPU box: bench.py over every BASELINE config (+ the per-farm-wind variants) -> one summary line each.
usage: python tools/all_configs.py [steps]"""
import json, os, subprocess, sys
steps = sys.argv[1] if len(sys.argv) > 1 else "40"
runs = [("cfg2", [], {}), ("cfg3", [], {}), ("cfg3b", [], {}), ("cfg4", [], {}), ("cfg4", [], {"WF_LL": "0"}),
        ("cfg4", [], {"WF_NO_PAIR_TABLE": "1"}), ("cfg5", [], {}), ("cfg5", [], {"WF_LL": "0"}), ("cfg5", ["--per-env-wind"], {})]
for cfg, extra, env in runs:
    cmd = [sys.executable, "bench.py", "--config", cfg, "--steps", steps, "--cpu-seconds", "1", "--no-env-leg"] + extra
    out = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True).stdout.strip().splitlines()
    try:
        d = json.loads(out[-1])
    except Exception:
        print(cfg, extra, env, "FAILED"); continue
    print(f'{cfg:5s} {" ".join(extra):15s} {" ".join(f"{k}={v}" for k, v in env.items()):20s} '
          f'{d["value"]:.3e} farm-steps/s  kernel {d["roofline"]["kernel_ms"]:.3f} ms  step {d["ms_per_step"]:.3f} ms  '
          f'host-synced {d.get("ms_per_step_host_synced", 0):.3f} ms  {d["config"]["kernel"]}  vgprs {d["config"]["vgprs"]} scratch {d["config"]["scratch_bytes"]}'
          f'  | vs oracle: power rel max {d["power_rel_err"]["max"]:.1e}, wind dir max {d["power_rel_err"]["wind_direction_abs_max_deg"]:.1e} deg, TI max {d["power_rel_err"]["ti_abs_max"]:.1e}')
