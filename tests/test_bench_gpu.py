"""GPU: bench.py's contract — the single-GPU line, and the multi-rank flow (self-launch through torch.distributed.run,
strong + weak legs, max-over-ranks timing) exercised with 2 ranks on whatever GPUs the box has (on a one-GPU box the
ranks share the device and rendezvous over gloo; bench.py selects that itself)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*args, timeout=900, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    d = _run("--config", "cfg2", "--steps", "5", "--warmup", "2", "--cpu-seconds", "0.5")
    assert d["metric"] == "farm_steps_per_sec" and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["value"] > 1e6 and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["env_batch_total"] == 4096 and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["frac_of_measured_copy"] > rf["frac"] and "traffic_source" in rf
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert d["power_rel_err"]["contract"] in ("ok", "flagged") and d["power_rel_err"]["max_unflagged"] <= 1e-4
    assert d["env_level"]["step"]["env_steps_per_sec"] > 0 and d["env_level"]["step_light"]["ms_per_step"] > 0
    # the env step beside the plain kernel on the env loop's own yaw state (round 6)
    assert d["env_level"]["kernel_on_env_state_ms"] > 0 and "over_kernel_same_state" in d["env_level"]["step"]


def test_two_ranks_self_launched_strong_and_weak():
    d = _run("--gpus", "2", "--config", "cfg2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-env-leg")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["env_batch_total"] == 4096 and d["config"]["env_batch_per_gpu"] == 2048
    assert d["value"] > 1e5
    wk = d["weak_scaling"]
    assert wk["env_batch_total"] == 8192 and wk["env_batch_per_gpu"] == 4096 and wk["value"] > 1e5
    assert "cpu_baseline" not in d
    d = _run("--gpus", "2", "--config", "cfg2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-env-leg",
             "--scaling", "weak")
    assert d["scaling"] == "weak" and d["config"]["env_batch_total"] == 8192 and d["strong_scaling"]["env_batch_total"] == 4096


def test_eight_rank_dry_run_of_baseline_config4():
    """BASELINE.json configs[3] at its real world size: `bench.py --gpus 8 --config cfg4` (HornsRev1 x 65536 farms sharded
    over 8 ranks).  No 8-GPU node is available to the build, so the ranks share this box's GPU(s) over gloo (bench.py
    selects that itself) — what is proven is the code path: every rank steps its own contiguous 8192-farm block, the
    blocks tile the batch, the kernel is the one the rounds model picks for an 8192-farm shard, and ONE line comes out."""
    from wfcrl_env_amd.sharding import shard_bounds

    # (WF_CALIBRATE=0: with eight ranks timing their kernel families on ONE shared GPU the per-handle calibration measures
    # the other ranks, not the kernels — on a real node every rank has a GPU of its own; here the rounds model's guess is checked)
    d = _run("--gpus", "8", "--config", "cfg4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-env-leg", timeout=1500,
             extra_env={"WF_CALIBRATE": "0"})
    c = d["config"]
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and c["env_batch_per_gpu"] == 8192 and c["env_batch_total"] == 65536
    assert sorted(s[0] for s in c["shards"]) == list(range(8))
    for r, dev, lo, hi in c["shards"]:
        assert (lo, hi) == shard_bounds(65536, r, 8) and hi - lo == 8192
    assert "wf_step_kernel<G=16,S=5>" in c["kernel"]  # the latency-bound choice for a chip a quarter full
    assert d["value"] > 1e6 and d["weak_scaling"]["env_batch_per_gpu"] == 65536
