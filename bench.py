#!/usr/bin/env python3
"""bench.py — farm-steps/s of the HIP wind-farm step (BASELINE.json metric) + roofline + CPU baseline.

  python bench.py [--gpus N --steps K --warmup W]            (N > 1: spawns one rank per GPU itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[3] — HornsRev1_Floris (80 turbines in the reference code,
data_cases.py:269-334), env_batch 65536 IN TOTAL, ws 8 m/s, wd 270 deg, random-walk yaw (dyaw ~ U(-5,5) clipped to
+-40).  It fits one GPU, so N = 1 runs the whole config on one device.  For N > 1 the default is
STRONG scaling — the config's total batch sharded into contiguous blocks of independent farms, 65536 / N per GPU, as
BASELINE configs[3-4] state it ("sharded across 8 MI355X") — and the same run also times the WEAK variant (the full
config batch on every GPU) and reports it beside the headline (`weak_scaling`); `--scaling weak` swaps the two.  No
data-path collective either way.  A "step" = one pass of the hot path (wf_step) over the batch, yaw already resident
in HBM, outputs left in HBM.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: measured device copy rate
VALU_PEAK_LANEOPS = 7.864e13   # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (157.3 TFLOP/s / 2)

CONFIGS = {
    # name: (layout, TOTAL env batch of the BASELINE config)
    "cfg2": ("Ablaincourt_", 4096),
    "cfg3": ("Turb16_Row5_", 16384),
    "cfg3b": ("Turb16_TCRWP_", 16384),  # configs[2] as named: first 16 TCRWP turbines
    "cfg4": ("HornsRev1_", 65536),
    "cfg5": ("HornsRev2_", 131072),
}


def valu_practical_peak(info, shared_wind=True):
    """The fp32 VALU ceiling for the kernel that ran (profiles/valu_practical.json: issue rates measured on this chip at the
    kernel's waves per SIMD, mixed by the kernel's own static share of transcendental instructions), or None."""
    try:
        vp = json.load(open(os.path.join(ROOT, "profiles", "valu_practical.json")))
    except Exception:
        return None
    G, S = info["lanes_per_env"], info["slots_per_lane"]
    if info.get("one_block_kernel"):
        want = f"ll_{G}x{S}_shared{1 if shared_wind else 0}_tab{1 if info.get('pair_table') else 0}_"
    else:
        want = f"slot_{G}x{S}_"
    cands = [k for k in vp["kernels"] if k.startswith(want) and "veer1" not in k]
    if not cands:
        return None
    k = vp["kernels"][sorted(cands, key=lambda n: ("mc1" not in n, "occ21" in n))[0]]
    waves = max(2, min(4, 512 // max(1, info["vgprs"])))
    r = vp[f"cycles_per_wave_instr_at_{waves}_waves_per_simd"]
    share = k["transcendental_share"]
    cyc = (1.0 - share) * r["plain_fp32"] + share * r["transcendental"]
    peak = 64.0 * 256 * 4 * 2.4e9 / cyc
    return {"peak": peak, "how": f"{waves} waves per SIMD: {r['plain_fp32']:.2f} cycles per plain fp32 wave-instruction, {r['transcendental']:.2f} per "
                                 f"transcendental (measured, {vp['source_rates']}); {100 * share:.1f} % of the kernel's {k['valu_static']} static VALU "
                                 f"instructions are transcendental -> {cyc:.2f} cycles per wave-instruction against the nominal 2"}


def effective_cpus() -> int:
    """CPUs this process may actually use: affinity mask, capped by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, n)


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a fresh child process tree
    (nothing in this process has touched the GPU) and relay rank 0's JSON line.  With fewer GPUs than ranks (a
    one-GPU box) the ranks share devices and rendezvous over gloo: a test mode, flagged in the JSON line."""
    import torch  # importing torch and counting devices does not initialise the GPU

    ndev = torch.cuda.device_count()
    env = dict(os.environ)
    if ndev < args.gpus:
        env.setdefault("BENCH_DIST_BACKEND", "gloo")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def load_layout(layout_name):
    with open(os.path.join(ROOT, "wfcrl-env_amd", "environments", "layouts.json")) as f:
        all_layouts = json.load(f)
    if layout_name == "Turb16_TCRWP_":
        t = all_layouts["Turb_TCRWP_"]
        return {"num_turbines": 16, "xcoords": t["xcoords"][:16], "ycoords": t["ycoords"][:16]}
    return all_layouts[layout_name]


def counter_profile(config: str, B: int, info: dict):
    """PMC-derived per-launch figures of the step kernel from the committed rocprofv3 passes (profiles/pmc_index.json:
    which file, which commit, which batch, WHICH KERNEL).  They are NOT measured by this run — counters need rocprofv3
    around the process — so they are reported with their source, and only when this run launched the kernel they were
    collected on at the batch they were collected at (a forced family, WF_LL=0, another pick: None)."""
    try:
        idx = json.load(open(os.path.join(ROOT, "profiles", "pmc_index.json")))
        e = idx.get(f"{config}_B{B}")
        if not e:
            return None
        if any(info.get(k) != v for k, v in e.get("kernel", {}).items()):
            return None
        pmc = json.load(open(os.path.join(ROOT, e["file"])))
        if "TCC_EA0_RDREQ_128B_sum" in pmc:  # the L2's requests to the fabric by size: exact bytes, no FETCH_SIZE correction
            rd = 128.0 * pmc["TCC_EA0_RDREQ_128B_sum"] + 64.0 * pmc["TCC_EA0_RDREQ_64B_sum"] + 32.0 * pmc["TCC_EA0_RDREQ_32B_sum"]
            wr = 64.0 * pmc["TCC_EA0_WRREQ_64B_sum"] + 32.0 * (pmc["TCC_EA0_WRREQ_sum"] - pmc["TCC_EA0_WRREQ_64B_sum"])
            nbytes, how = rd + wr, "TCC_EA0_RDREQ_{128B,64B,32B} / WRREQ_{64B,32B} request counts x their sizes"
        else:
            nbytes, how = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0, "(2 x FETCH_SIZE + WRITE_SIZE) KiB, the guide's gfx950 correction"
        return {"hbm_bytes": nbytes, "traffic_how": how, "insts_valu": pmc["SQ_INSTS_VALU"],
                "kernel_us": e.get("kernel_us"), "source": f'{e["file"]} (rocprofv3 --pmc, separate passes; kernel of commit '
                f'{e["commit"]}; not measured in this run)'}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg4", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the config's total batch sharded over the GPUs (default, BASELINE configs[3-4]); "
                         "weak = the full config batch on every GPU.  The other one is timed too and reported beside it.")
    ap.add_argument("--env-batch", type=int, default=0, help="total env batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-env-leg", action="store_true", help="skip the fused-env-step, env-level and per-farm-wind legs (profiling runs: only the headline kernel is launched)")
    ap.add_argument("--per-env-wind", action="store_true",
                    help="a wind per farm: cfg5 wd_b(t) = wd(t) + U(-10,10) (per-farm rotation + sort on the device every step); "
                         "other configs a fixed ws ~ U(6,12), wd ~ 270 + U(-10,10) per farm (the on-the-fly path)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import torch

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU (the driver's launch); BENCH_DIST_BACKEND=gloo + fewer GPUs than ranks is a test mode that
    # exercises the multi-rank flow on a single-GPU box (ranks then share devices)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    from wfcrl_env_amd.sharding import device_for_rank

    ndev = max(1, torch.cuda.device_count())
    shared_devices = backend != "nccl" and ndev < world
    local_rank = device_for_rank(local_rank, ndev, backend)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from wfcrl_env_amd.backend import WfStep
    from wfcrl_env_amd.sharding import shard_bounds

    layout_name, B_total = CONFIGS[args.config]
    if args.env_batch:
        B_total = args.env_batch
    lay = load_layout(layout_name)
    N = lay["num_turbines"]
    cfg_id = int(args.config[3])
    sweep = cfg_id == 5

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_leg(mode):
        """One timed leg.  strong: this rank owns the contiguous block shard_bounds(B_total, rank, world) of the
        config's farms; weak: it owns B_total farms of a world x B_total batch.  Returns the leg's figures and the
        live objects (handle, yaw ring, outputs) of rank-local use."""
        if mode == "strong":
            lo, hi = shard_bounds(B_total, rank, world)
            total = B_total
        else:
            lo, hi = shard_bounds(B_total * world, rank, world)
            total = B_total * world
        B = hi - lo
        w = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B, device_id=local_rank)
        w.set_stream(torch.cuda.current_stream().cuda_stream, True)
        # synthetic inputs, seeded per SURVEY §8d (1234 + cfg id) and per rank:
        #   cfg2: absolute yaw ~ U(-40,40), fixed wind;  cfg3/cfg4: random-walk yaw (dyaw ~ U(-5,5), clipped to +-40),
        #   fixed wind;  cfg5: random-walk yaw + wd(t) = 270 + 30 sin(2 pi t/200), shared by the farms (or per farm
        #   with --per-env-wind), re-set every step on the device (rotation + sort kernel, no host sync)
        gen = torch.Generator(device="cuda").manual_seed(1234 + cfg_id + 7919 * rank)
        ring = []
        yaw = torch.zeros((B, N), device="cuda", dtype=torch.float32)
        for _ in range(8):
            if cfg_id == 2:
                yaw = torch.rand((B, N), device="cuda", generator=gen) * 80 - 40
            else:
                yaw = (yaw + (torch.rand((B, N), device="cuda", generator=gen) * 10 - 5)).clamp_(-40, 40)
            ring.append(yaw.clone())
        if sweep:
            t_all = torch.arange(args.warmup + args.steps + 1, device="cuda", dtype=torch.float64)
            wd_t = 270.0 + 30.0 * torch.sin(2 * torch.pi * t_all / 200.0)
            nw = B if args.per_env_wind else 1
            jitter = (torch.rand(nw, device="cuda", generator=gen, dtype=torch.float64) * 20 - 10) if args.per_env_wind \
                else torch.zeros(1, device="cuda", dtype=torch.float64)
            ws_dev = torch.full((nw,), 8.0, device="cuda", dtype=torch.float64)

            def set_wind_at(t):
                w.set_wind(ws_dev, wd_t[t] + jitter)
        else:
            def set_wind_at(t):
                return None
        w.set_wind(8.0, 270.0)
        if args.per_env_wind and not sweep:  # a fixed wind per farm: the on-the-fly path (profiling runs)
            w.set_wind(torch.rand(B, device="cuda", generator=gen, dtype=torch.float64) * 6 + 6,
                       270.0 + torch.rand(B, device="cuda", generator=gen, dtype=torch.float64) * 20 - 10)
        # set-up, not warm-up: the handle times its kernel families once, before its first launch (wf_kernel_choice::calibrate);
        # the float64 re-solve of flagged farms is ON (the default of a handle: the shipped configuration is the measured one)
        out = w.step(ring[0])
        w.sync()
        for i in range(args.warmup):
            set_wind_at(i)
            w.step(ring[i % len(ring)], out)
        barrier()
        t0 = time.perf_counter()
        w.timing_begin()
        for i in range(args.steps):
            set_wind_at(args.warmup + i)
            w.step(ring[i % len(ring)], out)
        kern_ms = w.timing_end() / args.steps  # HIP events on the stream the kernel is launched on
        barrier()
        elapsed = time.perf_counter() - t0
        # host-synchronised per step (what a Python RL loop that reads every observation sees)
        t1 = time.perf_counter()
        nsync = min(args.steps, 20)
        for i in range(nsync):
            set_wind_at(args.warmup + i)
            w.step(ring[i % len(ring)], out)
            w.sync()
        sync_ms = (time.perf_counter() - t1) / nsync * 1e3
        shards = [[rank, local_rank, lo, hi]]
        ki = w.kernel_info()
        # every rank's kernel, beside its shard (each handle times its kernel families itself, so two ranks may
        # settle on different families for the same shard size — the line says so instead of quoting rank 0's only):
        # [lanes per farm, slots per lane, one-block kernel, farms of a mixed launch's main part, VGPRs]
        kern = [[ki["lanes_per_env"], ki["slots_per_lane"], ki["one_block_kernel"], ki["mixed_main_farms"], ki["vgprs"]]]
        if dist is not None:
            dev = "cuda" if backend == "nccl" else "cpu"
            t = torch.tensor([elapsed, kern_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, kern_ms = float(t[0]), float(t[1])
            mine = torch.tensor([rank, local_rank, lo, hi], device=dev, dtype=torch.int64)
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)  # (bookkeeping for the JSON line, outside the timed region: not a data-path collective)
            shards = [[int(v) for v in p_] for p_ in parts]
            mine_k = torch.tensor(kern[0], device=dev, dtype=torch.int64)
            parts_k = [torch.empty_like(mine_k) for _ in range(world)]
            dist.all_gather(parts_k, mine_k)
            kern = [[int(v) for v in p_] for p_ in parts_k]
        return dict(mode=mode, B=B, total=total, elapsed=elapsed, kern_ms=kern_ms, sync_ms=sync_ms, w=w, ring=ring, out=out,
                    shards=shards, shard_kernels=kern)

    main_leg = run_leg(args.scaling)
    other_leg = None
    if world > 1:
        main_leg["w"].sync()
        keep = main_leg if rank == 0 else None
        if keep is None:
            main_leg["w"].close()
        other_leg = run_leg("weak" if args.scaling == "strong" else "strong")
        other_leg["w"].close()
        # the collective part of the job is over: every rank leaves the group now, so that rank 0's remaining legs
        # (fused env step, env-level loop) run with no peer waiting on it
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    w, ring, out, B = main_leg["w"], main_leg["ring"], main_leg["out"], main_leg["B"]
    elapsed, kern_ms = main_leg["elapsed"], main_leg["kern_ms"]
    info = w.kernel_info()

    # the same farms through the fused env step (transition + budget gate + reward in the launch,
    # only reward + local wind observations written): reported beside the headline, not as `value`
    venv_ms = None
    env_level = None
    if not args.no_env_leg:
        try:
            w.env_config(load_coef=0.1)
            w.env_reset()
            act = [(r - ring[i - 1]) if i else r for i, r in enumerate(ring)]
            eout = w.env_step(act[0], want=("reward", "yaw", "wind_speed", "wind_direction"))
            w.sync()
            w.timing_begin()
            for i in range(args.steps):
                w.env_step(act[i % len(act)], want=("reward", "yaw", "wind_speed", "wind_direction"), out=eout)
            venv_ms = w.timing_end() / args.steps
        except Exception as e:  # pragma: no cover
            print(f"bench.py: fused env step leg failed: {e}", file=sys.stderr)
        # end to end from Python: the env object a learner holds — make("<layout>_Floris", env_batch=B) — stepped with
        # device-resident random actions; wall clock per step including every Python-side cost of VecWindFarmEnv
        try:
            from wfcrl_env_amd import environments as envs

            env_id = {"Turb16_TCRWP_": "Turb16_TCRWP_Floris"}.get(layout_name, layout_name + "Floris")
            acts = [(torch.rand((B, N), device="cuda") * 10 - 5) for _ in range(8)]
            env_level = {}
            # the env object as make() builds it by DEFAULT (round 6: two preallocated output buffer sets used alternately, power in
            # MW out of the kernel) and with reuse_buffers=False (fresh output tensors every step: the opt-out for callers that
            # keep what a step returned)
            for label, kw in (("default", {}), ("fresh_buffers", {"reuse_buffers": False})):
                env = envs.make(env_id, env_batch=B, max_num_steps=10 ** 9, load_coef=0.1, log=False, **kw)
                env.reset(seed=0, options={"wind_speed": 8.0, "wind_direction": 270.0})
                for name, fn in (("step", env.step), ("step_light", env.step_light)):
                    for i in range(3):
                        fn(acts[i])
                    torch.cuda.synchronize()
                    # steady-state rate of the asynchronous loop: events on the stream the env launches on (torch's current
                    # stream, which the handle adopts), over enough steps that the loop's one-off costs (first launch, the final
                    # host sync: ~0.5 ms, 3 % of a 20-step run) do not pass for per-step time; the wall clock is kept beside it
                    n_it = max(args.steps, 100)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t = time.perf_counter()
                    e0.record()
                    for i in range(n_it):
                        fn(acts[i % 8])
                    e1.record()
                    torch.cuda.synchronize()
                    wall_ms = (time.perf_counter() - t) / n_it * 1e3
                    ms = e0.elapsed_time(e1) / n_it
                    key = name + "_default_buffers" if label == "default" else name + "_fresh_buffers"
                    env_level[key] = {"ms_per_step": ms, "env_steps_per_sec": B / (ms * 1e-3), "over_kernel": ms / kern_ms - 1.0,
                                      "wall_ms_per_step": wall_ms, "steps_timed": n_it}
                if label == "default":
                    # the plain wf_step kernel ON THE ENV'S OWN YAW STATE (the loop's small angles behind the budget gate are a
                    # costlier solve than the headline's random walk): what the env step adds to the kernel, like for like
                    try:
                        yaw_env = torch.as_tensor(env.fi.env_get_state(as_torch=True)["yaw"]).to(torch.float32).reshape(B, N).contiguous()
                        ws_e, wd_e = env.fi.get_wind()
                        w2 = WfStep(lay["xcoords"], lay["ycoords"], env_batch=B)
                        w2.set_wind(ws_e, wd_e)
                        o2 = w2.step(yaw_env)
                        for _ in range(3):
                            w2.step(yaw_env, o2)
                        w2.sync()
                        w2.timing_begin()
                        for _ in range(n_it):
                            w2.step(yaw_env, o2)
                        same_ms = w2.timing_end() / n_it
                        w2.close()
                        env_level["kernel_on_env_state_ms"] = same_ms
                        for k2 in ("step_default_buffers", "step_light_default_buffers"):
                            env_level[k2]["over_kernel_same_state"] = env_level[k2]["ms_per_step"] / same_ms - 1.0
                    except Exception as e:  # pragma: no cover
                        print(f"bench.py: same-state kernel leg failed: {e}", file=sys.stderr)
                env.close()
            # (rounds 4-5 printed the reuse_buffers=True env under "step" / "step_light": that env is the default now)
            env_level["step"], env_level["step_light"] = env_level["step_default_buffers"], env_level["step_light_default_buffers"]
            env_level["what"] = (f'make("{env_id}", env_batch={B}).step / .step_light with device-resident random actions, '
                                 "device time per step between two events on the launch stream over max(steps, 100) steps (asynchronous "
                                 "launches; wall_ms_per_step: the same loop by the host clock, one sync at the end); "
                                 "*_default_buffers: the env as make() builds it (since round 6: outputs written into two preallocated "
                                 "buffer sets used alternately, power in MW out of the kernel); *_fresh_buffers: reuse_buffers=False "
                                 "(outputs allocated per step); over_kernel: against the plain wf_step kernel of the headline (random-walk yaw); "
                                 "over_kernel_same_state: against the plain wf_step kernel timed on the env loop's own yaw state and wind "
                                 "(kernel_on_env_state_ms) — what the env step adds, like for like")
        except Exception as e:  # pragma: no cover
            print(f"bench.py: env-level leg failed: {e}", file=sys.stderr)

    # Strict contract (north_star: 1e-4 on every farm): a handle runs with the float64 re-solve of the flagged farms ON by
    # default (wf_set_risk_resolve mode 1, reference interface.py:564 computes every step in float64), and the headline
    # `value` above was timed that way — the shipped configuration.  The same workload with the re-solve off (the float32
    # kernel on its own) and on is timed again here with HIP events, for every config, and reported under `extra`.
    nst = max(5, min(args.steps, 20))

    def both_modes(step_fn):
        """HIP-event ms per step of step_fn(i) with the re-solve off and on, and what the last strict step re-solved."""
        r = {}
        for mode, key in ((0, "float32_only"), (1, "with_float64_resolve")):
            w.set_risk_resolve(mode)
            for _ in range(2):  # (the handle times its kernels once, before the first launch of a wind regime: not in the timed region)
                step_fn(0)
            w.sync()
            w.timing_begin()
            for i in range(nst):
                step_fn(i)
            ms = w.timing_end() / nst
            r[key] = {"ms_per_step": ms, "farm_steps_per_sec": B / (ms * 1e-3)}
        st = w.resolve_stats()
        r["flagged_farm_frac_batch"] = float((st["raw_flags"] != 0).mean())
        r["n_resolved"] = st["n_resolved"]
        r["resolve_ms"] = r["with_float64_resolve"]["ms_per_step"] - r["float32_only"]["ms_per_step"]
        r["kernel"] = w.kernel_info()
        return r

    strict = None
    per_farm = None
    if not args.no_env_leg:
        try:
            def headline_step(i):
                if sweep:
                    w.set_wind(ws_sweep, wd_sweep[i])
                w.step(ring[i % len(ring)], out)

            if sweep:  # the timed leg's own wind process again (shared sweep, or + U(-10, 10) per farm)
                t_all = torch.arange(nst + 1, device="cuda", dtype=torch.float64)
                nw = B if args.per_env_wind else 1
                jit = (torch.rand(nw, device="cuda", dtype=torch.float64) * 20 - 10) if args.per_env_wind else torch.zeros(1, device="cuda", dtype=torch.float64)
                wd_sweep = [270.0 + 30.0 * torch.sin(2 * torch.pi * t_all[i] / 200.0) + jit for i in range(nst + 1)]
                ws_sweep = torch.full((nw,), 8.0, device="cuda", dtype=torch.float64)
            strict = both_modes(headline_step)
            strict["wind"] = "the timed leg's own wind: " + ("wd(t) sweep, re-set on the device every step" if sweep else "ws 8 m/s, wd 270 shared")
        except Exception as e:  # pragma: no cover
            print(f"bench.py: strict leg failed: {e}", file=sys.stderr)
            strict = None
    # The reference gives every env its own wind at reset (mdp.py:237-258: 8 Weibull(8) clipped to [3, 28] m/s,
    # N(270, 20) mod 360).  The same farms under that distribution — the on-the-fly kernel, about 2 % of the farms
    # flagged — without and with the float64 re-solve: reported under `extra`, never as `value`.
    if not args.per_env_wind and not args.no_env_leg:
        try:
            rngw = np.random.default_rng(1234 + cfg_id)
            ws_pf = np.clip(8 * rngw.weibull(8, B), 3, 28)
            wd_pf = rngw.normal(270, 20, B) % 360
            w.set_wind(ws_pf, wd_pf)
            per_farm = both_modes(lambda i: w.step(ring[i % len(ring)], out))
            per_farm["wind"] = "per farm: ws = clip(8 Weibull(8), 3, 28), wd = N(270, 20) mod 360 (reference wfcrl/mdp.py:237-258)"
            per_farm["_wind"] = (ws_pf, wd_pf)
        except Exception as e:  # pragma: no cover
            print(f"bench.py: per-farm-wind leg failed: {e}", file=sys.stderr)
            per_farm = None
        w.set_risk_resolve(1)
        w.set_wind(8.0, 270.0)

    value = main_leg["total"] * args.steps / elapsed
    algo_bytes = (32 * N + 8) * B  # read 4N yaw + 8 wind, write 28N outputs, per farm-step
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    cp = counter_profile(args.config + ("_per_env_wind" if args.per_env_wind else ""), B, info) if not sweep else None  # (pmc_index.json keys: cfg4_B65536, cfg4_per_env_wind_B65536)
    # VALU issue slots the kernel actually spent: SQ_INSTS_VALU wave-instructions x 64 lanes per launch (committed
    # rocprofv3 pass of THIS kernel at THIS batch, or nothing) over this run's kernel time.  (Rounds 1-3 also printed an
    # analytic "useful work" fraction; it counted every downstream pair and overstated the work once the kernel began
    # to skip far pairs — withdrawn: only issued slots are reported.)
    valu_achieved = (cp["insts_valu"] * 64.0 / (kern_ms * 1e-3)) if cp else None
    vp = valu_practical_peak(info, shared_wind=not args.per_env_wind)

    wl_wind = ("ws 8 m/s, wd(t) = 270 + 30 sin(2 pi t/200)" + (" + U(-10,10) per farm" if args.per_env_wind else " shared")
               if sweep else ("ws ~ U(6,12) m/s, wd ~ 270 + U(-10,10) per farm (fixed)" if args.per_env_wind else "ws 8 m/s, wd 270"))
    res = {"metric": "farm_steps_per_sec", "value": value, "unit": "farm-steps/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
           "ms_per_step_host_synced": main_leg["sync_ms"],
           "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{layout_name}Floris x env_batch {main_leg['total']} in total (BASELINE configs[{cfg_id - 1}]), "
                                  + wl_wind + (", yaw ~ U(-40,40)" if cfg_id == 2 else ", random-walk yaw"),
                      "risk_resolve": "on (wf_set_risk_resolve mode 1, the handle's default: flagged farms re-solved in float64 behind every step)",
                      "layout": layout_name.rstrip("_"),
                      "turbines": N, "env_batch_per_gpu": B, "env_batch_total": main_leg["total"],
                      # which farms each rank stepped: [rank, device, first farm, one past the last] (contiguous blocks,
                      # wfcrl_env_amd/sharding.py: shard_bounds)
                      "shards": main_leg["shards"],
                      # the kernel each rank's handle settled on, in the order of `shards`: [G, S, one-block kernel, main farms of a
                      # mixed launch, VGPRs]; `kernel` below is rank 0's
                      "shard_kernels": main_leg["shard_kernels"],
                      "shard_kernels_agree": len({tuple(k[:4]) for k in main_leg["shard_kernels"]}) == 1,
                      "parallelism": f"env-shard x{world}" + (f" ({world} ranks sharing {ndev} GPU(s) over gloo: test mode)" if shared_devices else ""),
                      "kernel": (f"wf_step_ll_kernel<G={info['lanes_per_env']},S={info['slots_per_lane']}> (one target block at a time, source log)"
                                 if info.get("one_block_kernel") else
                                 f"wf_step_kernel<G={info['lanes_per_env']},S={info['slots_per_lane']}>")
                                + (" + shared-wind pair table" if info.get("pair_table") else ""),
                      "vgprs": info["vgprs"], "scratch_bytes": info["scratch_bytes"]},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy": achieved / HBM_COPY_GBS,
                        "traffic": cp["hbm_bytes"] if cp else None,
                        # traffic / algorithmic bytes: what the kernel moves over what the path must move (the source log of the
                        # one-block kernel: written once per source and farm, re-read by every later block in reach)
                        "traffic_ratio": (cp["hbm_bytes"] / algo_bytes) if cp else None,
                        "traffic_source": cp["source"] if cp else None,
                        # what the number is: bytes the L2 requested from / sent to the fabric — an upper bound on HBM
                        # bytes (Infinity Cache hits are in it: rocprofv3 exposes no MALL counter on this box)
                        "traffic_kind": ("fabric bytes (L2 <-> EA): " + cp["traffic_how"]) if cp else None,
                        # what the kernel actually moves (the source log of the one-block kernel on top of the
                        # algorithmic bytes), as a fraction of the HBM peak at this run's kernel time
                        "traffic_frac_of_peak": (cp["hbm_bytes"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if cp else None,
                        "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": algo_bytes,
                        "note": "path is VALU-bound (arithmetic intensity ~2 kFLOP/B): see valu_roofline"},
           "valu_roofline": {"bound": "valu_fp32", "achieved": valu_achieved, "peak": VALU_PEAK_LANEOPS,
                             "unit": "lane-ops/s", "frac": (valu_achieved / VALU_PEAK_LANEOPS) if cp else None,
                             "frac_is": "ISSUED slots: SQ_INSTS_VALU x 64 lanes per launch / kernel time / (256 CU x 4 SIMD x 32 "
                                        "lanes x 2.4 GHz); idle lanes on the triangle and per-wave redundancy are in it",
                             "insts_valu_per_launch": cp["insts_valu"] if cp else None,
                             "source": cp["source"] if cp else None,
                             # the ceiling this chip reaches with the kernel's waves per SIMD and its own share of transcendentals
                             # (measured issue rates: profiles/valu_practical.json, tools/valu_practical.py) — the nominal peak, one
                             # wave-instruction per SIMD every 2 cycles, is not reachable at two waves per SIMD
                             "practical_peak": vp["peak"] if vp else None,
                             "frac_of_practical": (valu_achieved / vp["peak"]) if (cp and vp) else None,
                             "practical_peak_how": vp["how"] if vp else None},
           # which of the two roofs binds: the fraction of each the kernel reaches (fabric traffic over the HBM peak against issued
           # VALU slots over the practical ceiling)
           "binding_roof": (None if not (cp and vp) else
                            ("valu" if valu_achieved / vp["peak"] >= cp["hbm_bytes"] / (kern_ms * 1e-3) / 1e9 / HBM_COPY_GBS else "fabric"))}
    if other_leg is not None:
        res[f"{other_leg['mode']}_scaling"] = {
            "value": other_leg["total"] * args.steps / other_leg["elapsed"], "unit": "farm-steps/s",
            "ms_per_step": other_leg["elapsed"] / args.steps * 1e3, "env_batch_per_gpu": other_leg["B"],
            "env_batch_total": other_leg["total"], "kernel_ms": other_leg["kern_ms"]}
    if venv_ms is not None:
        res["fused_env_step"] = {"ms_per_step": venv_ms, "env_steps_per_sec_per_gpu": B / (venv_ms * 1e-3),
                                 "outputs": "reward[B], yaw/wind_speed/wind_direction[B,N]"}
    if env_level:
        res["env_level"] = env_level
    if not args.no_cpu_baseline and world == 1:  # CPU baseline + accuracy sample: rank 0 of the single-GPU run only
        from oracle import c_oracle

        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import parity

        nthreads = min(c_oracle.max_threads(), effective_cpus())
        ycpu = ring[0][: min(B, 4096)].cpu().numpy().astype(np.float64)
        # accuracy sample: this very batch under the wind of the timed leg against the float64 oracle, under the per-farm
        # contract of tests/parity.py (strict on every farm the kernel did not flag itself)
        ns = min(256, ycpu.shape[0])
        ws_s, wd_s = w.get_wind()  # the wind the timed leg ended on: 8 m/s / 270 deg, the last direction of cfg5's sweep, or
        ref = c_oracle.farm_step_batch(lay["xcoords"], lay["ycoords"], ws_s[:ns], wd_s[:ns], ycpu[:ns], margin=True)  # a wind per farm
        w.set_risk_resolve(0)  # the float32 kernel on its own, with its flags: what the per-farm contract of tests/parity.py judges
        got = w.step(ring[0], out)
        flags_all = w.risk_flags()
        w.sync()
        w.set_risk_resolve(1)
        g = {k: v[:ns].cpu().numpy().astype(np.float64) for k, v in got.items()}
        sm = parity.summarize(g, ref, flags_all[:ns])
        perr = np.abs(g["power"] - ref["power"]) / np.maximum(ref["power"], 1e3)
        res["power_rel_err"] = {"max": float(perr.max()), "p999": float(np.quantile(perr, 0.999)),
                                "frac_gt_1e-4": float((perr > 1e-4).mean()),
                                "max_unflagged": sm["worst_unflagged"]["power"],
                                "flagged_farm_frac_sample": sm["n_flagged"] / sm["n"],
                                "flagged_farm_frac_batch": float((flags_all != 0).mean()),
                                "contract": parity.classify(sm), "n_mismatch_flagged": sm["n_mismatch_flagged"],
                                "wind_speed_rel_max": float((np.abs(g["wind_speed"] - ref["wind_speed"]) / ref["wind_speed"]).max()),
                                "wind_direction_abs_max_deg": float(np.abs(g["wind_direction"] - ref["wind_direction"]).max()),
                                "ti_abs_max": float(np.abs(g["load"][..., 0] - ref["load"][..., 0]).max()),
                                "sample": f"{ns} envs x {N} turbines vs float64 oracle (oracle-pinned; the reference "
                                          "pins only its yaw = 0 notebook vector, tests/golden/kat1_demo_notebook.json); "
                                          "flagged = farms with a nonzero WF_RISK_* flag (include/wfstep.h)"}
        if per_farm:  # accuracy of the per-farm-wind leg: a sample of the batch that holds every kind of farm — the first
            # 256, plus up to 256 of the farms the float32 kernel flagged — against the oracle, without and with the re-solve
            ws_pf, wd_pf = per_farm.pop("_wind")
            w.set_wind(ws_pf, wd_pf)
            w.set_risk_resolve(0)
            g0 = {k: v.cpu().numpy() for k, v in w.step(ring[0], out).items()}
            fl0 = w.risk_flags()
            w.set_risk_resolve(1)
            g1 = {k: v.cpu().numpy() for k, v in w.step(ring[0], out).items()}
            fl1 = w.risk_flags()
            w.set_wind(8.0, 270.0)
            pick = np.unique(np.concatenate([np.arange(min(256, B)), np.flatnonzero(fl0 != 0)[:256]]))
            yall = ring[0].cpu().numpy().astype(np.float64)
            refp = c_oracle.farm_step_batch(lay["xcoords"], lay["ycoords"], ws_pf[pick], wd_pf[pick], yall[pick], margin=True)
            s0 = parity.summarize({k: v[pick] for k, v in g0.items()}, refp, fl0[pick])
            e1 = parity.errors({k: v[pick] for k, v in g1.items()}, refp)
            ok1 = parity.within(e1, parity.TOL, N)
            per_farm["accuracy"] = {
                "sample": f"{len(pick)} farms ({int((fl0[pick] != 0).sum())} of them flagged by the float32 kernel) x {N} turbines vs the float64 oracle",
                "float32_only": {"contract": parity.classify(s0), "n_flagged": s0["n_flagged"], "n_mismatch_flagged": s0["n_mismatch_flagged"],
                                 "n_bad_unflagged": s0["n_bad_unflagged"], "worst_flagged": s0["worst_flagged"],
                                 "worst_unflagged": s0["worst_unflagged"]},
                "with_float64_resolve": {"n_outside_tol": int((~ok1).sum()), "flags_left": int((fl1 != 0).sum()),
                                         "worst": {k: float(v.max()) for k, v in e1.items()}},
                "tolerances": parity.TOL}
        # timing: calibrate on a small sample (second call: the first one pays the thread-pool start), then ~cpu_seconds of work
        for _ in range(2):
            t = time.perf_counter()
            c_oracle.farm_step_batch(lay["xcoords"], lay["ycoords"], 8.0, 270.0, ycpu[: 16 * nthreads], nthreads=nthreads)
            per_env = (time.perf_counter() - t) / (16 * nthreads)
        n = int(max(4 * nthreads, min(ycpu.shape[0], args.cpu_seconds / per_env)))
        reps = max(1, int(args.cpu_seconds / (per_env * n)))
        t = time.perf_counter()
        for _ in range(reps):
            c_oracle.farm_step_batch(lay["xcoords"], lay["ycoords"], 8.0, 270.0, ycpu[:n], nthreads=nthreads)
        dt = time.perf_counter() - t
        res["cpu_baseline"] = {"value": n * reps / dt, "unit": "farm-steps/s", "cores": nthreads, "kind": "port",
                               "sample": f"{reps} x {n} farm-steps of the same workload, C float64 oracle "
                                         f"(OpenMP over envs, {nthreads} threads; os.cpu_count()={os.cpu_count()}, "
                                         f"usable={effective_cpus()}), {dt:.1f} s"}
        # the same algorithm in FLORIS' own execution style (NumPy, one process, one core)
        from oracle import floris_gch_numpy as onp

        t = time.perf_counter()
        k = 0
        while time.perf_counter() - t < 3.0:
            onp.farm_step(lay["xcoords"], lay["ycoords"], 8.0, 270.0, ycpu[k % ycpu.shape[0]])
            k += 1
        res["cpu_baseline_numpy"] = {"value": k / (time.perf_counter() - t), "unit": "farm-steps/s", "cores": 1,
                                     "kind": "port", "sample": f"{k} farm-steps, NumPy float64 oracle, single process"}
    if per_farm or strict:
        res["extra"] = {}
        if strict:
            res["extra"]["headline_wind"] = strict
        if per_farm:
            per_farm.pop("_wind", None)
            res["extra"]["per_farm_wind"] = per_farm
    print(json.dumps(res), flush=True)
    w.close()


if __name__ == "__main__":
    main()
