#!/usr/bin/env python3
"""MPC steps/s of the Koopman online-updated MPC hot path (lift -> RLS-EDMD update -> condensed QP ->
box-QP) on MI355X, BASELINE.json configs[1]: Duffing, 20-dim MLP lift, N = 20, batch = 4096
trajectories per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
           --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 20

A "step" is one closed-loop control step of every trajectory: kmpc_step (lift of the current state,
RLS update with the previous transition, condensed-QP build, exact box-QP solve) followed by the RK4
plant on the device; the whole loop is enqueued from C++ (kmpc_rollout), inputs are resident in HBM.
Multi-GPU: trajectories are independent, every rank owns its own 4096 (weak scaling), no collective on
the step path; the timed region is bracketed by barrier + synchronize and the MAX over ranks is used.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (step_kernel): algorithmic bytes
(SURVEY.md 8d formula, kmpc_algorithmic_bytes_per_step) x trajectories per launch / its average
duration measured with HIP events on the launch stream.  `cpu_baseline` times the NumPy oracle run the
way the reference runs (per-trajectory Python loop, SciPy L-BFGS-B on the shooting cost,
duffing.py:857-859) on a bounded sample of the same workload: one worker process per host core of the
box's CPU share (16), forked before the GPU is touched; the single-core figure is reported beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "koopman-online-updated-mpc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def _cpu_worker(args):
    """One host core: per-trajectory loop, lift -> solve -> plant -> RLS in the reference's order (duffing.py:823-1012)
    through the oracle, for `budget_s` seconds.  Returns (trajectory-steps done, seconds)."""
    os.environ["OMP_NUM_THREADS"] = "1"  # (one core means one core: no BLAS threads behind the loop)
    weights, A0, B0, C0, x0s, L, N, budget_s, solver, steps_per_traj = args
    from oracle import koopman_oracle as ko

    try:
        import threadpoolctl

        threadpoolctl.threadpool_limits(1)
    except Exception:
        pass
    lift = lambda x: ko.mlp_lift(weights, x)
    r = np.tile(np.array([[1.0], [0.0]]), (1, N))
    t0 = time.perf_counter()
    done = 0
    for t in range(x0s.shape[1]):
        ctl = ko.OracleController(lift, L, 2, N, -2.0, 2.0, A0, B0, C0, solver=solver)
        x = x0s[:, t].copy()
        for k in range(steps_per_traj):
            u, _, _ = ctl.step(x, r)
            x = ko.plant_step("duffing", x, u)
            done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return done, time.perf_counter() - t0


def cpu_baseline(weights, x0s, L, N, budget_s):
    """The reference's path on the host cores, timed BEFORE this process touches the GPU (the workers are forked):
    the NumPy oracle run the way the reference runs -- per-trajectory Python loop, SciPy L-BFGS-B on the shooting cost
    (duffing.py:857-859) -- on one core and on the box's CPU share (one trajectory stream per core, trajectories are
    independent); the exact-QP variant of the oracle on one core beside it.  Bounded by `budget_s` per leg."""
    import multiprocessing as mp

    from koopmpc.synth import offline_edmd
    from oracle import koopman_oracle as ko

    A0, B0, C0 = offline_edmd(lambda X: ko.mlp_lift(weights, X))  # the reference's one-off fit (duffing.py:152-177)
    spt = 8
    cores = max(1, min(16, os.cpu_count() or 1))  # a one-GPU box comes with 16 host cores
    one = _cpu_worker((weights, A0, B0, C0, x0s, L, N, budget_s / 3.0, "lbfgsb", spt))
    exact = _cpu_worker((weights, A0, B0, C0, x0s, L, N, budget_s / 3.0, "exact", spt))
    per = max(1, x0s.shape[1] // cores)
    jobs = [(weights, A0, B0, C0, x0s[:, i * per:(i + 1) * per], L, N, budget_s, "lbfgsb", spt) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    done = sum(d for d, _ in res)
    return {"all": (done / wall, done, wall, cores), "one": (one[0] / one[1], one[0], one[1]),
            "exact": (exact[0] / exact[1], exact[0], exact[1]), "spt": spt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="trajectories per GPU")
    ap.add_argument("--L", type=int, default=20)
    ap.add_argument("--N", type=int, default=20)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--cold-start", action="store_true",
                    help="start every QP at clip(0) like the reference instead of at the previous minimiser")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--spin-seconds", type=float, default=1.0,
                    help="untimed GPU activity on a scratch copy of the workload before the warm-up, so that the "
                         "clocks have left their idle state when the W warm-up steps start")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d "
                     "(WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    # ---- CPU baseline first (rank 0 at N = 1 only): its worker processes are forked before anything touches the GPU
    cpu = None
    if world == 1 and args.cpu_seconds > 0:
        from koopmpc.synth import initial_states as _init, random_mlp_weights as _rw

        cpu = cpu_baseline(_rw(2, 100, 3, args.L, seed=2024), _init(args.batch, seed=101), args.L, args.N, args.cpu_seconds)
    if not torch.cuda.is_available():
        if cpu is not None:
            print("cpu_baseline (no GPU here, nothing else measured): %s" % json.dumps(cpu), file=sys.stderr)
        sys.exit("bench.py needs a GPU: the hot path has no CPU implementation")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    from koopmpc import KoopmanMPC, max_over_ranks
    from koopmpc.synth import initial_states, offline_data, random_mlp_weights

    L, N, B = args.L, args.N, args.batch
    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    weights = random_mlp_weights(2, 100, 3, L, seed=2024)
    mpc = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=weights, dtype=dtype, threads=args.threads, device=dev,
                     cold_start=args.cold_start)
    # one-off offline EDMD fit (duffing.py:152-177) on the device: lift, MFMA Gram sums, p x p solve; identical on
    # every rank; the fitted model is handed to every trajectory
    A0, B0, C0 = [t.cpu().numpy() for t in mpc.offline_fit(*offline_data())]
    x0 = initial_states(B, seed=101 + rank)
    X = torch.tensor(x0, dtype=dtype, device=dev).contiguous()
    r = torch.tensor(np.tile(np.array([[1.0], [0.0]]), (1, N)), dtype=dtype, device=dev)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- warm-up (untimed), then EXACTLY --steps timed steps
    mpc.rollout("duffing", X, r, args.warmup, step0=0)
    torch.cuda.synchronize(dev)
    replay = args.steps <= 4096
    sd0, X0 = mpc.state_to(), X.clone()  # device-side snapshot of the state the timed region starts from
    # ---- bring the GPU out of its idle power state: a scratch controller runs the timed region's own steps from
    #      the snapshot (same launches as the timed one, so a rocprofv3 --stats average over the process is
    #      comparable with the number reported below); it does not touch the workload's controller
    if args.spin_seconds > 0:
        scratch = KoopmanMPC(n=2, L=L, N=N, batch=B, weights=weights, dtype=dtype, threads=args.threads, device=dev,
                             cold_start=args.cold_start)
        Xs = X.clone()
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spin_seconds:
            scratch.state_from(sd0)
            Xs.copy_(X0)
            scratch.rollout("duffing", Xs, r, args.steps, step0=args.warmup)
            torch.cuda.synchronize(dev)
        del scratch, Xs
    sync_all()
    t0 = time.perf_counter()
    mpc.rollout("duffing", X, r, args.steps, step0=args.warmup)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = max_over_ranks(time.perf_counter() - t0, device=dev)
    worst_status = int(mpc.status.max().item())
    newton_per_step = float(mpc.iters.double().mean().item()) / max(1, args.steps)
    newton_max = int(mpc.iters.max().item())

    # ---- kernel durations for the roofline: the timed region's steps replayed from the snapshot (same
    #      states, same models, bitwise the same controls) with HIP events around every lift / step kernel on
    #      the launch stream.  Kept out of the timed pass itself: the event packets cost ~10 % of a step.
    X_timed = X.clone()
    fused = mpc.rollout_is_fused()
    if replay:
        step0 = args.warmup
        nprof = args.steps
    else:
        step0 = args.warmup + args.steps
        nprof = 1000
    # (the first replay follows host-side work -- status read-back, snapshot restore -- and runs at a lower
    #  clock; it is discarded, the next two are averaged)
    for rep in range(3 if replay else 1):
        if replay:
            mpc.state_from(sd0)
            X.copy_(X0)
        if rep == (1 if replay else 0):
            mpc.profile(True)
        mpc.rollout("duffing", X, r, nprof, step0=step0)
    torch.cuda.synchronize(dev)
    pr = mpc.profile_read()
    mpc.profile(False)
    if replay and not torch.equal(X_timed, X):
        sys.exit("bench.py: the replayed region did not reproduce the timed region's states")
    # fused roll-out: ONE launch covers all nprof steps (lift inside); otherwise one lift + one step kernel per step
    steps_per_launch = nprof if fused else 1
    launch_ms = pr["step_ms"] / max(1, pr["count"]) * steps_per_launch
    lift_ms = pr["lift_ms"] / max(1, pr["count"])
    bytes_per_traj = mpc.algorithmic_bytes_per_step()
    bytes_per_launch = bytes_per_traj * B * steps_per_launch
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    x_ok = bool(torch.isfinite(X).all().item())

    if dist is not None:
        flag = torch.tensor([worst_status, 0 if x_ok else 1], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        worst_status, x_ok = int(flag[0].item()), int(flag[1].item()) == 0

    # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,
    # own-pattern calibration): collected offline with tools/one_phase.py, committed in profiles/
    traffic, traffic_note = None, "not collected for this configuration"
    tp = os.path.join(ROOT, "profiles", "r1_traffic_fused.json" if fused else "r1_traffic.json")
    if os.path.exists(tp) and (L, N, B, args.dtype) == (20, 20, 4096, "f64"):
        tj = json.load(open(tp))
        if not fused or tj.get("steps_per_launch") == steps_per_launch:
            traffic = tj["traffic_bytes_per_launch"]
            traffic_note = "bytes per launch, rocprofv3 PMC passes recorded in profiles/%s (FETCH x %.3f own-pattern calibration + WRITE)" % (
                os.path.basename(tp), tj["fetch_calibration"])

    if rank == 0:
        total = B * world
        out = {
            "metric": "MPC steps/s (lift+EDMD-update+QP, N=%d, %d-dim lift)" % (N, L),
            "value": total * args.steps / dt,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": "BASELINE cfg2: Duffing closed loop, %d-dim MLP lift (2-100-100-100-%d, random init seed 2024), "
                            "horizon N=%d, box +-2, %d trajectories per GPU x %d GPU(s), per-trajectory RLS, "
                            "RK4 plant on device, parameter switch at step 102; arithmetic in %s (the config line names fp32; the "
                            "reference computes in float64 and the 1e-6 bar on u needs it, DESIGN.md 4.1)" % (L, L, N, B, world, args.dtype),
                "global_batch": total,
                "parallelism": "trajectory-sharded x%d, no collective on the step path" % world,
                "qp": "exact box-QP (projected Newton), %s; mean Newton solves/step %.2f, worst trajectory total %d"
                      % ("each solve started at clip(0) like the reference" if args.cold_start else
                         "each solve started at the previous minimiser (the reference restarts at zeros: same minimiser, more work)",
                         newton_per_step, newton_max),
                "worst_qp_status": worst_status,
                "finite": x_ok,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("rollout_kernel (lift + RLS + condense + QP + plant, all %d steps in one launch), %d workgroups x 1024 threads"
                           % (steps_per_launch, (B + 15) // 16)) if fused else
                          ("step_kernel (RLS + condense + QP), %d blocks x %d threads"
                           % (B, mpc.cfg.threads or (256 if (L + 1) ** 2 > 2048 or N * N > 2048 else 64))),
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_note": traffic_note,
                "algorithmic_bytes_per_trajectory_step": bytes_per_traj,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "steps_per_launch": steps_per_launch,
                "avg_kernel_ms": launch_ms,
                "avg_lift_kernel_ms": lift_ms,
            },
        }
        if cpu is not None:
            v, done, secs, cores = cpu["all"]
            v1, done1, secs1 = cpu["one"]
            ve, donee, secse = cpu["exact"]
            out["cpu_baseline"] = {
                "value": v,
                "unit": "steps/s",
                "cores": cores,
                "kind": "port",
                "sample": "%d worker processes (one per core), each the first trajectories of its slice of the same workload x %d "
                          "closed-loop steps: %d trajectory-steps in %.1f s; NumPy oracle with SciPy L-BFGS-B exactly as "
                          "duffing.py:857-859, one BLAS thread per worker; host reports %d cores"
                          % (cores, cpu["spt"], done, secs, os.cpu_count()),
                "single_core_value": v1,
                "single_core_sample": "%d trajectory-steps in %.1f s" % (done1, secs1),
                "exact_qp_variant_single_core_value": ve,
            }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
