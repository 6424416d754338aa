#!/usr/bin/env python
"""bench.py — prior-regularised global bundle adjustment on MI355X (BASELINE.json metric).

One step = one full BA solve (Ceres-default LM until convergence) of the synthetic 200-image /
150k-landmark scene with reprojection + log-depth prior blocks (BASELINE config "C3"), starting
from the same perturbed state every step (device-to-device reset, inside the timed region).
The problem is resident in HBM before the timed region starts.  With N > 1 ranks the landmarks
are sharded and the ranks exchange the reduced camera system with an RCCL all-reduce every LM
iteration.  `--scaling weak` (default): each rank owns the configuration's landmark count around
the same cameras (N x 150 k landmarks at C3).  `--scaling strong`: the configuration's OWN landmarks
are cut into N contiguous ranges balanced by residual blocks (`mpsfm_amd.dist.shard_problem`) — the form
BASELINE config 4 names ("1000 images / 800k points ... landmark-sharded across 8 GPUs":
`--config C4 --gpus 8 --scaling strong`); the replicated dense solve bounds it.

Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("MPSFM_BENCH_SCALING", "weak"),
                    help="weak: every rank draws the configuration's landmark count; strong: the configuration's landmarks are sharded")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg")
    ap.add_argument("--kernel-reps", type=int, default=20, help="launches used to time the sweep / dense kernels")
    return ap.parse_args()


def main():
    args = parse()
    import torch

    from mpsfm_amd import capi
    from mpsfm_amd.synthetic import CONFIGS, algorithmic_bytes_sweep, algorithmic_flops_sweep, make_config

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal aids (several ranks on ONE GPU, where RCCL refuses duplicate devices): MPSFM_BENCH_DEVICE pins the
    # device ordinal, MPSFM_BENCH_BACKEND=gloo sums the device buffers through gloo.  Not used by the driver.
    local_rank = int(os.environ.get("MPSFM_BENCH_DEVICE", local_rank))
    backend = os.environ.get("MPSFM_BENCH_BACKEND", "nccl")
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libmpsfm_hip has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    keep = None
    force_dist = os.environ.get("MPSFM_FORCE_DIST") == "1"  # exercise the RCCL hook with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.scaling == "strong" and world > 1:
        from mpsfm_amd.dist import shard_problem

        full, _ = make_config(args.config, seed=0, shard=0)  # the same problem on every rank, cut by landmark range
        prob, (lm_lo, lm_hi) = shard_problem(full, rank, world)
        del full
    else:
        prob, _ = make_config(args.config, seed=0, shard=rank)
    collective = None
    alive = []  # the hook's callback object and the options block must outlive the handle

    def make_handle(native: bool):
        opts = capi.default_options(device=local_rank, stream=torch.cuda.current_stream().cuda_stream)
        name = None
        if dist is not None:
            from mpsfm_amd.dist import hook_options, make_torch_allreduce, use_native_rccl

            if native:
                use_native_rccl(opts)
                name = "native RCCL (ncclAllReduce on the solver's stream)"
            else:
                fn, keep = make_torch_allreduce()
                hook_options(opts, fn)
                alive.append((fn, keep))
                name = f"torch.distributed hook ({backend})"
        alive.append(opts)
        return capi.BAHandle(prob, opts), name

    # default: the library's own RCCL communicator (no Python inside the LM loop); MPSFM_BENCH_COLLECTIVE=hook (or a non-RCCL
    # rehearsal backend) goes through the torch.distributed hook.  The native communicator has not run with more than one rank
    # on the development box (one GPU): its first solve is checked — it must succeed on every rank and every rank must report
    # the same cost — and the run falls back to the hook when it does not.
    want_native = dist is not None and backend == "nccl" and os.environ.get("MPSFM_BENCH_COLLECTIVE", "rccl") == "rccl"
    h = None
    if want_native:
        ok, cost = 1.0, 0.0
        try:
            h, collective = make_handle(True)
            if os.environ.get("MPSFM_BENCH_INJECT_NATIVE_FAILURE"):  # tests: the fallback below must carry the run
                raise RuntimeError("injected failure of the native RCCL path")
            cost = float(h.solve()["final_cost"])
            ok = 1.0 if np.isfinite(cost) and cost > 0.0 else 0.0
        except Exception as e:  # noqa: BLE001 - any failure of the unverified path selects the verified one
            print(f"[bench] rank {rank}: native RCCL path failed ({e!r}); falling back to the torch.distributed hook", file=sys.stderr)
            ok = 0.0
        chk = torch.tensor([ok, cost, -cost], dtype=torch.float64, device="cuda")
        dist.all_reduce(chk, op=dist.ReduceOp.MIN)
        agreed = bool(chk[0].item() == 1.0) and abs(chk[1].item() + chk[2].item()) <= 1e-9 * abs(chk[1].item())  # min(cost) == max(cost)
        if not agreed:
            if rank == 0:
                print("[bench] native RCCL path not confirmed on every rank: using the torch.distributed hook", file=sys.stderr)
            if h is not None:
                h.close()
            h = None
    if h is None:
        h, collective = make_handle(False)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        h.reset_state()
        return h.solve()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    sums = [step() for _ in range(args.steps)]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    iters = sum(s["num_iterations"] for s in sums)
    revals = sum(s["num_residual_evals"] for s in sums)  # already the all-rank total
    last = sums[-1]

    # ---- roofline of the two priced kernels, timed live with HIP events on the solver's stream
    h.reset_state()
    reps = max(1, args.kernel_reps)
    sweeps, parts = [], []
    for _ in range(reps + 2):
        sweeps.append(h.sweep_once(1e4))
        parts.append(h.sweep_parts())
    sweep_ms = float(np.mean(sweeps[2:]))
    sweep_parts = {k: float(np.mean([p[k] for p in parts[2:]])) for k in ("dense_ms", "reduce_ms", "general_ms")}
    sweep_parts.update({k: parts[-1][k] for k in ("dense_chunks", "general_chunks", "long_tracks", "reduce_parts")})
    dense_ms = float(np.mean([h.dense_solve_once() for _ in range(reps + 2)][2:]))
    n = h.reduced_dim
    plan = h.dense_plan()
    bytes_sweep = algorithmic_bytes_sweep(prob, s_blocks=plan["s_blocks"])
    flops_sweep = algorithmic_flops_sweep(prob)
    flops_dense = n**3 / 3.0 + 2.0 * n * n
    roof_sweep = {
        "kernel": "track sweep: k_track_sweep_dense + k_reduce_slabs (+ the general / long-track kernels where a problem has such chunks)",
        "bound": "hbm", "achieved": bytes_sweep / (sweep_ms * 1e-3) / 1e9, "peak": 8000.0,
        "unit": "GB/s", "traffic": None, "algorithmic_bytes": bytes_sweep, "avg_ms": sweep_ms,
        # HIP-event times of the three launches of one sweep (each carries ~5 us of event overhead; avg_ms spans all of them)
        "launches": sweep_parts,
        # the same launches against the fp64 vector peak (SURVEY 8d prices the sweep by bytes; by arithmetic intensity it sits on the
        # compute side of the ridge): algorithmic flops, not issued ones
        "fp64": {"algorithmic_flops": flops_sweep, "achieved_TFLOPs": flops_sweep / (sweep_ms * 1e-3) / 1e12, "peak_TFLOPs": 78.6,
                 "frac": flops_sweep / (sweep_ms * 1e-3) / 1e12 / 78.6},
    }
    roof_sweep["frac"] = roof_sweep["achieved"] / roof_sweep["peak"]
    # what the factorisation really issues on the matrix pipe: 32x32x32 tile products of the symbolic factor (2 * 32^3 each) plus
    # the 32-column panel factorisations; the dense count above stands for the reduced system the caller handed over
    issued_dense = 65536.0 * plan["tile_products"] + plan["tile_columns"] * (32.0**3 / 3.0 + 2.0 * 32.0**3)
    roof_dense = {
        "kernel": "k_chol_level (+k_assemble, back substitution): reduced camera system, level-scheduled tile Cholesky", "bound": "mfma", "achieved": flops_dense / (dense_ms * 1e-3) / 1e12,
        "peak": 78.6, "unit": "TFLOP/s", "traffic": None, "algorithmic_flops": flops_dense, "avg_ms": dense_ms, "n": n,
        # one dense solve = this many launches; avg_ms is the HIP-event time of the whole sequence, to be compared with
        # sum(launches x rocprofv3 AverageNs) from profiles/rNN_kernel_stats.csv
        "launches_per_solve": ({"k_assemble": 1, "k_chol_level": plan["levels"], "k_inv_y": 1} if plan["inverse_accumulators"] else
                               {"k_assemble": 1, "k_chol_level": plan["levels"], "k_back_level": plan["backsub_launches"]}),
        "plan": plan,
        # the factorisation only touches the tiles the symbolic factorisation marks (camera pairs that share landmarks, plus
        # fill): the flops actually issued are fewer than the dense count `algorithmic_flops` (n^3/3 + 2 n^2)
        # that `achieved` is quoted on
        "note": "achieved = dense-equivalent flops / time; the factorisation skips structurally zero tiles",
        "issued_flops_estimate": issued_dense, "issued_TFLOPs": issued_dense / (dense_ms * 1e-3) / 1e12,
        "issued_frac": issued_dense / (dense_ms * 1e-3) / 1e12 / 78.6,
    }
    roof_dense["frac"] = roof_dense["achieved"] / roof_dense["peak"]
    t_lin, t_den, t_upd = last["time_linearize_s"], last["time_dense_s"], last["time_update_s"]
    dominant = roof_dense if t_den > t_lin else roof_sweep
    other = roof_sweep if dominant is roof_dense else roof_dense
    # HBM bytes per launch of the track sweep from the committed PMC passes (profiles/rNN_pmc_traffic.json:
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE); null if absent
    import glob

    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if cands:
        try:
            tr = json.load(open(cands[-1]))
            roof_sweep["traffic"] = tr.get("track_sweep_bytes_per_sweep", tr.get("k_track_sweep_bytes_per_launch"))
            roof_sweep["traffic_source"] = os.path.basename(cands[-1])
        except Exception:  # noqa: BLE001
            pass

    # matrix-pipe utilisation of the factorisation kernel from the committed counter pass (profiles/rNN_pmc_mfma.json:
    # SQ_VALU_MFMA_BUSY_CYCLES over the chip's SIMD-cycles while the kernel ran); null if absent
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")))
    if cands:
        try:
            mf = json.load(open(cands[-1]))
            ks = (mf.get("bench_C3") or {}).get("mpsfm::k_chol_level") or (mf.get("bench_C3") or {}).get("mpsfm::k_chol_step")
            if ks:
                roof_dense["mfma_util_pmc"] = ks.get("mfma_util")
                roof_dense["mfma_util_source"] = os.path.basename(cands[-1])
        except Exception:  # noqa: BLE001
            pass

    out = {
        "metric": f"BA residual-evals/sec (LM iterations/sec alongside), {CONFIGS[args.config][0]} imgs / {CONFIGS[args.config][1] // 1000}k pts "
                  + ("prior-BA" if CONFIGS[args.config][2] else "reprojection-only BA"),
        "value": revals / dt,
        "unit": "residual-block evals/s",
        "lm_iterations_per_s": iters / dt,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"{args.config}: {CONFIGS[args.config][0]} cameras, {CONFIGS[args.config][1]} landmarks per rank, "
                         f"{prob.n_obs} reprojection + {prob.n_dobs} log-depth blocks per rank, SoftL1/Cauchy, Ceres-default LM"
                         if not (args.scaling == "strong" and world > 1) else
                         f"{args.config}: {CONFIGS[args.config][0]} cameras, {CONFIGS[args.config][1]} landmarks IN TOTAL cut into {world} landmark "
                         f"ranges (rank 0: {prob.n_pts} landmarks, {prob.n_obs} reprojection + {prob.n_dobs} log-depth blocks), SoftL1/Cauchy, Ceres-default LM"),
            "parallelism": ((f"landmark-sharded x{world} ({args.scaling}: " + ("the configuration's landmarks split" if args.scaling == "strong" else "the configuration's landmark count per rank")
                             + (f"), {collective}" if collective else ")")) if (world > 1 or collective) else "single GPU"),
            "residual_blocks_total": last["num_residual_blocks"],
        },
        "solve": {
            "lm_iterations": last["num_iterations"], "successful": last["num_successful_steps"],
            "initial_cost": last["initial_cost"], "final_cost": last["final_cost"], "termination": last["termination"],
            "device_ms": {"track_sweep": 1e3 * t_lin, "dense_solve": 1e3 * t_den, "update_sweep": 1e3 * t_upd,
                          "total": 1e3 * last["time_total_s"]},
        },
        "roofline": dominant,
        "roofline_other": other,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # CPU baseline and side measurements at N=1 only
        out["cpu_baseline"] = cpu_baseline(args, last)
        try:
            out["extras"] = {"integration_290x387": integration_leg(local_rank)}
        except Exception as e:  # noqa: BLE001 - never lose the headline line to the side measurement
            out["extras"] = {"integration_290x387": {"error": repr(e)}}
        try:
            out["extras"]["one_shot_ms"] = one_shot_leg(local_rank)
        except Exception as e:  # noqa: BLE001
            out["extras"]["one_shot_ms"] = {"error": repr(e)}
        try:  # what one Optimizer.ba() call pays from host buffers (device table build + uploads + solve + read-back): never `value`
            out["one_shot_ms"] = {k: v.get("min") for k, v in out["extras"]["one_shot_ms"].items() if isinstance(v, dict)}
        except Exception:  # noqa: BLE001
            pass
        try:
            out["extras"]["other_configs_one_gpu"] = other_configs_leg(local_rank)
        except Exception as e:  # noqa: BLE001
            out["extras"]["other_configs_one_gpu"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def one_shot_leg(device):
    """mpsfm_ba_solve from host buffers (table build + upload + solve + download), what one Optimizer.ba() call pays:
    a local bundle (12 cameras / 4 k landmarks) and the C3 problem.  PCIe-inclusive, never the headline value."""
    from mpsfm_amd import capi
    from mpsfm_amd.synthetic import make_config, make_scene

    out = {}
    for name, base in (("local_12cam_4kpts", make_scene(12, 4000, True, seed=3)[0]), ("C3", make_config("C3")[0])):
        ts = []
        for _ in range(4):
            p = base.copy()
            t0 = time.perf_counter()
            s = capi.ba_solve(p, capi.default_options(device=device))
            ts.append(1e3 * (time.perf_counter() - t0))
        out[name] = {"min": min(ts[1:]), "lm_iterations": s["num_iterations"]}
    return out


def other_configs_leg(device):
    """Resident solves of the larger BASELINE configurations on ONE GPU (C4: 1000 cameras / 800 k landmarks, the shape the
    8-GPU run shards; C5 in shape: 300 cameras / 400 k landmarks): side measurements, never the headline value."""
    from mpsfm_amd import capi
    from mpsfm_amd.synthetic import make_config

    out = {}
    for name in ("C5", "C4"):
        prob, _ = make_config(name)
        with capi.BAHandle(prob, capi.default_options(device=device)) as h:
            ss = []
            for _ in range(3):
                h.reset_state()
                ss.append(h.solve())
            s = min(ss[1:], key=lambda x: x["time_total_s"])
            out[name] = {"ms_per_solve": 1e3 * s["time_total_s"], "lm_iterations": s["num_iterations"],
                         "lm_iterations_per_s": s["num_iterations"] / s["time_total_s"],
                         "residual_block_evals_per_s": s["num_residual_evals"] / s["time_total_s"],
                         "device_ms": {"track_sweep": 1e3 * s["time_linearize_s"], "dense_solve": 1e3 * s["time_dense_s"],
                                       "update_sweep": 1e3 * s["time_update_s"]},
                         "residual_blocks": s["num_residual_blocks"], "plan": h.dense_plan()}
    return out


def integration_leg(device):
    """Row f1 (depth-from-normals integration) at the reference's map size: HIP vs the SciPy oracle."""
    from mpsfm_amd import capi
    from mpsfm_amd.synthetic_maps import make_maps
    from oracle import integration_oracle as IO

    maps = make_maps(290, 387, seed=8, n_sparse=1500)
    nu = maps["normals_uncertainty"]
    nvar = np.stack([nu[..., 0, 0], nu[..., 1, 1], nu[..., 2, 2]], -1)
    args = (maps["depth_prior"], maps["depth_uncertainty"], maps["valid"], maps["normals"], nvar, maps["depth_init"], maps["K"],
            maps["kps"], maps["depth3d"], maps["zvars3d"])
    capi.integrate_depth(*args, device=device)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        d, s, *_ = capi.integrate_depth(*args, device=device)
        t.append(time.perf_counter() - t0)
    keys = ("depth_prior", "depth_uncertainty", "valid", "normals", "normals_uncertainty", "depth_init", "K", "kps", "depth3d", "zvars3d")
    t0 = time.perf_counter()
    do, _, _, info = IO.integrate(IO.IntInputs(**{k: maps[k] for k in keys}))
    t_cpu = time.perf_counter() - t0
    n_cg = sum(s["cg_iters"])
    return {"hip_wall_ms": 1e3 * min(t), "hip_device_ms": s["ms"], "scipy_oracle_ms": 1e3 * t_cpu, "irls": s["irls_iterations"],
            "cg_iterations": s["cg_iters"], "cg_iterations_oracle": info["cg_iters"], "max_rel_diff": float(np.max(np.abs(d / do - 1))),
            "algorithmic_GBps": 15 * 8 * 290 * 387 * n_cg / (s["ms"] * 1e-3) / 1e9}


def cpu_baseline(args, gpu_summary):
    """The oracle (CPU restatement of the Ceres path, OpenMP over all host cores) on the same
    single-rank problem: whole solves until the time budget is spent."""
    from mpsfm_amd.synthetic import make_config
    from oracle import cpu_oracle as O

    # size the OpenMP team to the CPUs this box really grants (affinity / cgroup quota), not to
    # the host's core count: oversubscribed spin locks make the oracle many times slower
    cores = int(os.environ.get("MPSFM_ORACLE_THREADS", "0")) or min(O.available_cpus(), 32)
    O.set_num_threads(cores)
    base, _ = make_config(args.config, seed=0, shard=0)
    t_used, n_solves, iters, revals, final = 0.0, 0, 0, 0, None
    while n_solves < 1 or (t_used < args.cpu_seconds and n_solves < 8):
        p = base.copy()
        t0 = time.perf_counter()
        s = O.solve(p)
        t_used += time.perf_counter() - t0
        n_solves += 1
        iters += s["num_iterations"]
        revals += s["num_residual_evals"]
        final = s["final_cost"]
    rel = abs(final - gpu_summary["final_cost"]) / final if args.gpus == 1 else None
    return {
        "value": revals / t_used, "unit": "residual-block evals/s", "lm_iterations_per_s": iters / t_used,
        "cores": cores, "kind": "port", "host_cpus_visible": os.cpu_count(),
        "sample": f"{n_solves} full solve(s) of the same {args.config} problem ({iters // n_solves} LM iterations each), {t_used:.1f} s",
        "final_cost": final, "final_cost_rel_diff_vs_gpu": rel, "ms_per_solve": 1e3 * t_used / n_solves,
    }


if __name__ == "__main__":
    main()
