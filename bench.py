#!/usr/bin/env python
"""bench.py -- SparseVFC EM-iteration throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

A "step" is ONE full SparseVFC EM iteration (E-step, weighted Gram + rhs, all-reduce, Cholesky solve, field
application, sigma^2 / gamma update - SURVEY.md Appendix A step 5) over the whole synthetic point cloud, with the
inputs resident in HBM.  Workload: BASELINE config 4 = 8 M cells, M = 3000 control points, float32 cells; it fits one
MI355X, so the same total problem is run at every N (cells block-sharded across ranks: strong scaling).
Rank 0 prints ONE JSON line.  Extra objects: ``roofline`` (dominant kernel = the MFMA Gram kernel, timed with HIP
events on its launch stream; ``traffic`` = the PMC figure parsed from the committed ``profiles/r04_pmc_traffic.json``),
``solve`` (the coefficient solve: path, Jacobi sweeps, ms, MFMA-tile TFLOP/s), ``f64`` (the SAME workload in float64
mode - the mode the 1e-5 parity clause is about - with its own roofline; N = 1 only), ``con_k`` (the materialised-kernel
HBM-write bandwidth), ``eval`` (the fused evaluator kernel and the wall time of Jacobian + curl on a 64^3 grid at the API), ``cpu_baseline`` (the float64 NumPy oracle on the host cores at N_cpu = 200 k and 100 k cells; its
``value`` is the rate its fitted t(N) = a N + b gives at the bench's own cell count; N = 1 only) and ``parity`` (the GPU
engine, float64 and float32, on exactly the 100 k-cell arrays of that CPU sample against the oracle's field after the
same 10 EM iterations, with the oracle's own lstsq-vs-eigh noise floor beside it; N = 1 only).  ``config`` (the object the
driver parses) also carries the figures the other BASELINE configurations are judged by: ``f64_value`` / ``f64_frac`` (the
reference-width run of the headline workload), ``c3_ms_per_step`` / ``c3_gram_frac`` (2 M x 2000), ``c2_ms_per_em_step``,
``c5_organ_ms_per_em_step``, ``c5_32_organs_wall_s`` (all 32 organs of config 5 through four streams of the one GPU),
``eval_frac`` (the fused evaluator kernel against the float64 VALU peak), ``solve_avg_ms`` / ``solve_warm_host_ms`` and ``rccl_ranks`` (ncclCommCount
of the communicator the step's collectives ran on).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "spateo-release_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X datasheet: FP64 matrix (v_mfma_f64_16x16x4_f64) dense peak
PEAK_F64_VALU_TFLOPS = 78.6   # MI355X datasheet: FP64 vector peak (the evaluators' arithmetic)
EVAL_F64_FLOP_PER_PAIR = 24.0 # kernel value (3 sub, 3 fma-chain, exp2 ~ 1) + 9 Jacobian + 3 field FMAs per (query, control point)
PEAK_HBM_GBPS = 8000.0
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")


def pmc_traffic(dtype, gram_mode, cached_u, gpus, cells_per_rank, ctrl):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE, gfx950 corrections applied; profiles/r04_pmc/, profiles/r03_pmc/ and profiles/r02_hbm_traffic_pmc.md explain the entries).  None if this
    configuration was not measured."""
    try:
        with open(PMC_TRAFFIC_FILE) as f:
            entries = json.load(f)["entries"]
    except (OSError, ValueError, KeyError):
        return None, None
    for e in entries:
        if (e["dtype"], e["gram_mode"], bool(e["cached_u"]), int(e["cells_per_rank"]), int(e["ctrl"])) == \
                (dtype, gram_mode, bool(cached_u), int(cells_per_rank), int(ctrl)):
            return float(e["fetch_bytes"]) + float(e["write_bytes"]), e.get("source")
    return None, None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def measured_traffic(cells, ctrl, dtype, lambda_, timeout_s=150):
    """HBM bytes per tile stage (= per EM iteration) of the dominant kernel, MEASURED IN THIS RUN: two child processes of
    this script (`--traffic-child`: the same workload, two EM iterations, jitter-Cholesky solve so that the eigensolver's
    launches do not slow the counter collection) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (one counter group
    per pass, --kernel-trace only), summed over the launches of gram_cached_kernel and divided by the iterations.
    Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: the counters are in KB; FETCH_SIZE reports
    half the bytes of wide coalesced streaming reads (x 2); WRITE_SIZE as reported.  None if rocprofv3 is unavailable or a
    pass fails (the caller then falls back to the committed profile and says so)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    # this process is itself running under a profiler / tool library: a nested counter pass would inherit it - skip
    if os.environ.get("HSA_TOOLS_LIB") or any(k_.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")) for k_ in os.environ):
        log("[bench] running under a profiler: no nested PMC passes (roofline.traffic from the committed profile)")
        return None
    iters = 2
    got = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="mvf_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "-d", tmp, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--traffic-child", str(iters), "--cells", str(cells), "--ctrl", str(ctrl),
                   "--dtype", dtype, "--lambda_", str(lambda_)]
            t0 = time.perf_counter()
            pr = subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL,
                                stderr=subprocess.PIPE, text=True)
            dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
            if pr.returncode != 0 or not dbs:
                log(f"[bench] PMC pass {ctr} failed (rc {pr.returncode}): {pr.stderr[-400:]}")
                return None
            con = sqlite3.connect(dbs[0])
            row = con.execute("select count(distinct dispatch_id), sum(value) from counters_collection where kernel_name "
                              "like '%gram_cached_kernel%' and counter_name = ?", (ctr,)).fetchone()
            con.close()
            if not row or not row[0]:
                log(f"[bench] PMC pass {ctr}: no gram_cached_kernel rows")
                return None
            got[ctr] = {"launches": int(row[0]), "raw_kb": float(row[1]), "pass_s": time.perf_counter() - t0}
        except Exception as exc:  # noqa: BLE001 - never let the counter pass take the bench line down
            log(f"[bench] PMC pass {ctr} failed: {exc!r}")
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    fetch = got["FETCH_SIZE"]["raw_kb"] * 1024.0 * 2.0 / iters
    write = got["WRITE_SIZE"]["raw_kb"] * 1024.0 / iters
    return {"bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
            "launches_per_iteration": got["FETCH_SIZE"]["launches"] / iters,
            "pass_seconds": [got["FETCH_SIZE"]["pass_s"], got["WRITE_SIZE"]["pass_s"]],
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace "
                      "only) over child processes running the same workload; counters in KB, FETCH_SIZE x 2 (gfx950: half "
                      "the bytes of wide streaming reads are tallied), summed over the tile-stage launches of one EM iteration"}


def _cpu_budget():
    """CPUs this process may actually use: the scheduler affinity capped by the container's CFS quota (cgroup v2 cpu.max, v1
    cpu.cfs_quota_us).  The GPU boxes of this pool report 256 CPUs and hold a quota of 16: thread pools sized by the CPU count
    (128 BLAS / torch threads) exhaust it within a scheduler period and the whole process is throttled for the rest of it."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / per
        except Exception:
            quota = None
    return n, quota, max(1, min(n, int(quota)) if quota else n)


def _throttle_stats():
    try:
        with open("/sys/fs/cgroup/cpu.stat") as fh:
            d = dict(line.split() for line in fh.read().strip().splitlines())
        return {"nr_periods": int(d.get("nr_periods", 0)), "nr_throttled": int(d.get("nr_throttled", 0)),
                "throttled_s": int(d.get("throttled_usec", 0)) / 1e6}
    except Exception:
        return None


def _eigh_solver(lhs, rhs, method=None):
    """The reference's solve with its LAPACK driver swapped for a mathematically identical one (truncated symmetric
    eigendecomposition, same eps * max|lambda| cut-off as gelsd): oracle-vs-oracle deviation = the reference noise floor."""
    w, q = np.linalg.eigh((lhs + lhs.T) / 2)
    keep = np.abs(w) > np.finfo(float).eps * np.abs(w).max()
    return (q[:, keep] / w[keep]) @ (q[:, keep].T @ rhs)


PARITY_STEPS = 10  # EM iterations of the parity leg (>= 10; the timing legs keep their 3)


def _cpu_steps(M, lambda_, n_cpu, steps, keep=False):
    """`steps` EM iterations of the float64 oracle on the C4 generator at n_cpu cells.  keep=True also returns the
    arrays and the oracle's state after the last step (the parity leg runs the GPU engine on exactly these arrays) and
    the same trajectory with the LAPACK driver swapped (the reference noise floor of this sample)."""
    from oracle import sparsevfc_oracle as svo
    from spateo_amd._synthetic import make_config

    X, V, _ = make_config("C4", N=n_cpu)
    valid, Xv, Yv, idx, ctrl, beta = svo.sparsevfc_setup(X, V, M=M, seed=0)
    K = svo.con_K(ctrl, ctrl, beta)
    t1 = time.perf_counter()
    U = svo.con_K(Xv, ctrl, beta)
    t_conk = time.perf_counter() - t1
    N, D = Yv.shape
    kw = dict(a=5, lambda_=lambda_, minP=1e-5, theta=0.75, lstsq_method="scipy")

    def trajectory(timed):
        Vc, C = np.zeros((N, D)), np.zeros((len(ctrl), D))
        s2, gamma, E, P = np.sum(Yv**2) / (N * D), 0.9, 1, None
        ts = []
        for _ in range(steps):
            t2 = time.perf_counter()
            P, E, tecr, C, Vc, s2, gamma = svo.em_step(U, K, Yv, Vc, C, s2, gamma, E, **kw)
            ts.append(time.perf_counter() - t2)
        return (ts if timed else None), dict(V=Vc, sigma2=s2, P=P, E=E)

    ts, state = trajectory(True)
    rec = (N, len(ctrl), float(np.median(ts)), t_conk, U.nbytes)
    if not keep:
        return rec, None
    orig = svo.lstsq_solver
    svo.lstsq_solver = _eigh_solver
    try:
        _, alt = trajectory(False)
    finally:
        svo.lstsq_solver = orig
    vmax = np.abs(state["V"]).max()
    dP = np.abs(alt["P"] - state["P"])
    floor = {"V": float(np.abs(alt["V"] - state["V"]).max() / vmax),
             "sigma2": float(abs(alt["sigma2"] - state["sigma2"]) / state["sigma2"]),
             "P": float(dP.max()), "P999": float(np.quantile(dP, 0.999)),
             "E": float(abs(alt["E"] - state["E"]) / abs(state["E"]))}
    return rec, dict(Xv=Xv, Yv=Yv, ctrl=ctrl, beta=float(beta), steps=steps, state=state, floor=floor)


def cpu_baseline(M, lambda_, n_cpu, n_target, steps=3):
    """The float64 NumPy/SciPy oracle (kind "port": dynamo is not installable) on a bounded sample of the SAME
    workload: the C4 generator at n_cpu cells and at n_cpu / 2 with the same M, median of `steps` EM steps each
    (N_cpu = 200 k at M = 3000).  The step time is t(N) = a N + b (b = the O(M^3) lstsq, which
    does not grow with N), fitted through the two sizes; `value` is the rate that fit gives at the bench's own cell count
    n_target - the honest denominator for a speed-up - and the raw sample rates are kept beside it.
    Returns (record, parity sample = the half-size arrays with the oracle's state after `steps` iterations)."""
    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    (N, Mc, t_step, t_conk, ubytes), _ = _cpu_steps(M, lambda_, n_cpu, steps)
    # the half-size run doubles as the oracle side of the parity leg: PARITY_STEPS iterations (median step time of all)
    (Nh, _, t_half, _, _), sample = _cpu_steps(M, lambda_, n_cpu // 2, max(steps, PARITY_STEPS), keep=True)
    a = (t_step - t_half) / (N - Nh)  # seconds per cell (the part that scales with N)
    b = t_step - a * N                # the N-independent part (lstsq of the M x M system)
    t_target = a * n_target + b if a > 0 else None
    rec = {
        "value": (n_target / t_target) if t_target else N / t_step,
        "unit": "cells/s",
        "cores": int(threads),
        "host_cpus": os.cpu_count(),
        "cfs_quota_cpus": _cpu_budget()[1],
        "kind": "port",
        "sample": f"float64 NumPy oracle (cdist+exp con_K, U.T*repmat(P) temporary, scipy.linalg.lstsq), C4 generator at "
                  f"N_cpu={N} and {Nh} cells, M={Mc}, median of {steps} / {max(steps, PARITY_STEPS)} EM steps "
                  f"({t_step:.2f} / {t_half:.2f} s/step); "
                  f"value = {n_target} cells / t({n_target}) of the fit t(N) = a N + b through the two sizes (the lstsq "
                  f"constant b amortised as it would be at the bench's size); con_K {t_conk:.2f} s = "
                  f"{ubytes / t_conk / 1e9:.2f} GB/s of output",
        "extrapolated_to_cells": int(n_target),
        "extrapolated_s_per_step": t_target,
        "sample_value": N / t_step,
        "ms_per_step": 1e3 * t_step,
        "half_sample": {"cells": Nh, "ms_per_step": 1e3 * t_half, "value": Nh / t_half},
        "linearity_ratio": (Nh / t_half) / (N / t_step),
        "fit_seconds_per_cell": a,
        "fit_constant_seconds": b,
        "asymptotic_cells_per_s": (1.0 / a) if a > 0 else None,
    }
    return rec, sample


def parity_on_sample(sample, lambda_, device):
    """The GPU engine (float64 and float32 cells) on EXACTLY the arrays of the CPU baseline's half-size sample - same
    control points, same beta, same number of EM iterations from the same initial state - against the oracle's field:
    a driver-run parity figure for the bench's own workload generator at M = 3000, lambda_ as benchmarked.
    floor = the oracle against itself with the LAPACK driver swapped (lstsq -> truncated eigh)."""
    import torch
    from spateo_amd.vectorfield import SparseVFCEngine

    st = sample["state"]
    vmax = float(np.abs(st["V"]).max())
    out = {"cells": int(len(sample["Xv"])), "ctrl": int(len(sample["ctrl"])), "em_steps": int(sample["steps"]),
           "lambda_": lambda_, "reference": "float64 NumPy oracle (scipy.linalg.lstsq), same arrays",
           "floor": sample["floor"]}
    for dtype in ("float64", "float32"):
        eng = SparseVFCEngine(sample["Xv"], sample["Yv"], sample["ctrl"], sample["beta"], dtype=dtype, device=device)
        eng.lstsq_method = "scipy"
        eng.init_state(gamma=0.9)
        for _ in range(sample["steps"]):
            E, _ = eng.em_step(a=5.0, lambda_=lambda_, minP=1e-5, theta=0.75)
        V, P, _ = eng.results()
        dP = np.abs(P - st["P"])
        fl = sample["floor"]
        errs = {"V": float(np.abs(V - st["V"]).max() / vmax),
                "sigma2": float(abs(eng.sigma2 - st["sigma2"]) / st["sigma2"]),
                "P": float(dP.max()), "P999": float(np.quantile(dP, 0.999)),
                "E": float(abs(E - st["E"]) / abs(st["E"]))}
        out["f64" if dtype == "float64" else "f32"] = {
            "V_rel_err": errs["V"], "sigma2_rel_err": errs["sigma2"], "P_max_abs_err": errs["P"],
            "P_q999_abs_err": errs["P999"], "E_rel_err": errs["E"],
            "V_err_over_floor": errs["V"] / max(fl["V"], 1e-300),
            "sigma2_err_over_floor": errs["sigma2"] / max(fl["sigma2"], 1e-300),
            "P_err_over_floor": errs["P"] / max(fl["P"], 1e-300),
            "P_q999_err_over_floor": errs["P999"] / max(fl["P999"], 1e-300),
            "E_err_over_floor": errs["E"] / max(fl["E"], 1e-300),
            "ctrl_used": int(eng.M),
        }
        eng.k.drop_ublk()
        del eng
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=8_000_000, help="total cells (default: BASELINE config 4)")
    ap.add_argument("--ctrl", type=int, default=3000, help="control points M")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--lambda_", type=float, default=0.02, help="Spateo's default regularisation")
    ap.add_argument("--cpu-cells", type=int, default=200_000,
                    help="sample size of the CPU baseline (BASELINE.md section 3: 200 k at M = 3000), also run at half of "
                         "it - the half-size arrays are also what the parity leg runs the GPU on (0 = skip both)")
    ap.add_argument("--no-f64", action="store_true", help="skip the float64-mode run of the same workload")
    ap.add_argument("--lstsq", default="scipy", choices=["scipy", "cholesky"],
                    help="coefficient solve: scipy = the reference's minimum-norm gelsd semantics (default); cholesky = "
                         "jitter-escalated Cholesky (non-reference fast mode)")
    ap.add_argument("--cache-u", default="auto", choices=["auto", "on", "off"],
                    help="materialise the float32 kernel values once (96 GB at 8M x 3000) and stream them in the Gram "
                         "kernel instead of regenerating them every EM iteration")
    ap.add_argument("--no-conk", action="store_true", help="skip the con_K bandwidth run")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: run the HEADLINE itself through the multi-rank protocol on a process group of one rank "
                         "(every collective of the EM step executes on RCCL; same arithmetic, bit-equal result)")
    ap.add_argument("--collective", default="torch", choices=["torch", "mvf"],
                    help="who issues the step's all-reduces: torch.distributed (RCCL under the nccl backend) or "
                         "mvf_allreduce_stats of the C ABI on the engine's own RCCL communicator")
    ap.add_argument("--no-measure-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes (roofline.traffic then comes from the committed profile)")
    ap.add_argument("--traffic-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--no-whole-fit", action="store_true", help="skip the whole-call (host arrays in, host dict out) timings")
    ap.add_argument("--no-c3", action="store_true", help="skip the BASELINE config 3 (2 M x 2000) run")
    ap.add_argument("--no-organs32", action="store_true", help="skip the 32-organ run of BASELINE config 5")
    ap.add_argument("--no-rccl-world1", action="store_true",
                    help="skip the extra N = 1 runs that execute the step's collectives on a one-rank RCCL communicator")
    args = ap.parse_args()

    # ONE JSON line on stdout: RCCL prints a version banner on STDOUT when a communicator is created (seen on the first RCCL
    # execution of this code, round 5) and other native libraries may do the same, so the process's fd 1 is pointed at stderr
    # for the whole run and only the final line goes to the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    # host thread pools sized to what the container grants (see _cpu_budget): BLAS through threadpoolctl, torch's intra-op pool
    n_aff, cpu_quota, cpu_use = _cpu_budget()
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_use = max(1, cpu_use // max(1, world_env))  # one process per GPU shares the quota
    try:
        from threadpoolctl import threadpool_limits

        _blas_limit = threadpool_limits(limits=cpu_use)  # (kept alive for the life of the process)
    except Exception:
        _blas_limit = None
    torch.set_num_threads(cpu_use)
    throttle0 = _throttle_stats()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit(
                f"--gpus {args.gpus} needs one process per GPU: launch with `python -m torch.distributed.run --nnodes=1 "
                f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus {args.gpus} ...`")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # developer override: exercise the multi-rank code path on a ONE-GPU box (all ranks on cuda:0, gloo collectives,
    # because RCCL refuses two ranks on one device).  Never set by the driver; numbers from it are meaningless.
    one_dev = os.environ.get("MVF_BENCH_ONE_DEVICE") == "1"
    if one_dev:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if one_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            try:  # eager communicator bound to this rank's GPU (what the one-rank runs of this code exercise)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
            except (TypeError, ValueError) as exc:  # a torch that does not take device_id: the lazy form
                log(f"[bench rank {rank}] init_process_group(device_id=...) refused ({exc!r}); retrying without it")
                dist.init_process_group("nccl", rank=rank, world_size=world)

    from spateo_amd._kernels import HipKernels
    from spateo_amd._synthetic import make_config
    from spateo_amd.vectorfield import SparseVFCEngine, shard_bounds, sparsevfc_preprocess

    # ---------------------------------------------------------------- N ranks really are N ranks on N different GPUs
    rccl_ranks = None
    if distributed:
        import socket

        props = torch.cuda.get_device_properties(local_rank)
        mine_id = (socket.gethostname(), int(local_rank), str(getattr(props, "uuid", "")), int(getattr(props, "pci_bus_id", -1)),
                   int(getattr(props, "pci_device_id", -1)))
        ids = [None] * world
        dist.all_gather_object(ids, mine_id)
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)
        if not one_dev:
            assert len({(h_, l_) for h_, l_, *_ in ids}) == world, f"{world} ranks do not own {world} distinct GPUs: {ids}"
            if all(u_ not in ("", "None") for _, _, u_, _, _ in ids):  # (the device UUID where this torch reports one)
                assert len({(h_, u_) for h_, _, u_, _, _ in ids}) == world, f"{world} ranks share a physical GPU: {ids}"
            # a communicator of the C ABI over the same ranks, created and queried before anything is timed: ncclCommCount
            # must say N (mvf_comm_info asks RCCL itself)
            from spateo_amd._comm import MvfComm

            probe = MvfComm(device, rank, world)
            try:
                rccl_ranks = int(probe.info()[0])
                t_ = torch.ones(1, dtype=torch.float64, device=device)
                probe.all_reduce(t_)
                torch.cuda.synchronize(device)
                assert rccl_ranks == world and float(t_.cpu()[0]) == float(world), (rccl_ranks, float(t_.cpu()[0]), world)
            finally:
                probe.close()
        if rank == 0:
            log(f"[bench] {world} ranks on devices {ids}; RCCL communicator size {rccl_ranks}")

    # ---------------------------------------------------------------- counter-pass child (see measured_traffic)
    if args.traffic_child > 0:
        Xc, Vc, _ = make_config("C4", N=args.cells)
        _, Xcv, Ycv, _, ctrl_c, beta_c = sparsevfc_preprocess(Xc, Vc, M=args.ctrl, seed=0)
        eng_c = SparseVFCEngine(Xcv, Ycv, ctrl_c, beta_c, dtype=args.dtype, device=device,
                                cache_u={"auto": "auto", "on": True, "off": False}[args.cache_u])
        eng_c.lstsq_method = "cholesky"
        eng_c.init_state(gamma=0.9)
        for _ in range(args.traffic_child):
            eng_c.em_step(a=5.0, lambda_=args.lambda_, minP=1e-5, theta=0.75)
        torch.cuda.synchronize()
        os.close(real_stdout)
        return

    # ---------------------------------------------------------------- synthetic workload (same on every rank)
    t0 = time.perf_counter()
    X, V, _ = make_config("C4", N=args.cells)
    M = args.ctrl
    if distributed:
        # as SparseVFC(distributed=True) does: rank 0 alone runs the O(N log N) host preprocessing and broadcasts the
        # control points; every rank only filters the finite rows and takes its block
        valid = np.where(np.isfinite(V.sum(1)))[0]
        Xv, Yv = X[valid], V[valid]
        box = [None]
        if rank == 0:
            _, _, _, idx0, ctrl0, beta0 = sparsevfc_preprocess(X, V, M=M, seed=0)
            box = [(ctrl0, beta0)]
        dist.broadcast_object_list(box, src=0)
        ctrl, beta = box[0]
    else:
        valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
    N = len(Xv)
    lo, hi = shard_bounds(N, rank, world)
    if rank == 0:
        log(f"[bench] generated + preprocessed N={N} M={len(ctrl)} beta={beta:.4g} in {time.perf_counter() - t0:.1f}s; "
            f"rank shard = {hi - lo} cells")
    n_loc = hi - lo
    Mc = len(ctrl)
    step_kw = dict(a=5.0, lambda_=args.lambda_, minP=1e-5, theta=0.75)

    def barrier():
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    def ensure_group():
        """A process group of ONE rank on RCCL (N = 1 runs that force the collectives); idempotent."""
        if not dist.is_initialized():
            import socket

            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                port = s_.getsockname()[1]
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ["MASTER_PORT"] = str(port)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(device))

    def run_mode(dtype, steps, warmup, force=False, collective="torch", data=None):
        """W warm-up + K timed EM iterations of the whole workload in one cell dtype; returns the timing record.
        data = (Xv, Yv, ctrl, beta): another single-rank workload through the same code (BASELINE config 3)."""
        kern = HipKernels(device, dtype)
        cache_u = {"auto": "auto", "on": True, "off": False}[args.cache_u]
        if force and collective == "torch":
            ensure_group()
        if data is None:
            Xr, Yr, ctrl_r, beta_r, N_r, n_loc_r = Xv[lo:hi], Yv[lo:hi], ctrl, beta, N, n_loc
        else:
            Xr, Yr, ctrl_r, beta_r = data
            N_r = n_loc_r = len(Xr)
        Mc_r = len(ctrl_r)
        eng = SparseVFCEngine(Xr, Yr, ctrl_r, beta_r, dtype=dtype, device=device,
                              distributed=(distributed or (force and collective == "torch")) and data is None,
                              n_total=N_r, kernels=kern, cache_u=cache_u, force_collectives=force,
                              collective=collective)
        eng.lstsq_method = args.lstsq
        eng.init_state(gamma=0.9)
        # the coefficient solve, bracketed by events on the launch stream (it synchronises internally once per Jacobi
        # sweep, so event time == wall time of the call)
        solve_events = []
        inner = eng._solve_all

        def timed_solve(ls2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            host = inner(ls2)
            e1.record()
            solve_events.append((e0, e1))
            return host

        eng._solve_all = timed_solve
        try:
            return _timed_steps(eng, kern, solve_events, dtype, steps, warmup, force, collective, N_r, n_loc_r, Mc_r)
        finally:  # (the RCCL communicator of the engine is destroyed on every path, not only on success)
            if eng.comm is not None:
                eng.comm.close()
            kern.drop_ublk()
            del eng, kern
            torch.cuda.empty_cache()

    def _timed_steps(eng, kern, solve_events, dtype, steps, warmup, force, collective, N_r, n_loc_r, Mc_r):
        for _ in range(warmup):
            eng.em_step(**step_kw)
        kern.gram_events = []
        eng.comm_events = [] if eng.multi else None
        solve_events.clear()
        n_sweeps0 = len(eng.solver_stats["sweeps"])
        barrier()
        t_start = time.perf_counter()
        for _ in range(steps):
            eng.em_step(**step_kw)
        barrier()
        elapsed = time.perf_counter() - t_start
        if distributed:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.cpu()[0])
        gram_ms = [e0.elapsed_time(e1) for e0, e1 in kern.gram_events]
        solve_ms = [e0.elapsed_time(e1) for e0, e1 in solve_events]
        kern.gram_events = None
        # The same solve behind a WARM host (untimed extra iterations): the bracket above starts on the device when the Gram stage
        # ends, while the host thread has been parked in the solve's first status read for the length of that stage (a second at
        # the headline) and wakes up late and cold; here the host synchronises first and the call is timed by the wall clock
        # (the call is host-synchronous).  Both are reported; `avg_ms` stays the bracket inside the timed steps.
        warm_ms = []
        if not eng.multi:
            timed = eng._solve_all

            def warm_solve(ls2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                host = timed(ls2)
                warm_ms.append(1e3 * (time.perf_counter() - t0))
                return host

            eng._solve_all = warm_solve
            try:
                for _ in range(2):
                    eng.em_step(**step_kw)
            finally:
                eng._solve_all = timed
            solve_events[:] = solve_events[: len(solve_ms)]
        ms_per_step = 1e3 * elapsed / steps
        gram_avg_ms = float(np.mean(gram_ms))
        Mg = int(eng.M)  # control points the Gram kernel works on
        alg_flops = float(n_loc_r) * Mg * (Mg + 1)  # symmetric Gram: n m (m+1) flops
        achieved = alg_flops / (gram_avg_ms * 1e-3) / 1e12
        peak = PEAK_F64_MFMA_TFLOPS  # both cell dtypes accumulate with v_mfma_f64_16x16x4_f64
        ctype = "float" if dtype == "float32" else "double"
        traffic, traffic_src = pmc_traffic(dtype, "f64acc", eng.cached_u, world, n_loc_r, Mg)
        rec = {
            "value": N_r * steps / elapsed,
            "ms_per_step": ms_per_step,
            "steps": steps,
            "warmup": warmup,
            "roofline": {
                "kernel": (f"gram_cached_kernel<{ctype}> (v_mfma_f64_16x16x4_f64, cached {dtype} U streamed from HBM)"
                           if eng.cached_u else f"gram_f64acc_kernel<{ctype}> (v_mfma_f64_16x16x4_f64)"),
                "bound": "mfma",
                "achieved": achieved,
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": achieved / peak,
                # HBM bytes per launch (FETCH_SIZE + WRITE_SIZE PMC passes with the gfx950 corrections) parsed from the
                # committed profiles/r04_pmc_traffic.json for exactly this (dtype, cells per rank, M); null otherwise
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_from_committed_profile": traffic is not None,  # a PMC pass of this build, not of this run
                "avg_kernel_ms": gram_avg_ms,
                "launches": len(gram_ms),
                "algorithmic_flops_per_launch": alg_flops,
                "share_of_step": gram_avg_ms / ms_per_step,
            },
            "solve": {
                "lstsq_method": eng.lstsq_method,
                "path": (("minimum-norm (pivoted-Cholesky factor of rank r; the invariant subspace below the eps*lambda_max "
                          "cut-off from block inverse iteration on 128 / 256 vectors (solve.block), three applications, deflated "
                          "solve; jacobi_sweeps = the block's Rayleigh-Ritz problem's)") if eng.mn_method == "deflated" else
                         ("minimum-norm (pivoted-Cholesky factor of rank r + one-sided block Jacobi on its r columns, "
                          "eps*lambda_max cut-off)") if eng.mn_method == "lowrank" else
                         "minimum-norm (Cholesky + one-sided block Jacobi eigensolver, eps*lambda_max cut-off)")
                        if eng.rank_deficient else "Cholesky (pivots certify full numerical rank)",
                "avg_ms": float(np.mean(solve_ms)),
                "warm_host_ms": (float(np.mean(warm_ms)) if warm_ms else None),
                "share_of_step": float(np.mean(solve_ms)) / ms_per_step,
                "jacobi_sweeps": eng.solver_stats["sweeps"][n_sweeps0:n_sweeps0 + steps],
                "kept_rank": eng.solver_stats["rank"][-1] if eng.solver_stats["rank"] else Mc_r,
                "factor_rank": (eng.solver_stats.get("factor_rank") or [None])[-1],
                "warm_start": bool(eng.warm_start),
                "jitter": eng.jitter,
            },
            "cached_u": bool(eng.cached_u),
            "ctrl_used": Mg,
            "sigma2_after": eng.sigma2,
            # whole-step rates over all ranks, U counted as materialised (2 s N M bytes, 2 N M^2 flop)
            "step_effective_GBps": 2.0 * (4 if dtype == "float32" else 8) * N_r * Mc_r / (ms_per_step * 1e-3) / 1e9,
            # flops the symmetric Gram kernel executes algorithmically, N M (M + 1), over the whole step's time
            "step_TFLOPs_NM_Mplus1": float(N_r) * Mc_r * (Mc_r + 1) / (ms_per_step * 1e-3) / 1e12,
        }
        # "MFMA utilisation on the solve" (north_star): MFMA-tile flops of the solve over its wall time.  Rank-revealing
        # Jacobi path: 2 r M^2 in the trailing updates of the pivoted factor + 8 r^2 M per Jacobi sweep (Gram + update tiles);
        # full width: M^3 / 3 (Cholesky) + 8 M^3 per sweep; Cholesky only: M^3 / 3.
        sv = rec["solve"]
        if eng.rank_deficient and sv["jacobi_sweeps"]:
            sw = float(np.mean([int(x) for x in sv["jacobi_sweeps"]]))
            if eng.mn_method == "deflated" and sv["factor_rank"] and sv["factor_rank"] >= 512:
                # pivoted factor 2 r M^2; S2 = L^T L 2 r^2 M; Cholesky + inverse rows of S2 4 r^3 / 3; S2^-1 2 r^3; the
                # block (b = 256): two Gram + two orthonormalising + three b x r x r products, Jacobi sweeps on b x b
                rr, bb = float(sv["factor_rank"]), float((eng.solver_stats.get("block") or [256])[-1] or 256)
                sv["block"] = int(bb)
                # (three applications of S2^-1 since round 5: one more b x r x r product and one more orthonormalisation)
                fl = (2.0 * rr * Mc_r * Mc_r + 2.0 * rr * rr * Mc_r + 10.0 / 3.0 * rr**3 + 12.0 * bb * bb * rr + 8.0 * bb * rr * rr +
                      sw * 8.0 * bb**3)
            elif eng.mn_method in ("lowrank", "deflated") and sv["factor_rank"]:
                rr = float(sv["factor_rank"])
                fl = 2.0 * rr * Mc_r * Mc_r + sw * 8.0 * rr * rr * Mc_r
            else:
                fl = Mc_r**3 / 3.0 + sw * 8.0 * float(Mc_r) ** 3
        else:
            fl = Mc_r**3 / 3.0
        sv["mfma_flops_per_solve"] = fl
        sv["TFLOPs"] = fl / (sv["avg_ms"] * 1e-3) / 1e12
        sv["frac_of_f64_mfma_peak"] = sv["TFLOPs"] / PEAK_F64_MFMA_TFLOPS
        sv["bound"] = "latency (dependent launches / rotation chains), not MFMA"
        if eng.multi:
            # what the first multi-GPU run needs to explain itself: per-rank Gram time and the collectives
            # per step: tri(G) (the big one, issued asynchronously: its interval also covers the rhs / quadform kernels and
            # the [R | stats] all-reduce it overlaps), [R | stats], and the two scalar-sized ones (E-step MIN, step end)
            nb_max = max(nb for _, _, nb in eng.comm_events)
            big = [(e0.elapsed_time(e1), nb) for e0, e1, nb in eng.comm_events if nb == nb_max]
            mid = [e0.elapsed_time(e1) for e0, e1, nb in eng.comm_events if 1024 < nb < nb_max]
            small = [e0.elapsed_time(e1) for e0, e1, nb in eng.comm_events if nb <= 1024]
            mine = {"rank": rank, "cells": n_loc_r, "gram_ms": gram_avg_ms, "solve_ms": float(np.mean(solve_ms)),
                    "allreduce_ms": float(np.mean([t for t, _ in big])) if big else None,
                    "allreduce_bytes": big[0][1] if big else 0,
                    "rhs_stats_allreduce_ms": float(np.mean(mid)) if mid else None,
                    "scalar_allreduce_ms": float(np.mean(small)) if small else None}
            # every rank also prints its own record on stderr BEFORE the gather: a run that dies in a collective still
            # leaves the per-rank figures in the log
            log(f"[bench rank {rank}] " + json.dumps(mine))
            allr = [None] * world
            try:
                if distributed:
                    dist.all_gather_object(allr, mine)
                else:
                    allr = [mine]
            except Exception as exc:  # noqa: BLE001
                log(f"[bench rank {rank}] gathering the per-rank records failed: {exc!r}")
                allr = [mine]
            rec["per_rank"] = allr
            rec["comm"] = {"collectives_per_step": len(eng.comm_events) / steps, "allreduce_bytes": mine["allreduce_bytes"],
                           "backend": ("mvf_allreduce_stats (C ABI, RCCL communicator of the engine)" if collective == "mvf"
                                       else f"torch.distributed {dist.get_backend()}"),
                           "ranks": world, "forced_on_one_rank": bool(force and world == 1),
                           "allreduce_ms_max": max((r_["allreduce_ms"] or 0.0) for r_ in allr),
                           "note": "event time on the compute stream from issuing a collective to the stream having waited "
                                   "for it: includes waiting for the slowest rank to arrive; the big one is asynchronous, "
                                   "its interval spans the rhs / quadform kernels and the [R | stats] all-reduce"}
        if eng.comm is not None:
            rec["rccl_ranks_mvf_comm_info"] = int(eng.comm.info()[0])  # ncclCommCount of the engine's communicator
        return rec

    try:
        if args.force_collectives and world != 1:
            raise SystemExit("--force-collectives is an N = 1 option (with N > 1 the collectives run anyway)")
        main_rec = run_mode(args.dtype, args.steps, args.warmup, force=args.force_collectives, collective=args.collective)
    except Exception as exc:
        # one JSON line per failing rank on stderr (which rank, what, where), then the error itself
        import traceback

        log(f"[bench rank {rank}] " + json.dumps({"rank": rank, "failed": True, "error": repr(exc), "cells": n_loc,
                                                  "traceback": traceback.format_exc().splitlines()[-6:]}))
        raise
    value, ms_per_step = main_rec["value"], main_rec["ms_per_step"]

    out = {
        "metric": "cells/s per SparseVFC EM iter",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32" if args.dtype == "float32" else "f64",
        "data": "synthetic",
        "config": {
            "workload": f"SparseVFC EM iteration, {N} cells x {Mc} control points, 3-D (BASELINE config 4: "
                        f"E9.5->E11.5 morphometric field), lambda_={args.lambda_}",
            "cells": N,
            "ctrl_points": Mc,
            "parallelism": f"cells block-sharded over {world} GPU(s), one all-reduce of [G|R|stats] per EM step",
            "sigma2_after": main_rec["sigma2_after"],
            "gram_mode": "f64acc",
            "cached_u": main_rec["cached_u"],
            "step_effective_GBps": main_rec["step_effective_GBps"],
            "step_TFLOPs_NM_Mplus1": main_rec["step_TFLOPs_NM_Mplus1"],
        },
        "roofline": main_rec["roofline"],
        "solve": main_rec["solve"],
        # developer knobs of libmvf / the host that change which kernel variant runs: none set = defaults
        "env": {k_: v_ for k_, v_ in sorted(os.environ.items()) if k_.startswith("MVF_")},
        "developer_options": __import__("spateo_amd")._lib.debug_options(),   # mvf_debug_option values != default
        # what the container grants the host side: thread pools (BLAS, torch) are sized to it, and the CPU baseline runs on it
        "host": {"cpus_visible": n_aff, "cfs_quota_cpus": cpu_quota, "threads_used": cpu_use, "throttle_at_start": throttle0},
    }
    if "comm" in main_rec and not distributed:
        out["per_rank"], out["comm"] = main_rec["per_rank"], main_rec["comm"]
        out["config"]["force_collectives"] = True
    if distributed:
        out["per_rank"], out["comm"] = main_rec["per_rank"], main_rec["comm"]
        out["config"]["parallelism"] = (f"cells block-sharded over {world} GPUs; per EM step the all-reduce of tri(G) "
                                        f"({main_rec['comm']['allreduce_bytes'] / 1e6:.1f} MB, overlapped with the rhs "
                                        f"kernels), [R | stats], and two scalar-sized ones (E-step MIN; sum P r + solver "
                                        f"agreement); redundant coefficient solve on every rank")

    # ---------------------------------------------------------------- the same workload in float64 mode (N = 1)
    if world == 1 and args.dtype == "float32" and not args.no_f64:
        k64, w64 = max(2, min(args.steps, 5)), 1
        r64 = run_mode("float64", k64, w64)
        out["f64"] = {"metric": out["metric"], "unit": "cells/s", "dtype": "f64", "value": r64["value"],
                      "ms_per_step": r64["ms_per_step"], "steps": k64, "warmup": w64, "roofline": r64["roofline"],
                      "solve": r64["solve"], "cached_u": r64["cached_u"], "sigma2_after": r64["sigma2_after"],
                      "note": "same cells, control points and lambda_ as the headline line, float64 cells and kernel "
                              "values (the mode the 1e-5 parity clause refers to)"}

    if world == 1 and args.dtype == "float32" and "f64" in out:
        out["config"]["f64_value"] = out["f64"]["value"]
        out["config"]["f64_ms_per_step"] = out["f64"]["ms_per_step"]
        out["config"]["f64_frac"] = out["f64"]["roofline"]["frac"]
    out["config"]["solve_avg_ms"] = main_rec["solve"]["avg_ms"]
    out["config"]["solve_warm_host_ms"] = main_rec["solve"]["warm_host_ms"]
    out["config"]["collectives_per_step"] = (main_rec.get("comm") or {}).get("collectives_per_step", 0.0)
    out["config"]["rccl_ranks"] = rccl_ranks if distributed else None   # (N = 1: filled by the rccl_world1 leg below)

    # ---------------------------------------------------------------- BASELINE config 3 (2 M x 2000) through the same code (N = 1)
    if rank == 0 and world == 1 and not args.no_c3 and args.cells >= 2_000_000:
        X3, V3, M3 = make_config("C3")
        _, X3v, Y3v, _, ctrl3, beta3 = sparsevfc_preprocess(X3, V3, M=M3, seed=0)
        del X3, V3
        k3 = max(3, min(args.steps, 10))
        r3 = run_mode(args.dtype, k3, 2, data=(X3v, Y3v, ctrl3, beta3))
        out["c3"] = {"workload": f"BASELINE config 3: {len(X3v)} cells x {len(ctrl3)} control points, lambda_={args.lambda_}",
                     "value": r3["value"], "ms_per_step": r3["ms_per_step"], "steps": k3, "warmup": 2, "dtype": out["dtype"],
                     "roofline": r3["roofline"], "solve": r3["solve"], "sigma2_after": r3["sigma2_after"]}
        out["config"]["c3_ms_per_step"] = r3["ms_per_step"]
        out["config"]["c3_gram_frac"] = r3["roofline"]["frac"]
        out["config"]["c3_solve_avg_ms"] = r3["solve"]["avg_ms"]
        del X3v, Y3v

    # ---------------------------------------------------------------- the step's collectives on RCCL with one rank (N = 1)
    if world == 1 and not args.no_rccl_world1 and not args.force_collectives:
        # The exchange of the multi-GPU path, executed: force_collectives runs the multi-rank protocol (split Gram stages,
        # asynchronous all-reduce of tri(G) overlapped with the rhs kernels, [R | stats], the E-step MIN and the closing
        # 14-double collective) on a communicator of ONE rank - RCCL accepts that on a one-GPU box - through both back ends.
        # Same arithmetic as the headline (a single rank's sum is the identity); the step-time difference to the headline
        # is what the protocol costs one rank before any xGMI transfer: the split launches, pack / unpack, and the
        # collectives' launch latency.
        kr, wr = max(2, min(args.steps, 3)), 1
        out["rccl_world1"] = {"note": "force_collectives on a one-rank RCCL communicator: every collective of the EM step "
                                      "executes on RCCL; NOT a multi-GPU measurement (no xGMI traffic)",
                              "headline_ms_per_step": ms_per_step, "steps": kr, "warmup": wr}
        for coll in ("torch", "mvf"):
            try:
                rr = run_mode(args.dtype, kr, wr, force=True, collective=coll)
                out["rccl_world1"][coll] = {"ms_per_step": rr["ms_per_step"], "comm": rr["comm"],
                                            "per_rank": rr["per_rank"], "sigma2_after": rr["sigma2_after"],
                                            "protocol_overhead_ms": rr["ms_per_step"] - ms_per_step}
                if "rccl_ranks_mvf_comm_info" in rr:  # ncclCommCount of the engine's own communicator (C-ABI path)
                    out["rccl_world1"][coll]["rccl_ranks"] = rr["rccl_ranks_mvf_comm_info"]
                    out["config"]["rccl_ranks"] = rr["rccl_ranks_mvf_comm_info"]
            except Exception as exc:  # noqa: BLE001 - the headline line must survive a failure of this extra
                import traceback

                out["rccl_world1"][coll] = {"failed": repr(exc), "traceback": traceback.format_exc().splitlines()[-4:]}
        if dist.is_initialized():
            dist.destroy_process_group()

    # ---------------------------------------------------------------- con_K HBM bandwidth (N = 1, rank 0)
    if rank == 0 and world == 1 and not args.no_conk:
        # BASELINE config 3: con_K roofline run (16 GB of float32 output); clipped to the workload when a developer run is smaller
        nk, mk = min(2_000_000, len(Xv)), min(2000, len(ctrl))
        xs = torch.from_numpy((Xv[:nk] - ctrl.mean(0)).astype(np.float32)).to(device)
        cs = torch.from_numpy((ctrl[:mk] - ctrl.mean(0)).astype(np.float32)).to(device)
        kf = HipKernels(device, "float32")
        from spateo_amd import _lib as _l

        Kmat = torch.empty(nk, mk, dtype=torch.float32, device=device)  # allocated once: time the kernel, not malloc
        stream = torch.cuda.current_stream(device).cuda_stream

        def run_conk():
            _l.check(kf.lib.mvf_con_k(xs.data_ptr(), nk, cs.data_ptr(), mk, 3, float(beta), Kmat.data_ptr(), _l.MVF_F32,
                                      stream), "mvf_con_k")

        run_conk()
        torch.cuda.synchronize()
        evs = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run_conk()
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
        del Kmat
        nbytes = 4.0 * (nk * mk + 3 * nk + 3 * mk)
        out["con_k"] = {"n": nk, "m": mk, "dtype": "f32", "ms": ms, "algorithmic_bytes": nbytes,
                        "GBps": nbytes / (ms * 1e-3) / 1e9, "peak_GBps": PEAK_HBM_GBPS,
                        "frac": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, "bound": "hbm (write)"}

    # ---------------------------------------------------------------- evaluator path (N = 1, rank 0): BASELINE config 2's shape
    if rank == 0 and world == 1 and not args.no_conk:
        from spateo_amd import _lib as _l
        from spateo_amd.vectorfield import SvcVectorField, _EVAL_ALL, clear_eval_cache

        me = 500  # Jacobian + curl on a 64^3 grid against 500 control points (the kernel's rate does not depend on C)
        rng = np.random.default_rng(0)
        lo_b, hi_b = Xv[:100_000].min(0), Xv[:100_000].max(0)
        grid = np.stack([g_.ravel() for g_ in np.meshgrid(*[np.linspace(lo_b[c_], hi_b[c_], 64) for c_ in range(3)],
                                                         indexing="ij")], axis=1)
        vfd = {"X_ctrl": ctrl[:me], "C": rng.standard_normal((me, 3)), "beta": float(beta)}
        out["eval"] = {"grid_points": len(grid), "ctrl": me, "pairs": len(grid) * me}
        for dt in ("float32", "float64"):
            vf = SvcVectorField(dtype=dt, device=device)
            vf.vf_dict = vfd
            walls = []
            for _ in range(5):
                clear_eval_cache()
                torch.cuda.synchronize()
                t_e = time.perf_counter()
                vf.get_Jacobian()(grid)
                vf.compute_curl(X=grid)
                walls.append(1e3 * (time.perf_counter() - t_e))
            # the same two calls with the CALLER KEEPING what they return (every output array of the next call at a fresh address:
            # no recycled host block, new page-locked blocks for the outputs) - what a user's loop over samples does
            kept, walls_kept = [], []
            for _ in range(5):
                clear_eval_cache()
                torch.cuda.synchronize()
                t_e = time.perf_counter()
                kept.append((vf.get_Jacobian()(grid), vf.compute_curl(X=grid)))
                walls_kept.append(1e3 * (time.perf_counter() - t_e))
            del kept
            ke = HipKernels(device, dt)
            cen = vfd["X_ctrl"].mean(0)
            x4e, c4e = ke.to_x4(grid, cen), ke.to_x4(vfd["X_ctrl"], cen)
            Cd = torch.from_numpy(np.ascontiguousarray(vfd["C"])).to(device)
            ke.eval(x4e, c4e, float(beta), Cd, _EVAL_ALL)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ke.eval(x4e, c4e, float(beta), Cd, _EVAL_ALL)
            e1.record()
            torch.cuda.synchronize()
            kms = e0.elapsed_time(e1) / 20
            tf_ = len(grid) * me * EVAL_F64_FLOP_PER_PAIR / (kms * 1e-3) / 1e12
            out["eval"][dt] = {"kernel_ms_all_quantities": kms, "Gpairs_per_s": len(grid) * me / kms / 1e6,
                               "kernel": f"eval_mfma_kernel<{'float' if dt == 'float32' else 'double'}> (all quantities, one launch)",
                               "f64_flop_per_pair": EVAL_F64_FLOP_PER_PAIR, "TFLOPs": tf_,
                               "frac_of_f64_valu_peak": tf_ / PEAK_F64_VALU_TFLOPS,
                               "jacobian_plus_curl_api_wall_ms": float(np.median(walls[1:])),
                               "jacobian_plus_curl_api_wall_results_kept_ms": float(np.median(walls_kept[1:])),
                               "first_call_api_wall_ms": walls[0]}
        clear_eval_cache()
        out["config"]["eval_frac"] = out["eval"]["float32" if args.dtype == "float32" else "float64"]["frac_of_f64_valu_peak"]
        out["config"]["eval_frac_f64_cells"] = out["eval"]["float64"]["frac_of_f64_valu_peak"]
        out["config"]["eval_api_wall_ms"] = out["eval"]["float32" if args.dtype == "float32" else "float64"]["jacobian_plus_curl_api_wall_ms"]

    # ---------------------------------------------------------------- BASELINE configs 2 and 5 (N = 1): ms per EM iteration
    if rank == 0 and world == 1 and not args.no_whole_fit:
        def small(cfg_name, n_small, m_small, seed_small):
            Xs, Vs, _ = make_config(cfg_name, N=n_small, seed=seed_small)
            _, Xsv, Ysv, _, ctrl_s, beta_s = sparsevfc_preprocess(Xs, Vs, M=m_small, seed=0)
            eng = SparseVFCEngine(Xsv, Ysv, ctrl_s, beta_s, dtype=args.dtype, device=device)
            eng.init_state(gamma=0.9)
            evs, inner = [], eng._solve_all

            def timed(ls2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                h_ = inner(ls2)
                e1.record()
                evs.append((e0, e1))
                return h_

            eng._solve_all = timed
            # warm-up: at least 4 iterations AND 0.4 s of them.  The solve at M = 500 is a chain of ~100 single-workgroup
            # launches: too little load to pull the GPU's clocks up by itself, so right after a host-bound stretch of this
            # script it ran 2 x slower for its first ~100 ms (measured: 4.5 ms / iteration behind the evaluator section,
            # 2.3 ms behind a Gram-heavy one) - the steady state is what this object reports
            t_w, n_w = time.perf_counter(), 0
            while n_w < 4 or time.perf_counter() - t_w < 0.4:
                eng.em_step(**step_kw)
                n_w += 1
            evs.clear()
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            for _ in range(30):
                eng.em_step(**step_kw)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t_s) / 30
            rec = {"cells": int(len(Xsv)), "ctrl": int(m_small), "dtype": args.dtype, "ms_per_em_step": ms,
                   "cells_per_s": len(Xsv) / (ms * 1e-3), "solve_ms": float(np.mean([a.elapsed_time(b) for a, b in evs])),
                   "solver": eng.mn_method if eng.rank_deficient else "cholesky",
                   "kept_rank": (eng.solver_stats["rank"] or [m_small])[-1],
                   "block": (eng.solver_stats.get("block") or [None])[-1]}
            eng.k.drop_ublk()
            return rec

        out["small_configs"] = {"note": "BASELINE configs 2 and 5 at their stated sizes, one EM iteration (steady state: "
                                        "30 timed after >= 0.4 s of warm-up iterations), lambda_ as the headline",
                                "c2_50k_x_500": small("C2", 50_000, 500, 2),
                                "c5_organ_250k_x_500": small("C2", 250_000, 500, 100)}
        # config 5 as it runs on one GPU of eight: 4 of the 32 organs, whole fits (host arrays in -> host dicts out, MaxIter 30,
        # run to convergence), one organ per HIP stream (replicas only: no collective) against the same four one after the other
        from spateo_amd.vectorfield import SparseVFC, SparseVFC_many

        organs = [make_config("C2", N=250_000, seed=100 + s_)[:2] + (None,) for s_ in range(4)]
        kw5 = dict(M=500, lambda_=args.lambda_, MaxIter=30, dtype=args.dtype, lstsq_method="scipy", device=device)
        SparseVFC_many(organs, n_streams=4, **kw5)  # every thread's first-use costs (kernel objects, workspaces, streams)
        t_5 = time.perf_counter()
        res5 = SparseVFC_many(organs, n_streams=4, **kw5)
        t_par = time.perf_counter() - t_5
        t_5 = time.perf_counter()
        seq5 = [SparseVFC(*o_, **kw5) for o_ in organs]
        t_seq = time.perf_counter() - t_5
        out["small_configs"]["c5_four_organs"] = {
            "organs": 4, "cells_each": 250_000, "ctrl": 500, "iterations": [int(r_["iteration"]) + 1 for r_ in res5],
            "four_streams_wall_s": t_par, "sequential_wall_s": t_seq,
            "identical_to_sequential": bool(all(np.array_equal(a_["V"], b_["V"]) for a_, b_ in zip(res5, seq5)))}
        del organs, res5, seq5
        out["config"]["c2_ms_per_em_step"] = out["small_configs"]["c2_50k_x_500"]["ms_per_em_step"]
        out["config"]["c5_organ_ms_per_em_step"] = out["small_configs"]["c5_organ_250k_x_500"]["ms_per_em_step"]
        # all 32 organs of BASELINE config 5 (250 k +- 10 % cells each, M = 500, seeds 100 + k) as whole fits through four
        # streams of this ONE GPU (on eight GPUs every rank takes four of them: replicas only, no collective)
        if not args.no_organs32:
            rng32 = np.random.default_rng(5)
            sizes = [int(250_000 * (0.9 + 0.2 * rng32.random())) for _ in range(32)]
            organs32 = [make_config("C2", N=sizes[s_], seed=100 + s_)[:2] + (None,) for s_ in range(32)]
            t_5 = time.perf_counter()
            res32 = SparseVFC_many(organs32, n_streams=4, **kw5)
            t32 = time.perf_counter() - t_5
            out["small_configs"]["c5_32_organs"] = {
                "organs": 32, "cells": sizes, "ctrl": 500, "streams": 4, "wall_s": t32,
                "em_iterations_total": int(sum(int(r_["iteration"]) + 1 for r_ in res32)),
                "cells_per_s_whole_fits": float(sum(sizes)) / t32,
                "note": "whole SparseVFC calls (host arrays in, host dicts out, MaxIter 30, run to convergence)"}
            out["config"]["c5_32_organs_wall_s"] = t32
            del organs32, res32

    # ---------------------------------------------------------------- whole calls: host arrays in -> host dict out (N = 1)
    if rank == 0 and world == 1 and not args.no_whole_fit:
        import spateo_amd._runtime as _rtm
        import spateo_amd.vectorfield as vfm

        def whole(Xw, Vw, Mw, max_iter):
            kw = dict(M=Mw, lambda_=args.lambda_, lstsq_method="scipy", seed=0, MaxIter=max_iter, dtype=args.dtype,
                      device=device)
            t_w = time.perf_counter()
            r_ = vfm.SparseVFC(Xw, Vw, None, **kw)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t_w
            _rtm.PROFILE_FITS = True
            try:
                vfm.SparseVFC(Xw, Vw, None, **kw)
                prof = dict(vfm.last_fit_profile())
            finally:
                _rtm.PROFILE_FITS = False
            its = int(r_["iteration"]) + 1
            return {"cells": int(len(Xw)), "ctrl": int(Mw), "dtype": args.dtype, "em_iterations": its, "wall_s": wall,
                    "split_of_a_second_call_with_phase_syncs": prof,
                    "overhead_s": wall - prof["em_s"], "em_ms_per_iteration": 1e3 * prof["em_s"] / its}

        X2, V2, M2 = make_config("C2")
        vfm.SparseVFC(X2, V2, None, M=M2, lambda_=args.lambda_, lstsq_method="scipy", MaxIter=2, dtype=args.dtype, device=device)
        out["whole_fit"] = {"note": "SparseVFC(X, Y, None, ...) as a user calls it: host NumPy arrays in, the reference's dict of "
                                    "host float64 arrays out (preprocessing, uploads, U cache, EM to convergence or MaxIter, "
                                    "downloads); wall_s = an unprofiled call, the split = a second call that synchronises at "
                                    "its phase boundaries",
                            "c2": whole(X2, V2, M2, 500),
                            "c4": whole(X, V, M, 4)}
        del X2, V2
    del X, V

    # ---------------------------------------------------------------- HBM traffic of the dominant kernel, measured (N = 1)
    if rank == 0 and world == 1 and not args.no_measure_traffic and not args.force_collectives:
        torch.cuda.empty_cache()
        mt = measured_traffic(N, Mc, args.dtype, args.lambda_)
        if mt is not None:
            rl = out["roofline"]
            rl["traffic_committed_profile"] = rl.get("traffic")
            rl["traffic"] = mt["bytes"]
            rl["traffic_source"] = mt["source"]
            rl["traffic_from_committed_profile"] = False
            rl["traffic_detail"] = {k_: mt[k_] for k_ in ("fetch_bytes", "write_bytes", "launches_per_iteration", "pass_seconds")}
            rl["traffic_over_algorithmic_bytes"] = mt["bytes"] / (2.0 * (4 if args.dtype == "float32" else 8) * N * Mc)

    # ---------------------------------------------------------------- CPU baseline (N = 1, rank 0, bounded sample)
    if rank == 0 and world == 1 and args.cpu_cells > 0:
        cb, sample = cpu_baseline(Mc, args.lambda_, args.cpu_cells, N)
        out["cpu_baseline"] = cb
        # against the fitted t(N) at the bench's own cell count (the lstsq constant amortised), not the small sample
        out["speedup_vs_cpu_baseline"] = value / cb["value"]
        out["speedup_vs_cpu_sample_rate"] = value / cb["sample_value"]
        # ------------------------------------------------------------ parity on the CPU sample's arrays (driver-run)
        out["parity"] = parity_on_sample(sample, args.lambda_, device)
        del sample

    if distributed:
        dist.barrier()
    sys.stdout.flush()
    if rank == 0:  # the line goes out BEFORE the process group is torn down: a hang in the teardown cannot lose it
        out["host"]["throttle_at_end"] = _throttle_stats()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
