#!/usr/bin/env python
"""bench.py -- SparseVFC EM-iteration throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

A "step" is ONE full SparseVFC EM iteration (E-step, weighted Gram + rhs, all-reduce, Cholesky solve, field
application, sigma^2 / gamma update - SURVEY.md Appendix A step 5) over the whole synthetic point cloud, with the
inputs resident in HBM.  Workload: BASELINE config 4 = 8 M cells, M = 3000 control points, float32 cells; it fits one
MI355X, so the same total problem is run at every N (cells block-sharded across ranks: strong scaling).
Rank 0 prints ONE JSON line.  Extra objects: ``roofline`` (dominant kernel = the MFMA Gram kernel, timed with HIP
events on its launch stream), ``con_k`` (the materialised-kernel HBM-write bandwidth) and ``cpu_baseline`` (the
float64 NumPy oracle on the host cores; N = 1 only, bounded sample).
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

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (spec; 155 measured)
PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X datasheet: FP64 matrix (v_mfma_f64_16x16x4_f64) dense peak
PEAK_HBM_GBPS = 8000.0
# (dtype, gram_mode, cached_u, gpus, cells, ctrl) -> corrected FETCH+WRITE bytes per launch of the dominant kernel
PMC_TRAFFIC_BYTES = {("float32", "f64acc", True, 1, 8_000_000, 3000): 3.33e12 + 1.14e9}        # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(M, lambda_, n_cpu, steps=2):
    """The float64 NumPy/SciPy oracle (kind "port": dynamo is not installable) on a bounded sample of the SAME
    workload: the C4 generator at n_cpu cells with the same M; cells/s per EM iteration is size independent at fixed
    M, which is what makes the sample comparable."""
    from oracle import sparsevfc_oracle as svo
    from spateo_amd._synthetic import make_config

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    X, V, _ = make_config("C4", N=n_cpu)
    t0 = time.perf_counter()
    valid, Xv, Yv, idx, ctrl, beta = svo.sparsevfc_setup(X, V, M=M, seed=0)
    K = svo.con_K(ctrl, ctrl, beta)
    t1 = time.perf_counter()
    U = svo.con_K(Xv, ctrl, beta)
    t_conk = time.perf_counter() - t1
    N, D = Yv.shape
    Vc, C = np.zeros((N, D)), np.zeros((len(ctrl), D))
    s2, gamma, E = np.sum(Yv**2) / (N * D), 0.9, 1
    ts = []
    for _ in range(steps):
        t2 = time.perf_counter()
        P, E, tecr, C, Vc, s2, gamma = svo.em_step(U, K, Yv, Vc, C, s2, gamma, E, a=5, lambda_=lambda_, minP=1e-5,
                                                   theta=0.75, lstsq_method="scipy")
        ts.append(time.perf_counter() - t2)
    t_step = float(np.median(ts))
    return {
        "value": N / t_step,
        "unit": "cells/s",
        "cores": int(threads),
        "host_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"float64 NumPy oracle (cdist+exp con_K, U.T*repmat(P) temporary, scipy.linalg.lstsq), C4 generator at "
                  f"N_cpu={N} cells, M={len(ctrl)}, median of {steps} EM steps ({t_step:.2f} s/step); con_K "
                  f"{t_conk:.2f} s = {U.nbytes / t_conk / 1e9:.2f} GB/s of output",
        "ms_per_step": 1e3 * t_step,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=8_000_000, help="total cells (default: BASELINE config 4)")
    ap.add_argument("--ctrl", type=int, default=3000, help="control points M")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--lambda_", type=float, default=0.02, help="Spateo's default regularisation")
    ap.add_argument("--cpu-cells", type=int, default=20_000, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--gram-mode", default="f64acc", choices=["f64acc", "f32mfma"],
                    help="float32 Gram kernel: f64acc = float32 operands + float64 MFMA accumulation (default, meets "
                         "the 1e-3 field tolerance); f32mfma = all-float32 MFMA (2x peak, noisier)")
    ap.add_argument("--cache-u", default="auto", choices=["auto", "on", "off"],
                    help="materialise the float32 kernel values once (96 GB at 8M x 3000) and stream them in the Gram "
                         "kernel instead of regenerating them every EM iteration")
    ap.add_argument("--no-conk", action="store_true", help="skip the con_K bandwidth run")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

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
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))

    from spateo_amd._kernels import HipKernels
    from spateo_amd._synthetic import make_config
    from spateo_amd.vectorfield import SparseVFCEngine, shard_bounds, sparsevfc_preprocess

    # ---------------------------------------------------------------- synthetic workload (same on every rank)
    t0 = time.perf_counter()
    X, V, _ = make_config("C4", N=args.cells)
    M = args.ctrl
    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
    N = len(Xv)
    lo, hi = shard_bounds(N, rank, world)
    if rank == 0:
        log(f"[bench] generated + preprocessed N={N} M={len(ctrl)} beta={beta:.4g} in {time.perf_counter() - t0:.1f}s; "
            f"rank shard = {hi - lo} cells")
    kern = HipKernels(device, args.dtype, gram_mode=args.gram_mode)
    cache_u = {"auto": "auto", "on": True, "off": False}[args.cache_u]
    if args.gram_mode == "f32mfma":
        cache_u = False
    eng = SparseVFCEngine(Xv[lo:hi], Yv[lo:hi], ctrl, beta, dtype=args.dtype, device=device, distributed=distributed,
                          n_total=N, kernels=kern, cache_u=cache_u)
    del X, V
    eng.init_state(gamma=0.9)
    step_kw = dict(a=5.0, lambda_=args.lambda_, minP=1e-5, theta=0.75)

    def barrier():
        torch.cuda.synchronize(device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        eng.em_step(**step_kw)
    kern.gram_events = []
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        eng.em_step(**step_kw)
    barrier()
    elapsed = time.perf_counter() - t_start
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.cpu()[0])
    gram_ms = [e0.elapsed_time(e1) for e0, e1 in kern.gram_events]
    kern.gram_events = None
    ms_per_step = 1e3 * elapsed / args.steps
    value = N * args.steps / elapsed

    # ---------------------------------------------------------------- roofline of the dominant kernel (rank 0's shard)
    n_loc = hi - lo
    Mc = len(ctrl)
    gram_avg_ms = float(np.mean(gram_ms))
    alg_flops = float(n_loc) * Mc * (Mc + 1)  # symmetric Gram: n m (m+1) flops
    achieved = alg_flops / (gram_avg_ms * 1e-3) / 1e12
    f32mfma = args.dtype == "float32" and args.gram_mode == "f32mfma"
    peak = PEAK_F32_MFMA_TFLOPS if f32mfma else PEAK_F64_MFMA_TFLOPS
    roofline = {
        "kernel": "gram_f32_kernel (v_mfma_f32_32x32x2_f32)" if f32mfma else
                  (f"gram_cached_kernel<{'float' if args.dtype == 'float32' else 'double'}> (v_mfma_f64_16x16x4_f64, "
                   f"cached {args.dtype} U streamed from HBM)" if eng.cached_u else
                   f"gram_f64acc_kernel<{'float' if args.dtype == 'float32' else 'double'}> (v_mfma_f64_16x16x4_f64)"),
        "bound": "mfma",
        "achieved": achieved,
        "peak": peak,
        "unit": "TFLOP/s",
        "frac": achieved / peak,
        # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected per the gfx950
        # notes of MI355X_MICROARCH.md; profiles/r01_hbm_traffic_pmc.md).  Only measured for the default workload.
        "traffic": PMC_TRAFFIC_BYTES.get((args.dtype, args.gram_mode, bool(eng.cached_u), world, N, Mc)),
        "avg_kernel_ms": gram_avg_ms,
        "launches": len(gram_ms),
        "algorithmic_flops_per_launch": alg_flops,
        "share_of_step": gram_avg_ms / ms_per_step,
    }

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
            "sigma2_after": eng.sigma2,
            "solve_jitter": eng.jitter,
            "solve_retries": eng.solve_retries,
            "gram_mode": args.gram_mode,
            "cached_u": bool(eng.cached_u),
            # whole-step rates over all ranks, U counted as materialised (2 s N M bytes, 2 N M^2 flop)
            "step_effective_GBps": 2.0 * (4 if args.dtype == "float32" else 8) * N * Mc / (ms_per_step * 1e-3) / 1e9,
            "step_TFLOPs_2NM2": 2.0 * N * Mc * Mc / (ms_per_step * 1e-3) / 1e12,
        },
        "roofline": roofline,
    }

    # ---------------------------------------------------------------- con_K HBM bandwidth (N = 1, rank 0)
    if rank == 0 and world == 1 and not args.no_conk:
        kern.drop_ublk()
        del eng
        torch.cuda.empty_cache()
        nk, mk = 2_000_000, 2000  # BASELINE config 3: con_K roofline run (16 GB of float32 output)
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

    # ---------------------------------------------------------------- CPU baseline (N = 1, rank 0, bounded sample)
    if rank == 0 and world == 1 and args.cpu_cells > 0:
        cb = cpu_baseline(Mc, args.lambda_, args.cpu_cells)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = value / cb["value"]

    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
