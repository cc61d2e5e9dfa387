"""world_size-2 ``gloo`` runs of the multi-GPU path on CPU: cells sharded across ranks (block shards of a replicated
input, or every rank bringing its own uneven rows), rank 0 alone preprocessing and broadcasting the control points, one
all-reduce of [tri(G) | R | stats] per EM step, the global min-non-zero rule of the E-step, outputs gathered on rank 0
or on every rank.  The device is replaced by
the oracle-backed test double (tests/_cpu_kernels.py); what is under test is the sharding / collective protocol."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, out_dir, mode="all"):
    for p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import spateo_amd as st
        from _cpu_kernels import CpuKernels
        from spateo_amd._synthetic import make_config

        X, V, _ = make_config("C2", N=601)  # odd: uneven shards
        if case == "zeros":
            V[598:] += 60.0  # gross outliers that live only in the LAST shard: exp(-r/2s2) underflows there only

        class Recording(CpuKernels):
            fills = []

            def estep_p(self, r, sigma2, gamma, a, dy, minP, theta, zero_fill, P_out, stats):
                Recording.fills.append(zero_fill)
                super().estep_p(r, sigma2, gamma, a, dy, minP, theta, zero_fill, P_out, stats)

            hints = []

            def solve(self, G, K, ls2, jitter, R, C_out, info, pivots=None):
                super().solve(G, K, ls2, jitter, R, C_out, info, pivots)
                if case == "minnorm_lr" and pivots is not None:
                    pivots[0] = 1e-14 * pivots[1]  # full rank NOT certified: every rank must switch to the min-norm solve
                if case == "disagree" and rank == 1 and pivots is not None:
                    pivots[0] = 1e-14 * pivots[1]  # ONLY rank 1 sees an uncertified rank: the ranks would part ways
                if case == "crash" and rank == 1 and len(Recording.fills) >= 2:
                    raise IndexError("not an MVFError: a plain bug on one rank in its second EM iteration")

            def solve_minnorm_lr(self, *a, **kw):
                Recording.hints.append(kw.get("rank_hint", 0))
                super().solve_minnorm_lr(*a, **kw)

        if case == "minnorm_lr":
            import spateo_amd.vectorfield as _v

            _v.SparseVFCEngine.minnorm_method = "lowrank"  # the rank-revealing solve regardless of M
        Grid = X[::30]
        kw = dict(M=25, lambda_=3.0, lstsq_method="scipy", MaxIter=6, seed=0)
        if case == "wide":
            V = np.column_stack([V, np.sin(X[:, 0] / 70), np.cos(X[:, 1] / 50)])  # Dy = 5: two column groups
        calls = {"unique": 0}
        import spateo_amd.preprocess as pre
        import spateo_amd.vectorfield as vfm

        orig_unique = pre.unique_rows

        def counting_unique(a, device=None):
            calls["unique"] += 1
            return orig_unique(a, device)

        pre.unique_rows = counting_unique
        vfm._rt._make_kernels = lambda device, dtype: Recording()  # the product's one kernel-binding seam
        if mode == "sharded":  # every rank brings ITS OWN rows: 401 + 200, not the block split
            lo, hi = (0, 401) if rank == 0 else (401, 601)
            got = st.SparseVFC(X[lo:hi], V[lo:hi], Grid, distributed=True, sharded_input=True, gather="root", **kw)
        elif case in ("disagree", "crash"):
            try:
                st.SparseVFC(X, V, Grid, distributed=True, gather=mode, **kw)
                msg = "no error"
            except Exception as exc:  # noqa: BLE001
                msg = f"{type(exc).__name__}: {exc}"
            with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
                f.write(msg)
            return
        else:
            got = st.SparseVFC(X, V, Grid, distributed=True, gather=mode, **kw)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), V=got["V"], P=got["P"], C=got["C"], grid_V=got["grid_V"],
                 sigma2=got["sigma2"], iteration=got["iteration"], E=got["E_traj"], fills=np.array(Recording.fills),
                 unique_calls=calls["unique"], valid_ind=got["valid_ind"], vfc=got["VFCIndex"],
                 hints=np.array(Recording.hints, dtype=np.int64))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["plain", "zeros", "minnorm_lr"])
def test_two_rank_gloo_matches_single_process(tmp_path, case):
    sys.path.insert(0, HERE)
    from oracle import sparsevfc_oracle as svo
    from spateo_amd._synthetic import make_config

    port = _free_port()
    mp.spawn(_worker, args=(2, port, case, str(tmp_path)), nprocs=2, join=True)
    X, V, _ = make_config("C2", N=601)
    if case == "zeros":
        V[598:] += 60.0
    ref = svo.SparseVFC(X, V, X[::30], M=25, lambda_=3.0, lstsq_method="scipy", MaxIter=6, seed=0)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # the min-non-zero fill value is GLOBAL: rank 0 (which holds no underflowing cell) sees the same value as rank 1
    np.testing.assert_array_equal(r0["fills"], r1["fills"])
    if case == "zeros":
        assert (r0["fills"] > 0).any()
    for k in ("V", "P", "C", "grid_V", "sigma2", "E"):
        np.testing.assert_array_equal(r0[k], r1[k])  # every rank ends with the identical, gathered result
    assert int(r0["iteration"]) == ref["iteration"]
    assert r0["V"].shape == ref["V"].shape == (601, 3) and r0["P"].shape == (601, 1)
    scale = np.abs(ref["V"]).max()
    assert np.abs(r0["V"] - ref["V"]).max() / scale < 1e-8
    assert np.abs(r0["grid_V"] - ref["grid_V"]).max() / scale < 1e-8
    np.testing.assert_allclose(r0["P"], ref["P"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(r0["E"], ref["E_traj"], rtol=1e-8)
    np.testing.assert_allclose(float(r0["sigma2"]), ref["sigma2"], rtol=1e-8)
    # the O(N log N) host preprocessing ran on rank 0 only
    assert int(r0["unique_calls"]) == 1 and int(r1["unique_calls"]) == 0
    if case == "minnorm_lr":
        # both ranks took the rank-revealing minimum-norm path in every iteration, with identical factor-rank hints
        # (the first call has none; afterwards the previous factor's 25 rows)
        np.testing.assert_array_equal(r0["hints"], r1["hints"])
        assert len(r0["hints"]) == int(r0["iteration"]) + 1 and r0["hints"][0] == 0 and (r0["hints"][1:] == 25).all()


def test_two_rank_gloo_divergent_solver_branches_raise_instead_of_hanging(tmp_path):
    """Rank 1 alone is made to distrust the Cholesky certificate: the per-step agreement check (MAX all-reduce of
    +/- the solver-decision signature) must raise MVFError on BOTH ranks in that very step."""
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, "disagree", str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        msg = (tmp_path / f"rank{r}.txt").read_text()
        assert "ranks disagree on the coefficient solve" in msg, msg


def test_two_rank_gloo_any_exception_on_one_rank_raises_on_all(tmp_path):
    """an exception that is NOT an MVFError (a plain bug) in one rank's solve used to skip the agreement
    collective and leave the other rank hanging in it.  Now it travels as the failure flag of the step's last collective:
    the failing rank re-raises its own error, the other one raises MVFError - both in that very step."""
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, "crash", str(tmp_path)), nprocs=2, join=True)
    assert "failed on another rank" in (tmp_path / "rank0.txt").read_text()
    assert (tmp_path / "rank1.txt").read_text().startswith("IndexError: not an MVFError")


def _many_worker(rank, world, port, out_dir, inject):
    for p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import spateo_amd.vectorfield as vfm
        from _cpu_kernels import CpuKernels
        from spateo_amd._synthetic import make_config

        vfm._rt._make_kernels = lambda device, dtype: CpuKernels()
        data = []
        for k in range(4):
            X, V, _ = make_config("C2", N=300 + 10 * k, seed=100 + k)
            data.append((X, V, None))
        if inject:
            orig = vfm.SparseVFC

            def failing(X, Y, Grid, **kw):
                if rank == 1 and len(X) == 310:   # organ 1: rank 1's first fit
                    raise IndexError("not an MVFError")
                return orig(X, Y, Grid, **kw)

            vfm.SparseVFC = failing
        try:
            res = vfm.SparseVFC_many(data, n_streams=2, distributed=True, M=20, lambda_=3.0, MaxIter=3, seed=0)
            msg = "ok " + " ".join(str(len(r["V"])) for r in res)
        except Exception as exc:  # noqa: BLE001 - the test reads what every rank saw
            msg = f"{type(exc).__name__}: {exc}"
        with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
            fh.write(msg)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_many_fits_an_exception_on_one_rank_raises_on_all(tmp_path):
    """`SparseVFC_many(distributed=True)` raised a worker's exception BEFORE the object gather - the other ranks
    waited in it for ever.  (results, error) now travel together: without a failure every rank returns all four organs; with
    one injected on rank 1 that rank re-raises its own exception and rank 0 raises MVFError naming rank 1 - nobody hangs."""
    sys.path.insert(0, HERE)
    mp.spawn(_many_worker, args=(2, _free_port(), str(tmp_path), False), nprocs=2, join=True)
    for r in (0, 1):
        assert (tmp_path / f"rank{r}.txt").read_text() == "ok 300 310 320 330"
    mp.spawn(_many_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    assert (tmp_path / "rank1.txt").read_text().startswith("IndexError: not an MVFError")
    m0 = (tmp_path / "rank0.txt").read_text()
    assert m0.startswith("MVFError") and "rank 1" in m0 and "not an MVFError" in m0, m0


def test_distributed_flag_requires_process_group():
    sys.path.insert(0, HERE)
    import spateo_amd as st
    from _cpu_kernels import CpuKernels

    X = np.random.default_rng(0).standard_normal((40, 3))
    with pytest.raises(RuntimeError, match="torch.distributed is not initialised"):
        st.SparseVFC(X, X, None, M=5, distributed=True)


def _force_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import spateo_amd as st
        import spateo_amd.vectorfield as vfm
        from _cpu_kernels import CpuKernels
        from spateo_amd._synthetic import make_config

        count = {"n": 0}
        orig = dist.all_reduce

        def counting(*a, **kw):
            count["n"] += 1
            return orig(*a, **kw)

        dist.all_reduce = counting
        vfm._rt._make_kernels = lambda device, dtype: CpuKernels()
        X, V, _ = make_config("C2", N=601)
        V[7] = np.nan
        kw = dict(M=25, lambda_=3.0, lstsq_method="scipy", MaxIter=6, seed=0)
        plain = st.SparseVFC(X, V, X[::30], **kw)
        n_plain = count["n"]
        forced = st.SparseVFC(X, V, X[::30], distributed=True, force_collectives=True, gather="all", **kw)
        np.savez(os.path.join(out_dir, "force.npz"), n_plain=n_plain, n_forced=count["n"] - n_plain,
                 iters=forced["iteration"] + 1, row_range=np.array(forced["row_range"]),
                 **{f"p_{k}": plain[k] for k in ("V", "P", "C", "grid_V", "sigma2", "E_traj", "VFCIndex")},
                 **{f"f_{k}": forced[k] for k in ("V", "P", "C", "grid_V", "sigma2", "E_traj", "VFCIndex")})
    finally:
        dist.destroy_process_group()


def test_force_collectives_runs_the_multi_rank_protocol_on_one_rank(tmp_path):
    """`force_collectives=True` on a process group of ONE rank: the rank-0 preprocessing + object broadcast, the four
    all-reduces of every EM step and the output gather all execute on the backend (gloo here; RCCL in
    tests/test_gpu_rccl.py), and - a single rank's sum being the identity - the result is that of the plain fit."""
    port = _free_port()
    mp.spawn(_force_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    z = np.load(tmp_path / "force.npz")
    assert int(z["n_plain"]) == 0                       # the plain single-process fit issues no collective at all
    assert int(z["n_forced"]) == 1 + 4 * int(z["iters"])  # init_state's sum P r + four per EM step
    # (the CPU test double's BLAS Gram is symmetric only to rounding and the packed triangle mirrors one half: 1e-13 apart;
    # the HIP Gram kernel's G is exactly symmetric and the RCCL run of this protocol IS bit-equal, tests/test_gpu_rccl.py)
    np.testing.assert_array_equal(z["p_VFCIndex"], z["f_VFCIndex"])
    for k in ("V", "P", "C", "grid_V", "sigma2", "E_traj"):
        np.testing.assert_allclose(z[f"p_{k}"], z[f"f_{k}"], rtol=1e-9, atol=1e-11)
    assert tuple(z["row_range"]) == (0, 600)


def test_force_collectives_needs_a_process_group_for_the_torch_collective():
    sys.path.insert(0, HERE)
    import spateo_amd.vectorfield as vfm
    from _cpu_kernels import CpuKernels

    X = np.random.default_rng(0).standard_normal((40, 3))
    with pytest.raises(ValueError, match="force_collectives"):
        vfm.SparseVFCEngine(X, X, X[:5], 0.1, kernels=CpuKernels(), force_collectives=True)
    with pytest.raises(ValueError, match="collective"):
        vfm.SparseVFCEngine(X, X, X[:5], 0.1, kernels=CpuKernels(), collective="mpi")
