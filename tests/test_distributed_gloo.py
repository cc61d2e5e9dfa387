"""world_size-2 ``gloo`` runs of the multi-GPU path on CPU: cells block-sharded across ranks, one all-reduce of
[G | R | stats] per EM step, the global min-non-zero rule of the E-step, gathered outputs.  The device is replaced by
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


def _worker(rank, world, port, case, out_dir):
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

        Grid = X[::30]
        kw = dict(M=25, lambda_=3.0, lstsq_method="scipy", MaxIter=6, seed=0)
        got = st.SparseVFC(X, V, Grid, distributed=True, _kernels=Recording(), **kw)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), V=got["V"], P=got["P"], C=got["C"], grid_V=got["grid_V"],
                 sigma2=got["sigma2"], iteration=got["iteration"], E=got["E_traj"], fills=np.array(Recording.fills))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["plain", "zeros"])
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


def test_distributed_flag_requires_process_group():
    sys.path.insert(0, HERE)
    import spateo_amd as st
    from _cpu_kernels import CpuKernels

    X = np.random.default_rng(0).standard_normal((40, 3))
    with pytest.raises(RuntimeError, match="torch.distributed is not initialised"):
        st.SparseVFC(X, X, None, M=5, distributed=True, _kernels=CpuKernels())
