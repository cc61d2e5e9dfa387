#!/usr/bin/env python
"""Writes tests/golden/stream_oracle_<case>.npz: the float64 STREAMED oracle (oracle/streamed_oracle.py - this repo's
restatement of dynamo's SparseVFC, U generated chunk by chunk; not the reference, which cannot run here) at the sizes
the benchmark and the 8-GPU split run at:

    c3_full   BASELINE config 3 at its stated size: 2 000 000 cells x 2000 control points, lambda_ = 0.02, 5 EM iterations
    c4_rank   one rank's share of config 4:         1 000 000 cells x 3000,                lambda_ = 0.02, 5 EM iterations
    c4_step   config 4 itself, ONE EM iteration:    8 000 000 cells x 3000,                lambda_ = 0.02
    c4_3step  config 4 itself, THREE EM iterations (round 6; floor = the eigh variant alone)

Stored per case: every `stride`-th cell of the final V and P, sigma^2 and the energy after every iteration, max |V|,
the control-point draw and beta, and the reference's own noise floors per quantity (tests/_floors.py's two deterministic
variants, evaluated on ALL cells): `eigh` (scipy.linalg.lstsq -> truncated symmetric eigendecomposition with the same
eps cut-off) and `sumorder` (the sums over cells made of a different number of sequential pieces).

    python tests/golden/make_stream_oracle.py c3_full|c4_rank|c4_step     (8 cores, round 4: 75 / 75 / 87 minutes, < 8 GB)
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), os.path.dirname(HERE)):
    sys.path.insert(0, p)

import _floors as F  # noqa: E402
from oracle import sparsevfc_oracle as svo  # noqa: E402
from oracle import streamed_oracle as so  # noqa: E402

CASES = {
    # name: generator config, cells, M, EM iterations, stride of the stored cells, chunk counts (base, sumorder variant)
    "c3_full": dict(cfg="C3", n=2_000_000, M=2000, steps=5, stride=64, chunks=(32, 13)),
    "c4_rank": dict(cfg="C4", n=1_000_000, M=3000, steps=5, stride=32, chunks=(16, 7)),
    "c4_step": dict(cfg="C4", n=8_000_000, M=3000, steps=1, stride=256, chunks=(128, 50)),
    # round 6: config 4 itself for THREE EM iterations (the rank-deficient regime starts at the second); floors from the eigh
    # variant only - every variant is a full three-iteration run of ~40 minutes per iteration on 8 cores
    "c4_3step": dict(cfg="C4", n=8_000_000, M=3000, steps=3, stride=256, chunks=(128, None)),
    # seconds-sized twin of the above for the CPU test of this script's plumbing
    "tiny": dict(cfg="C3", n=6_000, M=150, steps=3, stride=4, chunks=(5, 3)),
    "tiny3": dict(cfg="C3", n=6_000, M=150, steps=3, stride=4, chunks=(5, None)),
}
LAMBDA = 0.02
T0 = time.time()


def log(*a):
    print(f"[{time.time() - T0:7.0f}s]", *a, flush=True)


def progress(tag):
    last = [0.0]

    def p(what, done, total):
        if time.time() - last[0] > 60 or done == total:
            last[0] = time.time()
            log(tag, what, f"{done}/{total}")
    return p


def devs(got, ref):
    vmax = np.abs(ref["V"]).max()
    return {"V": float(np.abs(got["V"] - ref["V"]).max() / vmax),
            "sigma2": float(abs(got["sigma2"] - ref["sigma2"]) / ref["sigma2"]),
            "P": float(np.abs(got["P"] - ref["P"]).max()),
            "E": float(np.abs((got["E_traj"] - ref["E_traj"]) / ref["E_traj"]).max())}


def run(name):
    from spateo_amd._synthetic import make_config

    c = CASES[name]
    X, Y, _ = make_config(c["cfg"], N=c["n"])
    setup = svo.sparsevfc_setup(X, Y, M=c["M"], seed=0)
    log(name, "setup done: beta", setup[5])
    kw = dict(M=c["M"], lambda_=LAMBDA, MaxIter=c["steps"], ecr=0.0, seed=0, setup=setup)
    variants = {}
    if c["steps"] == 1:
        # one iteration: the eigh variant shares the assembled system with the base run
        em = so.StreamedEM(setup[1], setup[2], setup[4], setup[5], c["chunks"][0], progress=progress("base"))
        E, _ = em.step(lambda_=LAMBDA, keep_system=True)
        ref = dict(V=em.V, P=em.P, sigma2=em.sigma2, E_traj=np.array([E]), sigma2_traj=np.array([em.sigma2]), iteration=0)
        log("base done: sigma2", em.sigma2)
        C2 = F.eigh_solver(em.lhs, em.rhs)
        V2 = em.apply(C2)
        s2 = float(em.P[:, 0].dot(np.sum((em.Y - V2) ** 2, 1)) / (np.sum(em.P) * em.D))
        variants["eigh"] = devs(dict(V=V2, P=em.P, sigma2=s2, E_traj=ref["E_traj"]), ref)
        log("eigh", variants["eigh"])
        del em
    elif c["chunks"][1] is None:
        # several iterations, eigh variant only: the two runs share the FIRST iteration's assembled system (the variant
        # differs from the base run in the solver alone, and P of iteration 1 precedes every solve), then go their own ways
        def trajectory(em, E0, s0):
            E_t, s_t = [E0], [s0]
            for _ in range(c["steps"] - 1):
                E, _ = em.step(lambda_=LAMBDA)
                E_t.append(E)
                s_t.append(em.sigma2)
                log("  iteration", len(E_t), "sigma2", em.sigma2)
            return dict(V=em.V, P=em.P, sigma2=em.sigma2, E_traj=np.array(E_t), sigma2_traj=np.array(s_t),
                        iteration=c["steps"] - 1)

        em = so.StreamedEM(setup[1], setup[2], setup[4], setup[5], c["chunks"][0], progress=progress("base"))
        E1, tecr1 = em.step(lambda_=LAMBDA, keep_system=True)
        log("iteration 1 done: sigma2", em.sigma2)
        em2 = so.StreamedEM(setup[1], setup[2], setup[4], setup[5], c["chunks"][0], solver=F.eigh_solver,
                            progress=progress("eigh"))
        em2._buf = em._buf
        C2 = F.eigh_solver(em.lhs, em.rhs)
        V2 = em.apply(C2)
        s2 = float(em.P[:, 0].dot(np.sum((em.Y - V2) ** 2, 1)) / (np.sum(em.P) * em.D))
        em2.P, em2.E, em2.tecr, em2.C, em2.V, em2.sigma2, em2.gamma = em.P, em.E, em.tecr, C2, V2, s2, em.gamma
        em.lhs = em.rhs = None
        ref = trajectory(em, E1, em.sigma2)
        log("base done: sigma2", ref["sigma2_traj"])
        got = trajectory(em2, E1, s2)
        variants["eigh"] = devs(got, ref)
        log("eigh", variants["eigh"])
        del got, em, em2
    else:
        ref = so.SparseVFC_streamed(X, Y, chunks=c["chunks"][0], progress=progress("base"), **kw)
        log("base done: sigma2", ref["sigma2_traj"])
        got = so.SparseVFC_streamed(X, Y, chunks=c["chunks"][0], solver=F.eigh_solver, progress=progress("eigh"), **kw)
        variants["eigh"] = devs(got, ref)
        log("eigh", variants["eigh"])
        del got
    if c["chunks"][1] is not None:
        got = so.SparseVFC_streamed(X, Y, chunks=c["chunks"][1], progress=progress("sumorder"), **kw)
        variants["sumorder"] = devs(got, ref)
        log("sumorder", variants["sumorder"])
    st = c["stride"]
    out = dict(V=ref["V"][::st], P=ref["P"][::st], sigma2=ref["sigma2"], E_traj=ref["E_traj"],
               sigma2_traj=ref["sigma2_traj"], iteration=ref["iteration"], vmax=np.abs(ref["V"]).max(), stride=st,
               ctrl_idx=setup[3], beta=setup[5], n=c["n"], M=c["M"], lambda_=LAMBDA, steps=c["steps"])
    for q in ("V", "sigma2", "P", "E"):
        out[f"floor_{q}"] = max(v[q] for v in variants.values())
        for v, d in variants.items():
            out[f"var_{v}_{q}"] = d[q]
    return out


def main():
    for name in sys.argv[1:]:
        out = run(name)
        path = os.path.join(HERE, f"stream_oracle_{name}.npz")
        np.savez_compressed(path, **out)
        log("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
