#!/usr/bin/env python
"""CPU experiment (VERDICT r3 "next" #3): can the M-step be done on the control points that carry the numerical rank?

The float64 oracle is run with the control points restricted to a pivoted-Cholesky subset p of K = con_K(ctrl, ctrl)
(same beta, chosen once per fit; C is zero outside p, so V = U C holds exactly) and compared with the committed
full-M oracle fixtures of tests/golden/scale_oracle.npz through the per-quantity floors of tests/_floors.py.

    python tools/pivot_subset_experiment.py [case ...]      cases: m2000 m3000 m2000l3 m3000l3 c4
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from oracle import sparsevfc_oracle as svo  # noqa: E402
from spateo_amd._synthetic import make_config  # noqa: E402

EPS = np.finfo(float).eps


def pivoted_cholesky_subset(A, tol):
    """Greedy diagonally pivoted Cholesky of the PSD matrix A; stops when every remaining diagonal entry is <= tol.
    Returns the pivot order."""
    n = len(A)
    d = np.diag(A).copy()
    L = np.zeros((n, 0))
    piv = []
    Lbuf = np.zeros((n, n))
    r = 0
    while True:
        j = int(np.argmax(d))
        if d[j] <= tol or r == n:
            break
        col = A[:, j] - Lbuf[:, :r] @ Lbuf[j, :r]
        col /= np.sqrt(d[j])
        Lbuf[:, r] = col
        d -= col * col
        d[piv] = -np.inf
        d[j] = -np.inf
        piv.append(j)
        r += 1
    return np.array(piv)


def restricted_fit(X, Y, setup, p, kw):
    valid, Xv, Yv, idx, ctrl, beta = setup
    return svo.SparseVFC(X, Y, None, beta=beta, velocity_based_sampling=True, **kw, _ctrl_override=ctrl[p]) \
        if False else _fit_with_ctrl(Xv, Yv, ctrl[p], beta, kw)


def _fit_with_ctrl(X, Y, ctrl, beta, kw):
    """svo.SparseVFC's loop on given control points."""
    N, D = Y.shape
    M = len(ctrl)
    K = svo.con_K(ctrl, ctrl, beta)
    U = svo.con_K(X, ctrl, beta)
    V, C = np.zeros((N, D)), np.zeros((M, D))
    i, tecr, E, gamma = 0, 1, 1, 0.9
    sigma2 = np.sum((Y - V) ** 2) / (N * D)
    E_vec, s2 = [], []
    while i < kw["MaxIter"] and tecr > kw["ecr"] and sigma2 > 1e-8:
        P, E, tecr, C, V, sigma2, gamma = svo.em_step(U, K, Y, V, C, sigma2, gamma, E, a=5, lambda_=kw["lambda_"],
                                                      minP=1e-5, theta=0.75, lstsq_method="scipy")
        E_vec.append(E)
        s2.append(sigma2)
        i += 1
    return dict(V=V, P=P, sigma2=sigma2, E_traj=np.array(E_vec), iteration=i - 1, C=C)


def load_fixture(key):
    z = np.load(os.path.join(ROOT, "tests", "golden", "scale_oracle.npz"))
    return {n.split("|")[1]: z[n] for n in z.files if n.startswith(key + "|")}


def report(tag, got, fx, stride):
    dev = {"V": float(np.abs(got["V"][::stride] - fx["V"]).max() / float(fx["vmax"])),
           "sigma2": abs(got["sigma2"] - float(fx["sigma2"])) / float(fx["sigma2"]),
           "P": float(np.abs(got["P"][::stride] - fx["P"]).max()),
           "E": float(np.abs((got["E_traj"] - fx["E_traj"]) / fx["E_traj"]).max())}
    line = "; ".join(f"{q} {dev[q]:.2e} / floor {float(fx['floor_' + q][0]):.2e} (x{dev[q] / float(fx['floor_' + q][0]):.2f})"
                     for q in dev)
    print(f"{tag}: {line}", flush=True)
    return dev


CASES = {"m2000": ("C3", 20_000, 2000, 0.02, 1), "m3000": ("C3", 20_000, 3000, 0.02, 1),
         "m2000l3": ("C3", 20_000, 2000, 3.0, 1), "m3000l3": ("C3", 20_000, 3000, 3.0, 1),
         "c4": ("C4", 200_000, 3000, 0.02, 8)}


def main():
    for name in sys.argv[1:] or ["m2000", "m3000"]:
        cfg, n, M, lam, stride = CASES[name]
        X, Y, _ = make_config(cfg, N=n)
        setup = svo.sparsevfc_setup(X, Y, M=M, seed=0)
        ctrl, beta = setup[4], setup[5]
        K = svo.con_K(ctrl, ctrl, beta)
        lmax = np.linalg.eigvalsh(K)[-1]
        key = (f"c4_M{M}_lam{lam}_n{n}_s10" if cfg == "C4" else f"fit_M{M}_lam{lam}_n{n}_s10")
        fx = load_fixture(key)
        kw = dict(lambda_=lam, MaxIter=10, ecr=0.0)
        print(f"== {name}: {n} x {M}, lambda {lam}, lambda_max(K) {lmax:.3g}, numerical rank of K at eps: "
              f"{int((np.linalg.eigvalsh(K) > EPS * lmax).sum())}", flush=True)
        for tolf in [float(t) for t in os.environ.get("TOLS", "1 1e2 1e4 1e6").split()]:
            t0 = time.time()
            p = pivoted_cholesky_subset(K, tolf * EPS * lmax)
            got = _fit_with_ctrl(setup[1], setup[2], ctrl[p], beta, kw)
            report(f"  tol {tolf:g} eps lmax -> r = {len(p)} ({time.time() - t0:.0f}s)", got, fx, stride)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "A"):
    main()


# ---------------------------------------------------------------------------------------------------------------------
# second arm: the subset is read off the M-step matrix A = U^T P U + lambda sigma^2 K of EM iteration s0 + 1 (what the
# shipped rank-revealing solver finds every iteration: r ~ 0.28 M in the steady state) and kept for the rest of the fit
def fit_switching(X, Y, ctrl, beta, kw, s0, tolf):
    N, D = Y.shape
    M = len(ctrl)
    K = svo.con_K(ctrl, ctrl, beta)
    U = svo.con_K(X, ctrl, beta)
    V, C = np.zeros((N, D)), np.zeros((M, D))
    i, tecr, E, gamma = 0, 1, 1, 0.9
    sigma2 = np.sum((Y - V) ** 2) / (N * D)
    E_vec = []
    p = None
    while i < kw["MaxIter"]:
        if i == s0:
            P, _ = svo.get_P(Y, V, sigma2, gamma, 5)
            P = np.maximum(P, 1e-5)
            A = (U.T * P.T) @ U + kw["lambda_"] * sigma2 * K
            lmax = np.linalg.eigvalsh(A)[-1]
            p = pivoted_cholesky_subset(A, tolf * EPS * lmax)
            quad_full = np.trace(C.T @ K @ C)
            U, K, C = np.ascontiguousarray(U[:, p]), K[np.ix_(p, p)], np.zeros((len(p), D))
        P, E_new, tecr, Cn, V, sigma2, gamma = svo.em_step(U, K, Y, V, C, sigma2, gamma, E, a=5, lambda_=kw["lambda_"],
                                                           minP=1e-5, theta=0.75, lstsq_method="scipy")
        if i == s0:  # the energy's regulariser belongs to the previous (full) coefficients
            E_new += kw["lambda_"] / 2 * quad_full
        E, C = E_new, Cn
        E_vec.append(E)
        i += 1
    return dict(V=V, P=P, sigma2=sigma2, E_traj=np.array(E_vec), iteration=i - 1, r=len(p))


def main2():
    for name in sys.argv[2:]:
        cfg, n, M, lam, stride = CASES[name]
        X, Y, _ = make_config(cfg, N=n)
        setup = svo.sparsevfc_setup(X, Y, M=M, seed=0)
        key = (f"c4_M{M}_lam{lam}_n{n}_s10" if cfg == "C4" else f"fit_M{M}_lam{lam}_n{n}_s10")
        fx = load_fixture(key)
        kw = dict(lambda_=lam, MaxIter=10, ecr=0.0)
        print(f"== {name} (subset from the M-step matrix): {n} x {M}, lambda {lam}", flush=True)
        for s0 in [int(s) for s in os.environ.get("S0", "0 1 3").split()]:
            for tolf in [float(t) for t in os.environ.get("TOLS", "0.25").split()]:
                t0 = time.time()
                got = fit_switching(setup[1], setup[2], setup[4], setup[5], kw, s0, tolf)
                report(f"  switch after {s0} full iterations, tol {tolf:g} eps lmax(A) -> r = {got['r']} "
                       f"({time.time() - t0:.0f}s)", got, fx, stride)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "A":
    main2()
