"""CPU prototype (design probe, not product) of the COMPACT rank-revealing solve: after the pivoted Cholesky A ~ L L^T
(L m x r) the Jacobi iteration runs on the r x r triangular factor X = chol(L^T L)^T instead of on the r rows of length m
(one Cholesky-LR step ahead, rows 3.5 x shorter), and the truncated solve is C = L W S^-4 W^T (L^T R) with X = W S Z^T.
Checks: the Cholesky of L^T L exists, the field against scipy.linalg.lstsq / the eigh variant / the present path, sweeps.
python tools/lrproto_compact.py /tmp/proto/sys_*.npz 3,5 [sweeps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from jacobi_proto import EPS  # noqa: E402
from jacobi_proto import rr_pairs  # noqa: E402
from jacobi_proto2 import inner_jacobi_rel  # noqa: E402

d = np.load(sys.argv[1])
U = d["U"]


def pchol(A, tol):
    """exact greedy diagonal pivoting, no physical permutation; L (n x r), order"""
    n = A.shape[0]; dg = np.diag(A).copy(); order = []
    Lm = np.zeros((n, n)); r = 0; used = np.zeros(n, bool)
    while True:
        dm = np.where(used, -np.inf, dg); p = int(np.argmax(dm))
        if dm[p] <= tol: break
        c = A[:, p] - Lm[:, :r] @ Lm[p, :r]
        c[used] = 0.0
        c /= np.sqrt(c[p]); Lm[:, r] = c; dg -= c * c; used[p] = True; order.append(p); r += 1
    return Lm[:, :r].copy(), np.array(order), np.where(used, 0, dg).sum()


def onesided_rect(Xin, b=32, tol=None, max_sweeps=40):
    m, r = Xin.shape
    if tol is None: tol = np.sqrt(m) * EPS
    rp = -(-r // (2 * b)) * 2 * b
    X = np.zeros((m, rp)); X[:, :r] = Xin
    nb = rp // b; rounds = rr_pairs(nb); hist = []
    for sweep in range(max_sweeps):
        tot = 0
        for ri, rd in enumerate(rounds):
            order = np.array([x for pq in rd for x in pq])
            perm = (order[:, None] * b + np.arange(b)[None, :]).reshape(-1)
            Xp = X[:, perm].reshape(m, nb // 2, 2 * b)
            S = np.einsum("rja,rjb->jab", Xp, Xp)
            J, nrot = inner_jacobi_rel(S, 1, tol)
            tot += nrot
            if nrot == 0: continue
            X[:, perm] = np.einsum("rjc,jcd->rjd", Xp, J).reshape(m, rp)
        hist.append(tot)
        if tot == 0: break
    return X, hist


def field_err(C, Cref):
    V, Vr = U @ C, U @ Cref
    return np.linalg.norm(V - Vr) / np.linalg.norm(Vr)

its = [int(x) for x in sys.argv[2].split(",")]
for it in its:
    A = d[f"lhs{it}"]; R = d[f"rhs{it}"]; A = 0.5 * (A + A.T); Cref = d[f"C{it}"]
    wq, Qe = np.linalg.eigh(A); lmax = wq[-1]; keep = wq > EPS * lmax
    Ce = Qe[:, keep] @ ((Qe[:, keep].T @ R) / wq[keep][:, None])
    L, order, rem = pchol(A, 0.25 * EPS * lmax)
    r = L.shape[1]
    # present path: orthogonalise the columns of L (exact stand-in: SVD)
    Uu, s, _ = np.linalg.svd(L, full_matrices=False)
    k1 = s**2 > EPS * s[0]**2
    Ccur = Uu[:, k1] @ ((Uu[:, k1].T @ R) / (s[k1]**2)[:, None])
    # compact path
    S2 = L.T @ L
    Rt = np.linalg.cholesky(S2)            # S2 = Rt Rt^T (lower); raises if it does not exist
    X = Rt.T                                # rows to be orthogonalised: X X^T = Rt^T Rt (one LR step ahead of S2)
    Wx, sx, _ = np.linalg.svd(X.T, full_matrices=False)   # X^T = Wx sx Zx^T  ->  S2 = Rt Rt^T = X^T X = Wx sx^2 Wx^T
    k2 = sx**2 > EPS * sx[0]**2
    T = Wx[:, k2].T @ (L.T @ R)
    Ccmp = L @ (Wx[:, k2] @ (T / (sx[k2]**4)[:, None]))
    print(f"it {it}: r {r}, kept eigh {keep.sum()} / present {k1.sum()} / compact {k2.sum()}; "
          f"sigma^2 agreement present vs compact (kept): {np.abs(sx[k2][:min(k1.sum(), k2.sum())]**2 / s[k1][:min(k1.sum(), k2.sum())]**2 - 1).max():.2e}")
    print(f"    field vs lstsq: eigh-variant {field_err(Ce, Cref):.3e} | present {field_err(Ccur, Cref):.3e} | compact {field_err(Ccmp, Cref):.3e}"
          f" | compact vs present {field_err(Ccmp, Ccur):.3e}", flush=True)
    if len(sys.argv) > 3:
        t = time.time(); _, h1 = onesided_rect(L); print(f"    sweeps on the r columns of L (length m): {len(h1)} {h1} ({time.time() - t:.0f}s)", flush=True)
        t = time.time(); _, h2 = onesided_rect(X.T); print(f"    sweeps on the r rows of X (length r): {len(h2)} {h2} ({time.time() - t:.0f}s)", flush=True)
