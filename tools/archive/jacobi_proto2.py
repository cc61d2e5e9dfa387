"""NumPy prototype #2: eigen-decomposition of A + delta I = L L^T through ONE-SIDED block Jacobi on the Cholesky factor
(Veselic-Hari preconditioning): orthogonalise the columns of L; W = X Sigma^-1, lambda = sigma^2 - delta."""
import sys, time
import numpy as np
import scipy.linalg as sl
sys.path.insert(0, "."); sys.path.insert(0, "spateo-release_amd"); sys.path.insert(0, "tools")
from jacobi_proto import rr_pairs, collect, EPS

def inner_jacobi_rel(S, sweeps, tol):
    B, k, _ = S.shape
    J = np.broadcast_to(np.eye(k), (B, k, k)).copy()
    rounds = rr_pairs(k)
    nrot = 0
    for _ in range(sweeps):
        for rd in rounds:
            p = np.array([a for a, _ in rd]); q = np.array([b for _, b in rd])
            app = S[:, p, p]; aqq = S[:, q, q]; apq = S[:, p, q]
            act = np.abs(apq) > tol * np.sqrt(np.abs(app * aqq))
            act &= apq != 0
            safe = np.where(act, apq, 1.0)
            theta = (aqq - app) / (2 * safe)
            t = np.where(theta >= 0, 1.0, -1.0) / (np.abs(theta) + np.sqrt(theta * theta + 1))
            c = 1 / np.sqrt(t * t + 1); s = t * c
            c = np.where(act, c, 1.0); s = np.where(act, s, 0.0)
            nrot += int(act.sum())
            Sp, Sq = S[:, p, :].copy(), S[:, q, :].copy()
            S[:, p, :] = c[:, :, None] * Sp - s[:, :, None] * Sq
            S[:, q, :] = s[:, :, None] * Sp + c[:, :, None] * Sq
            Sp, Sq = S[:, :, p].copy(), S[:, :, q].copy()
            S[:, :, p] = c[:, None, :] * Sp - s[:, None, :] * Sq
            S[:, :, q] = s[:, None, :] * Sp + c[:, None, :] * Sq
            Jp, Jq = J[:, :, p].copy(), J[:, :, q].copy()
            J[:, :, p] = c[:, None, :] * Jp - s[:, None, :] * Jq
            J[:, :, q] = s[:, None, :] * Jp + c[:, None, :] * Jq
    return J, nrot

def onesided(L, b=32, inner_sweeps=1, tol=4 * EPS, max_sweeps=40, sort=True, verbose=False, Vacc=None):
    m = L.shape[0]
    mp = -(-m // (2 * b)) * 2 * b
    X = np.zeros((m, mp)); X[:, :m] = L
    V = None
    if Vacc is not None:
        V = np.zeros((m, mp)); V[:, :m] = Vacc
    if sort:
        o = np.argsort(-np.sum(X * X, 0), kind="stable"); X = X[:, o]
        if V is not None: V = V[:, o]
    nb = mp // b
    rounds = rr_pairs(nb)
    hist = []
    for sweep in range(max_sweeps):
        tot = 0
        for rd in rounds:
            order = np.array([x for pq in rd for x in pq])
            perm = (order[:, None] * b + np.arange(b)[None, :]).reshape(-1)
            Xp = X[:, perm].reshape(m, nb // 2, 2 * b)
            S = np.einsum("rja,rjb->jab", Xp, Xp)
            J, nrot = inner_jacobi_rel(S, inner_sweeps, tol)
            tot += nrot
            if nrot == 0: continue
            X[:, perm] = np.einsum("rjc,jcd->rjd", Xp, J).reshape(m, mp)
            if V is not None:
                Vp = V[:, perm].reshape(m, nb // 2, 2 * b)
                V[:, perm] = np.einsum("rjc,jcd->rjd", Vp, J).reshape(m, mp)
        hist.append(tot)
        if verbose: print("   sweep", sweep, "rot", tot)
        if tot == 0: break
    sig2 = np.sum(X * X, 0)
    return X, sig2, hist, V

def solve_from(X, sig2, delta, B, rcond=EPS):
    lam = sig2 - delta
    nz = sig2 > 0
    W = np.where(nz[None, :], X / np.sqrt(np.where(nz, sig2, 1.0))[None, :], 0.0)
    keep = nz & (np.abs(lam) > rcond * np.abs(lam).max())
    inv = np.where(keep, 1 / np.where(keep, lam, 1.0), 0.0)
    return W @ (inv[:, None] * (W.T @ B)), int(keep.sum()), W, lam

def chol_shift(A, rel=1e-13):
    md = np.mean(np.diag(A)); d = rel * md
    while True:
        try:
            return np.linalg.cholesky(A + d * np.eye(len(A))), d
        except np.linalg.LinAlgError:
            d *= 10

if __name__ == "__main__":
    N, M = int(sys.argv[1]), int(sys.argv[2])
    lam_ = float(sys.argv[3]); iters = int(sys.argv[4]); inner = int(sys.argv[5])
    U, systems = collect(N, M, lam_, iters)
    Wprev = None
    for it, (lhs, rhs) in enumerate(systems):
        lhs = 0.5 * (lhs + lhs.T)
        C_ref = sl.lstsq(lhs, rhs)[0]
        w, Q = np.linalg.eigh(lhs)
        keep = np.abs(w) > EPS * np.abs(w).max()
        C_eigh = Q @ (np.where(keep, 1 / np.where(keep, w, 1), 0)[:, None] * (Q.T @ rhs))
        F_ref = U @ C_ref; sc = np.abs(F_ref).max()
        print(f"iter {it}: lam range {w.max():.3e} .. {w.min():.3e} kept {keep.sum()}/{M} floor(lstsq vs eigh) {np.abs(U @ C_eigh - F_ref).max() / sc:.2e}")
        L, d = chol_shift(lhs)
        for sort in (True,):
            t = time.time()
            X, sig2, hist, _ = onesided(L, inner_sweeps=inner, sort=sort)
            C_j, kept, W, lam = solve_from(X, sig2, d, rhs)
            dev = np.abs(U @ C_j - F_ref).max() / sc
            lam_s = np.sort(lam)[-M:]
            nzc = sig2 > 0
            orth = np.abs(W[:, nzc].T @ W[:, nzc] - np.eye(nzc.sum())).max()
            print(f"   cold delta/mean {d/np.mean(np.diag(lhs)):.0e} sort={sort}: sweeps {len(hist)} rot {hist} kept {kept} dev vs lstsq {dev:.2e} "
                  f"eig err {np.abs(lam_s - w).max() / w.max():.1e} orth {orth:.1e} ({time.time()-t:.1f}s)")
        if Wprev is not None:
            t = time.time()
            A2 = Wprev.T @ lhs @ Wprev; A2 = 0.5 * (A2 + A2.T)
            L2, d2 = chol_shift(A2)
            X, sig2, hist, _ = onesided(L2, inner_sweeps=inner, sort=True)
            C2, kept, W2, lam = solve_from(X, sig2, d2, Wprev.T @ rhs)
            C_w = Wprev @ C2
            dev = np.abs(U @ C_w - F_ref).max() / sc
            print(f"   warm: sweeps {len(hist)} rot {hist} kept {kept} dev vs lstsq {dev:.2e} ({time.time()-t:.1f}s)")
        nzc = sig2 > 0
        Wprev = W[:, np.argsort(-lam)[:M]]
