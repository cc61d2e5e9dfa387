"""NumPy prototype of the block-Jacobi symmetric eigensolver behind mvf_solve_minnorm (design probe, CPU only).

Mirrors the GPU structure: blocks of b columns, round-robin tournament over block pairs, per pair a 2b x 2b cyclic
Jacobi (parallel ordering) that produces an orthogonal J; tiles are updated as J_I^T T J_Jc, V <- V J.
Reports sweeps to convergence (cold and warm-started across EM iterations) and the field deviation of the truncated
minimum-norm solve from scipy.linalg.lstsq (gelsd).
"""
import sys
import time

import numpy as np
import scipy.linalg as sl

sys.path.insert(0, ".")
sys.path.insert(0, "spateo-release_amd")
EPS = np.finfo(float).eps


def rr_pairs(n):
    """round-robin tournament: n even -> list of n-1 rounds, each n/2 disjoint (p, q) pairs"""
    idx = list(range(n))
    rounds = []
    for _ in range(n - 1):
        rounds.append([(min(idx[i], idx[n - 1 - i]), max(idx[i], idx[n - 1 - i])) for i in range(n // 2)])
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
    return rounds


def inner_jacobi(S, sweeps, tol_abs):
    """batched cyclic Jacobi on S (B, k, k) symmetric; returns (S', J) with S' = J^T S J.  Vectorised over batch and
    over the k/2 disjoint rotations of a round."""
    B, k, _ = S.shape
    J = np.broadcast_to(np.eye(k), (B, k, k)).copy()
    rounds = rr_pairs(k)
    nrot = 0
    for _ in range(sweeps):
        for rd in rounds:
            p = np.array([a for a, _ in rd])
            q = np.array([b for _, b in rd])
            app = S[:, p, p]
            aqq = S[:, q, q]
            apq = S[:, p, q]
            act = (np.abs(apq) > EPS * np.sqrt(np.abs(app * aqq))) & (np.abs(apq) > tol_abs)
            safe = np.where(act, apq, 1.0)
            theta = (aqq - app) / (2 * safe)
            t = np.where(theta >= 0, 1.0, -1.0) / (np.abs(theta) + np.sqrt(theta * theta + 1))
            c = 1 / np.sqrt(t * t + 1)
            s = t * c
            c = np.where(act, c, 1.0)
            s = np.where(act, s, 0.0)
            nrot += int(act.sum())
            # rows
            Sp, Sq = S[:, p, :].copy(), S[:, q, :].copy()
            S[:, p, :] = c[:, :, None] * Sp - s[:, :, None] * Sq
            S[:, q, :] = s[:, :, None] * Sp + c[:, :, None] * Sq
            Sp, Sq = S[:, :, p].copy(), S[:, :, q].copy()
            S[:, :, p] = c[:, None, :] * Sp - s[:, None, :] * Sq
            S[:, :, q] = s[:, None, :] * Sp + c[:, None, :] * Sq
            Jp, Jq = J[:, :, p].copy(), J[:, :, q].copy()
            J[:, :, p] = c[:, None, :] * Jp - s[:, None, :] * Jq
            J[:, :, q] = s[:, None, :] * Jp + c[:, None, :] * Jq
    return S, J, nrot


def block_jacobi(A, b=32, V0=None, inner_sweeps=2, max_sweeps=30, verbose=True, tol_rel=1.0):
    """returns (lam, V, sweeps).  A symmetric (m x m)."""
    m = A.shape[0]
    mp = -(-m // (2 * b)) * 2 * b
    Ap = np.zeros((mp, mp))
    Ap[:m, :m] = 0.5 * (A + A.T)
    V = np.eye(mp)
    if V0 is not None:
        V = V0.copy()
        Ap = V.T @ Ap @ V
        Ap = 0.5 * (Ap + Ap.T)
    nb = mp // b
    rounds = rr_pairs(nb)
    dmax = np.abs(np.diag(Ap)).max()
    tol_abs = tol_rel * EPS * dmax / 8
    hist = []
    for sweep in range(max_sweeps):
        off0 = np.sqrt(max(np.sum(Ap * Ap) - np.sum(np.diag(Ap) ** 2), 0.0))
        tot_rot = 0
        for rd in rounds:
            order = np.array([x for pq in rd for x in pq])  # block order: pairs adjacent
            perm = (order[:, None] * b + np.arange(b)[None, :]).reshape(-1)
            A4 = Ap[np.ix_(perm, perm)].reshape(nb // 2, 2 * b, nb // 2, 2 * b)
            S = np.stack([A4[i, :, i, :] for i in range(nb // 2)])
            S = 0.5 * (S + S.transpose(0, 2, 1))
            _, J, nrot = inner_jacobi(S.copy(), inner_sweeps, tol_abs)
            tot_rot += nrot
            if nrot == 0:
                continue
            # T_IJ <- J_I^T T_IJ J_J
            A4 = np.einsum("iab,iajc->ibjc", J, A4)
            A4 = np.einsum("ibjc,jcd->ibjd", A4, J)
            Ap[np.ix_(perm, perm)] = A4.reshape(mp, mp)
            Vp = V[:, perm].reshape(mp, nb // 2, 2 * b)
            V[:, perm] = np.einsum("rjc,jcd->rjd", Vp, J).reshape(mp, mp)
        Ap = 0.5 * (Ap + Ap.T)
        off1 = np.sqrt(max(np.sum(Ap * Ap) - np.sum(np.diag(Ap) ** 2), 0.0))
        hist.append((off0, off1, tot_rot))
        if verbose:
            print(f"   sweep {sweep}: off {off0:.3e} -> {off1:.3e}  rotations {tot_rot}")
        if tot_rot == 0:
            break
    return np.diag(Ap)[:].copy(), V, len(hist), hist


def minnorm_from_eig(lam, V, B, m, rcond=EPS):
    keep = np.abs(lam) > rcond * np.abs(lam).max()
    inv = np.where(keep, 1.0 / np.where(keep, lam, 1.0), 0.0)
    Vm = V[:m]
    return Vm @ (inv[:, None] * (Vm.T @ B)), int(keep.sum())


def collect(N, M, lambda_, iters, cfg="C2"):
    from oracle import sparsevfc_oracle as svo
    from spateo_amd._synthetic import make_config

    X, Y, _ = make_config(cfg, N=N)
    valid, Xv, Yv, idx, ctrl, beta = svo.sparsevfc_setup(X, Y, M=M, seed=0)
    K = svo.con_K(ctrl, ctrl, beta)
    U = svo.con_K(Xv, ctrl, beta)
    Nn, D = Yv.shape
    Vc, C = np.zeros((Nn, D)), np.zeros((M, D))
    s2, gamma, E = np.sum(Yv**2) / (Nn * D), 0.9, 1
    out = []
    for it in range(iters):
        P, _ = svo.get_P(Yv, Vc, s2, gamma, 5)
        P = np.maximum(P, 1e-5)
        UP = U.T * P.T
        lhs = UP @ U + lambda_ * s2 * K
        rhs = UP @ Yv
        out.append((lhs.copy(), rhs.copy()))
        P, E, tecr, C, Vc, s2, gamma = svo.em_step(U, K, Yv, Vc, C, s2, gamma, E, a=5, lambda_=lambda_, minP=1e-5,
                                                   theta=0.75, lstsq_method="scipy")
    return U, out


if __name__ == "__main__":
    N, M = int(sys.argv[1]), int(sys.argv[2])
    lam_ = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    inner = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    U, systems = collect(N, M, lam_, iters)
    Vprev = None
    for it, (lhs, rhs) in enumerate(systems):
        C_ref = sl.lstsq(lhs, rhs)[0]
        w, Q = np.linalg.eigh(lhs)
        keep = np.abs(w) > EPS * np.abs(w).max()
        C_eigh = Q @ (np.where(keep, 1 / np.where(keep, w, 1), 0)[:, None] * (Q.T @ rhs))
        F_ref = U @ C_ref
        sc = np.abs(F_ref).max()
        print(f"iter {it}: cond est {np.abs(w).max() / max(np.abs(w).min(), 1e-300):.2e}  rank kept {keep.sum()}/{M}  "
              f"floor(lstsq vs eigh) {np.abs(U @ C_eigh - F_ref).max() / sc:.2e}")
        for name, V0 in (("cold", None), ("warm", Vprev)):
            if V0 is None and name == "warm":
                continue
            t = time.time()
            lam, V, ns, hist = block_jacobi(lhs, b=32, V0=V0, inner_sweeps=inner, verbose=False)
            C_j, kept = minnorm_from_eig(lam, V, rhs, M)
            dev = np.abs(U @ C_j - F_ref).max() / sc
            ev = np.sort(lam)[-M:]
            print(f"   {name}: sweeps {ns} rot/sweep {[h[2] for h in hist]}  kept {kept}  field dev vs lstsq {dev:.2e}  "
                  f"eig err {np.abs(ev - w).max() / np.abs(w).max():.1e}  orth {np.abs(V.T @ V - np.eye(len(V))).max():.1e} "
                  f"({time.time() - t:.1f}s)")
            if name == "cold":
                Vcold = V
        Vprev = V if Vprev is not None else Vcold
