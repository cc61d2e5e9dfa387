"""CPU prototype (design probe, not product): one-sided block Jacobi on the pivoted-Cholesky factor - cold sweeps, sweeps
after 1-3 Cholesky-QR ("LR") steps, warm start through CholQR2 of the projected factor.
python tools/lrproto_lr3.py /tmp/proto/sys_*.npz 0.25 1,2,3,4 cold,lr,warm"""
import sys, numpy as np, scipy.linalg as sl, time
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from jacobi_proto import rr_pairs, EPS
from jacobi_proto2 import inner_jacobi_rel
d = np.load(sys.argv[1]); U = d["U"]; iters = int(d["iters"])
TOLF = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
its_run = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(iters))

def pchol(A, tol):
    """exact greedy diagonal pivoting, no physical permutation; L (n x r), order"""
    n = A.shape[0]; dg = np.diag(A).copy(); L = np.zeros((n, 0)); cols = []; order = []
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

def solve_from(X, R):
    sig2 = (X * X).sum(0); kk = sig2 > EPS * sig2.max()
    W = X[:, kk] / np.sqrt(sig2[kk]); return W @ ((W.T @ R) / sig2[kk][:, None]), W, sig2[kk]

def field_err(C, Cref):
    V, Vr = U @ C, U @ Cref
    return np.linalg.norm(V - Vr) / np.linalg.norm(Vr)

def cholqr2(Z):
    for _ in range(2):
        G = Z.T @ Z; Rr = np.linalg.cholesky(G).T; Z = sl.solve_triangular(Rr, Z.T, trans="T", lower=False).T
    return Z

Qprev = None
for it in its_run:
    A = d[f"lhs{it}"]; R = d[f"rhs{it}"]; A = 0.5 * (A + A.T); Cref = d[f"C{it}"]
    w = np.linalg.eigvalsh(A); lmax = w[-1]; tol = TOLF * EPS * lmax
    Q_, = (None,)
    wq, Qe = np.linalg.eigh(A); keep = wq > EPS * lmax; Ce = Qe[:, keep] @ ((Qe[:, keep].T @ R) / wq[keep][:, None])
    t = time.time(); L, order, remt = pchol(A, tol); t1 = time.time() - t; r = L.shape[1]
    print(f"it {it}: rank(eps) {keep.sum()} floor {field_err(Ce, Cref):.3e}; pchol r {r} ({t1:.1f}s) rem {remt/(EPS*lmax):.1f} eps*lmax", flush=True)
    cn = (L * L).sum(0); print("    col norm^2 of L: first %.2e last %.2e; monotone violations %d" % (cn[0], cn[-1], (np.diff(cn) > 0).sum()))
    if "cold" in sys.argv[4]:
        t = time.time(); X, hist = onesided_rect(L); Cj, W, s2 = solve_from(X, R)
        print(f"    cold: sweeps {len(hist)} rot {hist} kept {len(s2)} err vs lstsq {field_err(Cj, Cref):.3e} vs eigh {field_err(Cj, Ce):.3e} ({time.time()-t:.0f}s)", flush=True)
    if "lr" in sys.argv[4]:
        # LR preconditioning: k extra Cholesky-QR steps on the r x r level: L = Q1 L2^T ...
        Lk = L
        for k in range(1, 4):
            B = Lk.T @ Lk
            try: L2 = np.linalg.cholesky(B)
            except np.linalg.LinAlgError: print("    LR step", k, "cholesky failed"); break
            Lk = L2  # eigen(Lk Lk^T)=eigen(B); continue on r x r
            t = time.time(); X, hist = onesided_rect(Lk)
            print(f"    after {k} LR steps (r x r): sweeps {len(hist)} rot {hist} ({time.time()-t:.0f}s)", flush=True)
    if "warm" in sys.argv[4]:
        if Qprev is not None:
            Z = (Qprev.T @ L).T          # r x k : columns ~ sigma_i v_i
            nz = np.sqrt((Z * Z).sum(0)); o = np.argsort(-nz)[:r]; Z = Z[:, o] / nz[o]
            if Z.shape[1] < r:   # complete with unit vectors least represented
                miss = r - Z.shape[1]; lev = (Z * Z).sum(1); add = np.argsort(lev)[:miss]; E = np.zeros((r, miss)); E[add, np.arange(miss)] = 1; Z = np.concatenate([Z, E], 1)
            print("    warm: cond(Z) %.2e" % np.linalg.cond(Z))
            V0 = cholqr2(Z); print("    orth err %.2e" % np.abs(V0.T @ V0 - np.eye(r)).max())
            t = time.time(); X, hist = onesided_rect(L @ V0); Cw, W, s2 = solve_from(X, R)
            print(f"    warm: sweeps {len(hist)} rot {hist} kept {len(s2)} err vs lstsq {field_err(Cw, Cref):.3e} vs eigh {field_err(Cw, Ce):.3e} ({time.time()-t:.0f}s)", flush=True)
            Qprev = W
        else:
            Us, S, _ = np.linalg.svd(L, full_matrices=False); Qprev = Us[:, S**2 > EPS * S[0]**2]
