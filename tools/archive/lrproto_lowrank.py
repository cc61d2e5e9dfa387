"""CPU prototype (design probe, not product): truncation by pivoted Cholesky - exact greedy pivoting vs an un-pivoted
factor in maximin order - rank, trace of the remainder and field error against scipy.linalg.lstsq for tolerances
{1, 1/4, 1/16} eps lambda_max.  python tools/lrproto_lowrank.py /tmp/proto/sys_*.npz"""
import sys, numpy as np, scipy.linalg as sl, time
EPS = np.finfo(float).eps
d = np.load(sys.argv[1]); U = d["U"]; ctrl = d["ctrl"]; iters = int(d["iters"])
M = ctrl.shape[0]

def maximin(c):
    n = len(c); order = [int(np.argmin(((c - c.mean(0))**2).sum(1)))]
    dist = ((c - c[order[0]])**2).sum(1)
    for _ in range(n - 1):
        j = int(np.argmax(dist)); order.append(j); dist = np.minimum(dist, ((c - c[j])**2).sum(1))
    return np.array(order)

def pchol(A, tol, pivot=True, maxr=None):
    """right-looking (pivoted) Cholesky, stops when max remaining diag <= tol. returns L (M x r) in ORIGINAL row order, perm"""
    A = A.copy(); n = A.shape[0]; perm = np.arange(n); L = np.zeros((n, n)); dg = np.diag(A).copy()
    r = 0
    for j in range(n):
        if pivot:
            p = j + int(np.argmax(dg[j:]))
        else:
            p = j
        if dg[j:].max() <= tol: break
        if dg[p] <= 0: print("nonpos pivot at", j, dg[p], "max rem", dg[j:].max()); break
        if p != j:
            A[[j, p]] = A[[p, j]]; A[:, [j, p]] = A[:, [p, j]]; L[[j, p]] = L[[p, j]]; dg[[j, p]] = dg[[p, j]]; perm[[j, p]] = perm[[p, j]]
        col = A[j:, j] - L[j:, :j] @ L[j, :j]
        piv = col[0]
        L[j:, j] = col / np.sqrt(piv)
        dg[j+1:] -= L[j+1:, j]**2
        r = j + 1
    Lo = np.zeros((n, r)); Lo[perm] = L[:, :r]
    return Lo, perm, r, dg[r:].copy()

def field_err(C, Cref):
    V, Vr = U @ C, U @ Cref
    return np.linalg.norm(V - Vr) / np.linalg.norm(Vr)

mm = maximin(ctrl)
for it in range(iters):
    A = d[f"lhs{it}"]; R = d[f"rhs{it}"]
    A = 0.5 * (A + A.T)
    Cref = sl.lstsq(A, R)[0]
    w, Q = np.linalg.eigh(A)
    keep = np.abs(w) > EPS * np.abs(w).max()
    Ce = Q[:, keep] @ ((Q[:, keep].T @ R) / w[keep][:, None])
    print(f"it {it}: lmax {w.max():.3e} maxdiag {np.diag(A).max():.3e} rank(eps) {keep.sum()} floor(eigh vs lstsq) {field_err(Ce, Cref):.3e}")
    for tolf in (1.0, 0.25, 1/16.):
        tol = tolf * EPS * w.max()
        for name, order, piv in (("pivoted", np.arange(M), True), ("maximin-unpiv", mm, False)):
            Ao = A[np.ix_(order, order)]
            t = time.time(); L, perm, r, rem = pchol(Ao, tol, piv); 
            # SVD of L (M x r) : A ~ L L^T = Us S^2 Us^T
            Us, S, _ = np.linalg.svd(L, full_matrices=False)
            lam = S**2; k = lam > EPS * lam.max()
            Ro = R[order]
            Co = Us[:, k] @ ((Us[:, k].T @ Ro) / lam[k][:, None])
            C = np.zeros_like(Co); C[order] = Co
            resid = np.linalg.norm(Ao - L @ L.T, 2) if False else np.abs(rem).sum()
            print(f"   tol {tolf:6.3f}*eps*lmax {name:14s} r {r:5d} kept {k.sum():5d} trace(rem) {resid/ (EPS*w.max()):.2f} eps*lmax  err vs lstsq {field_err(C, Cref):.3e} vs eigh {field_err(C, Ce):.3e}", flush=True)
