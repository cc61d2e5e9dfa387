"""CPU prototype (design probe, not product): the previous EM iteration's pivot order as a hint (accept while the pivot
exceeds theta x the largest remaining diagonal entry) and its accumulated rotations as the Jacobi start.
python tools/lrproto_lr4.py /tmp/proto/sys_*.npz 0.1"""
import sys, numpy as np, scipy.linalg as sl, time
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from jacobi_proto import rr_pairs, EPS
from jacobi_proto2 import inner_jacobi_rel
d = np.load(sys.argv[1]); U = d["U"]; iters = int(d["iters"])
THETA = float(sys.argv[2])

def pchol_hint(A, tol, hint=None, theta=0.1):
    n = A.shape[0]; dg = np.diag(A).copy(); Lm = np.zeros((n, n)); r = 0; used = np.zeros(n, bool); order = []; hp = 0; followed = 0
    while True:
        dm = np.where(used, -np.inf, dg); pmax = int(np.argmax(dm))
        if dm[pmax] <= tol: break
        p = pmax
        if hint is not None:
            while hp < len(hint) and used[hint[hp]]: hp += 1
            if hp < len(hint) and dm[hint[hp]] > max(tol, theta * dm[pmax]): p = hint[hp]; followed += 1
        c = A[:, p] - Lm[:, :r] @ Lm[p, :r]; c[used] = 0.0
        c /= np.sqrt(c[p]); Lm[:, r] = c; dg -= c * c; used[p] = True; order.append(p); r += 1
    return Lm[:, :r].copy(), np.array(order), followed

def onesided_rect(Xin, V0=None, b=32, tol=None, max_sweeps=40):
    m, r = Xin.shape
    if tol is None: tol = np.sqrt(m) * EPS
    rp = -(-r // (2 * b)) * 2 * b
    X = np.zeros((m, rp)); X[:, :r] = Xin; V = np.eye(rp)
    if V0 is not None:
        k = min(V0.shape[0], rp); V[:k, :k] = V0[:k, :k]
        # V0 restricted must stay orthogonal: only valid if k == V0.shape[0] (pad) ; if truncated, re-orthogonalise by QR
        if V0.shape[0] > rp: V[:k, :k] = np.linalg.qr(V0[:k, :k])[0]
        X = X @ V
    nb = rp // b; rounds = rr_pairs(nb); hist = []
    for sweep in range(max_sweeps):
        tot = 0
        for rd in rounds:
            order = np.array([x for pq in rd for x in pq]); perm = (order[:, None] * b + np.arange(b)[None, :]).reshape(-1)
            Xp = X[:, perm].reshape(m, nb // 2, 2 * b); S = np.einsum("rja,rjb->jab", Xp, Xp)
            J, nrot = inner_jacobi_rel(S, 1, tol); tot += nrot
            if nrot == 0: continue
            X[:, perm] = np.einsum("rjc,jcd->rjd", Xp, J).reshape(m, rp)
            Vp = V[:, perm].reshape(rp, nb // 2, 2 * b); V[:, perm] = np.einsum("rjc,jcd->rjd", Vp, J).reshape(rp, rp)
        hist.append(tot)
        if tot == 0: break
    return X, V, hist

def field_err(C, Cref):
    V, Vr = U @ C, U @ Cref; return np.linalg.norm(V - Vr) / np.linalg.norm(Vr)

Vprev = None; oprev = None
for it in range(1, iters):
    A = d[f"lhs{it}"]; R = d[f"rhs{it}"]; A = 0.5 * (A + A.T); Cref = d[f"C{it}"]
    lmax = np.linalg.eigvalsh(A)[-1]; tol = 0.25 * EPS * lmax
    L, order, followed = pchol_hint(A, tol, oprev, THETA); r = L.shape[1]
    same = 0 if oprev is None else (order[:min(len(order), len(oprev))] == oprev[:min(len(order), len(oprev))]).mean()
    t = time.time(); X, V, hist = onesided_rect(L, Vprev)
    sig2 = (X * X).sum(0); kk = sig2 > EPS * sig2.max(); W = X[:, kk] / np.sqrt(sig2[kk]); C = W @ ((W.T @ R) / sig2[kk][:, None])
    print(f"it {it}: r {r} followed-hint {followed} same-position {same:.2f}; sweeps {len(hist)} rot {hist} err {field_err(C, Cref):.2e} ({time.time()-t:.0f}s)", flush=True)
    Vprev, oprev = V, order
