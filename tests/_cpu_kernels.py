"""TEST DOUBLE (never imported by the product): the ``HipKernels`` interface implemented on CPU tensors with the
float64 NumPy oracle, so that the HOST logic of the product - the EM driver, the restart loop, the AnnData wrappers,
the sharding / all-reduce protocol - can be exercised by the ``-m "not gpu"`` suite and by world_size-2 ``gloo`` runs.

It is the checker standing in for the device; it is not a fallback: the product constructs only ``HipKernels``
(``spateo_amd._runtime._make_kernels``) and fails loudly without a GPU.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg
import torch

from oracle import dg_oracle as dgo
from oracle import sparsevfc_oracle as svo

EVAL_V, EVAL_JAC, EVAL_DIV, EVAL_CURL, EVAL_ACC, EVAL_CURV, EVAL_TORS, EVAL_JDET = 1, 2, 4, 8, 16, 32, 64, 128


def _np(t):
    return t.detach().cpu().numpy()


def deflated_minnorm(L, R, cut, block, applications=2):
    """NumPy restatement of mvf_solve_minnorm_lrd's deflated solve (csrc/mvf_minnorm.hip, HISTORY.md 2.2.11) on the pivoted factor
    L (m x r, columns in pivot order):  C = sum over the eigenpairs of L L^T with eigenvalue > cut of q (q^T R) / lambda, WITHOUT
    the eigendecomposition of the whole factor.  Returns (C, number of deflated directions).
      S2 = L^T L = Rc Rc^T (Cholesky: the factor is graded), Minv = S2^-1;
      block inverse iteration started on the unit vectors of the `block` smallest pivots, Cholesky-QR after each application;
      Rayleigh-Ritz on H = Z S2 Z^T; W = Ritz vectors with theta <= cut; Pc = I - W^T W;
      C = L Pc Minv Pc Minv Pc L^T R  (the projections between the inverse applications keep the amplified rounding error of
      the dropped directions out of the result)."""
    r = L.shape[1]
    S2 = L.T @ L
    E = np.linalg.inv(np.linalg.cholesky(S2)).T          # Rc^-T
    Minv = E @ E.T

    def orthonormal_rows(Z):
        return np.linalg.inv(np.linalg.cholesky(Z @ Z.T)) @ Z

    Z = orthonormal_rows(Minv[r - block:])                # = Minv applied to the trailing unit vectors
    for _ in range(applications - 1):
        Z = orthonormal_rows(Z @ Minv)
    H = Z @ S2 @ Z.T
    theta, Yh = np.linalg.eigh((H + H.T) / 2)
    sel = theta <= cut
    W = Yh[:, sel].T @ Z

    def project(T):
        return T - W.T @ (W @ T)

    t = project(L.T @ R)
    t = project(Minv @ t)
    t = project(Minv @ t)
    return L @ t, int(sel.sum())


class CpuKernels:
    DEFL_BLOCK = 256  # mvf_minnorm.hip's DEFL_B; tests shrink it to reach the deflated route at CPU sizes
    def __init__(self, device=None, dtype="float64"):
        self.device = torch.device("cpu")
        self.dtype_name = "float64"
        self.tdtype = torch.float64
        self.gram_events = None

    # ---- helpers
    def to_x4(self, arr, center=None):
        a = np.asarray(arr, dtype=np.float64)
        if center is not None:
            a = a - np.asarray(center, dtype=np.float64)[None, : a.shape[1]]
        buf = np.zeros((a.shape[0], 4))
        buf[:, : a.shape[1]] = a
        return torch.from_numpy(buf)

    def h2d(self, a, tdtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(tdtype) if tdtype is not None and tdtype != t.dtype else t

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.tdtype)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.tdtype)

    # ---- kernels
    def lincomb3(self, out, a, A, b=0.0, B=None, c=0.0, C=None):
        v = a * A
        if B is not None:
            v = v + b * B
        if C is not None:
            v = v + c * C
        out.copy_(v)
        return out

    def con_k(self, x, y, beta, return_d=False, dtype=None):
        xn, yn = _np(x).astype(np.float64), _np(y).astype(np.float64)
        if return_d:
            K, D = svo.con_K(xn, yn, beta, return_d=True)
            return torch.from_numpy(np.atleast_2d(K)), torch.from_numpy(D)
        K = svo.con_K(xn, yn, beta)
        return torch.from_numpy(K.reshape(len(xn), len(yn)))

    def apply(self, x4, ctrl4, beta, C, y4=None, P=None, stats=None):
        X, ctrl = _np(x4)[:, :3], _np(ctrl4)[:, :3]
        n, m = len(X), len(ctrl)
        V = np.zeros((n, 4))
        if m:
            V[:, :3] = svo.con_K(X, ctrl, beta).reshape(n, m) @ _np(C)
        r = None
        if y4 is not None:
            rr = np.sum((_np(y4)[:, :3] - V[:, :3]) ** 2, 1)
            r = torch.from_numpy(rr)
            if P is not None:
                stats[0] += float(_np(P) @ rr)
        return torch.from_numpy(V), r

    def estep_min(self, r, sigma2):
        t1 = np.exp(-_np(r) / (2 * sigma2))
        nz = t1[t1 != 0]
        return torch.tensor([nz.min() if len(nz) else np.inf, float((t1 == 0).sum())], dtype=torch.float64)

    def estep_p(self, r, sigma2, gamma, a, dy, minP, theta, zero_fill, P_out, stats):
        rr = _np(r)
        if torch.is_tensor(zero_fill):
            zero_fill = float(zero_fill[0]) if np.isfinite(float(zero_fill[0])) else 0.0
        t1 = np.exp(-rr / (2 * sigma2))
        nzero = float((t1 == 0).sum())
        t1[t1 == 0] = zero_fill
        t2 = (2 * np.pi * sigma2) ** (dy / 2) * (1 - gamma) / (gamma * a)
        p = t1 / (t1 + t2)
        pf = np.maximum(p, minP)
        P_out.copy_(torch.from_numpy(pf))
        stats += torch.tensor([p @ rr, p.sum(), pf.sum(), float((pf > theta).sum()), nzero], dtype=torch.float64)

    def gram(self, x4, P, y4, ctrl4, beta, G, R, rhs_only=False, tiles_only=False):
        X, ctrl = _np(x4)[:, :3], _np(ctrl4)[:, :3]
        U = svo.con_K(X, ctrl, beta).reshape(len(X), len(ctrl))
        UP = U.T * _np(P)[None, :]
        if not rhs_only:
            G.copy_(torch.from_numpy(UP @ U))
        if not tiles_only:
            R.copy_(torch.from_numpy(UP @ _np(y4)[:, :3]))

    def solve(self, G, K, lambda_sigma2, jitter, R, C_out, info, pivots=None):
        A = _np(G) + lambda_sigma2 * _np(K)
        A = A + jitter * np.trace(A) / len(A) * np.eye(len(A))
        try:
            L = np.linalg.cholesky(A)
            c = scipy.linalg.cho_solve((L, True), _np(R))
            info.zero_()
            C_out.copy_(torch.from_numpy(c))
            if pivots is not None:
                d = np.diag(L) ** 2
                pivots.copy_(torch.tensor([d.min(), d.max()], dtype=torch.float64))
        except np.linalg.LinAlgError:
            info.fill_(1)

    def solve_minnorm(self, G, K, lambda_sigma2, shift, R, C_out, info, einfo, rcond=None, reuse=False, max_sweeps=60,
                      basis=None, warm=False):
        """Truncated minimum-norm solve through a symmetric eigendecomposition (what the device eigensolver computes)."""
        rc = np.finfo(float).eps if rcond is None else rcond
        if not reuse:
            A = _np(G) + lambda_sigma2 * _np(K)
            self._eig = np.linalg.eigh((A + A.T) / 2)
        w, q = self._eig
        keep = np.abs(w) > rc * np.abs(w).max()
        c = (q[:, keep] / w[keep]) @ (q[:, keep].T @ _np(R))
        C_out.copy_(torch.from_numpy(c))
        info.zero_()
        einfo[:6] = torch.tensor([1.0, keep.sum(), np.abs(w).max(), np.abs(w[keep]).min(), shift * np.trace(_np(G)) / len(w),
                                  w.min()], dtype=torch.float64)

    def solve_minnorm_lr(self, G, K, lambda_sigma2, R, C_out, info, einfo, rcond=None, reuse=False, max_sweeps=60,
                         rank_hint=0, tolf=0.25, deflate=False):
        """The device algorithm restated: greedy diagonally pivoted Cholesky stopped at tolf * eps * lambda_max, then the
        SVD of the m x r factor (what the one-sided Jacobi iteration converges to), truncated at rcond * sigma_max^2.
        deflate=True: mvf_solve_minnorm_lrd's route to the same solution (deflated_minnorm below) when the factor has at
        least 2 * DEFL_BLOCK columns, like the device (which uses a block of 256)."""
        rc = np.finfo(float).eps if rcond is None else rcond
        if not reuse:
            A = _np(G) + lambda_sigma2 * _np(K)
            A = (A + A.T) / 2
            m = len(A)
            if not np.isfinite(A).all():
                info.fill_(1)
                return
            lmax = max(float(np.linalg.eigvalsh(A)[-1]), float(np.diag(A).max()), 0.0)
            tol = tolf * np.finfo(float).eps * lmax
            dg = np.diag(A).copy()
            L = np.zeros((m, m))
            used = np.zeros(m, bool)
            r = 0
            order, pvals = [], []
            while r < m:
                dm = np.where(used, -np.inf, dg)
                p = int(np.argmax(dm))
                if not dm[p] > tol:
                    break
                c = A[:, p] - L[:, :r] @ L[p, :r]
                c[used] = 0.0
                if c[p] > 0.25 * tol:
                    c /= np.sqrt(c[p])
                    L[:, r] = c
                    dg -= c * c
                used[p] = True
                order.append(p)
                pvals.append(float(dm[p]))
                r += 1
            self._order = np.array(order, dtype=np.int64)
            self._pivot_values, self._pivot_tol = np.array(pvals), float(tol)
            u, s, _ = np.linalg.svd(L[:, :r], full_matrices=False) if r else (np.zeros((m, 0)), np.zeros(0), None)
            self._lr = (u, s * s, r)
            self._lr_factor = L[:, :r].copy()
        u, lam, r = self._lr
        info.zero_()
        if r == 0:
            C_out.zero_()
            einfo[:7] = 0.0
            return
        keep = lam > rc * lam.max()
        c = None
        if deflate and r >= 2 * self.DEFL_BLOCK:
            c, nsel = deflated_minnorm(self._lr_factor, _np(R), rc * lam.max(), self.DEFL_BLOCK)
            if nsel > self.DEFL_BLOCK - self.DEFL_BLOCK // 8:
                c = None  # block too small for what lies below the cut: the SVD route, like the device's Jacobi path
        if c is None:
            c = (u[:, keep] / lam[keep]) @ (u[:, keep].T @ _np(R))
        C_out.copy_(torch.from_numpy(c))
        o = 6 if reuse else 0
        einfo[o + 1 : o + 6] = torch.tensor([keep.sum(), lam.max(), lam[keep].min(), 0.0, lam.min()], dtype=torch.float64)
        if not reuse:
            einfo[0] = 1.0
            einfo[6] = float(r)

    def lr_pivot_order(self, m, with_values=False):
        if with_values:
            return self._order.copy(), self._pivot_values.copy(), self._pivot_tol
        return self._order.copy()

    def pinv_diag(self, x4, ctrl4, beta, rcond=None, lowrank=False):
        """diag(U pinv(A) U^T) from the decomposition of the last solve_minnorm_lr / solve_minnorm call."""
        rc = np.finfo(float).eps if rcond is None else rcond
        X, ctrl = _np(x4)[:, :3], _np(ctrl4)[:, :3]
        U = svo.con_K(X, ctrl, beta).reshape(len(X), len(ctrl))
        if lowrank:
            q, w, _ = self._lr
        else:
            w, q = self._eig
        keep = np.abs(w) > rc * np.abs(w).max() if len(w) else np.zeros(0, bool)
        Z = U @ q[:, keep]
        return torch.from_numpy((Z * Z / w[keep]).sum(1))

    def sym_pack(self, G, tri):
        g = _np(G)
        tri.copy_(torch.from_numpy(g[np.triu_indices(len(g))]))

    def sym_unpack(self, tri, G):
        m = G.shape[0]
        g = np.zeros((m, m))
        g[np.triu_indices(m)] = _np(tri)
        g = g + np.triu(g, 1).T
        G.copy_(torch.from_numpy(g))

    def quadform(self, K, C, out):
        c = _np(C)
        out[0] = float(np.trace(c.T @ _np(K) @ c))
        return out

    def integrate(self, x4, ctrl4, beta, C, dt, substeps, n_out, affine=None):
        """Same fixed-step RK4 as the device kernel, in NumPy (host-logic double; accuracy is checked on the GPU)."""
        X, ctrl, Cn = _np(x4)[:, :3].astype(np.float64), _np(ctrl4)[:, :3], _np(C)

        def f(q):
            v = svo.con_K(q, ctrl, beta).reshape(len(q), len(ctrl)) @ Cn
            if affine is not None:
                alpha, _, A, b = affine
                v = alpha * v + q @ np.asarray(A).T + np.asarray(b)[None, :]
            return v

        traj = np.empty((len(X), n_out, 3))
        x = X.copy()
        traj[:, 0] = x
        h = dt / substeps
        for t in range(1, n_out):
            for _ in range(substeps):
                k1 = f(x); k2 = f(x + 0.5 * h * k1); k3 = f(x + 0.5 * h * k2); k4 = f(x + h * k3)
                x = x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
            traj[:, t] = x
        return torch.from_numpy(traj)

    def eval(self, x4, ctrl4, beta, C, flags, affine=None):
        X, ctrl, Cn = _np(x4)[:, :3], _np(ctrl4)[:, :3], _np(C)
        vfd = {"X_ctrl": ctrl, "C": Cn, "beta": beta}
        n = len(X)
        out = {}
        v = svo.con_K(X, ctrl, beta).reshape(n, len(ctrl)) @ Cn
        J = dgo.Jacobian_rkhs_gaussian(X, vfd, vectorize=True)
        if affine is not None:
            alpha, jmul, A, b = affine
            v = np.asarray(alpha, dtype=float).reshape(1, -1) * v + X @ np.asarray(A).T + np.asarray(b)[None, :]  # alpha: scalar or (3,)
            J = jmul * J
        a = np.einsum("fin,ni->nf", J, v)
        if flags & EVAL_V:
            out[EVAL_V] = v
        if flags & EVAL_JAC:
            out[EVAL_JAC] = J
        if flags & EVAL_DIV:
            out[EVAL_DIV] = np.trace(J)
        if flags & EVAL_JDET:
            out[EVAL_JDET] = np.linalg.det(np.moveaxis(J, 2, 0))
        if flags & EVAL_CURL:
            out[EVAL_CURL] = np.stack([J[2, 1] - J[1, 2], J[0, 2] - J[2, 0], J[1, 0] - J[0, 1]], axis=1)
        if flags & EVAL_ACC:
            out[EVAL_ACC] = a
        if flags & EVAL_CURV:
            vv = np.sum(v * v, 1)
            va = np.sum(v * a, 1)
            out[EVAL_CURV] = (a * vv[:, None] - v * va[:, None]) / (np.sqrt(vv) ** 4)[:, None]
        if flags & EVAL_TORS:
            Ja = np.einsum("fin,ni->nf", J, a)
            aJa = np.sum(a * Ja, 1)
            den = np.sum(v * v, 1) * np.sum(a * a, 1)
            out[EVAL_TORS] = v * (aJa / den)[:, None]
        return {k: torch.from_numpy(np.ascontiguousarray(val)) for k, val in out.items()}
