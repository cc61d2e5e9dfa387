"""CPU ORACLE (test infrastructure, NOT product code): the SparseVFC EM of ``sparsevfc_oracle.py`` with the N x M kernel
matrix U streamed over cell chunks, so that the oracle reaches the sizes the benchmark and the 8-GPU split run at
(2 M x 2000, 1 M x 3000, 8 M x 3000) on a host that cannot hold U (192 GB at 8 M x 3000) or its M x N temporary.

Every arithmetic statement is the one of ``sparsevfc_oracle.em_step`` / ``SparseVFC`` (reference
call sites ``/root/reference/spateo/tdr/morphometrics/morphofield/sparsevfc.py:167,189-198,234``), applied chunk by chunk:

* ``U_c = exp(-beta * cdist(X_c, ctrl, "sqeuclidean"))``  - element-wise, so chunking does not change a single bit of U;
* ``UP_c = U_c.T * repmat(P_c.T, M, 1)``; ``lhs = sum_c UP_c.dot(U_c) + lambda sigma2 K``; ``rhs = sum_c UP_c.dot(Y_c)``
  - the products summed over the cells in ``chunks`` sequential pieces: exactly the ``sumorder`` form of
  ``tests/_floors.py`` (what a different BLAS thread count does to the reference's one ``dot`` call);
* ``C = scipy.linalg.lstsq(lhs, rhs)[0]`` through ``sparsevfc_oracle.lstsq_solver`` (or a swapped-in driver: the floors);
* ``V_c = U_c.dot(C)``; sigma2, gamma, energy, tecr and the stopping rule as in ``em_step``.

U is generated twice per EM iteration (Gram pass, V pass) and never stored.  Threads only fill disjoint row blocks of
U_c (element-wise work); every sum over cells is sequential and therefore reproducible.

PARITY STATUS: the same as ``sparsevfc_oracle.py`` - the EM loop restates dynamo 1.4.x and is **parity unpinned**
(dynamo is absent from /root/reference and not installable); ``tests/test_oracle.py`` pins THIS file to
``sparsevfc_oracle.SparseVFC`` (bit-identical P / E of the first step, identical fields to rounding) at sizes both run.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from scipy.spatial.distance import cdist

from . import sparsevfc_oracle as svo

__all__ = ["streamed_con_K", "StreamedEM", "SparseVFC_streamed"]

_BLOCK = 2048


def streamed_con_K(x, ctrl, beta, out=None, threads=None):
    """``svo.con_K(x, ctrl, beta)`` (cdist path) written block-wise into ``out`` by a thread pool (same values)."""
    n = len(x)
    out = np.empty((n, len(ctrl))) if out is None else out[:n]
    threads = threads or min(8, os.cpu_count() or 1)

    def fill(lo):
        d = cdist(x[lo : lo + _BLOCK], ctrl, "sqeuclidean")
        np.multiply(d, -beta, out=d)  # K = -beta * K   (gaussian_process.py:30)
        np.exp(d, out=out[lo : lo + _BLOCK])  # K = np.exp(K)  (gaussian_process.py:31)

    if threads == 1:
        for lo in range(0, n, _BLOCK):
            fill(lo)
    else:
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(fill, range(0, n, _BLOCK)))
    return out


class StreamedEM:
    """State of one streamed fit: X, Y (N x D), ctrl (M x D), beta; ``chunks`` = number of sequential pieces the sums
    over cells are made of (piece boundaries by ``np.linspace(0, N, chunks + 1)`` as in ``tests/_floors.chunked_dot``)."""

    def __init__(self, X, Y, ctrl, beta, chunks, solver=None, gamma=0.9, progress=None):
        self.X, self.Y, self.ctrl, self.beta = X, Y, ctrl, float(beta)
        self.N, self.D = Y.shape
        self.M = len(ctrl)
        self.edges = np.linspace(0, self.N, int(chunks) + 1).astype(int)
        self.solver = solver or (lambda lhs, rhs: svo.lstsq_solver(lhs, rhs, method="scipy"))
        self.K = svo.con_K(ctrl, ctrl, self.beta)
        self.V = np.zeros((self.N, self.D))
        self.C = np.zeros((self.M, self.D))
        self.E, self.tecr, self.gamma = 1, 1, gamma
        s2 = np.sum((Y - self.V) ** 2) / (self.N * self.D)
        self.sigma2 = 1e-7 if s2 < 1e-8 else s2
        self.P = None
        self._buf = np.empty((int(np.diff(self.edges).max()), self.M))
        self.progress = progress or (lambda *_: None)
        self.lhs = self.rhs = None

    def _chunks(self):
        for lo, hi in zip(self.edges[:-1], self.edges[1:]):
            if hi > lo:
                yield lo, hi, streamed_con_K(self.X[lo:hi], self.ctrl, self.beta, out=self._buf)

    def assemble(self, P, sigma2, lambda_):
        """lhs, rhs of the M-step for the floored P (N x 1)."""
        lhs = rhs = None
        for lo, hi, U in self._chunks():
            UP = U.T * np.tile(P[lo:hi].T, (self.M, 1))
            g, r = UP.dot(U), UP.dot(self.Y[lo:hi])
            lhs, rhs = (g, r) if lhs is None else (lhs + g, rhs + r)
            self.progress("gram", hi, self.N)
        return lhs + lambda_ * sigma2 * self.K, rhs

    def apply(self, C):
        V = np.empty((self.N, self.D))
        for lo, hi, U in self._chunks():
            V[lo:hi] = U.dot(C)
            self.progress("apply", hi, self.N)
        return V

    def step(self, a=5, lambda_=3, minP=1e-5, theta=0.75, keep_system=False):
        """One EM iteration = the body of ``svo.em_step``; returns (E, tecr)."""
        Y = self.Y
        E_old = self.E
        P, E = svo.get_P(Y, self.V, self.sigma2, self.gamma, a)
        E = E + lambda_ / 2 * np.trace(self.C.T.dot(self.K).dot(self.C))
        tecr = abs((E - E_old) / E)
        P = np.maximum(P, minP)
        lhs, rhs = self.assemble(P, self.sigma2, lambda_)
        if keep_system:
            self.lhs, self.rhs = lhs, rhs
        C = self.solver(lhs, rhs)
        V = self.apply(C)
        Sp = np.sum(P)
        sigma2 = float(P[:, 0].dot(np.sum((Y - V) ** 2, 1)) / (Sp * self.D))
        numcorr = len(np.where(P > theta)[0])
        gamma = numcorr / self.N
        gamma = 0.95 if gamma > 0.95 else (0.05 if gamma < 0.05 else gamma)
        self.P, self.E, self.tecr, self.C, self.V, self.sigma2, self.gamma = P, E, tecr, C, V, sigma2, gamma
        return E, tecr


def SparseVFC_streamed(X, Y, M=100, a=5, beta=None, ecr=1e-5, gamma=0.9, lambda_=3, minP=1e-5, MaxIter=500, theta=0.75,
                       seed=0, chunks=16, solver=None, progress=None, setup=None):
    """``svo.SparseVFC(X, Y, None, ..., lstsq_method="scipy")`` with U streamed; same dict (no grid keys)."""
    X = np.asarray(X, dtype=float)
    Y = np.asarray(Y, dtype=float)
    valid_ind, Xv, Yv, idx, ctrl, beta = setup or svo.sparsevfc_setup(X, Y, M=M, beta=beta, seed=seed)
    em = StreamedEM(Xv, Yv, ctrl, beta, chunks, solver=solver, gamma=gamma, progress=progress)
    i = 0
    tecr_vec = np.ones(MaxIter) * np.nan
    E_vec = np.ones(MaxIter) * np.nan
    s2_vec = np.ones(MaxIter) * np.nan
    while i < MaxIter and em.tecr > ecr and em.sigma2 > 1e-8:
        E, tecr = em.step(a=a, lambda_=lambda_, minP=minP, theta=theta)
        E_vec[i], tecr_vec[i], s2_vec[i] = E, tecr, em.sigma2
        i += 1
    return {"X": X, "valid_ind": valid_ind, "X_ctrl": ctrl, "ctrl_idx": idx, "Y": Y, "beta": beta, "V": em.V, "C": em.C,
            "P": em.P, "VFCIndex": np.where(em.P > theta)[0], "sigma2": em.sigma2, "grid": None, "grid_V": None,
            "iteration": i - 1, "tecr_traj": tecr_vec[:i], "E_traj": E_vec[:i], "sigma2_traj": s2_vec[:i]}
