#!/usr/bin/env python
"""CPU experiment (not a test): is an exact-integer ("Ozaki") Gram numerically admissible for this EM?

The integer scheme computes the EXACT Gram matrix of operands rounded to a fixed-point grid,
    A~ = round(sqrt(P) U * 2^b) / 2^b      (b = 7 bits x number of int8 slices),   G~ = A~^T A~ ,
i.e. a STRUCTURED perturbation of the operand, not noise added to G.  HISTORY.md 2.3 found structured perturbations benign
and unstructured ones (float32 accumulation: 1e-9 relative noise in G) fatal (1e-2 in the field).  This script runs the
float64 oracle's EM with G~ (and either the consistent rhs A~^T sqrt(P) Y~ or the reference's U^T P Y) and reports the
field deviation from the unmodified oracle next to the oracle's own noise floors.

    python tests/experiments/ozaki_numerics.py [N] [M] [lambda] [steps]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), os.path.dirname(HERE)):
    sys.path.insert(0, p)

import _floors as F  # noqa: E402
from oracle import sparsevfc_oracle as svo  # noqa: E402
from spateo_amd._synthetic import make_config  # noqa: E402


def run(U, K, Y, steps, lam, gram=None, f32_values=False):
    N, D = Y.shape
    M = U.shape[1]
    V, C = np.zeros((N, D)), np.zeros((M, D))
    s2, gamma, E = np.sum(Y**2) / (N * D), 0.9, 1
    for _ in range(steps):
        P, E = svo.get_P(Y, V, s2, gamma, 5)
        E = E + lam / 2 * np.trace(C.T @ K @ C)
        P = np.maximum(P, 1e-5)
        if gram is None:
            UP = U.T * P.T
            lhs, rhs = UP @ U + lam * s2 * K, UP @ Y
        else:
            G, R = gram(U, P[:, 0], Y)
            lhs, rhs = G + lam * s2 * K, R
        C = svo.lstsq_solver(lhs, rhs, "scipy")
        V = U @ C
        s2 = float(P[:, 0] @ np.sum((Y - V) ** 2, 1) / (P.sum() * D))
        g = np.count_nonzero(P > 0.75) / N
        gamma = min(0.95, max(0.05, g))
    return V, s2


def make_quantised(bits, consistent_rhs):
    q = float(2**bits)

    def gram(U, P, Y):
        A = np.sqrt(P)[:, None] * U
        Aq = np.round(A * q) / q
        G = Aq.T @ Aq
        if consistent_rhs:
            Yq = np.sqrt(P)[:, None] * Y
            sc = 2.0 ** np.ceil(np.log2(np.abs(Yq).max()))
            Yq = np.round(Yq / sc * q) / q * sc
            R = Aq.T @ Yq
        else:
            R = (U.T * P) @ Y
        return G, R

    return gram


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    lam = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    X, Vel, _ = make_config("C3", N=N)
    valid, Xv, Yv, idx, ctrl, beta = svo.sparsevfc_setup(X, Vel, M=M, seed=0)
    K = svo.con_K(ctrl, ctrl, beta)
    U = svo.con_K(Xv, ctrl, beta)
    ref, s2r = run(U, K, Yv, steps, lam)
    vmax = np.abs(ref).max()
    rel = lambda a: float(np.abs(a - ref).max() / vmax)  # noqa: E731
    orig = svo.lstsq_solver
    svo.lstsq_solver = F.eigh_solver
    try:
        alt, _ = run(U, K, Yv, steps, lam)
    finally:
        svo.lstsq_solver = orig
    print(f"N={N} M={M} lambda={lam} steps={steps}: reference floor (lstsq -> eigh) {rel(alt):.2e}")
    so, _ = run(U, K, Yv, steps, lam, gram=lambda U_, P_, Y_: (F.chunked_dot(U_.T * P_, U_), F.chunked_dot(U_.T * P_, Y_)))
    print(f"  reference floor (Gram summed in 7 chunks)            {rel(so):.2e}")
    for bits in (21, 28, 35, 42):
        for cons in (True, False):
            v, s2 = run(U, K, Yv, steps, lam, gram=make_quantised(bits, cons))
            print(f"  exact Gram of operands on a 2^-{bits} grid, rhs {'consistent' if cons else 'float64 U^T P Y'}: "
                  f"V dev {rel(v):.2e}, sigma2 rel {abs(s2 - s2r) / s2r:.2e}")
    # for contrast: unstructured noise of the same size added to G
    rng = np.random.default_rng(0)
    for eps in (2.0**-28, 1e-12):
        def noisy(U_, P_, Y_, eps=eps):
            UP = U_.T * P_
            G = UP @ U_
            Z = rng.standard_normal(G.shape)
            return G + eps * np.abs(G).max() * (Z + Z.T) / 2, UP @ Y_
        v, _ = run(U, K, Yv, steps, lam, gram=noisy)
        print(f"  unstructured symmetric noise {eps:.1e} x max|G| added to G: V dev {rel(v):.2e}")


if __name__ == "__main__":
    main()
