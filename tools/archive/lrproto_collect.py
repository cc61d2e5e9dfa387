"""CPU prototype helper (design probe for mvf_solve_minnorm_lr, not product): run the oracle EM loop and save the
linear systems (lhs, rhs, reference coefficients) of its iterations.  python tools/lrproto_collect.py N M lambda iters CFG
-> /tmp/proto/sys_CFG_N_M_lambda.npz"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
from oracle import sparsevfc_oracle as svo
from spateo_amd._synthetic import make_config
N, M, lam, iters, cfg = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
X, Y, _ = make_config(cfg, N=N)
valid, Xv, Yv, idx, ctrl, beta = svo.sparsevfc_setup(X, Y, M=M, seed=0)
K = svo.con_K(ctrl, ctrl, beta); U = svo.con_K(Xv, ctrl, beta)
Nn, D = Yv.shape
Vc, C = np.zeros((Nn, D)), np.zeros((M, D))
s2, gamma, E = np.sum(Yv**2) / (Nn * D), 0.9, 1
out = {}
for it in range(iters):
    P, _ = svo.get_P(Yv, Vc, s2, gamma, 5); P = np.maximum(P, 1e-5)
    UP = U.T * P.T
    out[f"lhs{it}"] = UP @ U + lam * s2 * K; out[f"rhs{it}"] = UP @ Yv
    P, E, tecr, C, Vc, s2, gamma = svo.em_step(U, K, Yv, Vc, C, s2, gamma, E, a=5, lambda_=lam, minP=1e-5, theta=0.75, lstsq_method="scipy")
    out[f"C{it}"] = C
    print(it, s2, flush=True)
np.savez(f"/tmp/proto/sys_{cfg}_{N}_{M}_{lam}.npz", U=U[:4000], ctrl=ctrl, beta=beta, iters=iters, **out)
