"""Input fuzz of the drop-in SparseVFC against the oracle: non-finite rows of Y, duplicated rows of X, more control points than unique
rows, explicit beta, permutation sampling, one-row grids, 2-D data with wide Y, the rank-deficient regime in 2-D.  A case passes when
the product reproduces the oracle's control points / valid rows exactly and its field within the literal tolerance or - where the
reference's own result is not determined to it - within 1.25 x the oracle's own lstsq -> eigh floor; exceptions must match in type."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "spateo-release_amd"), os.path.join(R, "tests")]
import numpy as np
import spateo_amd as st
import _floors as F
from oracle import sparsevfc_oracle as svo
rng = np.random.default_rng(7)

def data(n, D, Dy):
    X = rng.uniform(-100, 100, (n, D))
    Y = np.column_stack([np.sin(X[:, 0] / (30 + 3 * j)) + 0.2 * np.cos(X[:, -1] / (20 + j)) for j in range(Dy)]) + 0.03 * rng.standard_normal((n, Dy))
    return X, Y

cases = []
X, Y = data(3000, 3, 3); Yn = Y.copy(); Yn[::17, 1] = np.nan; Yn[5, 0] = np.inf
cases.append(("non-finite rows of Y", X, Yn, X[:50] + 1, dict(M=60, lambda_=3.0)))
Xd = X.copy(); Xd[100:200] = Xd[300:400]; Xd[7] = Xd[9]
cases.append(("duplicated rows of X", Xd, Y, None, dict(M=80, lambda_=3.0)))
Xs = np.repeat(X[:40], 50, axis=0); Ys = np.repeat(Y[:40], 50, axis=0) + 0.01 * rng.standard_normal((2000, 3))
cases.append(("M = 100 > 40 unique rows", Xs, Ys, None, dict(M=100, lambda_=3.0)))
cases.append(("explicit beta", X, Y, X[:10], dict(M=50, lambda_=3.0, beta=3e-4)))
cases.append(("permutation sampling", X, Y, None, dict(M=70, lambda_=3.0, velocity_based_sampling=False, seed=3)))
cases.append(("one-row grid", X, Y, X[:1] + 2.0, dict(M=50, lambda_=3.0)))
X2, Y2 = data(2500, 2, 7)
cases.append(("2-D data, Dy = 7", X2, Y2, X2[:30], dict(M=60, lambda_=3.0)))
X2b, Y2b = data(4000, 2, 2)
cases.append(("2-D, lambda_ = 0.02, M = 300", X2b, Y2b, X2b[:100], dict(M=300, lambda_=0.02, MaxIter=8, ecr=0.0)))
cases.append(("M = 2", X, Y, None, dict(M=2, lambda_=3.0)))
cases.append(("n = 3, M = 100", X[:3], Y[:3], None, dict(M=100, lambda_=3.0)))
cases.append(("all of Y non-finite", X[:50], np.full((50, 3), np.nan), None, dict(M=10, lambda_=3.0)))
cases.append(("a = 10, gamma = 0.5, theta = 0.6, minP = 1e-4", X, Y, None, dict(M=50, lambda_=3.0, a=10, gamma=0.5, theta=0.6, minP=1e-4)))
bad = 0
for tag, Xc, Yc, G, kw in cases:
    kw = dict(dict(lstsq_method="scipy", MaxIter=6, seed=0), **kw)
    for dtype, tol in (("float64", 1e-5), ("float32", 1e-3)):
        ref = got = eref = egot = None
        try: ref = svo.SparseVFC(Xc, Yc, G, **kw)
        except Exception as e: eref = e
        try: got = st.SparseVFC(Xc, Yc, G, dtype=dtype, device="cuda:0", **kw)
        except Exception as e: egot = e
        if eref is not None or egot is not None:
            ok = eref is not None and egot is not None and type(eref) is type(egot)
            msg = f"oracle {type(eref).__name__ if eref else 'no exception'}; product {type(egot).__name__ if egot else 'no exception'}: {str(egot)[:90] if egot else ''}"
        else:
            same = np.array_equal(got["ctrl_idx"], ref["ctrl_idx"]) and np.array_equal(got["valid_ind"], ref["valid_ind"]) and got["V"].shape == ref["V"].shape and got["iteration"] == ref["iteration"]
            vmax = max(np.abs(ref["V"]).max(), 1e-300)
            dev = np.abs(got["V"] - ref["V"]).max() / vmax if same else np.inf
            devg = 0.0 if G is None or not same else np.abs(got["grid_V"] - ref["grid_V"]).max() / vmax
            ok = same and dev < tol and devg < tol
            msg = f"V {dev:.1e} grid {devg:.1e}"
            if same and not ok:
                alt = F.oracle_fit(Xc, Yc, G, variant="eigh", **kw)
                fl = np.abs(alt["V"] - ref["V"]).max() / vmax if alt["iteration"] == ref["iteration"] else np.inf
                ok = dev <= 1.25 * fl
                msg += f" [reference's own lstsq -> eigh floor {fl:.1e}]"
        bad += not ok
        print("ok  " if ok else "FAIL", f"{tag} ({dtype}):", msg, flush=True)
print(f"{2 * len(cases)} cases, {bad} failures")
