"""Shared by the CPU and GPU suites: rebuild the BA_transform input dict from tests/golden/ref_align.npz."""
import numpy as np

CASES = (("n1", True, 1), ("n3", True, 3), ("raw", False, 1))


def ba_dict(g, normalize_c):
    return {
        "norm_dict": {k: g[f"ba_nd_{k}"] for k in ("scale_transformed", "mean_fixed", "mean_transformed")},
        "normalize_c": normalize_c,
        "inducing_variables": g["ba_inducing_variables"],
        "beta": float(g["ba_beta"]),
        "Coff": g["ba_Coff"],
        "R": g["ba_R"], "t": g["ba_t"], "optimal_R": g["ba_optimal_R"], "optimal_t": g["ba_optimal_t"],
        "init_R": g["ba_init_R"], "init_t": g["ba_init_t"],
    }


def check_ba(fn, g, rtol, **kw):
    """fn(vecfld, points, deformation_scale=...) -> (XAHat, velocities, optimal similarity) vs the reference."""
    for tag, norm_c, ds in CASES:
        hat, vel, opt = fn(ba_dict(g, norm_c), g[f"ba_{tag}_q"], deformation_scale=ds, **kw)
        for got, key in ((hat, "XAHat"), (vel, "vel"), (opt, "opt")):
            ref = g[f"ba_{tag}_{key}"]
            assert got.shape == ref.shape and got.dtype == np.float64
            assert np.abs(got - ref).max() <= rtol * np.abs(ref).max(), (tag, key)


def check_update_nonrigid(fn, g, dtype, device=None):
    """fn = spateo_amd.align.update_nonrigid against the goldens of the real Morpho_pairwise._update_nonrigid."""
    tol = {"float64": (1e-8, 1e-8, 5e-3), "float32": (2e-3, 1e-3, 2e-2)}[dtype]
    dtol = {"float64": (1e-8, 1e-3), "float32": (2e-3, 5e-2)}[dtype]  # SigmaDiag: well conditioned / rank deficient case
    out = {}
    for tag in ("a", "b"):
        r = fn(g[f"{tag}_coordsA"], g[f"{tag}_inducing_variables"], float(g[f"{tag}_beta"]), g[f"{tag}_K_NA"],
               g[f"{tag}_PXB_term"], float(g[f"{tag}_sigma2"]), float(g[f"{tag}_lambdaVF"]), dtype=dtype, device=device)
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
        assert r["Coff"].shape == g[f"{tag}_Coff"].shape and r["VnA"].shape == g[f"{tag}_VnA"].shape
        e_s = rel(r["SigmaInv"], g[f"{tag}_SigmaInv"])
        e_v = rel(r["VnA"], g[f"{tag}_VnA"])
        e_d = rel(r["SigmaDiag"], g[f"{tag}_SigmaDiag"])
        assert r["SigmaDiag"].shape == g[f"{tag}_SigmaDiag"].shape
        assert e_d < dtol[0 if tag == "a" else 1], (tag, e_d)
        out[tag] = (e_s, e_v, e_d)
        assert e_s < (1e-10 if dtype == "float64" else 1e-5), (tag, e_s)
        if tag == "a":  # well conditioned: the coefficients themselves are determined
            assert rel(r["Coff"], g["a_Coff"]) < tol[0]
            assert e_v < tol[1], e_v
        else:  # rank deficient (99 of 120 directions kept): the field, to the noise level of that system (6e-4)
            assert e_v < tol[2], e_v
    return out


def check_update_nonrigid_branches(fn, g, dtype, device=None):
    """The guidance ("nonrigid") and SVI branches (goldens c / d / e from the real method, morpho_class.py:1269-1294)."""
    tol = {"float64": 1e-8, "float32": 2e-3}[dtype]
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())  # noqa: E731
    out = {}
    for tag in ("c", "d", "e"):
        kw = {}
        if f"{tag}_X_AI" in g:
            kw["guidance"] = dict(X_AI=g[f"{tag}_X_AI"], X_BI=g[f"{tag}_X_BI"], R_AI=g[f"{tag}_R_AI"],
                                  weight=float(g[f"{tag}_guidance_weight"]), Sp=float(g[f"{tag}_Sp"]))
        if f"{tag}_SigmaInv_prev" in g:
            kw["svi"] = dict(step_size=float(g[f"{tag}_step_size"]), SigmaInv_prev=g[f"{tag}_SigmaInv_prev"],
                             PXB_prev=g[f"{tag}_PXB_prev"])
        r = fn(g[f"{tag}_coordsA"], g[f"{tag}_inducing_variables"], float(g[f"{tag}_beta"]), g[f"{tag}_K_NA"],
               g[f"{tag}_PXB_new"], float(g[f"{tag}_sigma2"]), float(g[f"{tag}_lambdaVF"]), dtype=dtype, device=device, **kw)
        errs = {q: rel(r[q], g[f"{tag}_{q}"]) for q in ("SigmaInv", "PXB_term", "Coff", "VnA", "SigmaDiag")}
        if "guidance" in kw:
            errs["V_AI"] = rel(r["V_AI"], g[f"{tag}_V_AI"])
        out[tag] = errs
        assert errs["SigmaInv"] < (1e-10 if dtype == "float64" else 1e-5), (tag, errs)
        assert errs["PXB_term"] < 1e-14, (tag, errs)
        for q in ("Coff", "VnA", "SigmaDiag", "V_AI"):
            if q in errs:
                assert errs[q] < tol, (tag, q, errs)
    return out
