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
