"""Deterministic synthetic 3-D point clouds + displacement fields of the BASELINE.json config sizes.

There is no network for datasets; ``bench.py`` and the tests draw their inputs from here:
points uniform in an ellipsoid (or a two-lobed "embryo" = union of two ellipsoids), a smooth analytic displacement
field (rotation about z + radial growth + sinusoidal shear), Gaussian noise and a few gross outliers.
"""
from __future__ import annotations

import numpy as np

__all__ = ["ellipsoid_cloud", "embryo_cloud", "displacement_field", "make_config", "CONFIGS"]


def ellipsoid_cloud(rng, n, axes, center=(0.0, 0.0, 0.0), dtype=np.float64):
    """n points uniform inside an axis-aligned ellipsoid."""
    g = rng.standard_normal((n, 3))
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    r = rng.random(n) ** (1.0 / 3.0)
    return ((g * r[:, None]) * np.asarray(axes) + np.asarray(center)).astype(dtype, copy=False)


def embryo_cloud(rng, n, axes=(2000.0, 1200.0, 900.0), dtype=np.float64):
    """Two-lobed cloud: union of two overlapping ellipsoids (head + trunk)."""
    n1 = int(0.6 * n)
    a = np.asarray(axes)
    trunk = ellipsoid_cloud(rng, n1, a * np.array([1.0, 0.8, 0.8]), center=(-0.35 * a[0], 0.0, 0.0), dtype=dtype)
    head = ellipsoid_cloud(rng, n - n1, a * np.array([0.55, 0.7, 0.75]), center=(0.75 * a[0], 0.15 * a[1], 0.0), dtype=dtype)
    return np.concatenate([trunk, head], axis=0)


def displacement_field(X, scale=1.0):
    """Smooth analytic field: rotation about z (omega = 0.01) + radial growth 0.02 x + sinusoidal shear (amp 2)."""
    L = float(np.abs(X).max()) or 1.0
    V = np.empty_like(X)
    V[:, 0] = -0.01 * X[:, 1] + 0.02 * X[:, 0] + 2.0 * np.sin(2 * np.pi * X[:, 2] / L)
    V[:, 1] = 0.01 * X[:, 0] + 0.02 * X[:, 1] + 2.0 * np.sin(2 * np.pi * X[:, 0] / L)
    V[:, 2] = 0.02 * X[:, 2] + 2.0 * np.cos(2 * np.pi * X[:, 1] / L)
    return V * scale


def _noisy(rng, X, noise, outlier_frac, outlier_sd):
    V = displacement_field(X)
    V /= np.sqrt(np.mean(V**2))  # unit rms: dynamo's outlier model (uniform density 1/a, a = 5) assumes O(1) velocities
    V += noise * rng.standard_normal(X.shape)
    n = len(X)
    k = int(outlier_frac * n)
    if k:
        out = rng.choice(n, size=k, replace=False)
        V[out] = outlier_sd * rng.standard_normal((k, X.shape[1]))
    return V


CONFIGS = {
    # name: (N, M, generator)
    "C2": dict(N=50_000, M=500, axes=(300.0, 200.0, 150.0), embryo=False, seed=2),
    "C3": dict(N=2_000_000, M=2000, axes=(2000.0, 1200.0, 900.0), embryo=True, seed=3),
    "C4": dict(N=8_000_000, M=3000, axes=(2000.0, 1200.0, 900.0), embryo=True, seed=4),
}


def make_config(name, N=None, noise=0.05, outlier_frac=0.05, outlier_sd=2.0, dtype=np.float64, seed=None):
    """Return (X, V, M) for a BASELINE config (optionally with N overridden, same generator).

    The displacement field is scaled to unit rms, noise and outliers are in those units.  (SURVEY.md 8d suggested
    noise 0.5 on an un-normalised field; with dynamo's default ``a = 5`` that makes every cell an "outlier" and the EM
    degenerate, so the generator keeps the geometry and the field shape but normalises the magnitudes.)"""
    cfg = CONFIGS[name]
    N = cfg["N"] if N is None else int(N)
    rng = np.random.default_rng(cfg["seed"] if seed is None else seed)
    if cfg["embryo"]:
        X = embryo_cloud(rng, N, cfg["axes"], dtype=dtype)
    else:
        X = ellipsoid_cloud(rng, N, cfg["axes"], dtype=dtype)
    V = _noisy(rng, X, noise, outlier_frac, outlier_sd)
    return X, V, cfg["M"]
