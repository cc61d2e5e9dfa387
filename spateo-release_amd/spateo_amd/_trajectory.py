"""Trajectory integration of a learned field (morphopath / construct_genesis): all RK4 steps of all
trajectories in one launch (`mvf_integrate`), dynamo's arc-length resampling on the host."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from . import _runtime as _rt
from ._runtime import _shared_kernels

# =====================================================================================================================
# trajectory integration (morphopath)
# =====================================================================================================================
def _default_t_end(X, V):
    """dynamo ``getTend``: extent of the data over the 1st percentile of the non-zero |velocity| entries."""
    V_abs = np.abs(np.asarray(V, dtype=float))
    V_abs = V_abs[np.isfinite(V_abs) & (V_abs > 0)]
    return float(np.max(X.max(0) - X.min(0)) / np.percentile(V_abs, 1))


def _hermite(tq, tk, xk, vk):
    """Cubic Hermite interpolation of trajectories: samples xk (n, K, d) with velocities vk at uniform times tk (K,),
    evaluated at per-trajectory times tq (n, Q) -> (n, Q, d).  O(h^4): with the fine RK4 samples this is the ODE's dense
    output to ~1e-8."""
    h = tk[1] - tk[0]
    u = (tq - tk[0]) / h
    i = np.clip(np.floor(u).astype(np.int64), 0, len(tk) - 2)
    w = (u - i)[..., None]
    rows = np.arange(xk.shape[0])[:, None]
    x0, x1, v0, v1 = xk[rows, i], xk[rows, i + 1], vk[rows, i], vk[rows, i + 1]
    h00, h10 = (1 + 2 * w) * (1 - w) ** 2, w * (1 - w) ** 2
    h01, h11 = w * w * (3 - 2 * w), w * w * (w - 1)
    return h00 * x0 + h10 * h * v0 + h01 * x1 + h11 * h * v1


def _arc_length_resample(tk, xk, vk, n_out, stop_tol=1e-5):
    """dynamo ``fate`` semantics on a finely sampled trajectory (``integrate_vf_ivp(..., sampling="arc_length")``): the
    integration ends where every |v| component drops below 1e-5 (its terminal event), the path is cut into n_out points
    EQUALLY SPACED IN ARC LENGTH, the times of those points come from linear interpolation along the polyline, and the
    states are the ODE solution at those times.  Returns (t (n, n_out), x (n, n_out, d))."""
    n, K, d = xk.shape
    slow = np.all(np.abs(vk) < stop_tol, axis=2)
    end = np.where(slow.any(1), slow.argmax(1), K - 1)  # first sample at rest, else the last one
    seg = np.linalg.norm(np.diff(xk, axis=1), axis=2)
    seg[np.arange(K - 1)[None, :] >= end[:, None]] = 0.0  # nothing moves after the terminal event
    s = np.concatenate([np.zeros((n, 1)), np.cumsum(seg, axis=1)], axis=1)
    L = s[:, -1]
    sq = np.linspace(0.0, 1.0, n_out)[None, :] * L[:, None]
    tq = np.empty((n, n_out))
    for r in range(n):  # monotone inverse s -> t, row by row (np.interp is 1-D)
        e = max(int(end[r]), 1)
        tq[r] = np.interp(sq[r], s[r, : e + 1], tk[: e + 1]) if L[r] > 0 else np.linspace(tk[0], tk[e], n_out)
    return tq, _hermite(tq, tk, xk, vk)


def integrate_field(vf_dict, init_states, t_end=None, interpolation_num=250, direction="forward", average=False,
                    nonrigid_only=False, substeps=4, dtype=None, device=None, max_cells_per_launch=1 << 16,
                    sampling="arc_length"):
    """Integrate dx/dt = v(x) from every row of ``init_states`` on the GPU (fused RK4 kernel).

    Returns ``(t, prediction)``: lists with one entry per trajectory, ``t[i]`` (n_t,), ``prediction[i]`` (n_t, d).
    ``sampling="arc_length"`` (dynamo ``fate``'s default, which ``morphopath`` inherits): ``interpolation_num`` points
    equally spaced in arc length along each path (twice as many for ``direction="both"``), every trajectory with its own
    times; the path ends early where the field is at rest (all |v| < 1e-5).  ``"uniform_time"``: ``interpolation_num``
    uniform times over [0, t_end] ("forward"), [-t_end, 0] ("backward") or both.
    ``average``: False | "origin" (one trajectory from the mean start) | "trajectory" / True (mean over cells per sample)."""
    dtype = dtype or _rt._DEFAULT_DTYPE
    X0 = np.asarray(init_states, dtype=np.float64)
    if X0.ndim == 1:
        X0 = X0[None, :]
    if direction not in ("forward", "backward", "both"):
        raise ValueError("direction must be one of 'forward', 'backward', 'both'")
    if sampling not in ("arc_length", "uniform_time"):
        raise ValueError("sampling must be 'arc_length' or 'uniform_time'")
    method = vf_dict.get("method", "sparsevfc")
    if t_end is None:
        t_end = _default_t_end(np.asarray(vf_dict["X"], dtype=float), vf_dict["V"])
    t_end = float(t_end)
    n_t = int(interpolation_num)
    if n_t < 2:
        raise ValueError("interpolation_num must be >= 2")
    if average == "origin":
        X0 = X0.mean(0, keepdims=True)
    d = X0.shape[1]
    k = _shared_kernels(device, dtype)
    if method == "gaussian_process":
        from .vectorfield import _gp_scalars  # (the GP variant's norm_dict rules live beside its evaluator)

        ctrl = np.asarray(vf_dict["inducing_variables"], dtype=np.float64)
        Cc = np.asarray(vf_dict["Coff"], dtype=np.float64)
        dg_ = ctrl.shape[1]
        sf, stt, mean_f, mean_t, _ = _gp_scalars(vf_dict, dg_)   # (d,) each: scalars or per-axis scales alike
        center = ctrl.mean(0)
        pad = lambda v: np.concatenate([np.asarray(v, dtype=float).reshape(-1), np.zeros(3 - dg_)])  # noqa: E731
        stt3 = np.concatenate([stt, np.ones(3 - dg_)])
        A = np.zeros((3, 3))
        if nonrigid_only:
            A[:dg_, :dg_] = np.diag((sf - stt) / 10000.0)
            b = np.zeros(3)
        else:
            R, tt = np.asarray(vf_dict["R"], dtype=float), np.asarray(vf_dict["t"], dtype=float).reshape(-1)
            A[:dg_, :dg_] = (sf[:, None] * R - np.diag(stt)) / 10000.0
            b = pad((sf * tt + mean_f - mean_t) / 10000.0)
        # integrate in normalised coordinates xn = (X - mean_t) / stt:  dxn_f/dt = v_f / stt_f
        affine = (pad(sf / 10000.0) / stt3, 1.0, A / stt3[:, None], (b + A @ pad(center)) / stt3)
        start = (X0 - mean_t) / stt
        to_world = lambda q: q * stt + mean_t  # noqa: E731
        vscale = stt
    else:
        ctrl = np.asarray(vf_dict["X_ctrl"], dtype=np.float64)
        Cc = np.asarray(vf_dict["C"], dtype=np.float64)
        center = ctrl.mean(0)
        affine = None
        start = X0
        to_world = lambda q: q  # noqa: E731
        vscale = 1.0
    if ctrl.shape[1] > 3 or Cc.shape[1] != ctrl.shape[1]:
        raise NotImplementedError("trajectory integration needs a field with Dy == D <= 3")
    C3 = np.zeros((len(ctrl), 3))
    C3[:, : Cc.shape[1]] = Cc
    Cd = k.h2d(C3)
    c4 = k.to_x4(ctrl, center)
    beta = float(vf_dict["beta"])
    arc = sampling == "arc_length"
    n_fine = 4 * n_t + 1 if arc else n_t  # dense RK4 samples the arc-length resampling works from
    dt = t_end / (n_fine - 1)
    tf = np.linspace(0.0, t_end, n_fine)

    def run(sign):
        """(times (n, n_t) or (n_fine,), states (n, n_t, d)) in world coordinates for one direction."""
        ts, xs = [], []
        for lo in range(0, len(start), max_cells_per_launch):
            x4 = k.to_x4(start[lo : lo + max_cells_per_launch], center)
            # arc-length mode samples 4x finer than the output, so `substeps` RK4 steps per OUTPUT interval become
            # max(2, substeps // 2) per fine interval (the default 4 -> 2, i.e. 8 per output interval; more on request)
            tr = k.integrate(x4, c4, beta, Cd, sign * dt, max(2, int(substeps) // 2) if arc else substeps, n_fine,
                             affine=affine)
            if not arc:
                xs.append(_rt._d2h(k, tr)[:, :, :d] + center[None, None, :d])
                continue
            # velocities at the dense samples (fused evaluator; same affine as the integrator), then dynamo's resampling
            pts = tr.reshape(-1, 3)
            p4 = torch.zeros(pts.shape[0], 4, dtype=k.tdtype, device=k.device)
            p4[:, :3] = pts.to(k.tdtype)
            vel = k.eval(p4, c4, beta, Cd, _lib.EVAL_V, affine=affine)[_lib.EVAL_V].reshape(tr.shape[0], n_fine, 3)
            xk = _rt._d2h(k, tr)[:, :, :d]
            tq, xq = _arc_length_resample(sign * tf, xk, sign * _rt._d2h(k, vel)[:, :, :d], n_t,
                                          stop_tol=1e-5 / float(np.max(vscale)))
            ts.append(tq)
            xs.append(xq + center[None, None, :d])
        x = to_world(np.concatenate(xs, axis=0))
        return (np.concatenate(ts, axis=0) if arc else sign * tf), x

    if direction == "forward":
        times, traj = run(+1.0)
    elif direction == "backward":
        times, traj = run(-1.0)
    else:
        (tb, back), (tfw, fwd) = run(-1.0), run(+1.0)
        if arc:  # dynamo doubles interpolation_num for "both": the backward half reversed, then the forward half
            traj = np.concatenate([back[:, ::-1], fwd], axis=1)
            times = np.concatenate([tb[:, ::-1], tfw], axis=1)
        else:
            traj = np.concatenate([back[:, :0:-1], fwd], axis=1)
            times = np.concatenate([tb[:0:-1], tfw])
    if average in ("trajectory", True):
        traj = traj.mean(0, keepdims=True)
        if arc:
            times = times.mean(0, keepdims=True)
    if arc:
        return [times[i].copy() for i in range(len(traj))], [traj[i] for i in range(len(traj))]
    return [times.copy() for _ in range(len(traj))], [traj[i] for i in range(len(traj))]


def genesis_states(vf_dict, init_states, time_vec, substeps=64, dtype=None, device=None):
    """The numeric core of ``construct_genesis`` (``spateo/tdr/models/models_migration/morphopath_model.py:138-148``):
    starting from ``init_states`` the cells are displaced step by step, ``pts <- odeint(f, pts, [0, time_vec[i]])[1]``
    for every entry of ``time_vec`` (each entry is the DURATION of that step, as in the reference's loop), and the
    positions after every step are returned as a list of (n, d) arrays (the reference's ``stages_X``).  One fused RK4
    launch per step (``substeps`` RK4 steps each) instead of one SciPy ``odeint`` call per cell and step."""
    dtype = dtype or _rt._DEFAULT_DTYPE
    pts = np.asarray(init_states, dtype=np.float64)
    ctrl = np.asarray(vf_dict["X_ctrl"], dtype=np.float64)
    Cc = np.asarray(vf_dict["C"], dtype=np.float64)
    d = ctrl.shape[1]
    if pts.ndim != 2 or pts.shape[1] != d or d > 3 or Cc.shape[1] != d:
        raise NotImplementedError("genesis_states needs (n, d) states and a field with Dy == D <= 3")
    k = _shared_kernels(device, dtype)
    center = ctrl.mean(0)
    C3 = np.zeros((len(ctrl), 3))
    C3[:, :d] = Cc
    Cd = k.h2d(C3)
    c4 = k.to_x4(ctrl, center)
    beta = float(vf_dict["beta"])
    stages = []
    for dt in np.asarray(time_vec, dtype=np.float64):
        if dt != 0.0:
            tr = k.integrate(k.to_x4(pts, center), c4, beta, Cd, float(dt), int(substeps), 2)
            pts = _rt._d2h(k, tr[:, 1, :d]) + center[None, :d]
        stages.append(pts.copy())
    return stages


# =====================================================================================================================
# batched independent fits (BASELINE config 5: 32 organs x ~250 k cells, M = 500) - replicas only, no collective
# =====================================================================================================================
