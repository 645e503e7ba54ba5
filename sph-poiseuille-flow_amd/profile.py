"""Velocity-profile monitors and the L2 metric.

compute_binned_profile_mean / compute_mid_channel_profile: SPH_Poiseuille.m:579-605.
final_profile: SPH_Poiseuille.m:617-623.  l2_error: SPH_Poiseuille_postprocess.m:37-42.
"""
from __future__ import annotations

import numpy as np


def compute_binned_profile_mean(y_values, u_values, y_min, y_max, n_bins):
    y_values = np.asarray(y_values, dtype=np.float64).ravel()
    u_values = np.asarray(u_values, dtype=np.float64).ravel()
    edges = np.linspace(y_min, y_max, n_bins + 1)
    y_mid = 0.5 * (edges[:-1] + edges[1:])
    # discretize(): bin k holds edges[k] <= y < edges[k+1]; the last bin also holds y == edges[end]
    inside = (y_values >= edges[0]) & (y_values <= edges[-1])
    bin_id = np.searchsorted(edges, y_values[inside], side="right") - 1
    bin_id = np.minimum(bin_id, n_bins - 1)
    sum_u = np.bincount(bin_id, weights=u_values[inside], minlength=n_bins).astype(np.float64)
    cnt_u = np.bincount(bin_id, minlength=n_bins).astype(np.float64)
    u_mean = sum_u / np.maximum(cnt_u, 1.0)
    u_mean[cnt_u == 0] = np.nan
    return y_mid, u_mean


def compute_mid_channel_profile(pos, u_x, DL, DH, mid_x, half_width, n_bins):
    x_wrap = np.mod(pos[:, 0], DL)
    dx_mid = np.abs(x_wrap - mid_x)
    dx_mid = np.minimum(dx_mid, DL - dx_mid)
    is_mid = dx_mid <= half_width
    if not np.any(is_mid):
        return compute_binned_profile_mean([], [], 0.0, DH, n_bins)
    return compute_binned_profile_mean(pos[is_mid, 1], u_x[is_mid], 0.0, DH, n_bins)


def n_profile_bins(DH, dp):
    """SPH_Poiseuille.m:234."""
    return max(20, int(np.floor(DH / dp + 0.5)))


def final_profile(pos_fluid, vel_x_fluid, prm):
    n_bins = n_profile_bins(prm.DH, prm.dp)
    y_mid, u_mean = compute_binned_profile_mean(pos_fluid[:, 1], vel_x_fluid, 0.0, prm.DH, n_bins)
    u_exact = prm.gravity_g / (2.0 * prm.nu) * y_mid * (prm.DH - y_mid)
    return y_mid, u_mean, u_exact


def l2_error(u_mean, u_exact):
    valid = ~np.isnan(u_mean)
    if not np.any(valid):
        raise ValueError("velocity profile bins are all empty")
    return float(np.sqrt(np.sum((u_mean[valid] - u_exact[valid]) ** 2)
                         / max(np.sum(u_exact[valid] ** 2), np.finfo(np.float64).eps)))
