"""Initial particle layout -- mirror of SPH_Poiseuille.m S3 and build_shell_wall_particles.m.

Fluid lattice [0,DL]x[0,DH] with y fastest (meshgrid + (:) ordering, SPH_Poiseuille.m:95-98), then
the bottom wall block, then the top wall block, each x-column-major with y fastest
(build_shell_wall_particles.m:22-31).  Arrays are column-major [n x 2] like MATLAB: numpy
order='F' so that pos[:,0] is the contiguous x column.
"""
from __future__ import annotations

import numpy as np


def _colon(a: float, step: float, b: float) -> np.ndarray:
    """MATLAB a:step:b for positive step: n = floor((b-a)/step + eps-tolerance) + 1 elements a + k*step."""
    n = int(np.floor((b - a) / step * (1.0 + 1e-12) + 1e-10)) + 1
    if n <= 0:
        return np.zeros(0)
    return a + step * np.arange(n, dtype=np.float64)


def build_shell_wall_particles(DL: float, DH: float, dp: float, wall_thickness: float):
    """build_shell_wall_particles.m:1-39 -> (pos_wall, wall_normal, wall_measure, wall_thickness_arr)."""
    if DL <= 0 or DH <= 0 or dp <= 0 or wall_thickness <= 0:
        raise ValueError("DL, DH, dp and wall_thickness must be positive")
    n_layers = int(round(wall_thickness / dp))
    if abs(n_layers * dp - wall_thickness) > 1e-12:
        raise ValueError("wall_thickness must be a whole multiple of dp")
    if n_layers < 1:
        raise ValueError("wall_thickness needs at least one particle layer")
    x_wall = _colon(dp / 2, dp, DL - dp / 2)
    y_bottom = _colon(-wall_thickness + dp / 2, dp, -dp / 2)
    y_top = _colon(DH + dp / 2, dp, DH + wall_thickness - dp / 2)

    def block(xs, ys):  # meshgrid(x,y) then (:) => y fastest
        X = np.repeat(xs, len(ys))
        Y = np.tile(ys, len(xs))
        return np.column_stack([X, Y])

    pos_bottom = block(x_wall, y_bottom)
    pos_top = block(x_wall, y_top)
    pos_wall = np.vstack([pos_bottom, pos_top])
    nb, ntp = len(pos_bottom), len(pos_top)
    wall_normal = np.vstack([np.tile([0.0, -1.0], (nb, 1)), np.tile([0.0, 1.0], (ntp, 1))])
    wall_measure = dp * np.ones(nb + ntp)
    wall_thickness_arr = dp * np.ones(nb + ntp)
    return pos_wall, wall_normal, wall_measure, wall_thickness_arr


def init_particles(prm):
    """SPH_Poiseuille.m:93-125.  Returns a dict of column-major float64 arrays and counts."""
    dp = prm.dp
    x_fluid = _colon(dp / 2, dp, prm.DL - dp / 2)
    y_fluid = _colon(dp / 2, dp, prm.DH - dp / 2)
    pos_fluid = np.column_stack([np.repeat(x_fluid, len(y_fluid)), np.tile(y_fluid, len(x_fluid))])
    n_fluid = pos_fluid.shape[0]
    pos_wall, wall_normal, wall_measure, wall_thickness_arr = build_shell_wall_particles(
        prm.DL, prm.DH, dp, prm.wall_thickness)
    n_wall = pos_wall.shape[0]
    n_total = n_fluid + n_wall
    F = lambda a: np.asfortranarray(a, dtype=np.float64)
    mass = np.concatenate([prm.rho0 * dp ** 2 * np.ones(n_fluid), prm.rho0 * wall_measure * wall_thickness_arr])
    B = np.zeros((n_total, 4), order="F")
    B[:, 0] = 1.0
    B[:, 3] = 1.0
    rho = prm.rho0 * np.ones(n_total)
    return dict(
        n_fluid=n_fluid, n_wall=n_wall, n_total=n_total,
        pos=F(np.vstack([pos_fluid, pos_wall])), vel=F(np.zeros((n_total, 2))),
        wall_vel=F(np.zeros((n_total, 2))), rho=rho, p=np.zeros(n_total), drho_dt=np.zeros(n_total),
        force=F(np.zeros((n_total, 2))), force_prior=F(np.zeros((n_total, 2))), mass=mass,
        Vol=mass / rho, B=B, wall_normal=wall_normal, wall_measure=wall_measure,
        wall_thickness_arr=wall_thickness_arr)


def developed_state(prm, parts, jitter=0.05, seed=12345):
    """Synthetic 'developed flow' start for throughput runs (labelled as such by callers): analytic
    parabola u_x = g/(2 nu) y (DH - y) on the fluid and a uniform +-jitter*dp position perturbation."""
    rng = np.random.default_rng(seed)
    nf = parts["n_fluid"]
    pos = parts["pos"].copy(order="F")
    vel = parts["vel"].copy(order="F")
    pos[:nf, :] += (rng.random((nf, 2)) * 2.0 - 1.0) * jitter * prm.dp
    pos[:nf, 0] = pos[:nf, 0] - np.floor(pos[:nf, 0] / prm.DL) * prm.DL
    y = pos[:nf, 1]
    vel[:nf, 0] = prm.gravity_g / (2.0 * prm.nu) * y * (prm.DH - y)
    return pos, vel
