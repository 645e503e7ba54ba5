"""Shared test helpers: seeded inputs and tolerance checks."""
import numpy as np


def make_case(cfgmod, geom, dp=0.05, DL=3.0, seed=1234, jitter=0.2, developed=True, **kw):
    """Lattice + uniform jitter (+- jitter*dp) + (optionally) a developed-flow-like state:
    parabolic u_x, small random u_y, non-zero drho_dt.  Deterministic for a given seed."""
    prm = cfgmod.params_from_values(dp=dp, DL=DL, **kw)
    parts = geom.init_particles(prm)
    nf, nt = parts["n_fluid"], parts["n_total"]
    rng = np.random.default_rng(seed)
    pos = parts["pos"].copy(order="F")
    vel = parts["vel"].copy(order="F")
    drho = parts["drho_dt"].copy()
    if jitter:
        pos[:nf] += (rng.random((nf, 2)) * 2 - 1) * jitter * prm.dp
        pos[:nf, 0] = pos[:nf, 0] - np.floor(pos[:nf, 0] / prm.DL) * prm.DL
    if developed:
        y = pos[:nf, 1]
        vel[:nf, 0] = prm.gravity_g / (2 * prm.nu) * y * (prm.DH - y) * (1 + 0.02 * rng.standard_normal(nf))
        vel[:nf, 1] = 0.02 * rng.standard_normal(nf)
        drho[:nf] = 0.5 * rng.standard_normal(nf)
    parts = dict(parts)
    parts.update(pos=pos, vel=vel, drho_dt=drho)
    return prm, parts


def canon_pairs(nb):
    """Sort a pair list by (i, j) so two lists can be compared as sets."""
    pi, pj = nb[0].astype(np.int64), nb[1].astype(np.int64)
    order = np.lexsort((pj, pi))
    return tuple(np.asarray(c)[order] for c in nb)


def assert_close(a, b, rtol=1e-11, atol_scale=1e-13, name=""):
    """|a-b| <= rtol*|b| + atol_scale*max|b| element-wise (differences are summation order only)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{name}: shape {a.shape} vs {b.shape}"
    scale = np.max(np.abs(b)) if b.size else 0.0
    err = np.abs(a - b)
    tol = rtol * np.abs(b) + atol_scale * max(scale, 1e-300) + 1e-300
    bad = err > tol
    assert not np.any(bad), (f"{name}: {int(bad.sum())}/{a.size} elements differ, max err {err.max():.3e} "
                             f"(scale {scale:.3e}) at {np.argwhere(bad)[:3].tolist()}")
