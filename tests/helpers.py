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


def assert_close(a, b, rtol=1e-11, atol_scale=1e-13, name="", atol=0.0):
    """|a-b| <= rtol*|b| + atol_scale*max|b| + atol element-wise.  The HIP kernels use the reference's
    formulas; differences come from summation order / FMA contraction only."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f"{name}: shape {a.shape} vs {b.shape}"
    assert np.all(np.isfinite(a)), f"{name}: non-finite values"
    scale = np.max(np.abs(b)) if b.size else 0.0
    err = np.abs(a - b)
    tol = rtol * np.abs(b) + atol_scale * max(scale, 1e-300) + atol + 1e-300
    bad = err > tol
    assert not np.any(bad), (f"{name}: {int(bad.sum())}/{a.size} elements differ, max err {err.max():.3e} "
                             f"(scale {scale:.3e}, atol {atol:.1e}) at {np.argwhere(bad)[:3].tolist()}")


def field_atol(prm, parts, nb, dt):
    """Absolute round-off floors per field.  The weakly-compressible EOS p = p0 (rho/rho0 - 1) with
    p0 = rho0 c_f^2 turns a 1-ulp density difference into p0*eps of pressure, which then propagates into
    force, velocity and drho_dt sums that may cancel to ~0 (e.g. on the pristine lattice).  Floors are
    1e3 ulps of that chain -- ten orders of magnitude below any formula error."""
    eps = 2.3e-16 * 1e3
    dWV = float(np.max(np.abs(nb[6]))) * float(np.max(parts["mass"])) / prm.rho0 if len(nb[6]) else 1.0
    m = float(np.min(parts["mass"]))
    vol = float(np.max(parts["mass"])) / prm.rho0
    vmax = float(np.max(np.abs(parts["vel"]))) + prm.gravity_g * dt
    a_rho = eps * prm.rho0
    a_p = eps * prm.p0
    a_F = 30 * dWV * vol * (a_p + eps * prm.mu * vmax / prm.h)
    a_v = a_F / m * dt + eps * vmax
    a_drho = 30 * dWV * prm.rho0 * (a_v + eps * vmax)
    return dict(rho=a_rho + 0.5 * dt * a_drho, p=a_p + prm.p0 / prm.rho0 * 0.5 * dt * a_drho, force=a_F,
                force_prior=a_F, vel=a_v, drho=a_drho, drho_dt=a_drho, pos=eps * prm.DL + dt * a_v, Vol=eps * vol,
                B=1e-12, zeros=0.0)
