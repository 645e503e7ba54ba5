"""-m gpu: every MEX-surface entry point of libsphx (HIP) against the CPU oracle on seeded inputs.

Tolerance: the reference computes in IEEE double; the HIP kernels use the same formulas and differ only
in summation order and FMA contraction, so per-mode outputs must agree to rtol 1e-11 (+1e-13 of the
field's magnitude) -- SURVEY.md section 8c's parity protocol.  Neighbour lists compare as sets: exact on
membership, 1e-14 relative on dx,dy,r and 1e-12 on W,dW.
"""
import numpy as np
import pytest

from helpers import assert_close, canon_pairs, field_atol, make_case

pytestmark = pytest.mark.gpu

CASES = [
    dict(dp=0.05, DL=3.0, jitter=0.0, developed=False),   # pristine lattice (config.ini as shipped)
    dict(dp=0.05, DL=3.0, jitter=0.2, developed=True),    # disordered + developed state
    dict(dp=0.04, DL=3.0, jitter=0.3, developed=True),    # DL/2h non-integer: seam handling
    dict(dp=0.025, DL=1.0, jitter=0.25, developed=True),  # short periodic box, many seam pairs
]


@pytest.fixture(scope="module", params=range(len(CASES)))
def case(request, cfgmod, geom, oracle):
    kw = CASES[request.param]
    prm, parts = make_case(cfgmod, geom, seed=100 + request.param, **kw)
    nb = oracle.neighbor_search(parts["pos"], parts["n_fluid"], parts["n_total"], prm.h, prm.DL)
    return prm, parts, nb


def test_neighbor_search_set_parity(case, mex):
    prm, parts, nb_ref = case
    nb = mex.sph_neighbor_search_mex(parts["pos"], parts["n_fluid"], parts["n_total"], prm.h, prm.DL)
    assert len(nb[0]) == len(nb_ref[0])
    a, b = canon_pairs(nb), canon_pairs(nb_ref)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # across the seam the oracle forms xi - (xj -+ DL) or (xi - xj) +- DL, the HIP path (xi -+ DL) - xj:
    # one rounding of magnitude DL apart
    for k, name in ((2, "dx"), (3, "dy"), (4, "r")):
        assert_close(a[k], b[k], rtol=1e-14, atol=4 * 2.3e-16 * prm.DL, name=name)
    for k, name in ((5, "W"), (6, "dW")):
        assert_close(a[k], b[k], rtol=1e-11, atol_scale=1e-12, name=name)
    # convention: fluid-fluid once with i<j, fluid particle always first
    nf = parts["n_fluid"]
    assert np.all(a[0] <= nf)
    ff = a[1] <= nf
    assert np.all(a[0][ff] < a[1][ff])


def _state(case, oracle):
    prm, parts, nb = case
    nf, nt = parts["n_fluid"], parts["n_total"]
    rho, Vol, B = oracle.density_correction(nb, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    return prm, parts, nb, nf, nt, rho, Vol, B


def test_density_correction(case, mex, oracle):
    prm, parts, nb, nf, nt, rho, Vol, B = _state(case, oracle)
    g_rho, g_Vol, g_B = mex.sph_physics_shell_mex("density_correction", *nb, parts["mass"], nf, nt, prm.rho0, prm.h,
                                                  prm.inv_sigma0)
    assert_close(g_rho, rho, name="rho")
    assert_close(g_Vol, Vol, name="Vol")
    assert_close(g_B, B, rtol=1e-10, atol_scale=1e-12, name="B")


def test_viscous_force(case, mex, oracle):
    prm, parts, nb, nf, nt, rho, Vol, B = _state(case, oracle)
    p6 = nb[:5] + (nb[6],)
    ref = oracle.viscous_force(nb, parts["vel"], Vol, B, prm.mu, prm.h, nf, nt, parts["mass"], parts["wall_vel"])
    got = mex.sph_physics_shell_mex("viscous_force", *p6, parts["vel"], Vol, B, prm.mu, prm.h, nf, nt, parts["mass"],
                                    parts["wall_vel"])
    assert_close(got, ref, rtol=1e-10, atol=field_atol(prm, parts, nb, 0.0)["force"], name="force")


def test_transport_correction(case, mex, oracle):
    prm, parts, nb, nf, nt, rho, Vol, B = _state(case, oracle)
    p6 = nb[:5] + (nb[6],)
    for coeff in (None, 0.30):
        ref = oracle.transport_correction(nb, Vol, B, parts["pos"], prm.h, nf, nt, 0.2 if coeff is None else coeff)
        args = (Vol, B, parts["pos"], prm.h, nf, nt) + (() if coeff is None else (coeff,))
        got = mex.sph_physics_shell_mex("transport_correction", *p6, *args)
        assert_close(got, ref, rtol=1e-13, atol_scale=1e-14, name="pos")


def _force_prior(prm, parts, nb, nf, nt, Vol, B, oracle):
    fp = oracle.viscous_force(nb, parts["vel"], Vol, B, prm.mu, prm.h, nf, nt, parts["mass"], parts["wall_vel"])
    fp[:nf, 0] += parts["mass"][:nf] * prm.gravity_g
    return fp


def test_integration_1st_2nd_verlet(case, mex, oracle):
    prm, parts, nb, nf, nt, rho, Vol, B = _state(case, oracle)
    p6 = nb[:5] + (nb[6],)
    fp = _force_prior(prm, parts, nb, nf, nt, Vol, B, oracle)
    dt = 0.25 * prm.h / (prm.c_f + 1.0)
    common = (Vol, B, rho, parts["mass"], parts["pos"], parts["vel"], parts["drho_dt"], fp, dt, nf, nt, prm.rho0,
              prm.p0, prm.c_f, parts["wall_vel"])
    ref1 = oracle.integration_1st(nb, *common)
    got1 = mex.sph_physics_shell_mex("integration_1st", *p6, *common)
    at = field_atol(prm, parts, nb, dt)
    for g, r, name in zip(got1, ref1, ("rho", "p", "pos", "force", "drho")):
        assert_close(g, r, rtol=1e-10, atol=at[name], name="int1." + name)
    rho_h, p_h, pos_h, force1, _ = ref1
    vel_new = parts["vel"].copy(order="F")
    vel_new[:nf] += (fp[:nf] + force1[:nf]) / parts["mass"][:nf, None] * dt
    ref2 = oracle.integration_2nd(nb, Vol, rho_h, pos_h, vel_new, dt, nf, nt, parts["wall_vel"])
    got2 = mex.sph_physics_shell_mex("integration_2nd", *p6, Vol, rho_h, pos_h, vel_new, dt, nf, nt, parts["wall_vel"])
    for g, r, name in zip(got2, ref2, ("pos", "drho", "zeros")):
        assert_close(g, r, rtol=1e-10, atol=at[name], name="int2." + name)
    assert not np.any(got2[2])
    refv = oracle.integration_verlet(nb, *common)
    gotv = mex.sph_physics_shell_mex("integration_verlet", *p6, *common)
    for g, r, name in zip(gotv, refv, ("rho", "p", "pos", "vel", "drho", "force")):
        assert_close(g, r, rtol=1e-10, atol=at[name], name="verlet." + name)


def test_advance_shell_step(case, mex, oracle):
    prm, parts, nb, nf, nt, rho, Vol, B = _state(case, oracle)
    dt = 0.25 * prm.h / (prm.c_f + 1.0)
    tail = (parts["mass"], parts["pos"], parts["vel"], parts["wall_vel"], rho, parts["drho_dt"], dt, nf, nt, prm.rho0,
            prm.p0, prm.c_f, prm.mu, prm.h, prm.inv_sigma0, prm.gravity_g)
    ref = oracle.advance_shell_step(nb, *tail)
    got = mex.sph_physics_shell_mex("advance_shell_step", *nb, *tail)
    names = ("rho", "p", "pos", "vel", "drho", "force", "force_prior", "Vol", "B")
    at = field_atol(prm, parts, nb, dt)
    for g, r, name in zip(got, ref, names):
        assert_close(g, r, rtol=1e-10, atol=at[name], name="advance." + name)


def test_wall_shear_monitor(case, mex, oracle):
    prm, parts, nb, nf, nt, rho, Vol, B = _state(case, oracle)
    p6 = nb[:5] + (nb[6],)
    ref = oracle.wall_shear_monitor(nb, parts["pos"], parts["vel"], parts["wall_vel"], Vol, B, nf, prm.DL, prm.DH,
                                    prm.mu, prm.h)
    got = mex.sph_physics_shell_mex("wall_shear_monitor", *p6, parts["pos"], parts["vel"], parts["wall_vel"], Vol, B,
                                    nf, prm.DL, prm.DH, prm.mu, prm.h)
    assert_close(np.array(got), np.array(ref), rtol=1e-10, atol_scale=1e-12, name="tau")


def test_error_ids(mex):
    with pytest.raises(mex.MexError) as e:
        mex.sph_physics_shell_mex("no_such_mode")
    assert e.value.identifier == "SPH:Physics:mode"
    with pytest.raises(mex.MexError) as e:
        mex.sph_physics_shell_mex("density_correction", 1, 2, 3)
    assert e.value.identifier == "SPH:Physics:density:nrhs"
    with pytest.raises(mex.MexError) as e:
        mex.sph_neighbor_search_mex(np.zeros((4, 2)), 5, 4, 0.1, 1.0)
    assert e.value.identifier == "SPH:Neighbor:count"
    with pytest.raises(mex.MexError) as e:
        mex.sph_neighbor_search_mex(np.zeros((4, 2)), 2, 4, -0.1, 1.0)
    assert e.value.identifier == "SPH:Neighbor:param"
