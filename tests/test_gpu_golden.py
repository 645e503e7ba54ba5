"""-m gpu: HIP path against the committed golden fixture (tests/golden/oracle_small.npz, produced by
tests/golden/make_golden.py) -- runs without building the oracle."""
import os

import numpy as np
import pytest

from helpers import assert_close, canon_pairs, field_atol

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "oracle_small.npz")


@pytest.fixture(scope="module")
def gold(cfgmod):
    g = np.load(GOLDEN)
    prm = cfgmod.params_from_values(dp=float(g["dp"]), DL=float(g["DL"]))
    nb = tuple(g["nb_" + n] for n in ("pair_i", "pair_j", "dx", "dy", "r", "W", "dW"))
    parts = dict(n_fluid=int(g["n_fluid"]), n_total=int(g["n_total"]), pos=g["pos"], vel=g["vel"], drho_dt=g["drho_dt"],
                 mass=g["mass"], wall_vel=g["wall_vel"])
    return g, prm, nb, parts


def test_golden_neighbor_list(gold, mex):
    g, prm, nb, parts = gold
    got = canon_pairs(mex.sph_neighbor_search_mex(parts["pos"], parts["n_fluid"], parts["n_total"], prm.h, prm.DL))
    ref = canon_pairs(nb)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    assert_close(got[4], ref[4], rtol=1e-14, atol=1e-15 * prm.DL, name="r")


def test_golden_modes(gold, mex):
    g, prm, nb, parts = gold
    nf, nt = parts["n_fluid"], parts["n_total"]
    p6 = nb[:5] + (nb[6],)
    dt = float(g["dt"])
    at = field_atol(prm, parts, nb, dt)
    rho, Vol, B = mex.sph_physics_shell_mex("density_correction", *nb, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    assert_close(rho, g["rho"], name="rho"); assert_close(Vol, g["Vol"], name="Vol")
    assert_close(B, g["B"], rtol=1e-10, atol_scale=1e-12, name="B")
    f = mex.sph_physics_shell_mex("viscous_force", *p6, parts["vel"], g["Vol"], g["B"], prm.mu, prm.h, nf, nt, parts["mass"],
                                  parts["wall_vel"])
    assert_close(f, g["viscous_force"], rtol=1e-10, atol=at["force"], name="viscous")
    tp = mex.sph_physics_shell_mex("transport_correction", *p6, g["Vol"], g["B"], parts["pos"], prm.h, nf, nt)
    assert_close(tp, g["transport_pos_default"], rtol=1e-13, atol_scale=1e-14, name="transport(0.2)")
    common = (g["Vol"], g["B"], g["rho"], parts["mass"], parts["pos"], parts["vel"], parts["drho_dt"], g["force_prior"], dt,
              nf, nt, prm.rho0, prm.p0, prm.c_f, parts["wall_vel"])
    for got, n in zip(mex.sph_physics_shell_mex("integration_verlet", *p6, *common), ("rho", "p", "pos", "vel", "drho", "force")):
        assert_close(got, g["verlet_" + n], rtol=1e-10, atol=at[n], name="verlet." + n)
    tail = (parts["mass"], parts["pos"], parts["vel"], parts["wall_vel"], g["rho"], parts["drho_dt"], dt, nf, nt, prm.rho0,
            prm.p0, prm.c_f, prm.mu, prm.h, prm.inv_sigma0, prm.gravity_g)
    names = ("rho", "p", "pos", "vel", "drho", "force", "force_prior", "Vol", "B")
    for got, n in zip(mex.sph_physics_shell_mex("advance_shell_step", *nb, *tail), names):
        assert_close(got, g["advance_" + n], rtol=1e-10, atol=at[n], name="advance." + n)
    tau = mex.sph_physics_shell_mex("wall_shear_monitor", *p6, parts["pos"], parts["vel"], parts["wall_vel"], g["Vol"], g["B"],
                                    nf, prm.DL, prm.DH, prm.mu, prm.h)
    assert_close(np.array(tau), g["tau"], rtol=1e-10, atol_scale=1e-12, name="tau")


def test_golden_resident_five_steps(gold, capi):
    g, prm, nb, parts = gold
    with capi.Context(prm, parts["n_fluid"], parts["n_total"], parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"],
                      parts["wall_vel"], t_end=1e9) as ctx:
        st = ctx.advance(1e9, max_steps=5)
        got = ctx.download()
        tb, tt, npairs = ctx.monitor(tau=True, pairs=True)
    assert abs(st["t"] - float(g["run5_t"])) < 1e-15
    for k in ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B"):
        assert_close(got[k], g["run5_" + k], rtol=1e-9, atol_scale=1e-10, name=k)
    assert npairs == float(g["run5_n_pairs"])
    assert_close(np.array([tb, tt]), g["run5_tau"], rtol=1e-8, atol_scale=1e-9, name="tau")
