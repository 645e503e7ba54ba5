"""The three MEX gateways (sph-poiseuille-flow_amd/matlab/*.c) compiled and RUN against a mock of MATLAB's C Matrix/MEX API
(tests/stubs/mex.h + mex_mock.c, driven by tests/mex_mock.py) -- there is no MATLAB in the image, so this is as close as
the gateways get to being exercised: argument order, arity / shape checks, error identifiers and the calls into libsphx
are the real code; only mxArray and mexErrMsgIdAndTxt are stand-ins.

CPU part : the gateways compile with -Wall -Wextra -Werror, report the same error identifiers as the Python mirror
           (mex_surface.py, itself a restatement of mex/sph_physics_mex.c:1735-1772 and sph_neighbor_search_mex.c's
           checks) for the same bad arguments, and surface libsphx's "no device" error instead of crashing.
GPU part : what a gateway returns is what mex_surface / capi return for the same inputs (same library, same kernels:
           equal to rounding of the unordered atomic sums, rtol 1e-12), for the neighbour search, all eight physics
           modes and the resident-context gateway.
"""
import numpy as np
import pytest

import mex_mock
from helpers import assert_close, canon_pairs, make_case

SOURCES = ("sph_neighbor_search_gateway.c", "sph_physics_shell_gateway.c", "sphx_ctx_mex.c")


@pytest.fixture(scope="module")
def gateways():
    return {s: mex_mock.Gateway(s) for s in SOURCES}


@pytest.mark.parametrize("source", SOURCES)
def test_gateway_compiles_without_warnings(source):
    mex_mock.compile_only(source)


def _ident(fn, *args):
    try:
        fn(*args)
    except Exception as e:                       # MexError of either surface
        return e.identifier
    return None


def test_physics_gateway_reports_the_mirrors_error_identifiers(gateways, mex):
    gw = gateways["sph_physics_shell_gateway.c"]
    z, pairs = np.zeros(3), [np.ones(2)] * 6
    nlhs = {k: v[1] for k, v in mex._MODES.items()}
    bad = [
        ("density_correction",) + (z,) * 5, ("viscous_force",) + (z,) * 3, ("transport_correction",) + (z,) * 3,
        ("integration_1st",), ("integration_2nd",), ("integration_verlet",), ("advance_shell_step",),
        ("wall_shear_monitor",), ("bogus",), (3.0,),
        ("viscous_force", *pairs, np.zeros((4, 3)), np.zeros(4), np.zeros((4, 4)), 0.1, 0.1, 2, 4, np.ones(4), np.zeros((4, 2))),
        ("viscous_force", *pairs, np.zeros((4, 2)), np.zeros(5), np.zeros((4, 4)), 0.1, 0.1, 2, 4, np.ones(4), np.zeros((4, 2))),
        ("viscous_force", *pairs, np.zeros((4, 2)), np.zeros(4), np.zeros((4, 3)), 0.1, 0.1, 2, 4, np.ones(4), np.zeros((4, 2))),
        ("viscous_force", np.ones(2), np.ones(3), *([np.ones(2)] * 4), np.zeros((4, 2)), np.zeros(4), np.zeros((4, 4)), 0.1, 0.1,
         2, 4, np.ones(4), np.zeros((4, 2))),
        ("transport_correction", *pairs, np.zeros(4), np.zeros((4, 4)), np.zeros((4, 2)), 0.1, 2, 4, -1.0),
        ("transport_correction", *pairs, np.zeros(4), np.zeros((4, 4)), np.zeros((3, 2)), 0.1, 2, 4),
        ("density_correction", *pairs, np.ones(3), np.ones(4), 2, 4, 1.0, 0.1, 1.0),
        ("density_correction", *pairs, np.ones(2), np.ones(4), 0, 4, 1.0, 0.1, 1.0),
        ("density_correction", *pairs, np.ones(2), np.ones(5), 2, 4, 1.0, 0.1, 1.0),
        ("density_correction", *pairs, np.ones(2), np.ones(4), 2, 4, -1.0, 0.1, 1.0),
        ("density_correction", *pairs, np.ones(2), np.ones(4), 2, 4, 1.0, 0.0, 1.0),
        ("integration_2nd", *pairs, np.zeros(4), np.zeros(4), np.zeros((4, 2)), np.zeros((4, 1)), 0.1, 2, 4, np.zeros((4, 2))),
        ("integration_1st", *pairs, np.zeros(4), np.zeros((4, 4)), np.zeros(3), *([z] * 12)),
        ("integration_verlet", *pairs, np.zeros(4), np.zeros((4, 4)), np.zeros(4), np.zeros(4), np.zeros((4, 2)), np.zeros((4, 2)),
         np.zeros(4), np.zeros((4, 2)), 0.1, 2, 4, 1.0, 0.0, 10.0, np.zeros((5, 2))),
        ("advance_shell_step", *pairs, np.ones(2), np.ones(4), np.zeros((4, 2)), np.zeros((4, 2)), np.zeros((4, 2)), np.ones(4),
         np.zeros(3), 0.1, 2, 4, *([1.0] * 7)),
        ("wall_shear_monitor", *pairs, np.zeros((4, 2)), np.zeros((4, 2)), np.zeros((4, 2)), np.ones(4), np.zeros((4, 4)), 2, -1.0,
         1.0, 0.1, 0.1),
    ]
    for args in bad:
        n = nlhs.get(args[0], 1) if isinstance(args[0], str) else 1
        want = _ident(mex.sph_physics_shell_mex, *args)
        assert want is not None and want.startswith("SPH:Physics:"), (args[0], want)
        assert _ident(gw, n, *args) == want, (args[0], want)
    assert _ident(gw, 1) == "SPH:Physics:nrhs"
    assert _ident(gw, 2, "density_correction", *([z] * 13)) == "SPH:Physics:density:nlhs"
    assert _ident(gw, 4, "integration_1st", *([z] * 21)) == "SPH:Physics:int1:nlhs"
    assert _ident(gw, 8, "advance_shell_step", *([z] * 23)) == "SPH:Physics:advance:nlhs"


def test_neighbor_gateway_reports_the_mirrors_error_identifiers(gateways, mex):
    gw = gateways["sph_neighbor_search_gateway.c"]
    bad = [(np.zeros((4, 3)), 2, 4, 0.1, 1.0), (np.zeros((4, 2)), 2, 5, 0.1, 1.0), (np.zeros((4, 2)), 5, 4, 0.1, 1.0),
           (np.zeros((4, 2)), 0, 4, 0.1, 1.0), (np.zeros((4, 2)), 2, 4, 0.1), (np.zeros((4, 2)), 2, 4, 0.1, 1.0, 1.0),
           (np.zeros((4, 2)), 2, 4, 0.1, -1.0), (np.zeros((4, 2)), 2, 4, -0.1, 1.0)]
    for args in bad:
        want = _ident(mex.sph_neighbor_search_mex, *args)
        assert want is not None and want.startswith("SPH:Neighbor:"), want
        assert _ident(gw, 7, *args) == want, want
    assert _ident(gw, 6, np.zeros((4, 2)), 2, 4, 0.1, 1.0) == "SPH:Neighbor:nlhs"


def test_context_gateway_checks_its_arguments(gateways):
    gw = gateways["sphx_ctx_mex.c"]
    assert _ident(gw, 0) == "SPHX:Ctx:cmd"
    assert _ident(gw, 0, 1.0) == "SPHX:Ctx:cmd"
    assert _ident(gw, 0, "bogus") == "SPHX:Ctx:cmd"
    assert _ident(gw, 1, "create", {"DL": 3.0}) == "SPHX:Ctx:nrhs"
    assert _ident(gw, 1, "advance", mex_mock.Handle(0)) == "SPHX:Ctx:nrhs"
    assert _ident(gw, 2, "advance", mex_mock.Handle(0), 1.0, 5) == "SPHX:Ctx:nlhs"
    assert _ident(gw, 1, "advance", 0.0, 1.0, 5) == "SPHX:Ctx:handle"
    assert _ident(gw, 1, "prepare", mex_mock.Handle(0), 5) == "SPHX:Ctx:nlhs"
    z2, z1 = np.zeros((4, 2)), np.zeros(4)
    assert _ident(gw, 1, "create", {"DL": 3.0}, 2, 4, z2, z2, z1, z1, z2, 0.0, 0) == "SPHX:Ctx:cfg"     # DH missing
    assert _ident(gw, 1, "create", {"DL": 3.0}, 2, 4, z2, "vel", z1, z1, z2, 0.0, 0) == "SPHX:Ctx:type"
    assert gw.lock_count() == 0


def test_library_errors_come_through_the_gateways(gateways, capi, cfgmod, geom):
    """On a box without a GPU every compute entry point of libsphx reports SPHX:NoDevice; the gateways must hand that
    identifier to mexErrMsgIdAndTxt, not crash or return garbage."""
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present: the no-device path cannot be shown here")
    prm, parts = make_case(cfgmod, geom, dp=0.1, DL=1.0, jitter=0.0, developed=False)
    args = (parts["pos"], parts["n_fluid"], parts["n_total"], prm.h, prm.DL)
    assert _ident(gateways["sph_neighbor_search_gateway.c"], 7, *args) == "SPHX:NoDevice"
    pairs = [np.ones(2)] * 7
    assert _ident(gateways["sph_physics_shell_gateway.c"], 3, "density_correction", *pairs, np.ones(4), 2, 4, 1.0, 0.1, 1.0) \
        == "SPHX:NoDevice"
    gw = gateways["sphx_ctx_mex.c"]
    assert _ident(gw, 1, "create", _cfg(prm), parts["n_fluid"], parts["n_total"], parts["pos"], parts["vel"],
                  parts["drho_dt"], parts["mass"], parts["wall_vel"], 0.0, 0) == "SPHX:NoDevice"
    assert gw.lock_count() == 0                 # a failed create must not leave the MEX file locked


def _cfg(prm, t_end=None):
    return dict(DL=prm.DL, DH=prm.DH, dp=prm.dp, h=prm.h, rho0=prm.rho0, mu=prm.mu, c_f=prm.c_f, p0=prm.p0,
                inv_sigma0=prm.inv_sigma0, gravity_g=prm.gravity_g, transport_coeff=prm.transport_coeff,
                t_end=prm.t_end if t_end is None else t_end, sort_interval=prm.sort_interval)


# ---------------------------------------------------------------------------------------------- GPU part
RTOL = 1e-12            # same kernels on the same inputs; only the order of the atomic sums may differ between two calls


@pytest.fixture(scope="module")
def case(cfgmod, geom):
    return make_case(cfgmod, geom, dp=0.04, DL=3.0, jitter=0.3, developed=True, seed=77)


def _same(got, want, name):
    got = got if isinstance(got, (list, tuple)) else [got]
    want = want if isinstance(want, (list, tuple)) else [want]
    assert len(got) == len(want), name
    for k, (g, w) in enumerate(zip(got, want)):
        g, w = np.asarray(g, dtype=np.float64), np.asarray(w, dtype=np.float64)
        assert g.shape == w.shape, (name, k, g.shape, w.shape)
        assert_close(g, w, rtol=RTOL, atol_scale=1e-13, name=f"{name}[{k}]")


@pytest.mark.gpu
def test_neighbor_gateway_returns_what_the_library_returns(gateways, mex, case):
    prm, parts = case
    args = (parts["pos"], parts["n_fluid"], parts["n_total"], prm.h, prm.DL)
    got = gateways["sph_neighbor_search_gateway.c"](7, *args)
    want = mex.sph_neighbor_search_mex(*args)
    assert len(got) == 7 and len(got[0]) == len(want[0]) > 0
    a, b = canon_pairs(tuple(got)), canon_pairs(tuple(want))
    for k in range(7):
        assert np.array_equal(a[k], b[k]), k        # same kernel, same arithmetic per pair: bit for bit once sorted


@pytest.mark.gpu
def test_physics_gateway_returns_what_the_library_returns(gateways, mex, case):
    prm, parts = case
    gw = gateways["sph_physics_shell_gateway.c"]
    nf, nt = parts["n_fluid"], parts["n_total"]
    nb = tuple(mex.sph_neighbor_search_mex(parts["pos"], nf, nt, prm.h, prm.DL))
    p6 = nb[:5] + (nb[6],)
    nlhs = {k: v[1] for k, v in mex._MODES.items()}

    def both(mode, *args):
        want = mex.sph_physics_shell_mex(mode, *args)
        got = gw(nlhs[mode], mode, *args)
        _same(got, want, mode)
        return want

    rho, Vol, B = both("density_correction", *nb, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    fp = both("viscous_force", *p6, parts["vel"], Vol, B, prm.mu, prm.h, nf, nt, parts["mass"], parts["wall_vel"])
    fp = np.array(fp[0] if isinstance(fp, (list, tuple)) else fp, order="F")
    fp[:nf, 0] += parts["mass"][:nf] * prm.gravity_g
    both("transport_correction", *p6, Vol, B, parts["pos"], prm.h, nf, nt)
    both("transport_correction", *p6, Vol, B, parts["pos"], prm.h, nf, nt, 0.3)
    dt = 0.25 * prm.h / (prm.c_f + 1.0)
    common = (Vol, B, rho, parts["mass"], parts["pos"], parts["vel"], parts["drho_dt"], fp, dt, nf, nt, prm.rho0, prm.p0,
              prm.c_f, parts["wall_vel"])
    rho_h, p_h, pos_h, force1, _ = both("integration_1st", *p6, *common)
    both("integration_verlet", *p6, *common)
    both("integration_2nd", *p6, Vol, rho_h, pos_h, parts["vel"], dt, nf, nt, parts["wall_vel"])
    both("advance_shell_step", *nb, parts["mass"], parts["pos"], parts["vel"], parts["wall_vel"], rho, parts["drho_dt"], dt,
         nf, nt, prm.rho0, prm.p0, prm.c_f, prm.mu, prm.h, prm.inv_sigma0, prm.gravity_g)
    both("wall_shear_monitor", *p6, parts["pos"], parts["vel"], parts["wall_vel"], Vol, B, nf, prm.DL, prm.DH, prm.mu, prm.h)


@pytest.mark.gpu
def test_context_gateway_runs_the_resident_loop(gateways, capi, case):
    prm, parts = case
    gw = gateways["sphx_ctx_mex.c"]
    nf, nt = parts["n_fluid"], parts["n_total"]
    state = (parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"])
    (h,) = gw(1, "create", _cfg(prm), nf, nt, *state, 0.0, 0)
    assert isinstance(h, mex_mock.Handle) and h.value != 0 and gw.lock_count() == 1
    try:
        gw(0, "prepare", h, 12)
        (st,) = gw(1, "advance", h, prm.t_end, 12)
        (st2,) = gw(1, "advance", h, prm.t_end, 12)
        fields = gw(9, "download", h)
        tau = gw(3, "monitor", h)
        stats = gw(3, "graph_stats", h)
    finally:
        gw(0, "destroy", h)
    assert gw.lock_count() == 0
    with capi.Context(prm, nf, nt, *state) as ctx:
        ctx.advance(prm.t_end, 12)
        want_st = ctx.advance(prm.t_end, 12)
        want = ctx.download()
        want_tau = ctx.monitor(tau=True, pairs=True)
    assert st["step"] == 12 and st2["step"] == 24 and st2["done"] == 0
    for k in ("t", "dt_last", "dt_next", "vmax"):
        assert_close(np.array(st2[k]), np.array(want_st[k]), rtol=1e-10, name="status." + k)
    order = ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B")
    for g, k in zip(fields, order):
        assert g.shape == want[k].shape, k
        scale = float(np.max(np.abs(want[k]))) or 1.0
        assert_close(g, want[k], rtol=1e-9, atol=1e-11 * scale, name="download." + k)
    assert_close(np.array(tau[:2]), np.array(want_tau[:2]), rtol=1e-8, atol=1e-12, name="tau")
    assert tau[2] == want_tau[2] > 0
    assert stats[0] >= 12 and stats[2] >= 1          # the prepared graph was replayed
