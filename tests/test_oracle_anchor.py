"""Pins for the CPU oracle (not gpu).

The reference ships no tests or golden vectors and cannot be built here (mex.h), so the restatement is
anchored on what IS recorded from the reference itself in BASELINE.md section 2 / SURVEY.md section 8:
  * the pair count of sph_neighbor_search_mex on the reference's own initial lattice at four
    resolutions (exact integers);
  * the analytic Poiseuille solution the reference validates against (postprocess L2 < 5 %);
  * steps-to-20 s (19 771 at dp = 0.05) and L2 at 20 s (1.42 %) -- chaotic at round-off, so a statistical
    anchor: oracle runs give 19 776-19 786 steps and L2 1.3-2.3 % over the last seconds.
and on the committed golden fixture (regression pin)."""
import os

import numpy as np
import pytest

from helpers import assert_close, canon_pairs

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "oracle_small.npz")


@pytest.mark.parametrize("dp,DL,n_fluid,n_wall,pairs", [(0.05, 3.0, 1200, 480, 12660), (0.04, 3.0, 1875, 600, 19575),
                                                         (0.025, 3.0, 4800, 960, 49320), (0.01, 6.0, 60000, 4800, 606600)])
def test_pair_count_matches_reference_probe(cfgmod, geom, oracle, dp, DL, n_fluid, n_wall, pairs):
    prm = cfgmod.params_from_values(dp=dp, DL=DL)
    parts = geom.init_particles(prm)
    assert (parts["n_fluid"], parts["n_wall"]) == (n_fluid, n_wall)
    nb = oracle.neighbor_search(parts["pos"], parts["n_fluid"], parts["n_total"], prm.h, prm.DL)
    assert len(nb[0]) == pairs
    i, j = nb[0].astype(int), nb[1].astype(int)
    assert i.min() >= 1 and i.max() <= n_fluid                       # wall particles never appear as pair_i
    ff = j <= n_fluid
    assert np.all(i[ff] < j[ff])                                      # fluid-fluid stored once, i < j
    assert len(set(zip(i.tolist(), j.tolist()))) == pairs             # seen_neighbor: no duplicates across the seam
    assert np.all(nb[4] < 2 * prm.h) and np.all(nb[4] > 1e-12)


def test_neighbor_set_equals_bruteforce_min_image(cfgmod, geom, oracle):
    """{(i,j): min-image r < 2h, r^2 > 1e-24}: ghost entries + dedup must produce exactly this set, also when
    DL/2h is not an integer and particles sit within 2h of both ends."""
    from helpers import make_case
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.0, jitter=0.3, seed=5, developed=False)
    nf, nt = parts["n_fluid"], parts["n_total"]
    nb = canon_pairs(oracle.neighbor_search(parts["pos"], nf, nt, prm.h, prm.DL))
    x, y = parts["pos"][:, 0], parts["pos"][:, 1]
    dx = x[:nf, None] - x[None, :]
    dx = np.where(dx > 0.5 * prm.DL, dx - prm.DL, np.where(dx < -0.5 * prm.DL, dx + prm.DL, dx))
    dy = y[:nf, None] - y[None, :]
    r2 = dx * dx + dy * dy
    ok = (r2 > 1e-24) & (r2 < (2 * prm.h) ** 2)
    jj = np.arange(nt)[None, :]
    ii = np.arange(nf)[:, None]
    ok &= (jj >= nf) | (jj > ii)
    bi, bj = np.nonzero(ok)
    assert np.array_equal(nb[0].astype(int) - 1, bi) and np.array_equal(nb[1].astype(int) - 1, bj)
    assert_close(nb[4], np.sqrt(r2[bi, bj]), rtol=1e-13, atol=1e-15, name="r")


def test_kernel_normalisation_and_w0(oracle):
    import ctypes as C
    h = 0.065
    W, dW = C.c_double(), C.c_double()
    f = oracle.lib().orc_kernel
    rs = np.linspace(1e-6, 2 * h, 20001)
    ws = []
    for r in rs:
        f(C.c_double(r), C.c_double(h), C.byref(W), C.byref(dW))
        ws.append(W.value)
    integral = np.trapezoid(np.array(ws) * 2 * np.pi * rs, rs)
    assert abs(integral - 1.0) < 1e-6                                  # 2-D cubic spline integrates to one
    f(C.c_double(0.0), C.c_double(h), C.byref(W), C.byref(dW))
    assert abs(W.value - 10.0 / (7.0 * np.pi * h * h)) < 1e-12 and dW.value == 0.0
    f(C.c_double(2 * h), C.c_double(h), C.byref(W), C.byref(dW))
    assert W.value == 0.0 and dW.value == 0.0


def test_lattice_density_is_rho0_and_B_is_identity_in_the_bulk(cfgmod, geom, oracle):
    prm = cfgmod.params_from_values(dp=0.05, DL=3.0)
    parts = geom.init_particles(prm)
    nf, nt = parts["n_fluid"], parts["n_total"]
    nb = oracle.neighbor_search(parts["pos"], nf, nt, prm.h, prm.DL)
    rho, Vol, B = oracle.density_correction(nb, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    assert np.all(rho[nf:] == prm.rho0) and np.all(B[nf:, 0] == 1) and np.all(B[nf:, 1] == 0)
    bulk = (parts["pos"][:nf, 1] > 0.3) & (parts["pos"][:nf, 1] < 0.7)
    assert np.allclose(rho[:nf][bulk], rho[:nf][bulk][0], rtol=1e-12)  # translation invariance incl. the seam
    assert abs(rho[:nf][bulk][0] / prm.rho0 - 1.0) < 0.02
    assert np.allclose(B[:nf][bulk][:, [1, 2]], 0.0, atol=1e-10)       # symmetric ring: no off-diagonal
    assert np.allclose(B[:nf][bulk][:, 0], B[:nf][bulk][:, 3], rtol=1e-10)


def test_start_up_matches_analytic_transient(cfgmod, geom, oracle, profmod):
    """Early on the channel accelerates uniformly (u ~ g t) and the profile approaches the analytic
    parabola the reference validates against: L2 falls 37 % -> ~13 % -> ~4.5 % over the first 3 s."""
    prm = cfgmod.params_from_values(dp=0.05, DL=3.0)
    parts = geom.init_particles(prm)
    nf = parts["n_fluid"]
    st = oracle.run(prm, parts, t_end=1.0, output_interval=1.0, enable_sort=False)
    assert 930 <= st["stats"]["steps"] <= 960                         # reference probe: ~19.8 k steps / 20 s
    u = st["vel"][:nf, 0]
    mid = np.abs(st["pos"][:nf, 1] - 0.5) < 0.1
    assert abs(u[mid].mean() - 0.628) < 0.02                          # analytic series solution at t = 1 s
    y, um, ue = profmod.final_profile(st["pos"][:nf], u, prm)
    assert 0.33 < profmod.l2_error(um, ue) < 0.41
    assert abs(st["stats"]["tau_bottom"] - st["stats"]["tau_top"]) < 0.02


def test_golden_fixture_reproduced(cfgmod, oracle):
    g = np.load(GOLDEN)
    prm = cfgmod.params_from_values(dp=float(g["dp"]), DL=float(g["DL"]))
    nf, nt = int(g["n_fluid"]), int(g["n_total"])
    nb = oracle.neighbor_search(g["pos"], nf, nt, prm.h, prm.DL)
    for k, name in enumerate(("pair_i", "pair_j", "dx", "dy", "r", "W", "dW")):
        assert np.array_equal(nb[k], g["nb_" + name]), name          # serial oracle: bit-exact
    rho, Vol, B = oracle.density_correction(nb, g["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    assert np.array_equal(rho, g["rho"]) and np.array_equal(B, g["B"])
    parts = dict(n_fluid=nf, n_total=nt, pos=g["pos"], vel=g["vel"], drho_dt=g["drho_dt"], mass=g["mass"],
                 wall_vel=g["wall_vel"])
    run = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=5, enable_sort=False)
    for k in ("pos", "vel", "drho_dt", "rho", "p"):
        assert np.array_equal(run[k], g["run5_" + k]), k


def test_omp_oracle_agrees_with_serial(cfgmod, geom, oracle):
    from helpers import make_case
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.5, jitter=0.2, seed=9)
    a = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=3, enable_sort=False)
    oracle.set_num_threads(4)
    b = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=3, enable_sort=False, omp=True)
    for k in ("pos", "vel", "drho_dt"):
        assert_close(b[k], a[k], rtol=1e-9, atol_scale=1e-11, name=k)


def test_sort_keeps_physics(cfgmod, geom, oracle):
    """The periodic cell re-sort (SPH_Poiseuille.m:272-278) only permutes rows."""
    from helpers import make_case
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.5, jitter=0.2, seed=9, sort_interval=2)
    a = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=5, enable_sort=False)
    b = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=5, enable_sort=True)
    inv = b["order"]
    assert not np.array_equal(inv, np.arange(len(inv)))
    for k in ("pos", "vel"):
        back = np.empty_like(b[k])
        back[inv] = b[k]
        assert_close(back, a[k], rtol=1e-9, atol_scale=1e-11, name=k)


# Realisations recorded in this container (round 4; oracle, lattice at rest -> t = 20 s; L2 of the y-binned u_x profile at
# t = 16, 17, 18, 19, 20 s | L2 of the profile AVERAGED over those five instants | steps):
#   dp 0.05 (reference: 19 771 steps, 1.42 % at 20 s)
#     serial          1.83 1.70 1.60 1.69 1.29 | 1.52 | 19 778 (stops at five output points: + 4 clipped steps)
#     8 OpenMP thr.   1.93 1.54 1.72 1.68 1.76 | 1.62 | 19 782
#                     1.66 1.65 2.37 2.32 1.98 | 1.88 | 19 777
#                     2.01 1.88 2.35 1.73 2.15 | 1.93 | 19 771
#   dp 0.04, c_f 10, transport_coeff 0.10 (reference: 16 895 steps, 2.75 % at 20 s)
#     serial          1.68 2.19 1.90 2.00 2.14 | 1.96 | 16 897
#     8 OpenMP thr.   2.39 2.53 2.41 2.49 2.46 | 2.44 | 16 895
#                     2.49 2.71 2.82 2.64 2.93 | 2.70 | 16 893
#                     3.11 3.08 3.07 3.12 2.94 | 3.05 | 16 896
# A single instant wanders by +-30 % within one realisation and between realisations (the flow is chaotic at round-off; with
# more than one OpenMP thread the reference's atomic scatter makes every run a new realisation) -- round 3 asserted one such
# instant of an 8-thread run inside 1.7 x the reference's and failed one CPU run in two.  Now: the SERIAL oracle, which is
# deterministic (same bits every run in this image), the hard window on the five-instant mean profile (0.6 .. 1.5 x the
# reference's figure: every realisation above lies inside 0.71 .. 1.36 x), a loose one on the last instant.
@pytest.mark.parametrize("kw, ref_steps, ref_L2", [
    (dict(dp=0.05, DL=3.0), 19771, 0.0142),                                   # config.ini as shipped
    (dict(dp=0.04, DL=3.0, c_f=10.0, transport_coeff=0.10), 16895, 0.0275),    # README-table constants
])
def test_full_runs_match_the_figures_recorded_from_the_reference(cfgmod, geom, oracle, profmod, kw, ref_steps, ref_L2):
    """steps to t = 20 s and L2(20 s) of the reference's own C code (compiled unmodified
    at survey time).  The step count is the sum of the dt sequence, i.e. of the whole max|v| history.  Two parameter
    sets: different sound speed (dt rule, EOS stiffness) and shifting strength.  ~35 s + ~45 s of one core."""
    prm = cfgmod.params_from_values(end_time=20.0, output_interval=20.0, **kw)
    parts = geom.init_particles(prm)
    nf = parts["n_fluid"]
    pos = vel = drho = None
    t, steps, profiles, L2s = 0.0, 0, [], []
    for t_out in (16.0, 17.0, 18.0, 19.0, 20.0):  # (every stop clips one dt: + 4 steps against the reference's single clip at t_end)
        st = oracle.run(prm, parts, t_end=t_out, output_interval=t_out, enable_sort=False, omp=False,
                        pos=pos, vel=vel, drho_dt=drho, t0=t, step0=steps)
        pos, vel, drho, t = st["pos"], st["vel"], st["drho_dt"], t_out
        steps += st["stats"]["steps"]
        y, um, ue = profmod.final_profile(st["pos"][:nf], st["vel"][:nf, 0], prm)
        profiles.append(um)
        L2s.append(profmod.l2_error(um, ue))
    assert abs(steps - ref_steps) <= 0.002 * ref_steps, (steps, ref_steps)
    L2_mean = profmod.l2_error(np.nanmean(np.array(profiles), axis=0), ue)
    assert 0.6 * ref_L2 < L2_mean < 1.5 * ref_L2, (L2_mean, L2s, ref_L2)       # the hard bound: the five-instant mean profile
    assert 0.4 * ref_L2 < L2s[-1] < 2.0 * ref_L2, (L2s, ref_L2)                # one instant: loose
    assert abs(st["stats"]["tau_bottom"] - 0.4) < 0.04 and abs(st["stats"]["tau_top"] - 0.4) < 0.04
