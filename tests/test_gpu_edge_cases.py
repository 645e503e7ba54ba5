"""-m gpu: edge cases the reference's guards exist for -- coincident particles (r^2 <= 1e-24,
neighbor.c:368), an isolated particle (singular KGC fallbacks, physics.c:335-339,354-356), empty pair lists,
a particle outside the wall rows, non-finite input (device status instead of a silent wrong answer) and
neighbour-list overflow."""
import numpy as np
import pytest

from helpers import assert_close, canon_pairs, field_atol, make_case

pytestmark = pytest.mark.gpu


def _case_with_defects(cfgmod, geom):
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.5, jitter=0.15, seed=21, developed=True)
    nf = parts["n_fluid"]
    pos = parts["pos"]
    pos[7] = pos[3]                       # exact duplicate: r^2 = 0 -> never a pair
    pos[11, 0], pos[11, 1] = pos[12, 0] + 3e-13, pos[12, 1]   # r = 3e-13 < 1e-12 -> filtered too
    pos[20] = (0.75, 0.5)                 # then clear a hole around it: isolated particle
    d = np.hypot(pos[:nf, 0] - 0.75, pos[:nf, 1] - 0.5)
    far = np.nonzero((d > 1e-9) & (d < 2.2 * prm.h))[0]
    pos[far, 1] = 0.08 + 0.02 * np.arange(len(far)) / max(len(far), 1)   # park them near the bottom wall
    pos[far, 0] = 0.05 + 1.4 * np.arange(len(far)) / max(len(far), 1)
    return prm, parts


def test_coincident_and_isolated_particles(cfgmod, geom, mex, oracle, capi):
    prm, parts = _case_with_defects(cfgmod, geom)
    nf, nt = parts["n_fluid"], parts["n_total"]
    ref = oracle.neighbor_search(parts["pos"], nf, nt, prm.h, prm.DL)
    got = mex.sph_neighbor_search_mex(parts["pos"], nf, nt, prm.h, prm.DL)
    a, b = canon_pairs(got), canon_pairs(ref)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    pairs = set(zip(a[0].astype(int).tolist(), a[1].astype(int).tolist()))
    assert (4, 8) not in pairs and (12, 13) not in pairs            # 1-based (3,7) and (11,12)
    assert not np.any(a[0] == 21) and not np.any(a[1] == 21)         # the isolated particle has no pairs
    rho, Vol, B = mex.sph_physics_shell_mex("density_correction", *ref, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    rho_r, Vol_r, B_r = oracle.density_correction(ref, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    assert_close(rho, rho_r, name="rho")
    assert_close(B, B_r, rtol=1e-10, atol_scale=1e-12, name="B")
    assert np.allclose(B[20], [1, 0, 0, 1])                          # det fallbacks -> identity
    # the resident loop keeps agreeing with the oracle on this state
    run = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=3, enable_sort=False)
    for lpp in (1, 4):
        with capi.Context(prm, nf, nt, parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"],
                          t_end=1e9, lanes_per_particle=lpp) as ctx:
            ctx.advance(1e9, max_steps=3)
            out = ctx.download()
        for k in ("pos", "vel", "rho", "drho_dt", "B"):
            assert_close(out[k], run[k], rtol=1e-9, atol_scale=1e-10, name=f"{k}/lpp={lpp}")


def test_empty_pair_list(cfgmod, geom, mex, oracle):
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.0, jitter=0.1, seed=2)
    nf, nt = parts["n_fluid"], parts["n_total"]
    e = np.zeros(0)
    nb0 = (e,) * 7
    rho, Vol, B = mex.sph_physics_shell_mex("density_correction", *nb0, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    rho_r, Vol_r, B_r = oracle.density_correction(nb0, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    assert_close(rho, rho_r, name="rho"); assert_close(B, B_r, name="B")
    f = mex.sph_physics_shell_mex("viscous_force", *nb0[:6], parts["vel"], Vol, B, prm.mu, prm.h, nf, nt, parts["mass"], parts["wall_vel"])
    assert not np.any(f)
    tau = mex.sph_physics_shell_mex("wall_shear_monitor", *nb0[:6], parts["pos"], parts["vel"], parts["wall_vel"], Vol, B, nf,
                                    prm.DL, prm.DH, prm.mu, prm.h)
    assert tuple(abs(t) for t in tau) == (0.0, 0.0)


def test_out_of_range_pair_indices_are_skipped_like_the_reference(cfgmod, geom, mex, oracle):
    """The reference `continue`s on ii<0, ii>=n_fluid, jj<0, jj>=n_total (physics.c:193,246,...)."""
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.0, jitter=0.1, seed=3)
    nf, nt = parts["n_fluid"], parts["n_total"]
    nb = list(oracle.neighbor_search(parts["pos"], nf, nt, prm.h, prm.DL))
    bad = [np.concatenate([c, c[:4]]) for c in nb]
    bad[0][-4:] = [0, nf + 1, 1, 2]          # i = 0 (invalid), i = a wall row (invalid as pair_i)
    bad[1][-4:] = [1, 2, 0, nt + 5]          # j = 0, j beyond n_total
    got = mex.sph_physics_shell_mex("density_correction", *bad, parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    ref = oracle.density_correction(tuple(bad), parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    want = oracle.density_correction(tuple(nb), parts["mass"], nf, nt, prm.rho0, prm.h, prm.inv_sigma0)
    for g, r, w, n in zip(got, ref, want, ("rho", "Vol", "B")):
        assert_close(g, r, rtol=1e-10, atol_scale=1e-12, name=n)
        assert_close(r, w, rtol=1e-12, name=n + "(bad rows ignored)")


def test_particle_outside_the_cell_rows(cfgmod, geom, capi, oracle):
    """A fluid particle above the top wall block is clamped into the last cell row (neighbor.c:275-276); it has
    no neighbours there and must not disturb anything else."""
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.5, jitter=0.1, seed=5, developed=True)
    nf, nt = parts["n_fluid"], parts["n_total"]
    parts["pos"][5] = (0.4, 2.5)
    parts["vel"][5] = (0.0, 0.0)
    run = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=2, enable_sort=False)
    with capi.Context(prm, nf, nt, parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"], t_end=1e9) as ctx:
        ctx.advance(1e9, max_steps=2)
        out = ctx.download(fields=("pos", "vel", "drho_dt"))
    for k in out:
        assert_close(out[k], run[k], rtol=1e-9, atol_scale=1e-10, name=k)


def test_non_finite_velocity_raises_device_status(cfgmod, geom, capi):
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.5, jitter=0.1, seed=5)
    parts["vel"][3, 0] = np.nan
    with capi.Context(prm, parts["n_fluid"], parts["n_total"], parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"],
                      parts["wall_vel"], t_end=1e9) as ctx:
        with pytest.raises(capi.SphxError) as e:
            ctx.advance(1e9, max_steps=4)
        assert e.value.code == capi.SPHX_ERR_DIVERGED


def test_neighbour_list_overflow_is_reported(cfgmod, geom, capi):
    """More than 64 neighbours inside 2h: the device raises SPHX_ERR_GRID instead of truncating silently."""
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=1.5, jitter=0.0, seed=5, developed=False)
    nf = parts["n_fluid"]
    rng = np.random.default_rng(0)
    parts["pos"][:90, 0] = 0.75 + 0.02 * rng.random(90)
    parts["pos"][:90, 1] = 0.50 + 0.02 * rng.random(90)
    with capi.Context(prm, nf, parts["n_total"], parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"],
                      parts["wall_vel"], t_end=1e9, lanes_per_particle=1) as ctx:
        with pytest.raises(capi.SphxError) as e:
            ctx.advance(1e9, max_steps=2)
        assert e.value.code == capi.SPHX_ERR_GRID


def _nearest_image_pairs(parts, prm):
    nf, nt = parts["n_fluid"], parts["n_total"]
    x, y = np.mod(parts["pos"][:, 0], prm.DL), parts["pos"][:, 1]
    dx = x[:nf, None] - x[None, :]
    dx = np.where(dx > 0.5 * prm.DL, dx - prm.DL, np.where(dx < -0.5 * prm.DL, dx + prm.DL, dx))
    dy = y[:nf, None] - y[None, :]
    r2 = dx * dx + dy * dy
    ok = (r2 > 1e-24) & (r2 < (2 * prm.h) ** 2)
    jj, ii = np.arange(nt)[None, :], np.arange(nf)[:, None]
    ok &= (jj >= nf) | (jj > ii)
    bi, bj = np.nonzero(ok)
    return bi, bj, np.sqrt(r2[bi, bj])


@pytest.mark.parametrize("DL", [0.7, 0.6])  # h = 0.13: 5.4 h and 4.6 h
def test_two_column_channel_matches_oracle(cfgmod, geom, mex, capi, oracle, DL):
    """4h <= DL < 6h: two cell columns, the left and the right neighbour column are the same one.  The reference finds
    every partner once through seen_neighbor (neighbor.c:342,383); so must the device search -- and the resident step
    built on it."""
    dp = 0.1
    prm, parts = make_case(cfgmod, geom, dp=dp, DL=DL, jitter=0.25, seed=9, developed=True)
    assert 4 * prm.h <= prm.DL < 6 * prm.h, (prm.DL, prm.h)
    nf, nt = parts["n_fluid"], parts["n_total"]
    a = canon_pairs(mex.sph_neighbor_search_mex(parts["pos"], nf, nt, prm.h, prm.DL))
    b = canon_pairs(oracle.neighbor_search(parts["pos"], nf, nt, prm.h, prm.DL))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert_close(a[4], b[4], rtol=1e-13, atol=1e-15 * prm.DL, name="r")
    ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=4, enable_sort=False)
    for lpp in (0, 4, 2):  # automatic (compact kernels) and the large-channel kernel forms (cell sweep with the duplicate column)
        with capi.Context(prm, nf, nt, parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"],
                          t_end=1e9, lanes_per_particle=lpp) as ctx:
            assert ctx.info()["n_cell_x"] == 2 and ctx.grid_policy()["rebuild_every"] == 1
            ctx.advance(1e9, max_steps=4)
            got = ctx.download()
        for k in ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B"):
            assert_close(got[k], ref[k], rtol=1e-9, atol_scale=1e-10, name=f"{k}/lpp={lpp}")


@pytest.mark.parametrize("n_cols_dp", [4, 2])
def test_one_column_channel_gives_the_nearest_image_of_every_pair(cfgmod, geom, mex, n_cols_dp):
    """DL < 4h (one cell column; at DL < 2h narrower than the kernel support): a pair can have two images within 2h.  The
    reference keeps whichever its cell scan meets first; the device search keeps the nearest one -- a period shorter
    than two kernel supports has no physical use, what matters is that the call is accepted (the reference accepts any
    DL > 0) and well defined."""
    dp = 0.1
    prm, parts = make_case(cfgmod, geom, dp=dp, DL=n_cols_dp * dp, jitter=0.2, seed=4, developed=False)
    assert prm.DL < 4 * prm.h
    nf, nt = parts["n_fluid"], parts["n_total"]
    a = canon_pairs(mex.sph_neighbor_search_mex(parts["pos"], nf, nt, prm.h, prm.DL))
    bi, bj, r = _nearest_image_pairs(parts, prm)
    assert np.array_equal(a[0].astype(int) - 1, bi) and np.array_equal(a[1].astype(int) - 1, bj)
    assert_close(a[4], r, rtol=1e-13, atol=1e-15, name="r")


def test_tall_columns_fall_back_to_the_compact_kernels(cfgmod, geom, capi, oracle):
    """The large-channel kernels keep fluid neighbours as 16-bit index differences, which presumes that a cell column holds
    well under 2^15 particles.  A channel 8 000 particles high and three columns long (dp = 1.25e-4, DH = 1, DL = 12 dp) does
    not: the automatic choice (4 lanes per particle at 96 k particles) must give way to 16 lanes and the compact kernels with
    their 32-bit lists, an explicit request for few lanes must be refused, and the result must still be the oracle's."""
    prm, parts = make_case(cfgmod, geom, dp=1.25e-4, DL=12 * 1.25e-4, DH=1.0, jitter=0.2, seed=5, developed=True)
    nf, nt = parts["n_fluid"], parts["n_total"]
    assert nf == 12 * 8000
    args = (prm, nf, nt, parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"])
    with capi.Context(*args, t_end=1e9) as ctx:
        assert ctx.tuning()["lanes_per_particle"] == 16, ctx.tuning()
        assert ctx.info()["n_cell_x"] <= 4
        st = ctx.advance(1e9, max_steps=3)
        got = ctx.download(fields=("pos", "vel", "drho_dt", "rho"))
    with pytest.raises(capi.SphxError) as ei:
        capi.Context(*args, t_end=1e9, lanes_per_particle=4)
    assert ei.value.identifier == "SPHX:Ctx:lpp"
    ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=3, enable_sort=False)
    assert st["step"] == 3 and abs(st["t"] - ref["stats"]["t"]) <= 1e-13 * ref["stats"]["t"]
    for k in ("pos", "vel", "drho_dt", "rho"):
        assert_close(got[k], ref[k], rtol=1e-9, atol_scale=1e-10, name=k)
