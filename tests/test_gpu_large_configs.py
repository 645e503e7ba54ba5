"""-m gpu: BASELINE.json configs[2..4] (C3 dp 0.01/DL 6, C4 dp 0.005/DL 12, C5 dp 0.002/DL 24) and one size in between on the DEFAULT
context against the oracle's time loop -- the code paths the small cases never reach: automatic lanes per
particle (4 / 2 / 2) with the large-channel kernels (entries ahead, fluid / wall loops, LDS tiles), the multi-block cell scan (> 8 192 cells: k_scan_tiles / k_scan_add), k_max_tiles (> 16 k
workgroups), the grid-stride re-binning kernels, size_t index products and, from 10^6 fluid particles, the
device-decided ("dynamic") re-binning with its in-place reorder (k_copyback).

Every case runs past the first scheduled re-binning (K = 8 at C3, 10 at C4 and at 1.25 M, 24 where the device re-bins by itself (from 2 x 10^6 particles) -- round 4: the interval no
longer sets the skin), so the list rebuilt from the re-binned layout is compared as well.  All nine step outputs, the dt sequence (through t), max|v|, the pair
count of the rebuilt neighbour structure and the wall shear are compared particle by particle at the same
tolerance as the small cases (rtol 1e-9 after <= 25 steps; reference loop: SPH_Poiseuille.m:250-292,
neighbor.c:312-392, physics.c:857-957).  The oracle is serial C: ~1 us per particle-step (C5: ~140 s).
"""
import numpy as np
import pytest

from helpers import assert_close, make_case

pytestmark = pytest.mark.gpu

FIELDS = ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B")
#        name   dp     DL    steps  expected policy
CASES = [("C3", 0.01, 6.0, 10, dict(lpp=4, dynamic=False, big_scan=False, fuse_ea=True, forms=dict(walk_kernels=True, lds_tiles=True, tiles_abe=False, coded_lists=False))),
         # ... and through 26 steps of the fused large-channel launch with its 16-bit lists: at least three re-binnings (K = 8;
         # this noisy synthetic state also outruns the skin, so a forced re-binning and its cool-down are in there as well)
         ("C3x26", 0.01, 6.0, 26, dict(lpp=4, dynamic=False, big_scan=False, fuse_ea=True, rebins=3)),
         # 194 k particles: 4 lanes per particle up to 220 k; a step is three launches (clock in the tail of E||A) up to 4 096
         # workgroups, i.e. at C4 too (round 3)
         ("M194k", 0.01, 18.0, 10, dict(lpp=4, dynamic=False, big_scan=True, fuse_ea=True)),
         ("M259k", 0.01, 24.0, 10, dict(lpp=2, dynamic=False, big_scan=True, fuse_ea=True)),
         ("C4", 0.005, 12.0, 11, dict(lpp=2, dynamic=False, big_scan=True, fuse_ea=True, forms=dict(tiles_abe=False, coded_lists=False))),
         # from 10^6 particles every pass stages an LDS tile and the lists name tile slots (slot-coded, round 3): the smallest
         # such channel and the largest configuration
         # the smallest channel past the fused E|A launch (4096 workgroups per pass = 524 k particles): tiles in every pass and
         # slot-coded lists start here since round 4 (10^6 before), on the host's schedule
         ("M590k", 0.0045, 12.0, 13, dict(lpp=2, dynamic=False, fuse_ea=False, forms=dict(lds_tiles=True, tiles_abe=True, coded_lists=True))),
         # (host-scheduled up to 2 x 10^6 particles since round 4: K = 10 on the wider skin)
         ("M1250k", 0.004, 20.0, 13, dict(lpp=2, dynamic=False, big_scan=True, fuse_ea=False, forms=dict(lds_tiles=True, tiles_abe=True, coded_lists=True))),
         ("C5", 0.002, 24.0, 25, dict(lpp=2, dynamic=True, big_scan=True, fuse_ea=False, forms=dict(lds_tiles=True, tiles_abe=True, coded_lists=True)))]


def _compare(name, prm, parts, n_steps, capi, oracle, expect, **ctx_kw):
    nf, nt = parts["n_fluid"], parts["n_total"]
    with capi.Context(prm, nf, nt, parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"],
                      t_end=1e9, **ctx_kw) as ctx:
        info, tun, pol, sched, forms = ctx.info(), ctx.tuning(), ctx.grid_policy(), ctx.schedule(), ctx.kernel_forms()
        st = ctx.advance(1e9, max_steps=n_steps)
        got = ctx.download()
        tb, tt, npairs = ctx.monitor(tau=True, pairs=True)
        pol_after = ctx.grid_policy()
        if "rebins" in expect:
            assert ctx.schedule()["rebins"] >= expect["rebins"], ctx.schedule()
    # the launch shape / grid policy this configuration is supposed to exercise
    if "lpp" in expect:
        assert tun["lanes_per_particle"] == expect["lpp"], tun
    for k, v in expect.get("forms", {}).items():
        assert forms[k] == v, (k, forms)
    if "fuse_ea" in expect:
        assert bool(sched["fuse_ea"]) == expect["fuse_ea"], sched
    if "dynamic" in expect:
        assert bool(sched["dynamic"]) == expect["dynamic"], sched
    if "big_scan" in expect:
        assert (info["n_cell_x"] * info["n_cell_y"] > 8192) == expect["big_scan"], info
    assert pol["rebuild_every"] < n_steps, (pol, n_steps)  # the run re-bins at least once
    ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=n_steps, enable_sort=False)
    assert st["step"] == n_steps == ref["stats"]["steps"]
    assert abs(st["t"] - ref["stats"]["t"]) <= 1e-13 * ref["stats"]["t"]
    assert abs(st["dt_last"] - ref["stats"]["dt_last"]) <= 1e-12 * ref["stats"]["dt_last"]
    assert abs(st["vmax"] - ref["stats"]["vmax"]) <= 1e-9 * ref["stats"]["vmax"]
    for k in FIELDS:
        assert_close(got[k], ref[k], rtol=1e-9, atol_scale=1e-10, name=f"{name}:{k}@{n_steps}")
    assert npairs == ref["stats"]["n_pairs_last"], (npairs, ref["stats"]["n_pairs_last"])
    assert_close(np.array([tb, tt]), np.array([ref["stats"]["tau_bottom"], ref["stats"]["tau_top"]]), rtol=1e-8,
                 atol_scale=1e-9, name=f"{name}:tau")
    assert np.all(got["pos"][:nf, 0] >= 0) and np.all(got["pos"][:nf, 0] <= prm.DL)
    return pol_after


@pytest.mark.parametrize("name,dp,DL,n_steps,expect", CASES, ids=[c[0] for c in CASES])
def test_default_context_matches_oracle(name, dp, DL, n_steps, expect, cfgmod, geom, capi, oracle):
    prm, parts = make_case(cfgmod, geom, dp=dp, DL=DL, jitter=0.2, seed=11, developed=True)
    _compare(name, prm, parts, n_steps, capi, oracle, expect)


def test_c4_dynamic_rebinning_matches_oracle(cfgmod, geom, capi, oracle):
    """0.5 M particles with the device-decided re-binning forced on and a skin small enough that the drift bound
    (not the schedule) triggers re-binnings inside the window: k_bin + the multi-block scan + k_copyback."""
    prm, parts = make_case(cfgmod, geom, dp=0.005, DL=12.0, jitter=0.2, seed=5, developed=True)
    pol = _compare("C4dyn", prm, parts, 6, capi, oracle, dict(lpp=2, big_scan=True), dynamic_rebin=1, rebuild_every=5, skin_h=0.05)
    assert pol["forced_rebuilds"] >= 1, pol  # re-binnings triggered by the drift bound


def test_device_decided_rebinning_at_1p25m_matches_oracle(cfgmod, geom, capi, oracle):
    """The smallest slot-coded channel with the device-decided re-binning asked for (the default from 2 x 10^6 particles):
    K = 24 on the thin skin, in-place reorder, far bits of the re-binned lists."""
    prm, parts = make_case(cfgmod, geom, dp=0.004, DL=20.0, jitter=0.2, seed=11, developed=True)
    _compare("M1250k-dyn", prm, parts, 25, capi, oracle,
             dict(lpp=2, dynamic=True, big_scan=True, fuse_ea=False, forms=dict(lds_tiles=True, tiles_abe=True, coded_lists=True)),
             dynamic_rebin=1)


def test_c3_lattice_start_matches_oracle(cfgmod, geom, capi, oracle):
    """The reference's own initial state (pristine lattice at rest): sums that cancel exactly on the lattice are
    compared with absolute floors (helpers.assert_close's atol_scale on the field's magnitude)."""
    prm = cfgmod.params_from_values(dp=0.01, DL=6.0)
    parts = geom.init_particles(prm)
    nf, nt = parts["n_fluid"], parts["n_total"]
    assert (nf, nt) == (60000, 64800)
    ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=3, enable_sort=False)
    with capi.Context(prm, nf, nt, parts["pos"], parts["vel"], parts["drho_dt"], parts["mass"], parts["wall_vel"],
                      t_end=1e9) as ctx:
        st = ctx.advance(1e9, max_steps=3)
        got = ctx.download()
        _, _, npairs = ctx.monitor(tau=False, pairs=True)
    assert st["step"] == 3 and abs(st["t"] - ref["stats"]["t"]) <= 1e-13 * ref["stats"]["t"]
    assert npairs == ref["stats"]["n_pairs_last"] == 606600  # recorded from the reference
    for k in ("pos", "vel", "rho", "Vol", "B"):
        assert_close(got[k], ref[k], rtol=1e-9, atol_scale=1e-10, name=f"C3lattice:{k}")
    scale = prm.p0 * prm.dp  # force scale of a pressure difference of p0 over one spacing, per unit volume
    for k in ("p", "drho_dt", "force", "force_prior"):
        assert_close(got[k], ref[k], rtol=1e-9, atol_scale=1e-10, atol=1e-9 * scale, name=f"C3lattice:{k}")
