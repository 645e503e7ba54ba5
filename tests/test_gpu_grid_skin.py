"""-m gpu: the cell grid with a skin (particles re-binned only every K-th step).

Between two grid builds the sweeps are centred on the cell a particle was BINNED into; the device clock
tracks the largest drift from the binning positions and stops the loop before it can exceed half the skin,
then the host re-bins ("forced rebuild"), re-bins every step for a short cool-down and resumes.  Whatever K and skin are, the physics must be the one
of the reference's rebuild-every-step loop (SPH_Poiseuille.m:250-292): only the summation order may
differ, so every schedule is compared with the oracle at the short-horizon tolerance of
test_gpu_resident.py, and identical schedules must give identical bits.
"""
import numpy as np
import pytest

from helpers import assert_close, canon_pairs, make_case

pytestmark = pytest.mark.gpu

FIELDS = ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B")


@pytest.fixture(scope="module")
def case(cfgmod, geom):
    return make_case(cfgmod, geom, dp=0.04, DL=3.0, jitter=0.3, seed=11, developed=True)


@pytest.fixture(scope="module")
def calm_case(cfgmod, geom):
    """Slightly disordered lattice at rest: nothing outruns the default skin in the first few hundred steps, so the
    static re-binning schedule runs undisturbed (tests of the graph bookkeeping)."""
    return make_case(cfgmod, geom, dp=0.04, DL=3.0, jitter=0.05, seed=11, developed=False)


def _ctx(capi, prm, parts, **kw):
    return capi.Context(prm, parts["n_fluid"], parts["n_total"], parts["pos"], parts["vel"], parts["drho_dt"],
                        parts["mass"], parts["wall_vel"], t_end=1e9, **kw)


def _pol(ctx):
    p = ctx.grid_policy()
    return p["rebuild_every"], p["skin"], p["forced_rebuilds"]


def _check(got, ref, tag):
    for k in FIELDS:
        assert_close(got[k], ref[k], rtol=1e-9, atol_scale=1e-10, name=f"{k}@{tag}")


@pytest.mark.parametrize("K", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("n_steps", [4, 7])
def test_any_rebuild_interval_matches_oracle(case, capi, oracle, K, n_steps):
    prm, parts = case
    ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=n_steps, enable_sort=False)
    with _ctx(capi, prm, parts, rebuild_every=K, lanes_per_particle=8) as ctx:
        pol = ctx.grid_policy()
        assert pol["rebuild_every"] == K and (pol["skin"] > 0) == (K > 1)
        st = ctx.advance(1e9, max_steps=n_steps)
        got = ctx.download()
        tb, tt, npairs = ctx.monitor(tau=True, pairs=True)
    assert st["step"] == n_steps
    _check(got, ref, f"K{K}")
    assert npairs == ref["stats"]["n_pairs_last"]
    assert_close(np.array([tb, tt]), np.array([ref["stats"]["tau_bottom"], ref["stats"]["tau_top"]]), rtol=1e-8,
                 atol_scale=1e-9, name="tau")


def test_undersized_skin_forces_rebuilds_and_stays_exact(case, capi, oracle):
    """A skin far too thin for K: the device must stop the loop by itself, the host re-bins and cools down."""
    prm, parts = case
    n_steps = 12
    ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=n_steps, enable_sort=False)
    with _ctx(capi, prm, parts, rebuild_every=8, skin_h=0.03, lanes_per_particle=4) as ctx:
        st = ctx.advance(1e9, max_steps=n_steps)
        pol = ctx.grid_policy()
        got = ctx.download()
        tb, tt, npairs = ctx.monitor(tau=True, pairs=True)
    assert st["step"] == n_steps
    assert pol["forced_rebuilds"] >= 1 and pol["rebuild_every"] == 8
    _check(got, ref, "forced")
    assert npairs == ref["stats"]["n_pairs_last"]
    assert_close(np.array([tb, tt]), np.array([ref["stats"]["tau_bottom"], ref["stats"]["tau_top"]]), rtol=1e-8,
                 atol_scale=1e-9, name="tau")


@pytest.mark.parametrize("kw", [dict(rebuild_every=5), dict(rebuild_every=8, skin_h=0.03), dict(rebuild_every=3, skin_h=0.05)])
def test_call_pattern_does_not_change_the_bits(case, capi, kw):
    """One advance, step-by-step advances and fire-and-forget batches follow the same rebuild schedule (it is
    a function of the step count and of device-side events only) -> identical bits."""
    prm, parts = case
    n = 23
    outs = []
    with _ctx(capi, prm, parts, steps_per_graph=4, **kw) as a:
        a.advance(1e9, max_steps=n)
        outs.append(a.download(fields=("pos", "vel", "drho_dt", "rho", "Vol")))
        pol_a = _pol(a)
    with _ctx(capi, prm, parts, steps_per_graph=4, **kw) as b:
        for _ in range(n):
            st = b.advance(1e9, max_steps=1)
        assert st["step"] == n
        outs.append(b.download(fields=("pos", "vel", "drho_dt", "rho", "Vol")))
        assert _pol(b) == pol_a
    with _ctx(capi, prm, parts, steps_per_graph=4, **kw) as c:
        c.enqueue_steps(9)
        c.enqueue_steps(6)
        st = c.sync()
        assert st["step"] == 15
        c.enqueue_steps(8)
        st = c.sync()
        assert st["step"] == n
        outs.append(c.download(fields=("pos", "vel", "drho_dt", "rho", "Vol")))
        assert _pol(c) == pol_a
    for o in outs[1:]:
        for k in outs[0]:
            assert np.array_equal(outs[0][k], o[k]), k


@pytest.mark.parametrize("lpp", [2, 4])
def test_large_channel_kernels_hand_out_the_last_step_of_any_batch(case, capi, lpp):
    """The large-channel kernels (few lanes per particle) write force, force_prior, rho and p in the LAST step of a batch only
    (FluidTmp::lazy_out): whatever the call pattern -- one batch, single steps, back-to-back batches without a sync, a target time
    instead of a step budget -- a download returns the outputs of the last executed step, bit for bit."""
    prm, parts = case
    n = 13
    fields = ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B")
    outs = []
    with _ctx(capi, prm, parts, lanes_per_particle=lpp, rebuild_every=5) as a:
        assert a.kernel_forms()["walk_kernels"]
        st_a = a.advance(1e9, max_steps=n)
        outs.append(a.download(fields=fields))
    with _ctx(capi, prm, parts, lanes_per_particle=lpp, rebuild_every=5) as b:
        for _ in range(n):
            b.advance(1e9, max_steps=1)
        outs.append(b.download(fields=fields))
    with _ctx(capi, prm, parts, lanes_per_particle=lpp, rebuild_every=5) as c:
        c.enqueue_steps(4)
        c.enqueue_steps(6)
        c.sync()
        mid = c.download(fields=fields)                      # outputs of step 10 ...
        c.enqueue_steps(3)
        c.sync()
        outs.append(c.download(fields=fields))
    with _ctx(capi, prm, parts, lanes_per_particle=lpp, rebuild_every=5) as d:
        d.advance(1e9, max_steps=10)
        ten = d.download(fields=fields)                      # ... are those of a batch that ends there
        st_d = d.advance(st_a["t"], max_steps=0)             # the rest by target time: the step that reaches it is the last
        assert st_d["step"] == n
        by_time = d.download(fields=fields)
    for k in fields:
        assert np.array_equal(mid[k], ten[k]), k
        for o in outs[1:]:
            assert np.array_equal(outs[0][k], o[k]), k
        # (the last dt of this one is clipped to the target: equal to rounding, not to the bit)
        scale = float(np.max(np.abs(outs[0][k]))) or 1.0
        assert_close(by_time[k], outs[0][k], rtol=1e-9, atol=1e-12 * scale, name="by time: " + k)
    assert np.any(outs[0]["force"] != ten["force"])          # (and they did change in between)


@pytest.mark.parametrize("lpp", [2, 4])
def test_download_without_a_sync_takes_the_steps_a_stopped_batch_owes(case, capi, lpp):
    """A batch of sphx_ctx_enqueue_steps on a skin far too thin stops early (drift bound, static schedule); its armed last step
    -- the only one that writes force / force_prior / rho / p on the large-channel kernels -- has not run.  download and monitor
    called straight away, without sphx_ctx_sync, must first take the owed steps like sync does: state AND outputs of step n."""
    prm, parts = case
    n = 23
    with _ctx(capi, prm, parts, lanes_per_particle=lpp, rebuild_every=8, skin_h=0.03) as a:
        assert a.kernel_forms()["walk_kernels"]
        a.enqueue_steps(n)
        st_a = a.sync()
        assert st_a["step"] == n and a.grid_policy()["forced_rebuilds"] > 0  # (the batch did stop on the way)
        ref = a.download(fields=FIELDS)
        tau_ref = a.monitor(tau=True, pairs=True)
    with _ctx(capi, prm, parts, lanes_per_particle=lpp, rebuild_every=8, skin_h=0.03) as b:
        b.enqueue_steps(n)
        got = b.download(fields=FIELDS)   # no sync
        assert b.sync()["step"] == n
    with _ctx(capi, prm, parts, lanes_per_particle=lpp, rebuild_every=8, skin_h=0.03) as c:
        c.enqueue_steps(n)
        tau_got = c.monitor(tau=True, pairs=True)   # no sync
        assert c.sync()["step"] == n
    for k in FIELDS:
        assert np.array_equal(got[k], ref[k]), k
    assert tau_got == tau_ref


def test_cool_downs_follow_a_fixed_schedule(case, capi):
    """Every forced rebuild starts a cool-down (re-binning every step for 16, 32, ... steps) at a step index set by
    the device-side event alone; how the host chunks its calls does not matter -> still identical bits."""
    prm, parts = case
    kw = dict(rebuild_every=6, skin_h=0.04, steps_per_graph=8, lanes_per_particle=8)
    n = 2300
    with _ctx(capi, prm, parts, **kw) as a:
        a.advance(1e9, max_steps=40)
        early = a.grid_policy()
        a.advance(1e9, max_steps=n - 40)
        A = a.download(fields=("pos", "vel", "drho_dt"))
        pol_a = _pol(a)
    assert early["rebuild_every"] == 6 and early["forced_rebuilds"] >= 2
    assert pol_a[2] > early["forced_rebuilds"]  # after each cool-down it tried the interval again (and was stopped again)
    with _ctx(capi, prm, parts, **kw) as b:
        done = 0
        for chunk in (700, 1, 323, 1000, 276):
            b.enqueue_steps(chunk)
            done += chunk
            if chunk != 1:
                assert b.sync()["step"] == done
        assert done == n and b.sync()["step"] == n
        Bd = b.download(fields=("pos", "vel", "drho_dt"))
        assert _pol(b) == pol_a
    for k in A:
        assert np.array_equal(A[k], Bd[k]), k


def test_graphs_replay_from_any_phase(calm_case, capi):
    """A caller with a fixed cadence (the reference logs every 20 steps, SPH_Poiseuille.m:285-291) enters the schedule
    at a different phase on every call: each (phase, length) gets its graph once, after that every call is pure
    replay -- and gives the bits of one long call."""
    prm, parts = calm_case
    with _ctx(capi, prm, parts, rebuild_every=8) as a:  # K = 8, 20-step calls: pos cycles 5, 1, 5, ... and lay flips
        a.advance(1e9, max_steps=5)                     # start misaligned (pos = 5)
        for _ in range(8):
            a.advance(1e9, max_steps=20)
        warm = a.graph_stats()
        for _ in range(8):
            st = a.advance(1e9, max_steps=20)
        hot = a.graph_stats()
        A = a.download(fields=("pos", "vel", "drho_dt"))
    assert st["step"] == 5 + 16 * 20
    assert hot["graphs_captured"] == warm["graphs_captured"]                 # nothing new to capture
    assert hot["slots_eager"] == warm["slots_eager"]                         # and nothing launched eagerly
    assert hot["slots_replayed"] - warm["slots_replayed"] == 8 * 20
    with _ctx(capi, prm, parts, rebuild_every=8) as b:
        b.advance(1e9, max_steps=5 + 16 * 20)
        Bd = b.download(fields=("pos", "vel", "drho_dt"))
    for k in A:
        assert np.array_equal(A[k], Bd[k]), k


def test_graphs_replay_after_a_forced_rebuild(case, capi):
    """A forced rebuild flips the state parity without taking a step; with an even K the context then never returns
    to the phase its first graph was captured from.  It must keep replaying graphs all the same."""
    prm, parts = case
    with _ctx(capi, prm, parts, rebuild_every=8, skin_h=0.03, lanes_per_particle=4) as ctx:
        ctx.advance(1e9, max_steps=12)
        assert ctx.grid_policy()["forced_rebuilds"] >= 1
    # a skin that is adequate most of the time: forced rebuilds are rare events followed by a 16-step cool-down
    with _ctx(capi, prm, parts, rebuild_every=8, skin_h=0.25, steps_per_graph=16) as ctx:
        ctx.advance(1e9, max_steps=3000)
        pol, g0 = ctx.grid_policy(), ctx.graph_stats()
        ctx.advance(1e9, max_steps=1600)
        g1 = ctx.graph_stats()
        forced_in_window = ctx.grid_policy()["forced_rebuilds"] - pol["forced_rebuilds"]
    replayed, eager = g1["slots_replayed"] - g0["slots_replayed"], g1["slots_eager"] - g0["slots_eager"]
    assert replayed + eager >= 1600
    # eager slots: cool-downs (<= 16..1024 steps per forced rebuild) and the empty slots behind each stop only
    assert replayed >= 1600 - forced_in_window * 1100 - 64 and (forced_in_window > 0 or eager == 0), (g0, g1, pol)


def test_prepare_steps_makes_the_next_batch_pure_replay(calm_case, capi):
    prm, parts = calm_case
    with _ctx(capi, prm, parts) as ctx:
        ctx.enqueue_steps(5)
        ctx.sync()
        ctx.prepare_steps(20)
        g0 = ctx.graph_stats()
        ctx.enqueue_steps(20)
        st = ctx.sync()
        g1 = ctx.graph_stats()
    assert st["step"] == 25
    assert g1["graphs_captured"] == g0["graphs_captured"] and g1["slots_eager"] == g0["slots_eager"]
    assert g1["slots_replayed"] - g0["slots_replayed"] == 20


@pytest.mark.parametrize("n_steps", [1, 2, 3, 4, 5, 6])
def test_pair_list_between_rebuilds(case, capi, oracle, n_steps):
    """The MEX-convention pair list taken from a stale (but still valid) grid equals a fresh search."""
    prm, parts = case
    with _ctx(capi, prm, parts, rebuild_every=5) as ctx:
        ctx.advance(1e9, max_steps=n_steps)
        nb = ctx.neighbor_list()
        pos = ctx.download(fields=("pos",))["pos"]
    ref = oracle.neighbor_search(pos, parts["n_fluid"], parts["n_total"], prm.h, prm.DL)
    a, b = canon_pairs(nb), canon_pairs(ref)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert_close(a[4], b[4], rtol=1e-13, atol=1e-15 * prm.DL, name="r")


def test_narrow_domain_falls_back_to_every_step(cfgmod, geom, capi):
    """DL too short for three skinned cell columns -> K = 1, no skin (the periodic sweep needs ncx >= 3)."""
    prm, parts = make_case(cfgmod, geom, dp=0.05, DL=0.45, jitter=0.1, seed=3, developed=True)
    with _ctx(capi, prm, parts, rebuild_every=6, skin_h=0.5) as ctx:
        pol = ctx.grid_policy()
        assert pol["rebuild_every"] == 1 and pol["skin"] == 0.0
        assert ctx.advance(1e9, max_steps=3)["step"] == 3


# ---- dynamic re-binning (the device decides when to re-bin; default from 2 x 10^6 particles, forced on here) -------------

@pytest.mark.parametrize("kw", [dict(rebuild_every=5), dict(rebuild_every=8, skin_h=0.03), dict(rebuild_every=3)])
@pytest.mark.parametrize("n_steps", [1, 4, 9, 17])
def test_dynamic_rebinning_matches_oracle(case, capi, oracle, kw, n_steps):
    """Same physics whatever triggers the re-binning: the K-th step, or -- skin far too thin -- the drift bound on
    (nearly) every step, decided on the device without the host."""
    prm, parts = case
    ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=n_steps, enable_sort=False)
    with _ctx(capi, prm, parts, dynamic_rebin=1, lanes_per_particle=4, **kw) as ctx:
        st = ctx.advance(1e9, max_steps=n_steps)
        got = ctx.download()
        tb, tt, npairs = ctx.monitor(tau=True, pairs=True)
        pol = ctx.grid_policy()
    assert st["step"] == n_steps
    _check(got, ref, f"dyn{kw}")
    assert npairs == ref["stats"]["n_pairs_last"]
    assert_close(np.array([tb, tt]), np.array([ref["stats"]["tau_bottom"], ref["stats"]["tau_top"]]), rtol=1e-8,
                 atol_scale=1e-9, name="tau")
    if kw.get("skin_h") and n_steps >= 4:
        assert pol["forced_rebuilds"] >= 1  # drift-triggered re-binnings, counted on the device


def test_dynamic_rebinning_is_chunk_invariant_and_repeatable(case, capi):
    prm, parts = case
    kw = dict(dynamic_rebin=1, rebuild_every=6, skin_h=0.06, steps_per_graph=4)
    n = 61
    outs = []
    for pattern in ("one", "single", "batches", "one"):
        with _ctx(capi, prm, parts, **kw) as c:
            if pattern == "one":
                c.advance(1e9, max_steps=n)
            elif pattern == "single":
                for _ in range(n):
                    c.advance(1e9, max_steps=1)
            else:
                for chunk in (7, 30, 1, 23):
                    c.enqueue_steps(chunk)
                assert c.sync()["step"] == n
            outs.append(c.download(fields=("pos", "vel", "drho_dt", "rho", "Vol", "B")))
            assert c.grid_policy()["forced_rebuilds"] >= 1
    for o in outs[1:]:
        for k in outs[0]:
            assert np.array_equal(outs[0][k], o[k]), k


def test_dynamic_pair_list_and_outputs_right_after_a_rebinning(case, capi, oracle):
    """Outputs of a step that ended with a re-binning are stored in the old ordering (reached through src_of)."""
    prm, parts = case
    for n_steps in (3, 4):  # K = 4: step 4 ends with a re-binning, step 3 does not
        ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=n_steps, enable_sort=False)
        with _ctx(capi, prm, parts, dynamic_rebin=1, rebuild_every=4) as ctx:
            ctx.advance(1e9, max_steps=n_steps)
            got = ctx.download()
            nb = ctx.neighbor_list()
        _check(got, ref, f"dyn@{n_steps}")
        refnb = oracle.neighbor_search(got["pos"], parts["n_fluid"], parts["n_total"], prm.h, prm.DL)
        a, b = canon_pairs(nb), canon_pairs(refnb)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
