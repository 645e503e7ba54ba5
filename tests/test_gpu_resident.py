"""-m gpu: the device-resident step loop (sphx_ctx_*) against the oracle's time loop.

Short horizons compare particle by particle (trajectories have not yet decorrelated): after 1, 3 and
10 steps every state field must match the oracle to rtol 1e-9 (round-off grows ~ x10 per few steps
through the stiff EOS, p0 = 225).  The dt sequence must match to 1e-13 relative.  Long horizons compare
statistics (profile, step count) -- see test_gpu_longrun.py.
"""
import numpy as np
import pytest

from helpers import assert_close, canon_pairs, make_case

pytestmark = pytest.mark.gpu


# (dp, DL, jitter, lanes per particle)
@pytest.fixture(scope="module", params=[(0.05, 3.0, 0.2, 16), (0.04, 3.0, 0.3, 1), (0.025, 1.5, 0.25, 4),
                                        (0.05, 3.0, 0.2, 2), (0.025, 1.5, 0.25, 32), (0.04, 3.0, 0.3, 8)])
def case(request, cfgmod, geom):
    dp, DL, jit, lpp = request.param
    prm, parts = make_case(cfgmod, geom, dp=dp, DL=DL, jitter=jit, seed=7, developed=True)
    return prm, parts, lpp


def _ctx(capi, prm, parts, lpp, **kw):
    return capi.Context(prm, parts["n_fluid"], parts["n_total"], parts["pos"], parts["vel"], parts["drho_dt"],
                        parts["mass"], parts["wall_vel"], lanes_per_particle=lpp, **kw)


@pytest.mark.parametrize("n_steps", [1, 3, 10])
def test_steps_match_oracle(case, capi, oracle, n_steps):
    prm, parts, lpp = case
    nf = parts["n_fluid"]
    ref = oracle.run(prm, parts, t_end=1e9, output_interval=1e9, max_steps=n_steps, enable_sort=False)
    with _ctx(capi, prm, parts, lpp, t_end=1e9) as ctx:
        st = ctx.advance(1e9, max_steps=n_steps)
        got = ctx.download()
        tb, tt, npairs = ctx.monitor(tau=True, pairs=True)
    assert st["step"] == n_steps
    assert abs(st["t"] - ref["stats"]["t"]) <= 1e-13 * ref["stats"]["t"]
    assert abs(st["dt_last"] - ref["stats"]["dt_last"]) <= 1e-12 * ref["stats"]["dt_last"]
    tol = dict(rtol=1e-9, atol_scale=1e-10)
    for k in ("pos", "vel", "rho", "p", "drho_dt", "force", "force_prior", "Vol", "B"):
        assert_close(got[k], ref[k], name=f"{k}@{n_steps}", **tol)
    assert npairs == ref["stats"]["n_pairs_last"]
    assert_close(np.array([tb, tt]), np.array([ref["stats"]["tau_bottom"], ref["stats"]["tau_top"]]), rtol=1e-8,
                 atol_scale=1e-9, name="tau")
    assert abs(st["vmax"] - ref["stats"]["vmax"]) <= 1e-9 * ref["stats"]["vmax"]
    assert np.all(got["pos"][:nf, 0] >= 0) and np.all(got["pos"][:nf, 0] <= prm.DL)


def test_bitwise_repeatable_and_lpp_consistent(case, capi):
    """The gather formulation has no atomics in the physics: two runs are bit-identical (our race
    detector); different lanes-per-particle only change the summation tree (tiny differences)."""
    prm, parts, lpp = case
    outs = []
    for l in (lpp, lpp, 8 if lpp != 8 else 2):
        with _ctx(capi, prm, parts, l, t_end=1e9) as ctx:
            ctx.advance(1e9, max_steps=6)
            outs.append(ctx.download(fields=("pos", "vel", "drho_dt")))
    for k in ("pos", "vel", "drho_dt"):
        assert np.array_equal(outs[0][k], outs[1][k]), k
        assert_close(outs[2][k], outs[0][k], rtol=1e-9, atol_scale=1e-10, name=k)


def test_graph_and_eager_agree(case, capi):
    """hipGraph replay (steps_per_graph=4) and per-step advance calls give identical bits."""
    prm, parts, lpp = case
    with _ctx(capi, prm, parts, lpp, t_end=1e9, steps_per_graph=4) as a:
        a.advance(1e9, max_steps=13)
        A = a.download(fields=("pos", "vel", "drho_dt"))
    with _ctx(capi, prm, parts, lpp, t_end=1e9, steps_per_graph=4) as b:
        for _ in range(13):
            st = b.advance(1e9, max_steps=1)
        assert st["step"] == 13
        Bd = b.download(fields=("pos", "vel", "drho_dt"))
    for k in A:
        assert np.array_equal(A[k], Bd[k]), k


def test_target_time_clipping(case, capi, oracle):
    """dt is clipped so the loop lands on the target (remain, SPH_Poiseuille.m:252) and stops there."""
    prm, parts, lpp = case
    dt0 = 0.25 * prm.h / (prm.c_f + 1.5)
    target = 7.3 * dt0
    ref = oracle.run(prm, parts, t_end=target, output_interval=target, enable_sort=False)
    with _ctx(capi, prm, parts, lpp, t_end=1e9) as ctx:
        st = ctx.advance(target)
        assert st["done"] == 1 and abs(st["t"] - target) < 1e-12
        assert st["step"] == ref["stats"]["steps"]
        st2 = ctx.advance(target)  # already there: nothing runs
        assert st2["step"] == st["step"]
        got = ctx.download(fields=("pos", "vel"))
    assert_close(got["vel"], ref["vel"], rtol=1e-9, atol_scale=1e-10, name="vel")


def test_ctx_neighbor_list_matches_oracle(case, capi, oracle):
    prm, parts, lpp = case
    with _ctx(capi, prm, parts, lpp, t_end=1e9) as ctx:
        ctx.advance(1e9, max_steps=2)
        nb = ctx.neighbor_list()
        pos = ctx.download(fields=("pos",))["pos"]
    ref = oracle.neighbor_search(pos, parts["n_fluid"], parts["n_total"], prm.h, prm.DL)
    a, b = canon_pairs(nb), canon_pairs(ref)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert_close(a[4], b[4], rtol=1e-13, atol=1e-15 * prm.DL, name="r")


def test_time_kernel_leaves_state_untouched(case, capi):
    """sphx_ctx_time_kernel replays one neighbour pass in a graph; it may only touch per-step temporaries."""
    prm, parts, lpp = case
    with _ctx(capi, prm, parts, lpp, t_end=1e9) as ctx:
        ctx.advance(1e9, max_steps=3)
        before = ctx.download(fields=("pos", "vel", "drho_dt"))
        for name in ("k_density", "k_kgc", "k_forces", "k_continuity"):
            assert ctx.time_kernel(name, reps=8) > 0.0
        try:  # contexts that run pass E and the next pass A in one launch can time that launch as well
            assert ctx.time_kernel("k_continuity_density", reps=8) > 0.0
        except capi.SphxError as e:
            assert e.identifier == "SPHX:Ctx:kernel"
        after = ctx.download(fields=("pos", "vel", "drho_dt"))
        st = ctx.advance(1e9, max_steps=2)
        assert st["step"] == 5
    for k in before:
        assert np.array_equal(before[k], after[k]), k
    with _ctx(capi, prm, parts, lpp, t_end=1e9) as ref:
        ref.advance(1e9, max_steps=5)
        want = ref.download(fields=("pos", "vel", "drho_dt"))
    with _ctx(capi, prm, parts, lpp, t_end=1e9) as ctx:
        ctx.advance(1e9, max_steps=3)
        ctx.time_kernel("k_forces", reps=4)
        ctx.advance(1e9, max_steps=2)
        got = ctx.download(fields=("pos", "vel", "drho_dt"))
    for k in want:
        assert np.array_equal(want[k], got[k]), k
