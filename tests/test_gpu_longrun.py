"""-m gpu: whole-run statistics of the device-resident driver against the analytic Poiseuille profile and
the figures recorded from the reference.  Trajectories are chaotic at round-off, so long
runs are compared through the binned profile, L2, the step count and the wall shear -- SURVEY.md section 8c.

Recorded from the reference probe: dp 0.05 -> 19 771 steps to 20 s, L2 1.42 %; dp 0.025 -> 39 496 steps,
L2 0.84 %.  Oracle realisations wander 1.3-2.3 % (dp 0.05) over the last seconds, hence the windows."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _record(name, res):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, f"longrun_{name}.json"), "w") as f:
            json.dump(dict(n_total=res.n_total, steps=res.steps, t=res.t, wall_seconds=res.wall_seconds, L2=res.L2_error,
                           L2_mean_last5=res.L2_time_mean(5) if len(res.full_profile_u) >= 5 else None,
                           particle_steps_per_s=res.particle_steps_per_s, tau_bottom=res.tau_bottom, tau_top=res.tau_top,
                           tau_target=res.tau_target, u_mean=np.nan_to_num(res.u_mean).tolist(), u_exact=res.u_exact.tolist()), f)
    except OSError:
        pass


def test_default_config_20s(cfgmod, driver):
    """config.ini as shipped (dp 0.05, c_f 15, 20 s, output every 1 s)."""
    prm = cfgmod.load_config(os.path.join(os.path.dirname(OUT), "config.ini"))
    res = driver.run(prm)
    _record("dp0.05", res)
    assert abs(res.t - 20.0) < 1e-9
    assert 19700 <= res.steps <= 19900, res.steps          # reference 19 771 (+ ~1 clipped step per output point)
    assert 0.008 < res.L2_error < 0.03, res.L2_error        # reference 1.42 %, oracle 1.3-2.3 %
    assert res.L2_error < 0.05                              # the reference's own pass criterion
    assert abs(res.tau_bottom - res.tau_target) < 0.05 and abs(res.tau_top - res.tau_target) < 0.05
    assert len(res.profile_times) == 21 and np.nanmax(res.mid_profile_u[-1]) > 0.9


def test_headline_config_dp0025_20s(cfgmod, driver):
    """BASELINE.json configs[1]: dp = 0.025, the full loop to 20 s; north-star bar L2 <= 1 % (reference 0.84 %)."""
    prm = cfgmod.params_from_values(dp=0.025, DL=3.0, end_time=20.0, output_interval=1.0)
    res = driver.run(prm)
    _record("dp0.025", res)
    assert 39300 <= res.steps <= 39800, res.steps           # reference 39 496
    # The north-star bar, L2 <= 1 %, is held on the profile averaged over the output points t = 16..20 s: a statistic of the
    # developed flow.  One instant is one chaotic realisation -- every kernel change is a new one; recorded over rounds 2-4 and
    # re-binning intervals K = 8..32: 0.77-0.95 % at t = 20 s against 0.72-0.82 % for the mean -- so the instant gets a soft
    # bound (1.2 %) and is reported (gpurun_out/longrun_dp0.025.json, bench.py's `accuracy`).
    assert res.L2_time_mean(last=5) <= 0.01, res.L2_time_mean(last=5)
    assert res.L2_error <= 0.012, res.L2_error
    assert abs(res.tau_bottom - res.tau_target) < 0.03


@pytest.mark.parametrize("c_f, tc, ref_steps, ref_L2", [(15.0, 0.30, 24714, 0.0136), (10.0, 0.10, 16895, 0.0275)])
def test_dp004_rows_of_the_reference_probe(cfgmod, driver, c_f, tc, ref_steps, ref_L2):
    """BASELINE.md section 2, dp = 0.04: the shipped constants (24 714 steps, L2 1.36 %) and the README-table
    constants c_f = 10, transport_coeff = 0.10 (16 895 steps, L2 2.75 %) -- a different sound speed (dt rule,
    EOS stiffness) and shifting strength than every other run here.  The reference probe clipped dt only at
    t_end; this driver also lands on every output point, so use one output interval."""
    prm = cfgmod.params_from_values(dp=0.04, DL=3.0, c_f=c_f, transport_coeff=tc, end_time=20.0, output_interval=20.0)
    res = driver.run(prm)
    _record(f"dp0.04_cf{c_f:g}_tc{tc:g}", res)
    assert abs(res.steps - ref_steps) <= 0.002 * ref_steps, (res.steps, ref_steps)   # dt sequence: vmax history
    assert 0.5 * ref_L2 < res.L2_error < 1.6 * ref_L2, (res.L2_error, ref_L2)          # one chaotic realisation each
    assert abs(res.tau_bottom - res.tau_target) < 0.04 and abs(res.tau_top - res.tau_target) < 0.04


def test_mex_engine_matches_resident_engine(cfgmod, driver):
    """The unmodified six-calls-per-step loop through the MEX-surface mirror and the resident loop agree."""
    prm = cfgmod.params_from_values(dp=0.05, DL=1.5, end_time=0.02, output_interval=0.01)
    a = driver.run(prm, engine="mex")
    b = driver.run(prm, engine="resident")
    assert a.steps == b.steps
    assert np.allclose(a.vel, b.vel, rtol=1e-8, atol=1e-10) and np.allclose(a.pos, b.pos, rtol=1e-10, atol=1e-12)
