"""Manual probe (not a test): a long stretch of the 6 M-particle case from the lattice at rest (stability of the
rebuild schedule, graph replay and forced-rebuild handling at scale).  python tools/probes/probe_soak.py [t_end]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("sph-poiseuille-flow_amd")
t_end = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4
prm = pkg.config.params_from_values(dp=0.002, DL=24.0, end_time=t_end, output_interval=t_end / 4)
res = pkg.driver.run(prm, log=lambda s: print(s, flush=True))
import numpy as np
nf = res.n_fluid
u = res.vel[:nf, 0]
mid = np.abs(res.pos[:nf, 1] - 0.5 * prm.DH) < 0.1 * prm.DH
# start-up of plane Poiseuille flow from rest (series solution), centre line
def u_centre(t, g=prm.gravity_g, nu=prm.nu, H=prm.DH):
    y = 0.5 * H
    u = g / (2 * nu) * y * (H - y)
    for k in range(0, 200):
        n = 2 * k + 1
        u -= 4 * g * H * H / (nu * np.pi ** 3 * n ** 3) * np.sin(n * np.pi * y / H) * np.exp(-n * n * np.pi ** 2 * nu * t / H ** 2)
    return u
print("centre band mean u_x %.5f, analytic centre-line start-up value %.5f, L2 of the binned profile vs the STEADY parabola %.4f"
      % (float(u[mid].mean()), u_centre(res.t), res.L2_error))
print(dict(n_total=res.n_total, steps=res.steps, t=res.t, wall=round(res.wall_seconds, 1), rate="%.3e" % res.particle_steps_per_s,
           umax=float(u.max()), g_t=prm.gravity_g * res.t, policy=res.grid_policy,
           finite=bool(np.isfinite(res.vel).all() and np.isfinite(res.pos).all())))
