"""Per-kernel graph-regime timings of one workload (manual probe): python tools/probes/probe_kernels.py C5 [steps]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sph-poiseuille-flow_amd")
cfg, geo, capi = pkg.config, pkg.geometry, pkg.capi
W = {"C2": dict(dp=0.025, DL=3.0), "C3": dict(dp=0.01, DL=6.0), "C4": dict(dp=0.005, DL=12.0), "C5": dict(dp=0.002, DL=24.0)}
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
prm = cfg.params_from_values(end_time=1e9, **W[name])
parts = geo.init_particles(prm)
pos, vel = geo.developed_state(prm, parts, jitter=0.05, seed=12345)
ctx = capi.Context(prm, parts["n_fluid"], parts["n_total"], pos, vel, parts["drho_dt"], parts["mass"], parts["wall_vel"], t_end=1e9)
dt = 1.0
if steps > 0:  # steps = 0: kernel timings on the initial state only (experiment builds with broken physics)
    ctx.enqueue_steps(40); ctx.sync()
    t0 = time.perf_counter(); ctx.enqueue_steps(steps); ctx.sync(); dt = time.perf_counter() - t0
steps = max(steps, 1)
out = {k: round(1e3 * ctx.time_kernel(k, reps=20), 2) for k in ("k_density", "k_kgc", "k_forces", "k_continuity")}
print(name, f"{1e6*dt/steps:.1f} us/step, {parts['n_total']*steps/dt:.4g} particle-steps/s; kernels (us):", out, ctx.tuning())
