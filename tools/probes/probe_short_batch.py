"""Where the time of a short exact batch goes (manual probe): host time of enqueue_steps(n) and of sync(), total per
batch for several n.  python tools/probes/probe_short_batch.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sph-poiseuille-flow_amd")
cfg, geo, capi = pkg.config, pkg.geometry, pkg.capi
prm = cfg.params_from_values(dp=0.025, DL=3.0, end_time=1e9)
parts = geo.init_particles(prm)
pos, vel = geo.developed_state(prm, parts, jitter=0.05, seed=12345)
ctx = capi.Context(prm, parts["n_fluid"], parts["n_total"], pos, vel, parts["drho_dt"], parts["mass"], parts["wall_vel"], t_end=1e9)
ctx.enqueue_steps(5); ctx.sync()

for n in (20, 40, 80, 160, 640):
    res = []
    for rep in range(12):
        ctx.prepare_steps(n)
        ctx.sync()  # (the disarmed replay that makes the graph resident is still running otherwise: round 4 fix)
        t0 = time.perf_counter(); ctx.enqueue_steps(n); t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        res.append((t1 - t0, t2 - t1, t2 - t0))
    res = res[2:]
    e = sum(r[0] for r in res) / len(res); s = sum(r[1] for r in res) / len(res); t = sum(r[2] for r in res) / len(res)
    print(f"n={n}: enqueue {e*1e6:.1f} us, sync {s*1e6:.1f} us, total {t*1e6:.1f} us = {t*1e6/n:.2f} us/step", ctx.graph_stats())
