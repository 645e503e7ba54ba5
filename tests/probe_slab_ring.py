"""In-process ring of slabs vs the single-GPU context on one device (manual probe): us/step and the ratio.
python tests/probe_slab_ring.py C4 2 40"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("sph-poiseuille-flow_amd")
slab = importlib.import_module("sph-poiseuille-flow_amd.slab")
cfg, geo, capi = pkg.config, pkg.geometry, pkg.capi
W = {"C2": dict(dp=0.025, DL=3.0), "C3": dict(dp=0.01, DL=6.0), "C4": dict(dp=0.005, DL=12.0), "C5": dict(dp=0.002, DL=24.0),
     "C4x2": dict(dp=0.005, DL=24.0)}
name, world, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
prm = cfg.params_from_values(end_time=1e9, **W[name])
parts = geo.init_particles(prm)
pos, vel = geo.developed_state(prm, parts, jitter=0.05, seed=12345)
engines = [slab.HipSlabEngine(prm, parts, r, world, 0, t_end=1e9, pos=pos, vel=vel, native=True) for r in range(world)]
slab.HipSlabEngine.group_run(engines, 8); [e.sync() for e in engines]
t0 = time.perf_counter(); slab.HipSlabEngine.group_run(engines, steps); [e.sync() for e in engines]; ring = (time.perf_counter() - t0) / steps
for e in engines: e.close()
ctx = capi.Context(prm, parts["n_fluid"], parts["n_total"], pos, vel, parts["drho_dt"], parts["mass"], parts["wall_vel"], t_end=1e9)
ctx.enqueue_steps(40); ctx.sync()
t0 = time.perf_counter(); ctx.enqueue_steps(steps); ctx.sync(); one = (time.perf_counter() - t0) / steps
print(f"{name}: ring of {world} slabs on one device {1e6*ring:.1f} us/step, single context {1e6*one:.1f} us/step, ratio {ring/one:.2f}")
