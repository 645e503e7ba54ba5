"""Manual probe (not a test): how fast can the Python step loop of one slab rank go when the exchange costs
nothing?  One process, rank 0 of a 2-slab decomposition, receive buffers fixed at "0 particles".  The physics is
meaningless (the halo empties out); the point is host time per step vs device time per step."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("sph-poiseuille-flow_amd")
slab = importlib.import_module("sph-poiseuille-flow_amd.slab")
import torch
cfg, geo = pkg.config, pkg.geometry
prm = cfg.params_from_values(end_time=1e9, dp=0.025, DL=6.0)
parts = geo.init_particles(prm)
pos, vel = geo.developed_state(prm, parts, jitter=0.05, seed=12345)
eng = slab.HipSlabEngine(prm, parts, 0, 2, 0, t_end=1e9, pos=pos, vel=vel)
with eng.stream_ctx():
    eng.local_vmax()
    eng.prepare(1e300, 100000)
    eng.recv_l.zero_(); eng.recv_r.zero_()
    for n in (200, 2000):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.compute()
            eng.finish()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"{n} steps: host enqueue {t_host/n*1e6:.1f} us/step, until device idle {t_all/n*1e6:.1f} us/step", flush=True)
